#!/usr/bin/env python3
"""bench.py -- agent-trajectory-samples/s of the DESIRE hot path on MI355X.

A "step" = one pass of the hot path (encode -> K-sample decode -> IOC score + one refinement)
over one batch of loader windows already resident in HBM.  Workload = BASELINE.json configs[1]:
32 agent slots per window, K=20, T_obs=8 / T_pred=40, H=128, L=128, fp32, social grid 4x4, scene
grid 64x64x32; `--windows` windows (DataLoader batch entries, each its own scene) per step per GPU.

    python bench.py                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: windows are independent scenes (social pooling never crosses a window), so they shard
across ranks with NO data-path collective; weak scaling (fixed windows per GPU).  Timing: barrier +
torch.cuda.synchronize() on both sides of exactly K steps, max over ranks, rank 0 prints one JSON
line LAST on stdout, kept under 4 KB (benchlib/emit.py: the complete record with every extra leg goes to
bench_full.json and to an earlier stdout line prefixed `#full `).  The line also carries `roofline` (dominant kernel k_ioc vs the fp32 MFMA peak, duration from
hipEvents on the launch stream over the timed steps) and `cpu_baseline` (the numpy oracle timed on
this host on a bounded sample: 8 windows = 5120 samples).  Default: 512 windows per step per GPU.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchlib.common import (BF16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, committed_traffic, ioc_flops_per_row,  # noqa: E402,F401
                             sdd_windows)
from benchlib.legs_alt import (bf16_config2_leg, config3_shape_leg, few_windows_leg, operand_form_legs, reference_defaults_leg, sdd_leg,  # noqa: E402,F401
                               training_step_leg, with_loader_leg)
from benchlib.emit import emit  # noqa: E402
from benchlib.legs_cpu import cpu_baseline  # noqa: E402
from benchlib.legs_dist import agent_sharded_comm, agent_sharded_setup, multi_rank_legs  # noqa: E402


def self_spawn(n_gpus):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE in the environment): re-execute this very command line under
    torch.distributed.run with N ranks on this node, rendezvous on 127.0.0.1 and a free port.  stdout is inherited, so the ONE JSON
    line rank 0 prints is this process's output; the exit code is the launcher's."""
    import socket
    import subprocess
    if os.environ.get("DESIRE_BENCH_ONE_GPU") != "1":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n_gpus:
            raise SystemExit("--gpus %d but %d GPU(s) visible on this node" % (n_gpus, have))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=None, help="loader windows (scenes) per step per GPU (128 = the size the per-kernel tables "
                                                             "in profiles/README.md were taken at; throughput saturates around 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed region of the headline: no alt / sdd / cpu_baseline / accuracy legs.  This is the command the "
                         "rocprofv3 sets under profiles/ are collected from, so that every kernel symbol appears at ONE launch size")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--bf16", action="store_true",
                    help="bf16 matrix operands in the IOC kernel (BASELINE configs[2] arithmetic; NOT the headline fp32 line)")
    ap.add_argument("--split", action="store_true",
                    help="split-bf16 operands (dims.bf16 = 2): every fp32 product runs as three bf16 MFMAs (hi.hi + lo.hi + hi.lo, fp32 "
                         "accumulate) in the kernels that have that form; fp32-equivalent results (~1e-5), NOT the headline line")
    ap.add_argument("--x6", action="store_true",
                    help="three bf16 pieces per fp32 operand, six bf16 MFMAs per product (dims.bf16 = 3) in the IOC kernel: fp32-class accuracy "
                         "from the bf16 matrix pipe; NOT the headline line")
    ap.add_argument("--mno", type=int, default=32, help="agent slots per window (configs[2]/[3]: 64)")
    ap.add_argument("--H", type=int, default=128, help="hidden width (configs[3]: 256)")
    ap.add_argument("--K", type=int, default=20, help="samples per agent (configs[3]: 50)")
    ap.add_argument("--grid", type=int, default=4, help="social grid side: 4 = the reference's flag (16 bins), 6 = the paper's 36 bins")
    ap.add_argument("--shard", choices=["scenes", "agents"], default="scenes",
                    help="multi-GPU partitioning: 'scenes' (default; windows are independent, no data-path collective) or "
                         "'agents' (the agents of EVERY scene block-sharded over the ranks: --mno slots per rank, hidden states "
                         "all-gathered over RCCL once per IOC step -- SURVEY.md 8(e) E1's prescribed form)")
    ap.add_argument("--nb", type=float, default=0.15,
                    help="social window (normalised units, square).  0.15 keeps every bin of every tile populated (the dense case the "
                         "headline is quoted on); the reference's flags -- 32 px on SDD frames -- are about 0.023, where most bins are empty")
    ap.add_argument("--compact", action="store_true",
                    help="row-compacted social pooling (dims.ioc_form = DESIRE_IOC_COMPACT, fp32, groups of up to 32 agents): the pooling MFMAs run "
                         "on the rows that have a neighbour in the bin only; NOT the headline (fewer flops are executed than the "
                         "dense formula credits)")
    ap.add_argument("--flags", type=int, default=0, help="dims.flags (DESIRE_FLAG_* of include/desire_hip.h), for A/B runs")
    ap.add_argument("--ioc_form", type=int, default=0, help="dims.ioc_form (DESIRE_IOC_* of include/desire_hip.h), for A/B runs")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step's launch sequence from a hipGraph (desire_graph_*; 1 GPU): for launch-bound shapes such as "
                         "`--windows 2` (configs[4] puts 2 windows on each of 8 GPUs)")
    ap.add_argument("--bn", choices=["frozen", "per_object", "batch"], default="frozen",
                    help="CVAE batch-norm: 'frozen' moving statistics (default, the headline) or 'per_object' = the reference graph's literal "
                         "phase=train on a batch of one object (dims.bn_mode = 1: per-sample moments, model/model.py:453-462,471-481)")
    ap.add_argument("--data", choices=["both", "synthetic", "sdd"], default="both",
                    help="'both' (default): the headline on dense synthetic windows (every slot present, every social bin populated) "
                         "plus an `sdd` object measured, outside the timed region, on real SDD bookstore windows; 'synthetic' skips it; "
                         "'sdd': the TIMED region itself runs on the real SDD windows (9 of 32 slots present, 32-px neighbourhood) -- with "
                         "--flags 4 / 12 the compacted paths -- for profiling them (profiles/collect_r05.sh); not the headline workload")
    ap.add_argument("--train", action="store_true",
                    help="time a TRAINING step instead (forward + backward + gradient all-reduce + clip + Adam + device repack); "
                         "not the BASELINE metric -- the default run is")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(a.gpus)
    if (a.split or a.x6) and (a.bf16 or a.compact or a.shard == "agents" or (a.split and a.x6) or (a.x6 and a.train)):
        raise SystemExit("--split / --x6 (split-bf16 operands in the IOC kernel) are forms of their own: not with --bf16 / --compact / --shard agents "
                         "/ each other (and --x6 is inference only)")
    if a.windows is None:
        # inference saturates around 512 windows; a training step keeps ~0.5 GB of activations per window (27 GB of it the
        # pooled operand at 128 windows), so it stays at the size its profile was taken at
        a.windows = 128 if (a.train or a.bf16 or a.shard == "agents") else 512

    import torch
    import torch.distributed as dist
    from desire_amd import _lib
    from desire_amd.spec import Dims, flops_per_sample, init_weights
    from desire_amd.synth import make_case

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (either launch N ranks with torch.distributed.run, or run `python bench.py --gpus N` "
                         "from a shell without WORLD_SIZE: it starts its own ranks)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    # DESIRE_BENCH_ONE_GPU=1: a smoke test of the multi-rank code path on a box with a single GPU (all ranks share cuda:0 and
    # rendezvous over gloo, because RCCL refuses two ranks on one device).  Never set for a measurement.
    one_gpu = os.environ.get("DESIRE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # launched by torch.distributed.run (RANK / MASTER_PORT in the environment): the process group is created at world_size 1 too, so that
    # `--gpus 1` under the launcher runs its barrier / all-reduces through RCCL like the N > 1 lines do (tests/test_gpu_nccl.py)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    d = Dims(n_scenes=a.windows, mno=a.mno, bf16=3 if a.x6 else 2 if a.split else int(a.bf16), bn_mode={"frozen": 0, "per_object": 1, "batch": 2}[a.bn], K=a.K, T_obs=8, T_pred=40, H=a.H, L=128, n_grids=1, grid_size=a.grid,
             nb_w=a.nb, nb_h=a.nb, sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1, ioc_form=8 if a.compact else a.ioc_form, flags=a.flags)
    w = init_weights(d, a.seed)
    past, fut, eps, grids, gos = make_case(d, seed=a.seed + 1 + rank, n_absent=0)
    if a.data == "sdd":
        W_IMG, H_IMG = 1424.0, 1088.0
        d = d.replace(nb_w=32.0 / W_IMG, nb_h=32.0 / H_IMG, sx=1.0 / W_IMG, sy=1.0 / H_IMG)
        past, fut, _ = sdd_windows(d.n_scenes, d.mno)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    if a.train:
        from desire_amd.dist import allreduce_mean_
        h.set_training(True)
        gflat = h.grad_tensor()

    if a.shard == "agents":
        if a.windows < 2 or a.windows % 2:
            raise SystemExit("--shard agents: an even number of windows per GPU (two micro-batches)")
        halves, sharded = agent_sharded_setup(d, w, grids_t, gos, past_t, fut_t, eps_t, rank, world, dev)

    def step():
        if a.shard == "agents":                    # per-agent stages locally, IOC with the neighbour all-gather per step
            for x in halves:
                x["h"].encode(x["past"].data_ptr(), x["fut"].data_ptr(), stream)
                x["h"].sample(x["eps"].data_ptr(), x["Y"].data_ptr(), stream)
            sharded.run([x["Y"] for x in halves], [x["score"] for x in halves])
            return
        h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), stream)
        if a.train:
            h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), stream)
            allreduce_mean_(gflat)
            h.clip_grads(10.0, stream=stream)
            h.adam_step(1e-4, stream=stream)

    if a.graph:
        if world != 1 or a.shard == "agents":
            raise SystemExit("--graph: single-GPU, scene-sharded runs only")
        side = torch.cuda.Stream()
        gstream = side.cuda_stream

        def body(st):                                # everything except Adam (its step size changes per call)
            h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), st)
            if a.train:
                h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), st)
                h.clip_grads(10.0, stream=st)
        torch.cuda.synchronize()
        body(gstream)                                # warm-up outside capture (lazy allocations)
        side.synchronize()
        h.graph_begin(gstream)
        body(gstream)
        gid = h.graph_end(gstream)

        def step():                                  # noqa: F811
            h.graph_launch(gid, gstream)
            if a.train:
                h.adam_step(1e-4, stream=gstream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    if not a.graph:                                  # per-kernel hipEvents are host-side records: not part of a replayed graph
        h.set_profiling(True)
        if a.shard == "agents":
            halves[0]["h"].set_profiling(True)
    # per-step device times for the median SURVEY.md D1 asks for: one event per step boundary on the launch stream (the
    # records are asynchronous and sit between kernels that are already serialised on that stream)
    ev_stream = side if a.graph else torch.cuda.current_stream()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record(ev_stream)
    for i in range(a.steps):
        step()
        marks[i + 1].record(ev_stream)
    fence()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps))
    h.set_profiling(False)
    prof = h.get_profile()
    if a.shard == "agents":
        halves[0]["h"].set_profiling(False)
        prof = halves[0]["h"].get_profile()
    ranks_seen = 1
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # n_gpus of the line = what the collective library saw, not what the command line asked for: every rank contributes
        # (1, rank + 1) to a sum over the job
        chk = torch.tensor([1.0, float(rank + 1)], device=dev, dtype=torch.float64)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        ranks_seen = int(round(float(chk[0].item())))
        if ranks_seen != dist.get_world_size() or int(round(float(chk[1].item()))) != world * (world + 1) // 2:
            raise SystemExit("all-reduce checksum: %s over %d ranks" % (chk.tolist(), world))
    if a.shard == "agents":
        Y = torch.cat([x["Y"] for x in halves]); score = torch.cat([x["score"] for x in halves])
    assert bool(torch.isfinite(Y).all()) and bool(torch.isfinite(score).all())
    comm = None
    if a.shard == "agents":
        comm = agent_sharded_comm(sharded, halves, fence, max(2, a.steps // 2), world, d.T_pred)
    # outside the timed region: the same steps through the opt-in row-compacted pooling, reported next to the headline
    alt = None
    if a.headline_only:
        a.no_cpu_baseline = True
        if a.data == "both":
            a.data = "synthetic"
    if world == 1 and not a.headline_only and a.data != "sdd" and not (a.train or a.bf16 or a.split or a.x6 or a.graph or a.compact) and a.shard == "scenes" and a.mno <= 32 and a.H <= 128:
        alt = operand_form_legs(a, d, w, h, step, past_t, fut_t, eps_t, grids_t, gos, Y, score, stream)
        def leg(name, fn, *args, **kw):            # an optional leg must never cost the driver its headline line
            try:
                alt[name] = fn(*args, **kw)
            except Exception as e:                 # noqa: BLE001 -- reported in the line itself
                alt[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                torch.cuda.synchronize()
        leg("bf16_config2", bf16_config2_leg, d, w, a.seed, dev, a.steps, with_accuracy=not a.no_cpu_baseline)
        leg("reference_defaults", reference_defaults_leg, a.seed, dev, a.steps)
        leg("training_step", training_step_leg, d, a.seed, dev, a.steps)
        leg("few_windows", few_windows_leg, d, a.seed, dev)
        leg("with_loader", with_loader_leg, d, w, a.seed, dev, a.steps)
        leg("config3_shape", config3_shape_leg, a.seed, dev, a.steps)

    # outside the timed region: the same path on REAL SDD windows (BASELINE configs[1] names "SDD bookstore"): tiled bookstore/video6
    # windows with their absent slots and the reference's 32-px neighbourhood (train.py:68-70) on the 1424 x 1088 frame
    sdd = None
    if world == 1 and not (a.train or a.graph or a.compact) and a.shard == "scenes" and a.mno >= 32 and a.data == "both":
        sdd = sdd_leg(a, d, w, grids_t, gos, eps_t, Y, score, stream, dev)

    per_kernel = {}
    for name, ms in prof:
        per_kernel.setdefault(name, []).append(ms)
    kern_ms = {k: float(np.mean(v)) for k, v in per_kernel.items()}
    ioc_ms = kern_ms.get("ioc")
    ioc_tflops = ioc_flops_per_row(d) * d.R / (ioc_ms * 1e-3) / 1e12 if ioc_ms else None
    whole_tflops = flops_per_sample(d) * d.R * a.steps / dt / 1e12
    # the decoder hoists the constant-input half of its per-step contraction out of the time loop (x_z is the same at every step,
    # model/model.py:280): SURVEY.md D4 credits T*6H*2H, the kernel executes T*6H*H + 6H*H
    executed_per_sample = flops_per_sample(d) - (d.T_pred - 1) * 6.0 * d.H * d.H
    whole_exec_tflops = executed_per_sample * d.R * a.steps / dt / 1e12

    peak = BF16_MFMA_PEAK_TFLOPS if a.bf16 else FP32_MFMA_PEAK_TFLOPS
    algorithmic_bytes = d.R * (2 * d.T_pred * 2 * 4 + 4) + d.A * d.H * 4
    traffic = committed_traffic(d, a.bf16) if a.mno == 32 and a.grid == 4 and a.data != "sdd" and not (a.compact or a.split or a.x6) else None
    assert traffic is None or traffic >= algorithmic_bytes, (traffic, algorithmic_bytes)
    if rank == 0 and a.train:
        samples = d.R * world * a.steps
        fwd = sum(v for k, v in kern_ms.items() if not k.startswith("bwd_"))
        bwd = sum(v for k, v in kern_ms.items() if k.startswith("bwd_"))
        emit({
            "metric": "TRAINING agent-trajectory-samples/sec (K=20, T_pred=40; fwd+bwd+allreduce+clip+Adam+repack)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 state / activations / gradients; matrix products as three bf16 MFMAs on split operands (IOC forward, IOC BPTT, weight-gradient reductions, large data-gradient convolutions)" if a.split else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] shapes, training step; %d windows/step/GPU%s%s" % (a.windows, "; launch sequence replayed from a hipGraph" if a.graph else "",
                                   "; dims.bf16 = 2: k_ioc_x3 forward (fp32 saves), k_ioc_bwd_x3, k_gemm_tn2_xp, k_conv_gather_x3, six-product sample generation; the remaining backward kernels fp32" if a.split else ""),
                       "windows_per_gpu": a.windows, "rows_per_gpu": d.R, "parallelism": "scene-sharded x%d, flat-gradient all-reduce" % world},
            "forward_ms": fwd, "backward_ms": bwd, "kernel_ms": kern_ms,
            "whole_step_tflops_3x_forward_credit": 3 * whole_tflops})
    elif rank == 0:
        samples = d.R * world * a.steps
        out = {
            "metric": "agent-trajectory-samples/sec (K=20, T_pred=40)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "step_ms_median": step_ms[len(step_ms) // 2], "step_ms_min": step_ms[0],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 operands (IOC kernel), f32 accumulate/state; other kernels f32" if a.bf16 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SDD-like synthetic windows, 32 agent slots/window, K=20, "
                                   "T_obs=8/T_pred=40, H=128, L=128, fp32, posterior CVAE, IOC 1 refinement, "
                                   "social grid 4x4, scene grid 64x64x32; %d windows/step/GPU" % a.windows,
                       "windows_per_gpu": a.windows, "rows_per_gpu": d.R, "parallelism": ("scene-sharded x%d" % world) if a.shard == "scenes" else
                                      ("agent-sharded x%d: %d slots/rank of %d-agent scenes, RCCL all-gather of [R_loc, H] per IOC step" % (world, d.mno, d.mno * world)),
                       "flops_per_sample": flops_per_sample(d), "collective_backend": dist.get_backend() if use_dist else None},
            "roofline": {"bound": "mfma", "kernel": "k_ioc_bf16<128,16,32,1,false>" if a.bf16 else "k_ioc<%d,16,32,32,false,%s>" % (d.H, "true" if a.compact else "false"), "achieved": ioc_tflops,
                         "peak": peak, "unit": "TFLOP/s", "frac": (ioc_tflops / peak) if ioc_tflops else None,
                         "traffic": traffic, "traffic_over_algorithmic": (traffic / algorithmic_bytes) if traffic else None,
                         "traffic_unit": "bytes/launch", "traffic_source": ("from_profile: %s (rocprofv3 --pmc passes of `bench.py --headline-only` at this "
                                                                           "launch size, committed; not measured in this run)" % committed_traffic.source)
                                                                          if traffic else "no committed PMC profile of this launch class",
                         "algorithmic_hbm_bytes_per_launch": algorithmic_bytes,
                         "kernel_ms": ioc_ms,
                         "algorithmic_flops_per_launch": ioc_flops_per_row(d) * d.R,
                         "whole_path_tflops": whole_tflops, "whole_path_frac": whole_tflops / peak,
                         "whole_path_frac_executed": whole_exec_tflops / peak,
                         "whole_path_note": "whole_path_frac credits SURVEY.md D4's formula (%.1f MF/sample); whole_path_frac_executed counts what the "
                                            "kernels execute (%.1f MF/sample: the decoder's constant-input contraction is hoisted)"
                                            % (flops_per_sample(d) / 1e6, executed_per_sample / 1e6)},
            "kernel_ms": kern_ms,
        }
        if a.data == "sdd":
            out["data"] = "SDD bookstore/video6 windows (committed 160-frame slice, tiled)"
            out["config"]["workload"] = ("NOT the headline workload: real SDD bookstore/video6 windows, %d slots/window with ~9 present, K=%d, T 8/40, H=%d, "
                                         "32-px neighbourhood, dims.flags = %d; %d windows/step/GPU" % (d.mno, d.K, d.H, a.flags, a.windows))
            out["roofline"]["note"] = "achieved / frac credit the dense formula over ALL slots; on these windows most rows are padding or skipped"
            out["roofline"]["traffic"] = None
        if a.compact:
            out["config"]["workload"] += "; row-compacted pooling (opt-in)"
            out["roofline"]["note"] = ("row-compacted pooling executes fewer flops than the dense formula credits: achieved / frac are "
                                       "dense-equivalent figures, not utilisation")
        if a.nb != 0.15:      # sparse windows: bins that are empty across a whole tile are skipped, so fewer flops are EXECUTED
            out["config"]["workload"] += "; social window %.3g (non-default: sparse bins)" % a.nb
            out["roofline"]["note"] = ("achieved / frac credit the dense algorithm's flops; with --nb below 0.15 part of the social "
                                       "contraction is skipped (exact zeros), so frac can exceed 1 and is not a utilisation figure")
        if a.x6:
            out["metric"] += " -- split-bf16 (3 pieces, 6 products) operands in the IOC kernel"
            out["dtype"] = "bf16x6 (three bf16 pieces per fp32 operand, six bf16 MFMAs per product, f32 accumulate/state) in the IOC kernel; other kernels f32"
            out["config"]["workload"] += "; IOC contractions on the bf16 matrix pipe with three-piece operands (dims.bf16 = 3)"
            out["roofline"].update({"kernel": ("k_ioc_x6r2<%d,16,32,false>" if d.n_scenes * d.K * d.mno >= 256 * 64 else "k_ioc_x3<%d,16,32,false,3>") % d.H, "peak": BF16_MFMA_PEAK_TFLOPS / 6.0,
                                    "frac": (ioc_tflops / (BF16_MFMA_PEAK_TFLOPS / 6.0)) if ioc_tflops else None, "traffic": None,
                                    "traffic_source": "not collected for this form",
                                    "note": "achieved = fp32-equivalent (algorithmic) flops / kernel time; peak = dense bf16 MFMA peak / 6 "
                                            "(six bf16 products per fp32 product)"})
            out["roofline"]["whole_path_frac"] = whole_tflops / (BF16_MFMA_PEAK_TFLOPS / 6.0)
            out["roofline"]["whole_path_frac_executed"] = whole_exec_tflops / (BF16_MFMA_PEAK_TFLOPS / 6.0)
        if a.split:
            out["metric"] += " -- split-bf16 (3-product) operands in the IOC kernel"
            out["dtype"] = "bf16x3 (hi+lo split of fp32 operands, three bf16 MFMAs per product, f32 accumulate/state) in the IOC kernel; other kernels f32"
            out["config"]["workload"] += "; IOC contractions on the bf16 matrix pipe with split operands (dims.bf16 = 2)"
            out["roofline"].update({"kernel": ("k_ioc_step<%d,16,32,2>" if d.H == 256 else "k_ioc_x3<%d,16,32,false,2>") % d.H, "peak": BF16_MFMA_PEAK_TFLOPS / 3.0,
                                    "frac": (ioc_tflops / (BF16_MFMA_PEAK_TFLOPS / 3.0)) if ioc_tflops else None, "traffic": None,
                                    "traffic_source": "not collected for this form",
                                    "note": "achieved = fp32-equivalent (algorithmic) flops / kernel time; peak = dense bf16 MFMA peak / 3 "
                                            "(three bf16 products per fp32 product); whole_path fractions are against the same figure "
                                            "although only the IOC kernel runs in this form"})
            out["roofline"]["whole_path_frac"] = whole_tflops / (BF16_MFMA_PEAK_TFLOPS / 3.0)
            out["roofline"]["whole_path_frac_executed"] = whole_exec_tflops / (BF16_MFMA_PEAK_TFLOPS / 3.0)
        if a.bf16:
            out["metric"] += " -- bf16 operands"
            out["config"]["workload"] = ("BASELINE configs[2] arithmetic (bf16 MFMA operands, fp32 accumulate / state): synthetic windows, %d agent "
                                         "slots/window%s, K=%d, T_obs=8/T_pred=40, H=%d, scene grid 64x64x32; %d windows/step/GPU = %d agents"
                                         % (d.mno, " (cluster form: %d workgroups per group)" % (d.mno // 32) if d.mno > 64 else "", d.K, d.H, a.windows, d.A))
            out["roofline"]["kernel"] = ("k_ioc_bf16_cl<%d>" % d.H) if d.mno > 64 else "k_ioc_bf16<%d,16,32,%d>" % (d.H, 2 if d.mno > 32 else 1)
        elif a.mno != 32 or a.H != 128 or a.K != 20 or a.grid != 4:
            out["config"]["workload"] = ("non-default shape: %d agent slots/window, K=%d, H=%d, social grid %dx%d, T_obs=8/T_pred=40, fp32; "
                                         "%d windows/step/GPU" % (d.mno, d.K, d.H, a.grid, a.grid, a.windows))
            if a.split or a.x6:      # H = 256 and scenes of more than 128 agents: the step-wise kernel, one launch per step
                if d.H == 256 or d.mno > 128:
                    out["roofline"]["kernel"] = "k_ioc_step<%d,16,32,%d>" % (d.H, 3 if a.x6 else 2)
            else:
                out["roofline"]["kernel"] = ("k_ioc_step<%d,16,32,0>" % d.H) if d.mno > 128 else \
                                            ("k_ioc_cl<%d,16,32,false>" % d.H) if (d.mno > 64 or d.H == 256) else "k_ioc<%d,16,32,%d,false,false>" % (d.H, d.mno)
        if comm:
            out["comm"] = comm
        if a.bn == "per_object":
            out["config"]["workload"] += "; CVAE batch-norm with per-object statistics (the reference's batch-of-one phase=train)"
        if a.bn == "batch":
            out["config"]["workload"] += "; CVAE batch-norm with whole-batch statistics (phase=train over everything the call batches)"
        if alt:
            out["alt"] = alt
        if sdd:
            out["sdd"] = sdd
        if world == 1 and not a.no_cpu_baseline and a.data != "sdd":
            out["cpu_baseline"] = cpu_baseline(d, a.seed)
            out["accuracy"] = out["cpu_baseline"].pop("accuracy")
            assert out["accuracy"]["max_abs_err_Y0"] < 1e-3 and out["accuracy"]["max_abs_err_Y"] < 1e-3, out["accuracy"]
    # N > 1, default partitioning: the agent-sharded form (north_star's "RCCL all-gather only for the social-pooling neighbour exchange", over
    # collectives and over peer buffers) and BASELINE configs[3] / configs[4] at their named multi-GPU shapes are measured as well, OUTSIDE the timed
    # region, into the same line (benchlib/legs_dist.py: multi_rank_legs; one watchdog bounds them all)
    if world > 1 and a.shard == "scenes" and not (a.train or a.graph):
        multi_rank_legs(out if rank == 0 else None, a, d, w, grids_t, rank, world, dev, stream, fence,
                        lambda: emit(out))
    if rank == 0 and not a.train:
        emit(out)
    if use_dist:
        try:
            dist.destroy_process_group()
        except Exception:                                         # noqa: BLE001 -- a peer that died in an extra leg: the line is already out
            pass


if __name__ == "__main__":
    main()
