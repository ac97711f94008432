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
line.  The line also carries `roofline` (dominant kernel k_ioc vs the fp32 MFMA peak, duration from
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

FP32_MFMA_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0


def ioc_flops_per_row(d):
    """Algorithmic FLOPs of k_ioc per row (one (agent,k) trajectory), SURVEY.md 8(d) D4 IOC terms."""
    H, E, B, T = d.H, d.E, d.B, d.T_pred
    return d.iters * (T * (6.0 * H * (E + H) + 2.0 * B * H * H + 2 * H + 4 * d.E_v) + 2.0 * H * 2 * T)


def committed_traffic(windows, bf16=False):
    """HBM bytes per k_ioc launch from the committed rocprofv3 PMC passes (profiles/, collected from this very
    command at the same windows/step in separate --pmc runs): 2 x FETCH_SIZE (gfx950 counts wide reads at half,
    MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes.  None when no matching profile is committed."""
    name = {(128, False): "r01_final_bench_pmc_per_kernel.json", (128, True): "r01_bf16_bench_pmc_per_kernel.json",
            (512, False): "r01_final_bench_w512_pmc_per_kernel.json"}.get((windows, bool(bf16)))
    path = os.path.join(ROOT, "profiles", name) if name else None
    if path is None or not os.path.exists(path):
        return None
    with open(path) as fh:
        pmc = json.load(fh)
    for name, c in pmc.items():
        hit = ("k_ioc_bf16ILi128" in name) if bf16 else (("k_iocILi128" in name or name.startswith("void k_ioc<128"))
                                                          and "ELb0ELb1E" not in name)     # (not the opt-in compact form of the `alt` pass)
        if hit and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    return None


def cpu_baseline(d_full, seed):
    """The CPU restatement (oracle, NOT TF1 -- the reference cannot run), timed on this host on a bounded sample.  `value` is
    the faster of the two restatements: oracle/desire_torch.py (torch fp32, batched GEMMs on every host thread -- the fair
    one); the numpy oracle the parity tests check against is reported next to it."""
    import torch
    from oracle import desire_oracle as O                      # cpu_baseline leg: allowed importer
    from oracle import desire_torch as OT
    from desire_amd.spec import init_weights
    from desire_amd.synth import make_case
    tr = lambda x: np.ascontiguousarray(x.transpose(1, 0, 2, 3).reshape(x.shape[1], -1, 3))
    def torch_run(n_windows):
        d = d_full.replace(n_scenes=n_windows, n_grids=1)
        w = init_weights(d, seed)
        past, fut, eps, grids, gos = make_case(d, seed=seed + 1)
        OT.DT = torch.float32
        wt = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in w.items()}
        with torch.no_grad():
            t0 = time.perf_counter()
            OT.forward_loss(tr(past), tr(fut), eps, grids, gos, wt, d)
            return d.R, time.perf_counter() - t0
    nthr0 = torch.get_num_threads()
    try:
        torch_run(1)                                           # thread pools / allocator warm
        best = None
        for nt in sorted({min(nthr0, c) for c in (8, 16, 32, 64, 128)}):      # many small GEMMs: the widest pool is not the fastest
            torch.set_num_threads(nt)
            r8, t8 = torch_run(8)
            if best is None or t8 < best[2]:
                best = (nt, r8, t8)
        threads, r8, t8 = best
        torch.set_num_threads(threads)
        n = int(min(64, max(8, 8 * 12.0 / max(t8, 1e-3))))     # about 12 s of CPU work, at most 64 windows (host memory)
        rt, tt = torch_run(n) if n > 8 else (r8, t8)
    finally:
        OT.DT = torch.float64
        torch.set_num_threads(nthr0)
    d = d_full.replace(n_scenes=4, n_grids=1)
    w = init_weights(d, seed)
    past, fut, eps, grids, gos = make_case(d, seed=seed + 1)
    t0 = time.perf_counter()
    ref = O.forward(tr(past), tr(fut), eps, grids, gos, w, d)
    dt = time.perf_counter() - t0
    # accuracy gate of the metric: the HIP path on the SAME 4 windows against that oracle run.  Sample
    # generation end to end; the IOC pass from the oracle's own Y0, so that a neighbour sitting within 1e-7 of a bin edge
    # cannot land in different bins on the two sides.
    from desire_amd import _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    tt_ = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    hh = _lib.Handle(d)
    hh.set_weights(w)
    p_t, f_t, e_t, g_t = tt_(past), tt_(fut), tt_(eps), tt_(grids)
    hh.set_scene_grids(g_t.data_ptr(), gos)
    Yg = torch.zeros((d.R, d.T_pred, 2), device=dev); sg = torch.zeros((d.R,), device=dev)
    hh.encode(p_t.data_ptr(), f_t.data_ptr())
    hh.sample(e_t.data_ptr(), Yg.data_ptr())
    torch.cuda.synchronize()
    e_y0 = float(np.abs(Yg.cpu().numpy() - ref["Y0"]).max())
    Yg.copy_(tt_(ref["Y0"].astype(np.float32)))
    hh.ioc_refine(Yg.data_ptr(), sg.data_ptr())
    torch.cuda.synchronize()
    dY = Yg.cpu().numpy() - ref["Y"]
    accuracy = {"max_abs_err_Y0": e_y0, "max_abs_err_Y": float(np.abs(dY).max()), "ade_vs_oracle": float(np.sqrt((dY ** 2).sum(-1)).mean()),
                "gate": 1e-3, "units": "normalised frame coordinates", "sample": "%d samples (4 windows), HIP path vs oracle/desire_oracle.py" % d.R}
    hh.close()
    # the reference's own structure (model/model.py:211): one object at a time, batch dimension 1, for the
    # sample-generation stages (the IOC stage needs the whole group and stays batched above)
    d1 = d.replace(n_scenes=1, mno=1, iters=1)
    n_obj = 16
    t1 = time.perf_counter()
    for a_ in range(n_obj):
        e1 = eps.reshape(d.n_scenes, d.K, d.mno, d.L)[0, :, a_].reshape(d1.R, d.L)
        pn = O.normalise(tr(past)[:, a_:a_ + 1], d1)
        fn = O.normalise(tr(fut)[:, a_:a_ + 1], d1)
        Hx = O.gru_encode(pn, w, "enc_x"); Hy = O.gru_encode(fn, w, "enc_y")
        vin = O.relu(np.concatenate([Hx, Hy], -1) @ w["fc_c/w"] + w["fc_c/b"])
        mu, ls = O.vae_encoder(vin, w, d.L)
        z = O.rows_from_agents(mu, d1) + np.sqrt(np.exp(O.rows_from_agents(ls, d1))) * e1
        xh = O.vae_decoder(z, w)
        Hr = O.rows_from_agents(Hx, d1)
        xz = O.softmax(O.relu(xh @ w["mask_fc/w"] + w["mask_fc/b"])) * Hr
        O.decode(xz, Hr, O.rows_from_agents(pn[-1], d1), w, d1)
    dt1 = (time.perf_counter() - t1) / n_obj
    return {"accuracy": accuracy, "value": rt / tt, "unit": "agent-trajectory-samples/s", "cores": int(threads), "kind": "port",
            "sample": "oracle/desire_torch.py forward (torch fp32, batched, %d threads: the fastest of 8..128) on %d windows = %d samples, %.1f s; "
                      "CPU restatement, not TF1 (reference graph does not build)" % (threads, rt // (d.K * d.mno), rt, tt),
            "numpy_oracle": {"value": d.R / dt, "unit": "agent-trajectory-samples/s",
                             "note": "oracle/desire_oracle.py (the parity checker) on 4 windows = %d samples, %.1f s" % (d.R, dt)},
            "per_object_loop": {"value": d.K / dt1, "unit": "agent-trajectory-samples/s",
                                "note": "sample-generation stages only, one object at a time like model/model.py:211 "
                                        "(%d objects x K=%d, %.2f s each)" % (n_obj, d.K, dt1)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=None, help="loader windows (scenes) per step per GPU (128 = the size the per-kernel tables "
                                                             "in profiles/README.md were taken at; throughput saturates around 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--bf16", action="store_true",
                    help="bf16 matrix operands in the IOC kernel (BASELINE configs[2] arithmetic; NOT the headline fp32 line)")
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
                    help="row-compacted social pooling (DESIRE_IOC_VARIANT=8, fp32, groups of up to 32 agents): the pooling MFMAs run "
                         "on the rows that have a neighbour in the bin only; NOT the headline (fewer flops are executed than the "
                         "dense formula credits)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step's launch sequence from a hipGraph (desire_graph_*; 1 GPU): for launch-bound shapes such as "
                         "`--windows 2` (configs[4] puts 2 windows on each of 8 GPUs)")
    ap.add_argument("--train", action="store_true",
                    help="time a TRAINING step instead (forward + backward + gradient all-reduce + clip + Adam + device repack); "
                         "not the BASELINE metric -- the default run is")
    a = ap.parse_args()
    if a.windows is None:
        # inference saturates around 512 windows; a training step keeps ~0.5 GB of activations per window (27 GB of it the
        # pooled operand at 128 windows), so it stays at the size its profile was taken at
        a.windows = 128 if (a.train or a.bf16 or a.shard == "agents") else 512

    if a.compact:
        os.environ["DESIRE_IOC_VARIANT"] = "8"
    import torch
    import torch.distributed as dist
    from desire_amd import _lib
    from desire_amd.spec import Dims, flops_per_sample, init_weights
    from desire_amd.synth import make_case

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    # DESIRE_BENCH_ONE_GPU=1: a smoke test of the multi-rank code path on a box with a single GPU (all ranks share cuda:0 and
    # rendezvous over gloo, because RCCL refuses two ranks on one device).  Never set for a measurement.
    one_gpu = os.environ.get("DESIRE_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    d = Dims(n_scenes=a.windows, mno=a.mno, bf16=int(a.bf16), K=a.K, T_obs=8, T_pred=40, H=a.H, L=128, n_grids=1, grid_size=a.grid,
             nb_w=a.nb, nb_h=a.nb, sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1)
    w = init_weights(d, a.seed)
    past, fut, eps, grids, gos = make_case(d, seed=a.seed + 1 + rank, n_absent=0)
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
        from desire_amd.dist import ShardedIoc
        sharded = ShardedIoc(h, rank, world)

    def step():
        if a.shard == "agents":                    # per-agent stages locally, IOC with the neighbour all-gather per step
            h.encode(past_t.data_ptr(), fut_t.data_ptr(), stream)
            h.sample(eps_t.data_ptr(), Y.data_ptr(), stream)
            sharded.run(Y, score)
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
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    if not a.graph:                                  # per-kernel hipEvents are host-side records: not part of a replayed graph
        h.set_profiling(True)
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
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(Y).all()) and bool(torch.isfinite(score).all())
    # outside the timed region: the same steps through the opt-in row-compacted pooling, reported next to the headline
    alt = None
    if world == 1 and not (a.train or a.bf16 or a.graph or a.compact) and a.shard == "scenes" and a.mno <= 32 and a.H <= 128:
        os.environ["DESIRE_IOC_VARIANT"] = "8"                # read by the library at every launch
        try:
            step(); torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(max(2, a.steps // 2)):
                step()
            torch.cuda.synchronize()
            alt_dt = (time.perf_counter() - ta) / max(2, a.steps // 2)
            alt = {"row_compacted_pooling": {"value": d.R / alt_dt, "ms_per_step": alt_dt * 1e3, "unit": "samples/s",
                                             "note": "opt-in (DESIRE_IOC_VARIANT=8 / --compact): same results up to fp32 summation order; "
                                                     "executes fewer flops than the dense formula, hence not the headline"}}
        finally:
            del os.environ["DESIRE_IOC_VARIANT"]

    per_kernel = {}
    for name, ms in prof:
        per_kernel.setdefault(name, []).append(ms)
    kern_ms = {k: float(np.mean(v)) for k, v in per_kernel.items()}
    ioc_ms = kern_ms.get("ioc")
    ioc_tflops = ioc_flops_per_row(d) * d.R / (ioc_ms * 1e-3) / 1e12 if ioc_ms else None
    whole_tflops = flops_per_sample(d) * d.R * a.steps / dt / 1e12

    peak = BF16_MFMA_PEAK_TFLOPS if a.bf16 else FP32_MFMA_PEAK_TFLOPS
    if rank == 0 and a.train:
        samples = d.R * world * a.steps
        fwd = sum(v for k, v in kern_ms.items() if not k.startswith("bwd_"))
        bwd = sum(v for k, v in kern_ms.items() if k.startswith("bwd_"))
        print(json.dumps({
            "metric": "TRAINING agent-trajectory-samples/sec (K=20, T_pred=40; fwd+bwd+allreduce+clip+Adam+repack)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1] shapes, training step; %d windows/step/GPU%s" % (a.windows, "; launch sequence replayed from a hipGraph" if a.graph else ""),
                       "windows_per_gpu": a.windows, "rows_per_gpu": d.R, "parallelism": "scene-sharded x%d, flat-gradient all-reduce" % world},
            "forward_ms": fwd, "backward_ms": bwd, "kernel_ms": kern_ms,
            "whole_step_tflops_3x_forward_credit": 3 * whole_tflops}))
    elif rank == 0:
        samples = d.R * world * a.steps
        out = {
            "metric": "agent-trajectory-samples/sec (K=20, T_pred=40)",
            "value": samples / dt, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "step_ms_median": step_ms[len(step_ms) // 2], "step_ms_min": step_ms[0],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 operands (IOC kernel), f32 accumulate/state; other kernels f32" if a.bf16 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: SDD-like synthetic windows, 32 agent slots/window, K=20, "
                                   "T_obs=8/T_pred=40, H=128, L=128, fp32, posterior CVAE, IOC 1 refinement, "
                                   "social grid 4x4, scene grid 64x64x32; %d windows/step/GPU" % a.windows,
                       "windows_per_gpu": a.windows, "rows_per_gpu": d.R, "parallelism": ("scene-sharded x%d" % world) if a.shard == "scenes" else
                                      ("agent-sharded x%d: %d slots/rank of %d-agent scenes, RCCL all-gather of [R_loc, H] per IOC step" % (world, d.mno, d.mno * world)),
                       "flops_per_sample": flops_per_sample(d)},
            "roofline": {"bound": "mfma", "kernel": "k_ioc_bf16<128,16,32,1>" if a.bf16 else "k_ioc<128,16,32>", "achieved": ioc_tflops,
                         "peak": peak, "unit": "TFLOP/s", "frac": (ioc_tflops / peak) if ioc_tflops else None,
                         "traffic": committed_traffic(a.windows, a.bf16), "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                         "algorithmic_hbm_bytes_per_launch": d.R * (2 * d.T_pred * 2 * 4 + 4) + d.A * d.H * 4,
                         "kernel_ms": ioc_ms,
                         "algorithmic_flops_per_launch": ioc_flops_per_row(d) * d.R,
                         "whole_path_tflops": whole_tflops, "whole_path_frac": whole_tflops / peak},
            "kernel_ms": kern_ms,
        }
        if a.compact:
            out["config"]["workload"] += "; row-compacted pooling (opt-in)"
            out["roofline"]["note"] = ("row-compacted pooling executes fewer flops than the dense formula credits: achieved / frac are "
                                       "dense-equivalent figures, not utilisation")
        if a.nb != 0.15:      # sparse windows: bins that are empty across a whole tile are skipped, so fewer flops are EXECUTED
            out["config"]["workload"] += "; social window %.3g (non-default: sparse bins)" % a.nb
            out["roofline"]["note"] = ("achieved / frac credit the dense algorithm's flops; with --nb below 0.15 part of the social "
                                       "contraction is skipped (exact zeros), so frac can exceed 1 and is not a utilisation figure")
        if alt:
            out["alt"] = alt
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(d, a.seed)
            out["accuracy"] = out["cpu_baseline"].pop("accuracy")
            assert out["accuracy"]["max_abs_err_Y0"] < 1e-3 and out["accuracy"]["max_abs_err_Y"] < 1e-3, out["accuracy"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
