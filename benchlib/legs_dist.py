"""bench.py: the multi-rank legs (agent-sharded IOC over RCCL all-gathers or peer buffers, BASELINE configs[3] / configs[4] shapes over N ranks)."""
import json
import os
import sys
import time

import numpy as np

from .common import (BF16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, committed_traffic, ioc_flops_per_row,  # noqa: F401
                     sdd_windows)


def agent_sharded_setup(d, w, grids_t, gos, past_t, fut_t, eps_t, rank, world, dev):
    """SURVEY.md 8(e) E1's prescribed partitioning: the agents of EVERY scene block-sharded over the ranks (d.mno slots per rank).  Two
    micro-batches (half of the rank's windows each, own handle): their IOC steps alternate on the compute stream while the per-step
    neighbour all-gathers run on a communication stream (dist.PipelinedShardedIoc)."""
    import torch
    from desire_amd import _lib
    from desire_amd.dist import PipelinedShardedIoc, ShardedIoc
    dh = d.replace(n_scenes=d.n_scenes // 2)
    halves = []
    for i in range(2):
        hs = slice(i * dh.n_scenes, (i + 1) * dh.n_scenes)
        hh = _lib.Handle(dh); hh.set_weights(w); hh.set_scene_grids(grids_t.data_ptr(), gos[hs])
        er = eps_t.view(d.n_scenes, -1, d.L)[hs].reshape(-1, d.L).contiguous()
        halves.append(dict(h=hh, past=past_t[hs].contiguous(), fut=fut_t[hs].contiguous(), eps=er,
                           Y=torch.zeros((dh.R, d.T_pred, 2), device=dev), score=torch.zeros((dh.R,), device=dev)))
    return halves, PipelinedShardedIoc([ShardedIoc(x["h"], rank, world) for x in halves])


def agent_sharded_comm(sharded, halves, fence, nrep, world, t_pred):
    """Exposed communication of the agent-sharded IOC: the loop as it is, then the same loop with the collectives taken out (every
    step re-uses one gathered buffer: timing only)."""
    sent, recv = sharded.comm_bytes_per_step(world)

    def ioc_only():
        sharded.run([x["Y"] for x in halves], [x["score"] for x in halves])
    ioc_only(); fence()
    tc = time.perf_counter()
    for _ in range(nrep):
        ioc_only()
    fence()
    with_comm = (time.perf_counter() - tc) / nrep
    saved = [p.gather for p in sharded.parts]
    cache = {}
    for i, p in enumerate(sharded.parts):
        def stale(tn, i=i, g=saved[i]):
            key = (i, tuple(tn.shape))
            if key not in cache:
                cache[key] = g(tn)
            return cache[key]
        p.gather = stale
    ioc_only(); fence()
    tc = time.perf_counter()
    for _ in range(nrep):
        ioc_only()
    fence()
    no_comm = (time.perf_counter() - tc) / nrep
    for p, g in zip(sharded.parts, saved):
        p.gather = g
    out = {"bytes_sent_per_rank_per_ioc_step": sent, "bytes_received_per_rank_per_ioc_step": recv, "ioc_steps_per_pass": t_pred,
           "ioc_ms_with_collectives": with_comm * 1e3, "ioc_ms_collectives_removed": no_comm * 1e3,
           "exposed_comm_ms": max(0.0, (with_comm - no_comm) * 1e3),
           "note": "two micro-batches per rank: the all-gather of one runs on a communication stream while the other computes its step"}
    # the same pass over PEER buffers (desire_peer_*: regions mapped through hipIpc, one call per pass, a one-wave wait kernel between the
    # steps, no collective and no host in the step loop); the micro-batches run one after the other on the launch stream
    # Between real GPUs the mapped regions are reached over xGMI -- a path no box of this build ever had (one GPU per box).  The leg is ON by
    # default (VERDICT r04 missing 1: the day an 8-GPU node runs the plain command, the peer design must be in the line); what protects the line
    # is bench.py's watchdog around all multi-rank legs, which prints it without the leg that hung.  DESIRE_BENCH_PEER_LEG=0 switches it off.
    if world > 1 and os.environ.get("DESIRE_BENCH_PEER_LEG") == "0":
        out["peer_buffers"] = {"skipped": "DESIRE_BENCH_PEER_LEG=0"}
        return out
    try:
        import torch
        from desire_amd.dist import PeerShardedIoc
        peers = [PeerShardedIoc(x["h"], p.rank, world) for x, p in zip(halves, sharded.parts)]
        st = torch.cuda.current_stream().cuda_stream

        def peer_only():              # one stream, the micro-batches one after the other: every rank issues them in the same order, and a
            for pr, x in zip(peers, halves):      # pass's parked wait kernel can then never sit in front of work a peer is waiting for
                pr.run(x["Y"], x["score"], st)
        peer_only(); fence()
        tc = time.perf_counter()
        for _ in range(nrep):
            peer_only()
        fence()
        out["peer_buffers"] = {"ioc_ms": (time.perf_counter() - tc) / nrep * 1e3,
                               "note": "dist.PeerShardedIoc: hidden states read in place from the peers' exchange regions (hipIpc; xGMI between "
                                       "GPUs), progress counters instead of collectives, %d launches per pass enqueued at once" % (3 * t_pred + 5)}
        for pr in peers:
            pr.close()
    except Exception as exc:                                  # noqa: BLE001 -- the leg must not cost the line
        out["peer_buffers"] = {"error": repr(exc)[:300]}
    return out


def multi_rank_legs(out, a, d, w, grids_t, rank, world, dev, stream, fence, emit):
    """Everything `bench.py --gpus N` (N > 1, default flags) measures besides the scene-sharded headline, OUTSIDE its timed region, into the same
    line: (1) `agent_sharded`: north_star's partitioning on configs[1] dims -- d.mno slots per rank of (d.mno * N)-agent scenes, the IOC over
    RCCL all-gathers (exposed-communication ms) and over peer buffers; (2) `alt.config3`: BASELINE configs[3]'s shape -- 32 scenes x 64 agents
    sharded 64/N per rank, K = 50, H = 256, fp32 and split operands; (3) `alt.config4_train`: BASELINE configs[4] -- 512 agents per step over the N
    ranks, forward + backward + flat-gradient all-reduce + clip + Adam, with the all-reduce timed alone.  `out` is rank 0's line (None elsewhere);
    every leg is guarded, and ONE watchdog bounds them all: if a collective or a peer wait hangs, `emit()` prints the line with the legs that did
    finish and the process exits."""
    import threading
    import torch
    import torch.distributed as dist
    from desire_amd import _lib
    from desire_amd.dist import allreduce_mean_
    from desire_amd.spec import Dims, init_weights
    from desire_amd.synth import make_case
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    done = threading.Event()
    state = {"leg": "agent_sharded"}

    def bail():
        if done.is_set():
            return
        if rank == 0:
            out.setdefault("alt", {})
            tgt = out if state["leg"] == "agent_sharded" else out["alt"]
            tgt[state["leg"]] = {"error": "timed out (watchdog %d s): a collective or a peer wait did not complete; the scene-sharded headline above is unaffected" % budget}
            emit()
        os._exit(0)
    budget = int(os.environ.get("DESIRE_BENCH_LEG_TIMEOUT", "240"))
    wd = threading.Timer(float(budget), bail)
    wd.daemon = True
    wd.start()
    # a rank that DIES in one of these legs (a GPU fault across xGMI is an abort, not an exception) makes the launcher send SIGTERM to the others:
    # rank 0 then still owes the driver its line.  The main thread may sit in a C call (a synchronise that will never return), where a Python signal
    # handler cannot run, so the signal is routed to a wake-up pipe and a helper thread prints the line with the legs that did finish.
    if rank == 0:
        import signal
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        try:
            signal.set_wakeup_fd(wfd)
            signal.signal(signal.SIGTERM, lambda *_: None)

            def on_term():
                os.read(rfd, 1)
                if done.is_set():
                    return
                out.setdefault("alt", {})
                tgt = out if state["leg"] == "agent_sharded" else out["alt"]
                tgt[state["leg"]] = {"error": "terminated by the launcher while this leg ran (another rank died); the scene-sharded headline above is unaffected"}
                emit()
                os._exit(0)
            threading.Thread(target=on_term, daemon=True).start()
        except ValueError:                                        # not the main thread (bench.py imported and driven from elsewhere): no handler
            pass
    fault_at = os.environ.get("DESIRE_BENCH_FAULT_AT")              # test hook: the LAST rank kills itself when that leg starts (tests/test_gpu_bench.py)

    def maybe_fault(name):
        if fault_at == name and rank == world - 1:
            os.kill(os.getpid(), 9)

    broken = {"v": False}

    def leg_fence():
        """barrier + synchronise between legs; a peer that has died turns it into an exception: no further legs then, the line goes out as it is"""
        try:
            fence()
        except Exception:                                         # noqa: BLE001
            broken["v"] = True

    def put(name, val, top=False):
        if rank == 0:
            if top:
                out[name] = val
            else:
                out.setdefault("alt", {})[name] = val

    # ---- (1) agent-sharded IOC at configs[1] dims --------------------------------------------------------------------------------------
    if d.mno * world <= 256 and os.environ.get("DESIRE_BENCH_NO_AGENT_LEG") != "1":
        maybe_fault("agent_sharded")
        try:
            da = d.replace(n_scenes=32)
            past_a, fut_a, eps_a, _, gos_a = make_case(da, seed=a.seed + 101 + rank, n_absent=0)
            halves, sharded = agent_sharded_setup(da, w, grids_t, gos_a, t(past_a), t(fut_a), t(eps_a), rank, world, dev)
            for x in halves:
                x["h"].encode(x["past"].data_ptr(), x["fut"].data_ptr(), stream)
                x["h"].sample(x["eps"].data_ptr(), x["Y"].data_ptr(), stream)
            leg = agent_sharded_comm(sharded, halves, fence, 3, world, d.T_pred)
            ok = all(bool(torch.isfinite(x["Y"]).all()) for x in halves)
            leg.update({"windows_per_gpu": da.n_scenes, "rows_per_gpu": da.R, "agents_per_scene_over_all_ranks": d.mno * world, "finite": ok,
                        "samples_per_s_ioc_only": da.R * world / (leg["ioc_ms_with_collectives"] * 1e-3)})
            put("agent_sharded", leg, top=True)
            for x in halves:
                x["h"].close()
        except Exception as exc:                                  # noqa: BLE001 -- the headline must survive a failure of an extra leg
            put("agent_sharded", {"error": repr(exc)[:300]}, top=True)
        leg_fence()

    # ---- (2) BASELINE configs[3]: 32 scenes x 64 agents, K = 50, H = 256, agents sharded over the ranks --------------------------------------
    state["leg"] = "config3"
    if not broken["v"] and 64 % world == 0 and os.environ.get("DESIRE_BENCH_NO_CONFIG3_LEG") != "1":
        maybe_fault("config3")
        res = {}
        try:
            m_loc = 64 // world
            for tag, mode in (("fp32", 0), ("split_bf16x3", 2)):
                d3 = Dims(n_scenes=32, mno=m_loc, K=50, T_obs=8, T_pred=40, H=256, L=128, n_grids=1, grid_size=4, nb_w=0.15, nb_h=0.15,
                          sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1, bf16=mode)
                w3 = init_weights(d3, a.seed)
                # the SAME 64-agent scenes on every rank (seed without the rank), this rank's block of slots cut out of them
                dfull = d3.replace(mno=64)
                past3, fut3, _, grids3, gos3 = make_case(dfull, seed=a.seed + 301, n_absent=0)
                sl = slice(rank * m_loc, (rank + 1) * m_loc)
                eps3 = np.random.default_rng(a.seed + 302 + rank).standard_normal((d3.R, d3.L)).astype(np.float32)
                g3 = t(grids3)
                halves, sharded = agent_sharded_setup(d3, w3, g3, gos3, t(past3[:, :, sl]), t(fut3[:, :, sl]), t(eps3), rank, world, dev)

                def whole():
                    for x in halves:
                        x["h"].encode(x["past"].data_ptr(), x["fut"].data_ptr(), stream)
                        x["h"].sample(x["eps"].data_ptr(), x["Y"].data_ptr(), stream)
                    sharded.run([x["Y"] for x in halves], [x["score"] for x in halves])
                whole(); fence()
                tc = time.perf_counter()
                for _ in range(3):
                    whole()
                fence()
                wdt = (time.perf_counter() - tc) / 3
                leg = agent_sharded_comm(sharded, halves, fence, 3, world, d3.T_pred)
                ok = all(bool(torch.isfinite(x["Y"]).all()) and bool(torch.isfinite(x["score"]).all()) for x in halves)
                res[tag] = {"ms_per_step": wdt * 1e3, "value": d3.R * world / wdt, "unit": "samples/s (all ranks)", "finite": ok,
                            "ioc_ms_with_collectives": leg["ioc_ms_with_collectives"], "ioc_ms_collectives_removed": leg["ioc_ms_collectives_removed"],
                            "exposed_comm_ms": leg["exposed_comm_ms"], "peer_buffers": leg.get("peer_buffers"),
                            "bytes_received_per_rank_per_ioc_step": leg["bytes_received_per_rank_per_ioc_step"]}
                for x in halves:
                    x["h"].close()
                del halves, sharded
                torch.cuda.empty_cache()
            res["shape"] = "BASELINE configs[3]: 32 scenes x 64 agents (2048 agents), K=50, H=256, T 8/40; %d slots per rank over %d ranks, %d rows per rank" % (m_loc, world, 32 * 50 * m_loc)
        except Exception as exc:                                  # noqa: BLE001
            res["error"] = repr(exc)[:300]
        put("config3", res)
        leg_fence()

    # ---- (3) BASELINE configs[4]: training, 512 agents per step over the ranks -----------------------------------------------------------------
    state["leg"] = "config4_train"
    if not broken["v"] and os.environ.get("DESIRE_BENCH_NO_CONFIG4_LEG") != "1":
        maybe_fault("config4_train")
        res = {}
        try:
            n_win = max(1, 16 // world)                               # 16 windows x 32 slots = 512 agents per step
            for tag, mode in (("fp32", 0), ("split_bf16x3", 2)):
                d4 = d.replace(n_scenes=n_win, bf16=mode, flags=0, ioc_form=0)
                past4, fut4, eps4, _, gos4 = make_case(d4, seed=a.seed + 401 + rank, n_absent=0)
                h4 = _lib.Handle(d4)
                h4.set_weights(w)
                h4.set_scene_grids(grids_t.data_ptr(), gos4)
                h4.set_training(True)
                gflat = h4.grad_tensor()
                p4, f4, e4 = t(past4), t(fut4), t(eps4)
                Y4 = torch.zeros((d4.R, d4.T_pred, 2), device=dev); s4 = torch.zeros((d4.R,), device=dev)

                def one():
                    h4.forward(p4.data_ptr(), f4.data_ptr(), e4.data_ptr(), Y4.data_ptr(), s4.data_ptr(), stream)
                    h4.backward(p4.data_ptr(), f4.data_ptr(), e4.data_ptr(), stream)
                    allreduce_mean_(gflat)
                    h4.clip_grads(10.0, stream=stream)
                    h4.adam_step(1e-4, stream=stream)
                for _ in range(2):
                    one()
                fence()
                tc = time.perf_counter()
                for _ in range(4):
                    one()
                fence()
                sdt = (time.perf_counter() - tc) / 4
                tc = time.perf_counter()
                for _ in range(4):
                    allreduce_mean_(gflat)
                fence()
                adt = (time.perf_counter() - tc) / 4
                terms = h4.train_loss(f4.data_ptr(), stream)
                res[tag] = {"ms_per_step": sdt * 1e3, "allreduce_ms": adt * 1e3, "allreduce_bytes": int(gflat.numel()) * 4,
                            "value": d4.R * world / sdt, "unit": "samples/s trained (all ranks)", "loss": float(terms["loss"]),
                            "finite": all(np.isfinite(float(v)) for v in terms.values())}
                h4.close()
                del h4, gflat
                torch.cuda.empty_cache()
            res["shape"] = "BASELINE configs[4]: 512 agents per step = 16 windows x 32 slots, K=20, T 8/40, H=128; %d window(s) per rank over %d ranks; forward + backward + flat-gradient all-reduce + clip + Adam + repack" % (n_win, world)
        except Exception as exc:                                  # noqa: BLE001
            res["error"] = repr(exc)[:300]
        put("config4_train", res)
        leg_fence()
    done.set()
    wd.cancel()
