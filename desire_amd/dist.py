"""Multi-GPU sharding of the hot path: one process per GPU, windows (scenes) block-sharded.

Social pooling never crosses a window (zero-comm alternative), so the
data path needs no collective; the only communication is an optional all_gather of the finished
trajectories/scores.  Works with backend "nccl" (= RCCL over xGMI on ROCm) and with "gloo" on
CPU tensors (used by tests/test_dist_gloo.py, world_size 2)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def shard_windows(n_windows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of windows owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or n_windows < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_windows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch: Sequence, rank: int, world: int) -> List:
    lo, hi = shard_windows(len(batch), rank, world)
    return list(batch[lo:hi])


def gather_results(local, n_windows: int, group=None):
    """All-gather per-window results (first dim = local windows) into the global window order.
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_windows(n_windows, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("local shard has %d windows, expected %d" % (local.shape[0], hi - lo))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: hi - lo] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: h - l] for o, (l, h) in zip(out, sizes)], dim=0)


def allreduce_mean_(flat, group=None, force: bool = False):
    """In-place mean of ONE flat gradient tensor over the data-parallel ranks (no-op when torch.distributed is not
    initialised or world_size == 1; `force` issues the collective at world_size 1 too: tests/test_gpu_nccl.py runs RCCL on a one-GPU box).  Each rank's loss is a mean over ITS present agents, so this is the usual
    data-parallel estimate of the global-batch gradient.  RCCL over xGMI on GPU tensors, gloo on CPU tensors."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return flat
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.mul_(1.0 / world)
    return flat


# ---- agent-sharded IOC: the north_star's "RCCL all-gather only for the social-pooling neighbour exchange" -------------
def all_gather_stack(t, group=None, force: bool = False):
    """[...] on every rank -> [world, ...] on every rank (rank-major), one all_gather_into_tensor (RCCL on GPU tensors,
    gloo on CPU tensors); the identity stack when torch.distributed is not initialised or the world is one rank (`force`: the
    collective is issued at world_size 1 as well -- how a one-GPU box executes the RCCL path, tests/test_gpu_nccl.py)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return t.unsqueeze(0).contiguous()
    world = dist.get_world_size(group)
    t = t.contiguous()
    flat = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenated form: every backend takes it
    dist.all_gather_into_tensor(flat, t, group=group)
    return flat.view((world,) + tuple(t.shape))


class ShardedIoc:
    """IOC scoring / refinement when the AGENTS of every scene are block-sharded over the ranks (rank g owns slots
    [g*m_loc, (g+1)*m_loc) of every scene; the handle is created with mno = m_loc).  Encoders, CVAE and decoder are
    per-agent and run locally (Handle.encode / Handle.sample); here positions, last observed positions and presence
    flags are gathered once per pass and the hidden states once per step -- T_pred all-gathers of [R_loc, H] fp32 per
    pass, the only data-path collective of the whole framework.  `gather` defaults to all_gather_stack; tests inject a
    stand-in to run several virtual ranks in one process.

    Scene-sharding (dist.shard_windows) needs no collective at all and is what bench.py uses; this form exists for
    scenes that must be split across GPUs."""

    def __init__(self, handle, rank: int, nranks: int, gather=None, group=None):
        self.h, self.rank, self.nranks = handle, int(rank), int(nranks)
        self.gather = gather or (lambda t: all_gather_stack(t, group))

    def local_state(self):
        """(Hx rows [R_loc, H], p_last [A_loc, 2], valid [A_loc] uint8, Y0 [R_loc, T, 2]) views / copies on the device."""
        d = self.h.dims
        H = max(d.H, 64)                                       # physical width of the recurrent tile (d_dim 16 / 32 run zero-padded)
        HxHy = self.h.device_tensor("HxHy")[: d.A * 2 * H].view(d.A, 2 * H)
        Hx = HxHy[:, : H].reshape(d.n_scenes, 1, d.mno, H).expand(d.n_scenes, d.K, d.mno, H).reshape(d.R, H).contiguous()
        p_last = self.h.device_tensor("p_last")[: d.A * 2].view(d.A, 2)
        valid = self.h.device_tensor("valid", "|u1")[: d.A]
        Y0 = self.h.device_tensor("Y0")[: d.R * d.T_pred * 2].view(d.R, d.T_pred, 2)
        return Hx, p_last, valid, Y0

    def prepare(self, Y_loc):
        """Gathers what is fixed during a pass; returns the per-pass context."""
        import torch
        d = self.h.dims
        Hx, p_last, valid, _ = self.local_state()
        return {"plast_all": self.gather(p_last.contiguous()), "valid_all": self.gather(valid.contiguous()),
                "Yall": self.gather(Y_loc.contiguous()), "hst": Hx.clone(),
                "score": torch.zeros(d.R, device=Y_loc.device, dtype=torch.float32)}

    def step(self, ctx, t: int, Hall, stream: int = 0):
        self.h.ioc_step(t, self.rank, self.nranks, ctx["Yall"].data_ptr(), ctx["plast_all"].data_ptr(), ctx["valid_all"].data_ptr(),
                        Hall.data_ptr(), ctx["hst"].data_ptr(), ctx["score"].data_ptr(), stream)

    def finish(self, ctx, Y_loc, score_loc, stream: int = 0):
        self.h.ioc_finish(ctx["hst"].data_ptr(), ctx["score"].data_ptr(), Y_loc.data_ptr(), score_loc.data_ptr(), stream)

    def run(self, Y_loc, score_loc):
        """Y_loc [R_loc, T, 2] (in: decoded, out: refined), score_loc [R_loc]; all d.iters passes."""
        import torch
        d = self.h.dims
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(d.iters):
            ctx = self.prepare(Y_loc)
            for t in range(d.T_pred):
                Hall = self.gather(ctx["hst"])                      # the neighbour exchange
                self.step(ctx, t, Hall, stream)
            self.finish(ctx, Y_loc, score_loc, stream)
        return Y_loc, score_loc


class PeerShardedIoc:
    """Agent-sharded IOC over PEER buffers (include/desire_hip.h: desire_peer_*): the ranks map each other's exchange regions once
    (hipIpc; xGMI between GPUs) and a pass is ONE call that enqueues every step -- the step kernels read the neighbours' hidden states in
    place and a one-wave kernel waits on the peers' progress counters, so there is no collective and no host round trip per step (the
    ShardedIoc loop above issues T_pred all-gathers from Python).  Bit-identical to ShardedIoc.  `exchange(obj) -> list` hands the 64-byte
    handles round: default torch.distributed.all_gather_object (any backend); `barrier()` default torch.distributed.barrier."""

    def __init__(self, handle, rank: int, nranks: int, exchange=None, barrier=None, group=None):
        import torch.distributed as dist
        self.h, self.rank, self.nranks = handle, int(rank), int(nranks)
        if handle.dims.mno * nranks > 256:
            raise ValueError("agent-sharded IOC: at most 256 agents per scene over all ranks")

        def _exchange(obj):
            out = [None] * nranks
            dist.all_gather_object(out, obj, group=group)
            return out
        mine = handle.peer_export()
        handles = (exchange or _exchange)(mine) if nranks > 1 else [mine]
        for peer in range(nranks):
            handle.peer_open(self.rank, self.nranks, peer, handles[peer])
        if nranks > 1:
            (barrier or (lambda: dist.barrier(group=group)))()       # every rank has mapped every region before anybody publishes

    def run(self, Y_loc, score_loc, stream: int = 0, sync: bool = False):
        """Y_loc [R_loc, T, 2] (in: decoded, out: refined), score_loc [R_loc]; all dims.iters passes, stream-ordered, no host sync.
        sync=True: wait for the pass and raise if one of its bounded peer waits gave up (desire_peer_status) -- without it a timed-out pass
        is only reported by the NEXT run."""
        import torch
        st = stream or torch.cuda.current_stream().cuda_stream
        self.h.ioc_peer_pass(Y_loc.data_ptr(), score_loc.data_ptr(), st)
        if sync:
            torch.cuda.synchronize()
            if self.h.peer_timed_out():
                raise RuntimeError("peer-buffer IOC pass: a peer never arrived (bounded wait gave up); Y / score of this pass are invalid")
        return Y_loc, score_loc

    def close(self) -> None:
        self.h.peer_close()


class PipelinedShardedIoc:
    """The same agent-sharded IOC with the per-step neighbour all-gather HIDDEN behind compute ("overlap
    step-t all-gather with ... the local rows").  The rank's scenes are split into micro-batches (one ShardedIoc / handle each,
    normally two); their steps alternate on the compute stream while the all-gathers run on a separate communication stream:

        comm   :  gather A(t)   gather B(t)   gather A(t+1)   gather B(t+1) ...
        compute:               step A(t)     step B(t)        step A(t+1)  ...

    gather m(t+1) waits (event) for step m(t) and runs while the OTHER micro-batch computes; step m(t) waits for gather m(t).
    A step is data-dependent on its own gather (social pooling needs every neighbour's h_{t-1}), so a single batch cannot
    overlap with itself -- two independent halves can, with results bit-identical to ShardedIoc.run.  Per rank and step the
    collective moves R_loc * H * 4 bytes in and world times that out (configs[3]: 13.1 MB, about 86 us on one xGMI link per
    peer); a micro-batch's step kernel at that shape runs several hundred microseconds, so the transfer is covered."""

    def __init__(self, parts):
        import torch
        self.parts = list(parts)
        self.comm = torch.cuda.Stream()

    def run(self, Ys, scores):
        import torch
        compute = torch.cuda.current_stream()
        d = self.parts[0].h.dims
        for _ in range(d.iters):
            ctxs = [p.prepare(Y) for p, Y in zip(self.parts, Ys)]      # per-pass gathers (positions, flags), on the compute stream
            ready = []
            for _m in self.parts:
                e = torch.cuda.Event(); e.record(compute); ready.append(e)
            for t in range(d.T_pred):
                got = []
                for m, p in enumerate(self.parts):
                    with torch.cuda.stream(self.comm):
                        self.comm.wait_event(ready[m])                  # h_{t-1} of micro-batch m is final
                        Hall = p.gather(ctxs[m]["hst"])                 # RCCL all-gather (torch.distributed) on the comm stream
                        Hall.record_stream(compute)
                        e = torch.cuda.Event(); e.record(self.comm)
                    got.append((Hall, e))
                for m, p in enumerate(self.parts):
                    compute.wait_event(got[m][1])
                    p.step(ctxs[m], t, got[m][0], compute.cuda_stream)
                    e = torch.cuda.Event(); e.record(compute); ready[m] = e
            for p, c, Y, sc in zip(self.parts, ctxs, Ys, scores):
                p.finish(c, Y, sc, compute.cuda_stream)
        return Ys, scores

    def comm_bytes_per_step(self, world: int):
        """(sent, received) bytes per rank per IOC step: the hidden states of the local rows out, everybody's in."""
        n = 0
        for p in self.parts:
            d = p.h.dims
            n += d.R * max(d.H, 64) * 4
        return n, n * world
