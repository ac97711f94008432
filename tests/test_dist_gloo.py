"""N>1 path on CPU: world_size-2 gloo.  The compute itself needs a GPU; what is covered here is the
sharding + gather logic bench.py / DESIREModel use between ranks (windows block-sharded, results
all-gathered in global window order, max-over-ranks timing reduction)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from desire_amd.dist import gather_results, shard_batch, shard_windows


def test_shard_windows_covers_everything():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_windows(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        shard_windows(4, 2, 2)


def _worker(rank, world, port, n_windows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = [np.full((3, 2), i, np.float32) for i in range(n_windows)]
        mine = shard_batch(batch, rank, world)
        # stand-in for the per-window hot-path result: f(window) known in closed form
        local = torch.stack([torch.as_tensor(b) * 2 + 1 for b in mine]) if mine else torch.zeros((0, 3, 2))
        full = gather_results(local, n_windows)
        t = torch.tensor([0.5 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)            # bench.py's max-over-ranks step time
        dist.barrier()
        q.put((rank, full.numpy(), float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_windows", [5, 8])
def test_gloo_world2_gather_matches_single_process(n_windows):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_windows, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.stack([np.full((3, 2), i, np.float32) * 2 + 1 for i in range(n_windows)])
    for rank, full, tmax in res:
        np.testing.assert_array_equal(full, want)
        assert tmax == 1.5


# ---- data-parallel training step: each rank's gradient of ITS windows, one flat all-reduce, identical Adam update ----
def _flat(grads, names):
    return torch.cat([torch.as_tensor(np.asarray(grads[k], np.float32)).reshape(-1) for k in names])


def _shard_grads(rank, world, n_windows):
    """Oracle (torch autograd, CPU) gradient of the training loss on rank's block of windows."""
    from desire_amd.spec import init_weights
    from oracle import desire_torch as OT
    from tests.helpers import make_case, small_dims, to_oracle_layout
    lo, hi = shard_windows(n_windows, rank, world)
    d = small_dims(n_scenes=hi - lo, mno=4, K=2, T_obs=3, T_pred=3, n_grids=1, H=64, L=16)
    dfull = small_dims(n_scenes=n_windows, mno=4, K=2, T_obs=3, T_pred=3, n_grids=1, H=64, L=16)
    w = init_weights(dfull, 5)
    past, fut, eps, grids, gos = make_case(dfull, seed=6, n_absent=1)
    rows = slice(lo * d.K * d.mno, hi * d.K * d.mno)
    _, g = OT.loss_and_grads(to_oracle_layout(past[lo:hi]), to_oracle_layout(fut[lo:hi]), eps[rows], grids, gos[lo:hi], w, d)
    return w, g


def _train_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from desire_amd.dist import allreduce_mean_
        w, g = _shard_grads(rank, world, 2)
        names = sorted(w)
        flat = _flat(g, names)
        allreduce_mean_(flat)
        q.put((rank, flat.numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world2_gradient_mean_is_the_data_parallel_gradient():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0], res[1])                       # every rank applies the same update
    w, g0 = _shard_grads(0, 2, 2)
    _, g1 = _shard_grads(1, 2, 2)
    names = sorted(w)
    want = (_flat(g0, names) + _flat(g1, names)).numpy() / 2
    np.testing.assert_allclose(res[0], want, rtol=1e-6, atol=1e-9)
    assert np.abs(want).max() > 0


def test_allreduce_mean_is_a_noop_without_a_process_group():
    from desire_amd.dist import allreduce_mean_
    t = torch.arange(4.0)
    assert allreduce_mean_(t) is t and torch.equal(t, torch.arange(4.0))


# ---- the neighbour exchange of the agent-sharded IOC: rank-major stack of every rank's tensor ----
def _stack_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from desire_amd.dist import all_gather_stack
        h = torch.arange(6, dtype=torch.float32).reshape(3, 2) + 100 * rank          # "hidden states" of my 3 rows
        v = torch.tensor([1, 0, 1], dtype=torch.uint8) * (rank + 1)                  # presence flags travel as uint8
        q.put((rank, all_gather_stack(h).numpy(), all_gather_stack(v).numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world2_all_gather_stack_is_rank_major():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stack_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    base = np.arange(6, dtype=np.float32).reshape(3, 2)
    for rank, hs, vs in res:
        np.testing.assert_array_equal(hs, np.stack([base, base + 100]))
        np.testing.assert_array_equal(vs, np.stack([np.array([1, 0, 1], np.uint8), np.array([2, 0, 2], np.uint8)]))


def test_all_gather_stack_without_process_group_is_the_identity_stack():
    from desire_amd.dist import all_gather_stack
    t = torch.arange(4.0).reshape(2, 2)
    assert torch.equal(all_gather_stack(t), t[None])
