"""Agent-sharded IOC over peer buffers (desire_peer_* / dist.PeerShardedIoc): no collective and no host in the step loop.  Real
PROCESSES -- 2, 4 and 8 of them, all on this box's one GPU -- exchange their regions through hipIpc handles and must reproduce the
host-stepped ShardedIoc loop (gloo all-gathers) bit for bit.  (Ranks as several handles of ONE process on one device are not a
supported arrangement: a pass parks a wait kernel on its stream, and the runtime multiplexes a process's streams onto four hardware
queues -- two ranks that land on the same queue wait for each other until the time-out.  Seen while writing this test.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims
from tests.test_gpu_sharded_ioc import _setup_rank

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_rank_peer_pass_equals_ioc_refine():
    import torch
    from desire_amd.dist import PeerShardedIoc
    d = small_dims(K=3)
    w = init_weights(d, 23)
    past, fut, eps, grids, gos = make_case(d, seed=24, n_absent=2)
    h, keep = _setup_rank(torch, d, w, past, fut, eps, grids, gos)
    Y0 = h.read_buffer("Y0", (d.R, d.T_pred, 2))
    Ya = torch.as_tensor(Y0.copy(), device="cuda"); sa = torch.zeros(d.R, device="cuda")
    h.ioc_refine(Ya.data_ptr(), sa.data_ptr())
    Yb = torch.as_tensor(Y0.copy(), device="cuda"); sb = torch.zeros(d.R, device="cuda")
    PeerShardedIoc(h, 0, 1).run(Yb, sb, sync=True)
    assert not h.peer_timed_out()
    torch.cuda.synchronize()
    assert float((Ya - Yb).abs().max()) < 2e-6 and float((sa - sb).abs().max()) < 2e-5


@pytest.mark.parametrize("case,world", [("small", 2), ("two_passes", 2), ("h64", 2), ("sparse_mno64", 2), ("small", 4), ("config3", 8)])
def test_processes_exchange_through_hipipc(case, world):
    """Real processes, one per rank, all on this box's single GPU (RCCL refuses that; hipIpc does not): regions exported as
    hipIpcMemHandles, handed round with one all_gather_object over gloo, mapped with hipIpcOpenMemHandle.  config3 = BASELINE
    configs[3]'s sharding: 64-agent scenes over EIGHT ranks, K = 50, H = 256, T_pred = 40."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29541 + world), os.path.join(ROOT, "tests", "peer_worker.py"), case]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert p.stdout.count("peer == gathered: True") == world, p.stdout[-2000:]


def test_peer_api_refuses_out_of_order_use():
    import torch
    from desire_amd import _lib
    d = small_dims(K=2)
    h = _lib.Handle(d)
    h.set_weights(init_weights(d, 1))
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); s = torch.zeros(d.R, device="cuda")
    with pytest.raises(_lib.DesireError):
        h.ioc_peer_pass(Y.data_ptr(), s.data_ptr())                       # nothing exported / opened
    with pytest.raises(_lib.DesireError):
        h.peer_open(0, 2, 1, b"\0" * 64)                                  # before export
    h.peer_export()
    with pytest.raises(_lib.DesireError):
        h.peer_open(0, 9, 0)                                              # more than 8 ranks
    h.peer_open(0, 2, 0)
    with pytest.raises(_lib.DesireError):
        h.ioc_peer_pass(Y.data_ptr(), s.data_ptr())                       # rank 1 not attached yet
    h.peer_close()


def test_peer_pass_replays_from_a_hipgraph():
    """A pass is nothing but kernel launches (epoch bump, waits, publish, steps, counter bumps, finish): captured once with
    desire_graph_begin / _end and replayed, it gives the same bits as the direct call -- the epoch lives in device memory, so every
    replay is a new pass of the protocol."""
    import torch
    from desire_amd.dist import PeerShardedIoc
    d = small_dims(K=3, T_pred=9)
    w = init_weights(d, 23)
    past, fut, eps, grids, gos = make_case(d, seed=24, n_absent=2)
    h, keep = _setup_rank(torch, d, w, past, fut, eps, grids, gos)
    Y0 = torch.as_tensor(h.read_buffer("Y0", (d.R, d.T_pred, 2)).copy(), device="cuda")
    peer = PeerShardedIoc(h, 0, 1)
    Ya, sa = Y0.clone(), torch.zeros(d.R, device="cuda")
    peer.run(Ya, sa)                                                       # direct (also the warm-up: lazy allocations)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    Yb, sb = Y0.clone(), torch.zeros(d.R, device="cuda")
    torch.cuda.synchronize()
    h.graph_begin(side.cuda_stream)
    peer.run(Yb, sb, side.cuda_stream)
    gid = h.graph_end(side.cuda_stream)
    for _ in range(3):
        Yb.copy_(Y0); sb.zero_()
        torch.cuda.synchronize()
        h.graph_launch(gid, side.cuda_stream)
        side.synchronize()
        assert torch.equal(Yb, Ya) and torch.equal(sb, sa)
    peer.close()
