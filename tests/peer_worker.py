"""Worker of tests/test_gpu_peer_ioc.py::test_two_processes_exchange_through_hipipc: launched 2 / 4 / 8 times by torch.distributed.run (gloo
rendezvous, BOTH ranks on cuda:0 -- RCCL refuses two ranks on one device, hipIpc does not).  Each rank owns its block of the agent slots of
every scene, runs the per-agent stages locally, then refines (a) with the host-stepped ShardedIoc loop over gloo all-gathers and (b)
with PeerShardedIoc -- regions mapped through hipIpcOpenMemHandle, one call per pass, no collective.  Both must agree bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from desire_amd import _lib
    from desire_amd.dist import PeerShardedIoc, ShardedIoc
    from desire_amd.spec import init_weights
    from tests.helpers import make_case, small_dims
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    case = sys.argv[1] if len(sys.argv) > 1 else "small"
    if case == "config3":      # BASELINE configs[3] at its per-scene shape: 64-agent scenes over the ranks, K = 50, H = 256, T_pred = 40 (two scenes)
        d = small_dims(mno=64, n_scenes=2, K=50, H=256, L=128, T_obs=8, T_pred=40, n_grids=1, nb_w=0.3, nb_h=0.3)
    elif case == "h64":
        d = small_dims(mno=32, n_scenes=2, K=3, H=64, T_pred=7, n_grids=1)
    elif case == "sparse_mno64":
        d = small_dims(mno=64, n_scenes=1, K=2, n_grids=1, nb_w=0.04, nb_h=0.04)
    else:
        d = small_dims(mno=32, n_scenes=3, K=3, T_pred=9, n_grids=1, iters=2 if case == "two_passes" else 1)
    m_loc = d.mno // world
    w = init_weights(d, 21)
    past, fut, eps, grids, gos = make_case(d, seed=22, n_absent=3)
    sl = slice(rank * m_loc, (rank + 1) * m_loc)
    d_loc = d.replace(mno=m_loc)
    epsr = eps.reshape(d.n_scenes, d.K, d.mno, d.L)[:, :, sl].reshape(-1, d.L)
    h = _lib.Handle(d_loc)
    h.set_weights(w)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    keep = dict(past=t(past[:, :, sl]), fut=t(fut[:, :, sl]), eps=t(epsr), grids=t(grids))
    h.set_scene_grids(keep["grids"].data_ptr(), gos)
    h.encode(keep["past"].data_ptr(), keep["fut"].data_ptr())
    Y0 = torch.zeros((d_loc.R, d.T_pred, 2), device="cuda")
    h.sample(keep["eps"].data_ptr(), Y0.data_ptr())
    torch.cuda.synchronize()
    Ya, sa = Y0.clone(), torch.zeros(d_loc.R, device="cuda")
    ShardedIoc(h, rank, world).run(Ya, sa)                                # T_pred all-gathers per pass, issued from Python
    torch.cuda.synchronize()
    peer = PeerShardedIoc(h, rank, world)
    outs = []
    for _ in range(3):                                                    # three passes in a row: the epoch / parity protocol across passes
        Yb, sb = Y0.clone(), torch.zeros(d_loc.R, device="cuda")
        peer.run(Yb, sb)
        outs.append((Yb, sb))
    torch.cuda.synchronize()
    assert not h.peer_timed_out()                                         # desire_peer_status: the passes this rank has waited for all completed
    Yc, sc = Y0.clone(), torch.zeros(d_loc.R, device="cuda")
    peer.run(Yc, sc, sync=True)                                           # the checked form: raises if a bounded wait had given up
    outs.append((Yc, sc))
    dist.barrier()
    ok = all(torch.equal(Yb, Ya) and torch.equal(sb, sa) for Yb, sb in outs) and bool(torch.isfinite(Ya).all()) and float((Ya - Y0).abs().max()) > 0
    peer.close()
    print("rank %d peer == gathered: %s (max |dY| %.3e)" % (rank, ok, float((Ya - Y0).abs().max())), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
