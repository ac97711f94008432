"""Few tiles per launch (a handful of windows): k_ioc's bin-split form -- several workgroups per 32-row tile, each contracting its share
of the social bins, partial e_r sums exchanged through global memory once per step (kernels_rnn.hip: k_ioc NSPL, api.hip:
ioc_bin_split).  Checked against the plain form of the same kernel (dims.ioc_split = 1 keeps one workgroup per tile), against
the oracle, and for run-to-run determinism."""
import numpy as np
import pytest

from desire_amd.spec import Dims, init_weights
from tests.helpers import make_case, small_dims
from tests.test_gpu_parity import oracle_forward, run_gpu, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(n_scenes=1, mno=32, K=20, T_obs=8, T_pred=40, n_grids=1, nb_w=0.2, nb_h=0.2),     # literal BASELINE configs[1]: one window
    dict(mno=16, n_scenes=3, K=5),                                                         # ragged last tile
    dict(H=64, T_pred=7, K=3),
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2),                                            # 36 bins
    dict(bin_mode=1, grid_size=4, nb_w=0.45, nb_h=0.04, K=2),                              # log-polar bins
    dict(nb_w=0.04, nb_h=0.04, K=2),                                                       # sparse windows: some members get no bin at all
])
def test_bin_split_matches_the_plain_form(torch_cuda, kw):
    d = Dims(sx=1 / 1400.0, sy=1 / 1100.0, **kw) if "n_grids" in kw else small_dims(**kw)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6, n_absent=min(3, d.mno - 1))
    _, Yp, sp = run_gpu(torch_cuda, d.replace(ioc_split=1), w, past, fut, eps, grids, gos)
    outs = {}
    for cap in (2, 3, 4):
        d = d.replace(ioc_split=cap)
        _, Y, s = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
        assert np.abs(Y - Yp).max() < 2e-6, (cap, np.abs(Y - Yp).max())
        assert np.abs(s - sp).max() < 2e-5 * max(1.0, np.abs(sp).max())
        _, Y2, s2 = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
        np.testing.assert_array_equal(Y, Y2)                 # fixed member order: deterministic
        np.testing.assert_array_equal(s, s2)
        outs[cap] = Y
    assert any(not np.array_equal(outs[c], Yp) for c in outs)         # (the split really ran: partial sums group differently)


def test_bin_split_against_the_oracle(torch_cuda):
    d = small_dims(T_pred=40, K=4)
    w = init_weights(d, 3)
    past, fut, eps, grids, gos = make_case(d, seed=4, n_absent=3)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    _, Y, s = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    assert np.abs(Y - ref["Y"]).max() < 1e-5
    assert np.abs(s - ref["score"]).max() < 5e-3


def test_two_passes_run_the_plain_form(torch_cuda):
    """A second refinement pass reads the first pass's refined positions, which only member 0 of a tile holds: iters > 1 never splits."""
    d = small_dims(iters=2, K=2)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6, n_absent=2)
    _, Ya, sa = run_gpu(torch_cuda, d.replace(ioc_split=1), w, past, fut, eps, grids, gos)
    _, Yb, sb = run_gpu(torch_cuda, d.replace(ioc_split=4), w, past, fut, eps, grids, gos)
    np.testing.assert_array_equal(Ya, Yb)
    np.testing.assert_array_equal(sa, sb)


def test_bin_split_forward_replays_from_a_hipgraph(torch_cuda):
    """One window (literal configs[1]) captured once and replayed: the arrival counters of the bin-split IOC are reset by a fill KERNEL in
    the captured sequence (a memset node was once seen to run out of order on replay: DESIGN.md 6a) -- a counter left at the previous
    pass's value would let every member read its peers' slots before they are written.  Thirty replays, bit-identical to the direct call."""
    import torch
    from desire_amd import _lib
    d = Dims(n_scenes=1, mno=32, K=20, T_obs=8, T_pred=40, n_grids=1, nb_w=0.2, nb_h=0.2, sx=1 / 1400.0, sy=1 / 1100.0)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6, n_absent=3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    h = _lib.Handle(d); h.set_weights(w); h.set_scene_grids(g.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
    side = torch.cuda.Stream(); sp = side.cuda_stream
    torch.cuda.synchronize()
    h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr(), sp)      # warm-up outside capture (lazy allocations)
    side.synchronize()
    Y_ref, s_ref = Y.clone(), sc.clone()
    h.graph_begin(sp)
    h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr(), sp)
    gid = h.graph_end(sp)
    for _ in range(30):
        Y.zero_(); sc.zero_()
        torch.cuda.synchronize()
        h.graph_launch(gid, sp)
        side.synchronize()
        assert torch.equal(Y, Y_ref) and torch.equal(sc, s_ref)
    assert float(Y_ref.abs().max()) > 0
    h.close()


def test_split_switched_off_restores_batch_size_invariance(torch_cuda):
    """dims.ioc_split = 1 (include/desire_hip.h): every launch runs the plain kernel, and a window's results are bit-identical whether it
    is run alone or inside a larger batch -- the invariance the bin-split regime (the default, ioc_split = 0) trades for latency.  The
    same switch flipped on a LIVE handle (desire_set_option) takes effect at the next call."""
    from desire_amd import _lib
    d = Dims(n_scenes=4, mno=32, K=4, T_obs=8, T_pred=12, n_grids=1, nb_w=0.2, nb_h=0.2, sx=1 / 1400.0, sy=1 / 1100.0, ioc_split=1)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6, n_absent=3)
    _, Y4, s4 = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    d1 = d.replace(n_scenes=1); r1 = d1.R
    h1, Y1, s1 = run_gpu(torch_cuda, d1, w, past[:1], fut[:1], eps[:r1], grids, gos[:1])
    assert np.array_equal(Y4[:r1], Y1) and np.array_equal(s4[:r1], s1)
    _, Y1s, s1s = run_gpu(torch_cuda, d1.replace(ioc_split=0), w, past[:1], fut[:1], eps[:r1], grids, gos[:1])
    assert np.abs(Y1s - Y1).max() < 2e-6                                            # the default regroups the partial sums of one window
    with pytest.raises(_lib.DesireError):
        h1.set_option("ioc_split", 9)
    with pytest.raises(_lib.DesireError):
        h1.set_option("no_such_switch", 1)
    h1.set_option("ioc_split", 0)
    assert h1.dims.ioc_split == 0
