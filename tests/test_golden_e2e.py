"""Committed build-oracle goldens on real SDD inputs (tests/golden/e2e_cfg{0,1}.npz, made by
tests/golden/make_e2e_golden.py): BASELINE configs[0] (4 agents, K=1, T=8/12) and configs[1]
(32 slots, K=20, T=8/40, H=128).  CPU: the oracle still reproduces them (drift guard).
GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

from desire_amd.spec import Dims, init_weights

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(tag):
    g = np.load(os.path.join(HERE, f"e2e_{tag}.npz"))
    n, mno, K, To, Tp, H, L = (int(v) for v in g["dims"])
    sx, sy, nbw, nbh = (float(v) for v in g["scale"])
    d = Dims(n_scenes=n, mno=mno, K=K, T_obs=To, T_pred=Tp, H=H, L=L, sx=sx, sy=sy, nb_w=nbw, nb_h=nbh, n_grids=1)
    seed = int(g["seed"])
    rng = np.random.default_rng(seed)
    eps = rng.standard_normal((d.R, d.L)).astype(np.float32)
    grids = rng.uniform(-1, 1, (d.n_grids, d.Gh, d.Gw, d.C)).astype(np.float32)
    return d, g, eps, grids, np.zeros(n, np.int32), init_weights(d, seed)


@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_oracle_reproduces_goldens(tag):
    from oracle import desire_oracle as O
    d, g, eps, grids, gos, w = load_case(tag)
    if tag == "cfg1":
        d = d.replace(K=2)                      # keep the CPU suite fast: first 2 of the 20 draws
        eps = eps.reshape(1, 20, d.mno, d.L)[:, :2].reshape(-1, d.L)
    tr = lambda x: np.ascontiguousarray(x.transpose(1, 0, 2, 3).reshape(x.shape[1], -1, 3))
    ref = O.forward(tr(g["past"]), tr(g["fut"]), eps, grids, gos, w, d)
    np.testing.assert_allclose(ref["Hx"], g["Hx"], atol=1e-6)
    np.testing.assert_allclose(ref["Y0"], g["Y0"][: d.R], atol=1e-5)
    np.testing.assert_allclose(ref["Y"], g["Y"][: d.R], atol=1e-4)
    np.testing.assert_allclose(ref["score"], g["score"][: d.R], atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_hip_reproduces_goldens(tag):
    import torch
    from desire_amd import _lib
    d, g, eps, grids, gos, w = load_case(tag)
    h = _lib.Handle(d)
    h.set_weights(w)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past, fut, eps_t, grids_t = t(g["past"]), t(g["fut"]), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    h.forward(past.data_ptr(), fut.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.abs(h.read_buffer("Hx", (d.A, d.H)) - g["Hx"]).max() < 2e-4
    assert np.abs(h.read_buffer("z_mean", (d.A, d.L)) - g["z_mean"]).max() < 2e-4
    assert np.abs(h.read_buffer("Y0", (d.R, d.T_pred, 2)) - g["Y0"]).max() < 1e-3
    valid = np.repeat((g["past"][:, -1, :, 0] != 0)[:, None, :], d.K, axis=1).reshape(-1)
    if float(g["bin_margin"]) > 1e-5:             # no pair close enough to a bin edge to flip on a 1e-6 difference
        assert np.abs(Y.cpu().numpy() - g["Y"])[valid].max() < 1e-3
        assert np.abs(score.cpu().numpy() - g["score"])[valid].max() < 5e-3
    # IOC on the golden decoder output: bins identical by construction, every row compared
    Y.copy_(t(g["Y0"]))
    h.ioc_refine(Y.data_ptr(), score.data_ptr(), s)
    torch.cuda.synchronize()
    assert np.abs(Y.cpu().numpy() - g["Y"]).max() < 1e-3
    assert np.abs(score.cpu().numpy() - g["score"]).max() < 5e-3


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cfg0", "cfg1"])
def test_unanchored_end_to_end_statistic_on_the_sdd_goldens(tag):
    """every IOC parity check above re-anchors on the oracle's Y0, and the un-anchored compare is asserted only
    when no pair sits near a bin edge.  This bounds, by a TEST, how often the full HIP chain (its own Y0 -> its own IOC pass) lands a
    row elsewhere than the oracle's chain on the real-SDD goldens: the fraction of present rows whose refined trajectory differs by
    more than 1e-3 anywhere must stay under 1 % (a 1e-7 difference in Y0 can move a neighbour across a bin edge or a position
    across a scene cell; DESIGN.md 4-split measured 0.3 % of rows for 1e-7 perturbations on the dense bench batch), and the rows
    that do NOT flip agree to the usual 1e-3."""
    import torch
    from desire_amd import _lib
    d, g, eps, grids, gos, w = load_case(tag)
    h = _lib.Handle(d)
    h.set_weights(w)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past, fut, eps_t, grids_t = t(g["past"]), t(g["fut"]), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    h.forward(past.data_ptr(), fut.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    valid = np.repeat((g["past"][:, -1, :, 0] != 0)[:, None, :], d.K, axis=1).reshape(-1)
    err = np.abs(Y.cpu().numpy() - g["Y"]).reshape(d.R, -1).max(1)[valid]
    flipped = err > 1e-3
    frac = float(flipped.mean())
    print("%s: %d present rows, %d differ by > 1e-3 un-anchored (%.3f %%); others max %.2e" % (tag, valid.sum(), flipped.sum(), 100 * frac,
                                                                                                 err[~flipped].max() if (~flipped).any() else 0.0))
    assert frac < 0.01, (tag, frac)
    assert err[~flipped].max() < 1e-3
    serr = np.abs(score.cpu().numpy() - g["score"])[valid][~flipped]
    assert serr.max() < 5e-3
