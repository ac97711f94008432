"""bf16-operand IOC kernel (dims.bf16 = 1, BASELINE configs[2]) against the oracle.

Two references: the oracle with the SAME operand rounding (bf16_round where values enter a matrix product) pins the
kernel's logic -- the register-chained pooling, the chain-order weight packs, the transposed h image -- and the plain
fp32 oracle bounds what bf16 operands cost in accuracy.  Tolerances are stated per assertion."""
import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout
from tests.test_gpu_parity import oracle_forward, run_gpu, torch_cuda  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kw", [
    dict(),                                              # 32-agent groups, H=128
    dict(mno=16, n_scenes=3, K=5),                       # two groups per 32-row block, ragged last tile
    dict(mno=64, n_scenes=1, K=2, n_grids=1),            # one group spans both row blocks of the tile
    dict(H=64, T_pred=7, K=3),
    dict(H=256, K=3, n_scenes=1, n_grids=1, T_pred=10),
    dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2),
    dict(mno=8, n_scenes=5, K=3),
    dict(mno=1, n_scenes=3, K=2, n_absent=0),
    dict(T_pred=40, K=2),
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2),          # 36 bins
    dict(nb_w=0.04, nb_h=0.04, K=2),                     # sparse windows: empty bins skipped per tile
    dict(grid_size=6, nb_w=0.08, nb_h=0.08, K=2, mno=64, n_scenes=1, n_grids=1),
    # groups larger than one workgroup: the cluster form (kernels_bf16_cl.hip), mno/32 workgroups exchanging bf16 h^T tiles
    dict(mno=96, n_scenes=1, K=2, n_grids=1),
    dict(mno=128, n_scenes=1, K=2, n_grids=1, T_pred=40),           # BASELINE configs[2]: 128 agents, T_pred = 40
    dict(mno=128, n_scenes=2, K=3, H=64, T_pred=9),
    dict(mno=96, n_scenes=1, K=2, n_grids=1, H=256, T_pred=6),
    dict(mno=128, n_scenes=1, K=2, n_grids=1, grid_size=6, nb_w=0.5, nb_h=0.5, T_pred=8),
    dict(mno=128, n_scenes=1, K=1, n_grids=1, nb_w=0.05, nb_h=0.05, n_absent=40),   # sparse windows, many absent slots
])
def test_ioc_bf16_matches_rounding_oracle(torch_cuda, kw):
    from oracle import desire_oracle as O
    kw = dict(kw)
    n_absent = kw.pop("n_absent", 3)
    d32 = small_dims(**kw)
    d16 = d32.replace(bf16=1)
    w = init_weights(d32, 3)
    past, fut, eps, grids, gos = make_case(d32, seed=4, n_absent=min(n_absent, d32.mno - 1))
    ref32 = oracle_forward(d32, w, past, fut, eps, grids, gos)
    ref16 = oracle_forward(d32, w, past, fut, eps, grids, gos, Y_override=ref32["Y0"], ioc_q=O.bf16_round)
    _, Y, score = run_gpu(torch_cuda, d16, w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    dY_ref = ref16["Y"] - ref32["Y0"]
    err = np.abs(Y - ref16["Y"]).max()
    err32 = np.abs(Y - ref32["Y"]).max()
    print("bf16 kernel vs rounding oracle %.2e | vs fp32 oracle %.2e | |dY|max %.2e" % (err, err32, np.abs(dY_ref).max()))
    # same rounding points; what is left is fp32 accumulation order and the occasional operand that rounds the other way
    # (one bf16 ulp = 2^-8 relative), fed back through T_pred recurrent steps: 7e-3 of the refinement offset scale (all cases
    # but the 2x2-grid / wide-window one sit below 1e-3; there, with ~8 neighbours per bin, rounding alone moves the result by
    # 1.8e-2 and the two pooling forms of the kernel -- split over bins or over columns -- land at 1.15e-2 and 0.92e-2)
    scale = max(1.0, float(np.abs(dY_ref).max()))
    assert err < 7e-3 * scale, (err, err32)
    assert np.abs(score - ref16["score"]).max() < 2e-2 * max(1.0, np.abs(ref16["score"]).max())
    # accuracy cost of bf16 operands against exact fp32 (measured 1e-3 .. 2e-2 over these cases): 3e-2 of the scale
    assert err32 < 3e-2 * scale, err32


def test_ioc_bf16_second_refinement_pass_runs(torch_cuda):
    """iters = 2 re-bins the agents from the refined positions; a 1e-3 position difference can move an agent across a
    bin / cell edge, so the second pass is only required to stay close in the mean (same caveat as the fp32 tests'
    bin margin), not elementwise."""
    from oracle import desire_oracle as O
    d32 = small_dims(iters=2, K=2)
    w = init_weights(d32, 3)
    past, fut, eps, grids, gos = make_case(d32, seed=4, n_absent=3)
    ref32 = oracle_forward(d32, w, past, fut, eps, grids, gos)
    _, Y, score = run_gpu(torch_cuda, d32.replace(bf16=1), w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    assert np.isfinite(Y).all() and np.isfinite(score).all()
    assert np.abs(Y - ref32["Y"]).mean() < 2e-2


@pytest.mark.parametrize("variant", ["4", "6"])
def test_ioc_bf16_cluster_equals_one_workgroup_form(torch_cuda, variant):
    """64 agents fit one workgroup (the 64-row tile) AND two cluster members (dims.ioc_form 4 = DESIRE_IOC_CLUSTER: pooling split over
    columns, 6 = DESIRE_IOC_CLUSTER_BINS: over bins): same neighbour-chunk order, same chain-ordered weights -> one pass agrees up to fp32 summation order.
    A second pass re-bins from positions that differ by that rounding, so it is compared in the mean (as for every two-pass
    bf16 check); it exercises the pass-end hand-off."""
    d32 = small_dims(mno=64, n_scenes=2, K=3, n_grids=1, T_pred=9)
    w = init_weights(d32, 3)
    past, fut, eps, grids, gos = make_case(d32, seed=4, n_absent=5)
    ref32 = oracle_forward(d32, w, past, fut, eps, grids, gos)
    _, Y1, s1 = run_gpu(torch_cuda, d32.replace(bf16=1), w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    _, Y1b, _ = run_gpu(torch_cuda, d32.replace(bf16=1, iters=2), w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    _, Y2, s2 = run_gpu(torch_cuda, d32.replace(bf16=1, ioc_form=int(variant)), w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    _, Y2b, _ = run_gpu(torch_cuda, d32.replace(bf16=1, iters=2, ioc_form=int(variant)), w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    assert np.isfinite(Y2).all() and np.isfinite(Y2b).all()
    # the forms sum the per-bin partial products in different orders; an e_r that lands on the other side of a bf16 rounding
    # boundary (2^-8 relative) then moves the result like in the rounding-oracle test: 3e-3 of the offset scale
    tol = 3e-3 * max(1.0, float(np.abs(Y1 - ref32["Y0"]).max()))
    assert np.abs(Y2 - Y1).max() < tol, np.abs(Y2 - Y1).max()
    assert np.abs(s2 - s1).max() < 1e-2 * max(1.0, np.abs(s1).max())
    assert np.abs(Y2b - Y1b).mean() < 2e-3


def test_bf16_is_inference_only_and_validated(torch_cuda):
    from desire_amd import _lib
    d = small_dims(bf16=1)
    h = _lib.Handle(d)
    h.set_weights(init_weights(d, 0))
    with pytest.raises(_lib.DesireError):
        h.set_training(True)
    with pytest.raises(_lib.DesireError):
        _lib.Handle(small_dims(bf16=4))                   # (2 / 3 = split operands, three / six products: tests/test_gpu_split.py)


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3, L=64)])
@pytest.mark.parametrize("fused", [True, False])
def test_cvae_decoder_bf16_convs_match_rounding_oracle(torch_cuda, kw, fused):
    """deconv2 / deconv3 (/ deconv4 when fused) with bf16 operands: d2, d3 and xhat against the oracle's decoder with the
    same operand rounding (fed with the kernel's own z so the comparison isolates these layers), and against plain fp32.
    Default = deconv3+deconv4 fused (d3 never exists); dims.flags = DESIRE_FLAG_NO_FUSE34 keeps the separate kernels (fp32 deconv4)."""
    from oracle import desire_oracle as O
    from desire_amd.spec import FLAG_NO_FUSE34
    d32 = small_dims(**kw)
    d16 = d32.replace(bf16=1, flags=0 if fused else FLAG_NO_FUSE34)
    w = init_weights(d32, 5)
    past, fut, eps, grids, gos = make_case(d32, seed=6, n_absent=2)
    h, _, _ = run_gpu(torch_cuda, d16, w, past, fut, eps, grids, gos)
    z = h.read_buffer("z", (d32.R, d32.L))
    ql = ("deconv1", "deconv2", "deconv3", "deconv4") if fused else ("deconv1", "deconv2", "deconv3")
    xhat_q, layers_q = O.vae_decoder(z, w, return_layers=True, q=O.bf16_round, q_layers=ql)
    xhat_f, layers_f = O.vae_decoder(z, w, return_layers=True)
    checks = [("d1", layers_q[0], layers_f[0], 2048), ("d2", layers_q[1], layers_f[1], 4096), ("xhat", xhat_q, xhat_f, 1024)]
    if not fused:
        checks.insert(1, ("d3", layers_q[2], layers_f[2], 8192))
    for name, lq, lf, n in checks:
        got = h.read_buffer(name, (d32.R, n))
        eq = np.abs(got - lq.reshape(d32.R, n)).max()
        ef = np.abs(got - lf.reshape(d32.R, n)).max()
        scale = max(1.0, float(np.abs(lf).max()))
        print("%s: vs rounding oracle %.2e, vs fp32 %.2e (|x|max %.2f)" % (name, eq, ef, np.abs(lf).max()))
        assert eq < 2e-3 * scale, (name, eq)          # same rounding points: accumulation order + rare 1-ulp operand flips
        assert ef < 3e-2 * scale, (name, ef)          # cost of bf16 operands over a K = 25*128 / 25*64 contraction
    # mask fc (bf16 operands) on the kernel's own xhat / Hx
    xhat = h.read_buffer("xhat", (d32.R, 1024))
    Hr = O.rows_from_agents(h.read_buffer("Hx", (d32.A, d32.H)), d32)
    q = O.bf16_round
    xz_q = O.softmax(O.relu(q(xhat) @ q(w["mask_fc/w"]) + w["mask_fc/b"])) * Hr
    xz_f = O.softmax(O.relu(xhat @ w["mask_fc/w"] + w["mask_fc/b"])) * Hr
    xz = h.read_buffer("xz", (d32.R, d32.H))
    print("xz: vs rounding oracle %.2e, vs fp32 %.2e (|x|max %.3f)" % (np.abs(xz - xz_q).max(), np.abs(xz - xz_f).max(), np.abs(xz_f).max()))
    assert np.abs(xz - xz_q).max() < 2e-5 and np.abs(xz - xz_f).max() < 2e-4


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3), dict(H=256, K=3, n_scenes=1, n_grids=1, T_pred=10),
                                dict(T_pred=40, K=2)])
def test_decoder_bf16_matches_rounding_oracle(torch_cuda, kw):
    """k_decoder_bf16 on the kernel's own x_z / Hx against the oracle decoder with the same operand rounding."""
    from oracle import desire_oracle as O
    d32 = small_dims(**kw)
    w = init_weights(d32, 7)
    past, fut, eps, grids, gos = make_case(d32, seed=8, n_absent=2)
    h, _, _ = run_gpu(torch_cuda, d32.replace(bf16=1), w, past, fut, eps, grids, gos)
    xz = h.read_buffer("xz", (d32.R, d32.H))
    Hx = h.read_buffer("Hx", (d32.A, d32.H))
    pn = O.normalise(to_oracle_layout(past), d32)
    Hr, pl = O.rows_from_agents(Hx, d32), O.rows_from_agents(pn[d32.T_obs - 1], d32)
    Yq = O.decode(xz, Hr, pl, w, d32, q=O.bf16_round)
    Yf = O.decode(xz, Hr, pl, w, d32)
    Y0 = h.read_buffer("Y0", (d32.R, d32.T_pred, 2))
    eq, ef = np.abs(Y0 - Yq).max(), np.abs(Y0 - Yf).max()
    print("Y0: vs rounding oracle %.2e, vs fp32 %.2e" % (eq, ef))
    assert eq < 1e-3, eq          # normalised coordinates; same rounding points
    assert ef < 5e-3, ef          # cost of bf16 recurrent operands on the decoded trajectory



def test_bf16_path_is_window_independent_and_deterministic(torch_cuda):
    """BASELINE configs[1] shapes, 16 windows: windows are independent scenes, so running them together or in two halves
    must give bit-identical refined trajectories and scores (also across repeated runs) -- the property scene-sharding
    over GPUs relies on, here for the bf16-operand kernels."""
    from desire_amd.spec import Dims
    from desire_amd.synth import make_case as mk
    n = 16
    d = Dims(n_scenes=n, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4, nb_w=0.15, nb_h=0.15,
             sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1, bf16=1)
    w = init_weights(d, 0)
    past, fut, eps, grids, gos = mk(d, seed=1, n_absent=0)
    _, Y, s = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    _, Yr, sr = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    np.testing.assert_array_equal(Y, Yr)
    np.testing.assert_array_equal(s, sr)
    assert np.isfinite(Y).all() and np.isfinite(s).all()
    rows = d.K * d.mno
    for lo, hi in ((0, n // 2), (n // 2, n)):
        dd = d.replace(n_scenes=hi - lo)
        _, Yh, sh = run_gpu(torch_cuda, dd, w, past[lo:hi], fut[lo:hi], eps[lo * rows:hi * rows], grids, gos[lo:hi])
        np.testing.assert_array_equal(Yh, Y[lo * rows:hi * rows])
        np.testing.assert_array_equal(sh, s[lo * rows:hi * rows])


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3, L=64)])
def test_cvae_encoder_bf16_convs_match_rounding_oracle(torch_cuda, kw):
    """conv2 / conv3 of the CVAE encoder with bf16 operands, on the kernel's own vae_in."""
    from oracle import desire_oracle as O
    d32 = small_dims(**kw)
    w = init_weights(d32, 9)
    past, fut, eps, grids, gos = make_case(d32, seed=10, n_absent=2)
    h, _, _ = run_gpu(torch_cuda, d32.replace(bf16=1), w, past, fut, eps, grids, gos)
    vin = h.read_buffer("vae_in", (d32.A, 1024))
    mu_q, ls_q, lq = O.vae_encoder(vin, w, d32.L, q=O.bf16_round, return_layers=True)
    mu_f, ls_f, lf = O.vae_encoder(vin, w, d32.L, return_layers=True)
    for name, a_q, a_f, n in (("c2", lq[1], lf[1], 4096), ("c3", lq[2], lf[2], 2048), ("z_mean", mu_q, mu_f, d32.L),
                              ("z_log_sigma_sq", ls_q, ls_f, d32.L)):
        got = h.read_buffer(name, (d32.A, n))
        eq, ef = np.abs(got - a_q.reshape(d32.A, n)).max(), np.abs(got - a_f.reshape(d32.A, n)).max()
        scale = max(1.0, float(np.abs(a_f).max()))
        print("%s: vs rounding oracle %.2e, vs fp32 %.2e (|x|max %.2f)" % (name, eq, ef, np.abs(a_f).max()))
        assert eq < 2e-3 * scale and ef < 3e-2 * scale, (name, eq, ef)


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3), dict(H=256, K=2, n_scenes=1, n_grids=1, T_pred=10),
                                dict(T_pred=40, K=2)])
def test_encoders_bf16_match_rounding_oracle(torch_cuda, kw):
    from oracle import desire_oracle as O
    d32 = small_dims(**kw)
    w = init_weights(d32, 13)
    past, fut, eps, grids, gos = make_case(d32, seed=14, n_absent=2)
    h, _, _ = run_gpu(torch_cuda, d32.replace(bf16=1), w, past, fut, eps, grids, gos)
    pn, fn = O.normalise(to_oracle_layout(past), d32), O.normalise(to_oracle_layout(fut), d32)
    for name, seq, pfx in (("Hx", pn, "enc_x"), ("Hy", fn, "enc_y")):
        got = h.read_buffer(name, (d32.A, d32.H))
        eq = np.abs(got - O.gru_encode(seq, w, pfx, q=O.bf16_round)).max()
        ef = np.abs(got - O.gru_encode(seq, w, pfx)).max()
        print("%s: vs rounding oracle %.2e, vs fp32 %.2e" % (name, eq, ef))
        assert eq < 1e-3 and ef < 1e-2, (name, eq, ef)


@pytest.mark.parametrize("kw", [dict(K=2), dict(mno=64, n_scenes=1, K=2, n_grids=1), dict(mno=128, n_scenes=1, K=1, n_grids=1, T_pred=10)])
def test_ioc_bf16_error_budget_per_pass(torch_cuda, kw):
    """The bf16 IOC error budget, PER refinement pass (two passes were only compared in the mean).  Pass p is isolated
    by starting it from the ORACLE's pass-(p-1) output, so a bin that flipped in an earlier pass cannot leak in:
        against the oracle with the kernel's rounding points    <= 7e-3 of the offset scale   (logic)
        against the exact fp32 oracle                           <= 3e-2 of the offset scale   (what bf16 operands cost)
    per pass; errors of successive passes add (the second pass starts from positions that already carry the first one's)."""
    from oracle import desire_oracle as O
    d32 = small_dims(**kw)
    w = init_weights(d32, 9)
    past, fut, eps, grids, gos = make_case(d32, seed=10, n_absent=3)
    ref0 = oracle_forward(d32, w, past, fut, eps, grids, gos)
    Yin = ref0["Y0"]
    for p in range(2):
        r32 = oracle_forward(d32, w, past, fut, eps, grids, gos, Y_override=Yin)
        r16 = oracle_forward(d32, w, past, fut, eps, grids, gos, Y_override=Yin, ioc_q=O.bf16_round)
        _, Y, score = run_gpu(torch_cuda, d32.replace(bf16=1), w, past, fut, eps, grids, gos, Y_in=Yin)
        scale = max(1.0, float(np.abs(r16["Y"] - Yin).max()))
        e16, e32 = float(np.abs(Y - r16["Y"]).max()), float(np.abs(Y - r32["Y"]).max())
        print("pass %d: vs rounding oracle %.2e, vs fp32 %.2e (offset scale %.2e)" % (p + 1, e16, e32, scale))
        assert e16 < 7e-3 * scale and e32 < 3e-2 * scale, (p, e16, e32)
        assert np.abs(score - r16["score"]).max() < 2e-2 * max(1.0, np.abs(r16["score"]).max())
        Yin = r32["Y"].astype(np.float32)              # the next pass starts from the exact first-pass result
