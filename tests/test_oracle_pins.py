"""Pins for the CPU oracle (the reference holds no tests or golden vectors for the model path, so the
oracle is checked against INDEPENDENT formulas: torch conv kernels, scipy's bivariate normal, a
hand-computed TF-order GRU step, closed-form KLD, the deconv size rule of the reference)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from desire_amd.spec import Dims, flops_per_sample, init_weights, weight_shapes
from oracle import desire_oracle as O


def test_conv2d_matches_torch_same_and_valid():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 32, 32, 1)).astype(np.float32)
    w = rng.standard_normal((5, 5, 1, 32)).astype(np.float32)
    got = O.conv2d(x, w, 2, "SAME")                          # TF SAME: pad 1 before, 2 after
    ref = F.conv2d(F.pad(torch.tensor(x).permute(0, 3, 1, 2), (1, 2, 1, 2)), torch.tensor(w).permute(3, 2, 0, 1), stride=2)
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), atol=2e-5)
    x = rng.standard_normal((2, 8, 8, 64)).astype(np.float32)
    w = rng.standard_normal((5, 5, 64, 128)).astype(np.float32)
    got = O.conv2d(x, w, 1, "VALID")
    ref = F.conv2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1))
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), atol=5e-4)


def test_conv2d_transpose_matches_torch_and_reference_size_rule():
    rng = np.random.default_rng(1)
    # size rule utils/convolutional_vae_util.py:154-157: 1 -> 4 -> 8 -> 16 -> 32
    assert [O.deconv_out_size(1, 4, 1, "VALID"), O.deconv_out_size(4, 5, 1, "VALID"),
            O.deconv_out_size(8, 5, 2, "SAME"), O.deconv_out_size(16, 5, 2, "SAME")] == [4, 8, 16, 32]
    x = rng.standard_normal((2, 8, 8, 64)).astype(np.float32)
    w = rng.standard_normal((5, 5, 32, 64)).astype(np.float32)      # [kh,kw,out,in]
    got = O.conv2d_transpose(x, w, 2, "SAME")
    ref = F.conv_transpose2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1), stride=2, padding=1)
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy()[:, :16, :16], atol=2e-4)
    x = rng.standard_normal((2, 4, 4, 128)).astype(np.float32)
    w = rng.standard_normal((5, 5, 64, 128)).astype(np.float32)
    got = O.conv2d_transpose(x, w, 1, "VALID")
    ref = F.conv_transpose2d(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1))
    np.testing.assert_allclose(got, ref.permute(0, 2, 3, 1).numpy(), atol=3e-4)
    # conv2d_transpose is the adjoint of conv2d: <conv(x), y> == <x, conv_T(y)>
    xx = rng.standard_normal((1, 16, 16, 4)).astype(np.float64)
    ww = rng.standard_normal((5, 5, 4, 6)).astype(np.float64)
    yy = rng.standard_normal((1, 8, 8, 6)).astype(np.float64)
    lhs = (O.conv2d(xx, ww, 2, "SAME") * yy).sum()
    rhs = (xx * O.conv2d_transpose(yy, ww.transpose(0, 1, 2, 3), 2, "SAME")).sum()
    assert abs(lhs - rhs) < 1e-9 * max(1.0, abs(lhs))


def test_gru_cell_tf_order_hand_computed():
    # H=1, x scalar: every number by hand.  Wg rows [x; h], columns [r, u]; bias_start 1.0
    x, h = np.array([[0.5]]), np.array([[-0.25]])
    Wg, bg = np.array([[0.2, -0.4], [0.6, 0.8]]), np.array([1.0, 1.0])
    Wc, bc = np.array([[1.5], [-2.0]]), np.array([0.1])
    r = 1 / (1 + np.exp(-(0.5 * 0.2 + -0.25 * 0.6 + 1.0)))
    u = 1 / (1 + np.exp(-(0.5 * -0.4 + -0.25 * 0.8 + 1.0)))
    c = np.tanh(0.5 * 1.5 + (r * -0.25) * -2.0 + 0.1)          # reset applied BEFORE the candidate matmul
    want = u * -0.25 + (1 - u) * c                              # u gates the OLD state
    got = O.gru_cell(x, h, Wg, bg, Wc, bc)
    assert abs(got[0, 0] - want) < 1e-12
    # and it is NOT torch.nn.GRU's convention (which applies r after the matmul and z to the new state)
    torch_style = (1 - u) * -0.25 + u * np.tanh(0.5 * 1.5 + r * (-0.25 * -2.0) + 0.1)
    assert abs(got[0, 0] - torch_style) > 1e-3


def test_bivariate_normal_matches_scipy_and_loss_utils():
    from scipy.stats import multivariate_normal
    rng = np.random.default_rng(2)
    for _ in range(20):
        mux, muy = rng.normal(size=2)
        sx, sy = rng.uniform(0.3, 2.0, 2)
        rho = rng.uniform(-0.9, 0.9)
        x, y = rng.normal(size=2)
        cov = [[sx * sx, rho * sx * sy], [rho * sx * sy, sy * sy]]
        want = multivariate_normal.pdf([x, y], mean=[mux, muy], cov=cov)
        assert abs(O.normal_2d_pdf(x, y, mux, muy, sx, sy, rho) - want) < 1e-12
    assert O.reconstr_loss(0., 0., 1., 1., 0., np.array([0.]), np.array([0.])) == pytest.approx(np.log(2 * np.pi))
    assert O.reconstr_loss(0., 0., 1e-3, 1e-3, 0., np.array([50.]), np.array([50.])) == pytest.approx(-np.log(1e-20))
    mux, muy, sx, sy, corr = O.get_coef(np.array([[0.1, 0.2, 0.0, np.log(2.0), 0.5]]))
    assert (sx[0, 0], sy[0, 0]) == pytest.approx((1.0, 2.0)) and corr[0, 0] == pytest.approx(np.tanh(0.5))


def test_kld_closed_forms():
    z = np.zeros((4, 8))
    assert O.kld_loss(z, z) == 0.0
    assert O.kld_loss(np.ones((2, 3)), z[:2, :3]) == pytest.approx(1.5)           # 0.5*sum(mu^2)
    ls = np.full((1, 5), np.log(4.0))
    assert O.kld_loss(z[:1, :5], ls) == pytest.approx(-0.5 * 5 * (1 + np.log(4.0) - 4.0))


def test_neighbor_bins_known_answers():
    G, nb = 4, 0.5                              # dyadic numbers: every step exact in fp32
    pos = np.zeros((1, 6, 2), np.float32)
    pos[0, 0] = [0.5, 0.5]
    pos[0, 1] = [0.25, 0.25]                    # exactly low,low  -> cell (0,0) -> bin 0
    pos[0, 2] = [0.75, 0.5]                     # x == high        -> excluded
    pos[0, 3] = [0.7, 0.7]                      # 0.45/0.5*4 = 3.6 -> cell (3,3) -> bin 15
    pos[0, 4] = [0.5, 0.5]                      # coincident       -> cell (2,2) -> bin 10
    pos[0, 5] = [0.5625, 0.3125]                # cx = 2.5 -> 2, cy = 0.5 -> 0 -> bin 2
    valid = np.ones((1, 6), bool)
    b = O.neighbor_bins(pos, valid, nb, nb, G)[0, 0]
    assert list(b) == [-1, 0, -1, 15, 10, 2]
    valid[0, 4] = False
    assert O.neighbor_bins(pos, valid, nb, nb, G)[0, 0, 4] == -1
    cy, cx = O.scene_cell(np.array([[0.0, 0.0], [0.999, 0.5], [1.0, -0.1], [0.015625, 0.03125]], np.float32), 64, 64)
    assert list(cy) == [0, 32, 0, 2] and list(cx) == [0, 63, 63, 1]


def test_temporal_conv_channel_order_and_quirk():
    """O1 (model/model.py:116-133): depthwise, output channel c*100+q, input channels are (id, x)."""
    rng = np.random.default_rng(3)
    T, M = 8, 3
    td = rng.standard_normal((1, M, T, 3)).astype(np.float32)
    wt = rng.standard_normal((1, T, 2, 100)).astype(np.float32)
    bt = rng.standard_normal(200).astype(np.float32)
    out = O.temporal_conv(td, wt, bt)
    assert out.shape == (1, M, 1, 200)
    m, c, q = 1, 1, 37
    want = max(0.0, float((td[0, m, :, c] * wt[0, :, c, q]).sum() + bt[c * 100 + q]))
    assert out[0, m, 0, c * 100 + q] == pytest.approx(want, rel=1e-5)


def test_forward_shapes_and_batch_bn_mode_runs():
    d = Dims(n_scenes=1, mno=8, K=2, T_obs=4, T_pred=5, nb_w=0.5, nb_h=0.5)
    w = init_weights(d, 0)
    assert set(w) == set(weight_shapes(d))
    rng = np.random.default_rng(4)
    past = np.zeros((4, 8, 3), np.float32); past[..., 0] = np.arange(1, 9); past[..., 1:] = rng.random((4, 8, 2))
    fut = np.zeros((5, 8, 3), np.float32); fut[..., 0] = np.arange(1, 9); fut[..., 1:] = rng.random((5, 8, 2))
    eps = rng.standard_normal((d.R, d.L)).astype(np.float32)
    grids = rng.uniform(-1, 1, (1, 64, 64, 32)).astype(np.float32)
    a = O.forward(past, fut, eps, grids, [0], w, d)
    b = O.forward(past, fut, eps, grids, [0], w, d, bn_mode="batch")
    assert a["Y"].shape == (16, 5, 2) and a["score"].shape == (16,)
    assert np.isfinite(b["Y"]).all() and np.abs(a["xhat"] - b["xhat"]).max() > 1e-4
    # prior mode ignores the future
    dp = d.replace(posterior=0)
    c = O.forward(past, None, eps, grids, [0], w, dp)
    np.testing.assert_array_equal(c["z"], eps)
    assert 50e6 < flops_per_sample(Dims(mno=32, K=20)) < 56e6


def test_ref_compat_literal_graph_plumbing(golden_dir):
    """BASELINE configs[0]: the reference graph as written, on a real loader batch (CPU plumbing, no GPU):
    H = d_dim = 16 = 2*T, 7 decoder steps, one eps per object, raw pixels, per-object train-phase BN."""
    import os
    from desire_amd.spec import weight_shapes
    g = np.load(os.path.join(golden_dir, "loader_bookstore6_T8.npz"))
    x, y = g["x"][0][0], g["y"][0][0]                       # [T=8, MNO=32, 3] loader layout (time-major)
    inp, tgt = x.transpose(1, 0, 2), y.transpose(1, 0, 2)   # placeholders are object-major (model/model.py:91-105)
    H, T, L = 16, 8, 128
    rng = np.random.default_rng(0)
    w = {}
    def gru(p, n_in):
        w[p + "/gates/kernel"] = (rng.standard_normal((n_in + H, 2 * H)) * 0.05).astype(np.float32)
        w[p + "/gates/bias"] = np.ones(2 * H, np.float32)
        w[p + "/candidate/kernel"] = (rng.standard_normal((n_in + H, H)) * 0.05).astype(np.float32)
        w[p + "/candidate/bias"] = np.zeros(H, np.float32)
    gru("enc_x", 2); gru("enc_y", 2); gru("dec", H)
    big = init_weights(Dims(T_obs=T, T_pred=T), 1)
    for k, v in big.items():
        if k.startswith(("vae_enc", "vae_dec", "temporal")):
            w[k] = v
    w["fc_c/w"] = (rng.standard_normal((2 * H, 1024)) / np.sqrt(2 * H)).astype(np.float32); w["fc_c/b"] = np.zeros(1024, np.float32)
    w["mask_fc/w"] = (rng.standard_normal((1024, H)) / 32).astype(np.float32); w["mask_fc/b"] = np.zeros(H, np.float32)
    eps = rng.standard_normal((32, L)).astype(np.float32)
    out = O.forward_ref_compat(inp, tgt, eps, w, H=H, L=L)
    assert out["rho"].shape == (32, 200) and out["output_states"].shape == (32, 7, 8, 2)
    assert out["feature_pooling"].shape == (32, 7, 8, 200) and np.isfinite(out["feature_pooling"]).all()
    # O10/O11 are pure re-reads of the decoder outputs
    k, t, obj = 3, 5, 9
    assert out["feature_pooling"][obj, k, t, 150] == out["output_states"][obj, k, t, 1] * out["rho"][obj, 150]
    # batch-of-one train-phase BN makes objects independent: running one object alone gives the same row
    one = O.forward_ref_compat(inp[obj:obj + 1], tgt[obj:obj + 1], eps[obj:obj + 1], w, H=H, L=L)
    np.testing.assert_allclose(one["output_states"][0], out["output_states"][obj], atol=1e-6)
    with pytest.raises(ValueError):
        O.forward_ref_compat(inp, tgt, eps, w, H=32, L=L)


def test_logpolar_bins_known_answers():
    """Hand-checked cases of the log-polar layout (G = 4 rings x 4 sectors, r_min = 0.125, r_max = 0.5: ring radii
    0.125*4^(k/4)... -> thresholds 0.1768, 0.25, 0.3536, 0.5).  Sector boundaries at 0, 90, 180, 270 degrees."""
    from oracle import desire_oracle as O
    G = 4
    tab = O.logpolar_table(0.125, 0.5, G)
    np.testing.assert_allclose(np.sqrt(tab[:4]), [0.125 * 4 ** 0.25, 0.25, 0.125 * 4 ** 0.75, 0.5], rtol=1e-6)
    c = np.float32([0.5, 0.5])
    pts = np.float32([[0.5, 0.5],          # the centre itself
                      [0.6, 0.55],         # d = 0.112 -> ring 0, first quadrant -> sector 0      -> bin 0
                      [0.5, 0.7],          # d = 0.2   -> ring 1, on the +y axis: cross(dir_1, v) = 0 >= 0, sector 1 -> bin 5
                      [0.2, 0.5],          # d = 0.3   -> ring 2, on the -x axis -> sector 2     -> bin 10
                      [0.5, 0.05],         # d = 0.45  -> ring 3, on the -y axis -> sector 3     -> bin 15
                      [0.9, 0.9],          # d = 0.566 >= r_max -> outside
                      [0.75, 0.5],         # d = 0.25 exactly on the ring-1/2 threshold (d2 >= t) -> ring 2, +x axis -> sector 0 -> bin 8
                      [0.45, 0.45]])       # d = 0.0707 < r_min -> ring 0 (the inner disc belongs to ring 0), third quadrant -> sector 2 -> bin 2
    pts[0] = c
    valid = np.ones(len(pts), bool)
    bins = O.neighbor_bins_logpolar(pts, valid, G, tab)[0]          # seen from agent 0 (the centre)
    assert bins.tolist() == [-1, 0, 5, 10, 15, -1, 8, 2]
    valid[3] = False
    assert O.neighbor_bins_logpolar(pts, valid, G, tab)[0][3] == -1  # absent agents are never pooled
    # coincident agents: v = 0 -> ring 0, no sector test succeeds -> sector 0
    two = np.float32([[0.3, 0.3], [0.3, 0.3]])
    assert O.neighbor_bins_logpolar(two, np.ones(2, bool), G, tab).tolist() == [[-1, 0], [0, -1]]
