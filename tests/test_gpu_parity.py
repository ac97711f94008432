"""GPU parity: the HIP path (through the C ABI, via ctypes) against the CPU oracle on identical
seeded inputs.  Tolerances: fp32 trajectory coordinates 1e-3 abs in normalised units (the bar
BASELINE.json states); intermediates 2e-4 abs; integer indices bit-exact."""
import numpy as np
import pytest

from desire_amd.spec import Dims, init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout

pytestmark = pytest.mark.gpu

TOL_Y = 1e-3
TOL_MID = 2e-4


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch


def run_gpu(torch, d, w, past, fut, eps, grids, gos, Y_in=None):
    from desire_amd import _lib
    h = _lib.Handle(d)
    h.set_weights(w)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past_t, eps_t, grids_t = t(past), t(eps), t(grids)
    fut_t = t(fut) if d.posterior else None
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    if Y_in is None:
        h.forward(past_t.data_ptr(), fut_t.data_ptr() if fut_t is not None else 0, eps_t.data_ptr(),
                  Y.data_ptr(), score.data_ptr(), stream)
    else:
        h.encode(past_t.data_ptr(), fut_t.data_ptr() if fut_t is not None else 0, stream)
        Y.copy_(t(Y_in))
        h.ioc_refine(Y.data_ptr(), score.data_ptr(), stream)
    torch.cuda.synchronize()
    return h, Y.cpu().numpy(), score.cpu().numpy()


def oracle_forward(d, w, past, fut, eps, grids, gos, **kw):
    from oracle import desire_oracle as O
    return O.forward(to_oracle_layout(past), to_oracle_layout(fut) if d.posterior else None, eps, grids, gos, w, d, **kw)


def test_stagewise_parity_small(torch_cuda):
    from oracle import desire_oracle as O
    d = small_dims()
    w = init_weights(d, 1)
    past, fut, eps, grids, gos = make_case(d, seed=2)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    h, Y, score = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    A, R = d.A, d.R
    shapes = {"Hx": (A, d.H), "Hy": (A, d.H), "vae_in": (A, d.V), "z_mean": (A, d.L), "z_log_sigma_sq": (A, d.L),
              "z": (R, d.L), "d1": (R, 2048), "d2": (R, 4096), "d3": (R, 8192), "xhat": (R, 1024), "xz": (R, d.H),
              "Y0": (R, d.T_pred, 2)}
    report = {}
    for name, shp in shapes.items():
        got = h.read_buffer(name, shp)
        report[name] = float(np.abs(got - ref[name].reshape(shp)).max())
    print("max abs err per stage:", report)
    for name, err in report.items():
        assert err < (TOL_Y if name == "Y0" else TOL_MID), (name, err, report)
    # IOC stage on identical inputs (the oracle's decoder output) -> identical bins by construction
    margin = O.bin_margin(ref["Y0"].reshape(d.n_scenes * d.K, d.mno, d.T_pred, 2).transpose(0, 2, 1, 3), d.nb_w, d.nb_h, d.grid_size)
    ref2 = oracle_forward(d, w, past, fut, eps, grids, gos)
    _, Y2, score2 = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    assert np.abs(Y2 - ref2["Y"]).max() < TOL_Y
    assert np.abs(score2 - ref2["score"]).max() < 5e-3
    # end to end (GPU decoder output feeds GPU IOC); bins can only differ if a pair sits within ~1e-6 of an edge
    if margin > 1e-5:
        assert np.abs(Y - ref["Y"]).max() < TOL_Y
        assert np.abs(score - ref["score"]).max() < 5e-3


@pytest.mark.parametrize("kw", [
    dict(posterior=0),                                   # prior sampling: z = eps, no future
    dict(mno=16, n_scenes=3, K=5),                       # 4 groups per tile, ragged last tile (R=240)
    dict(mno=64, n_scenes=1, K=2, n_grids=1),            # one group per tile
    dict(H=64, T_pred=7, K=3),                           # smaller hidden, odd horizon
    dict(iters=2, K=2),                                  # two refinement passes
    dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2),          # 2x2 social grid, wide window
    dict(H=256, K=3, n_scenes=1, n_grids=1, T_pred=10),  # BASELINE configs[3] hidden width (mno <= 32 in this round)
    dict(mno=8, n_scenes=5, K=3),                        # 4 groups per 32-row tile, R = 120 (ragged last tile)
    dict(mno=1, n_scenes=3, K=2, n_absent=0),            # lone agents: social pooling sees nobody
    dict(K=1, T_pred=1, n_scenes=1, n_grids=1),          # degenerate horizon / single draw
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2),          # 36 social bins (the paper's count)
    dict(grid_size=5, nb_w=0.4, nb_h=0.4, K=2, mno=64, n_scenes=1, n_grids=1),   # 25 bins, cluster-free 64-row tile
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2, H=64, mno=16, n_scenes=3),
    dict(nb_w=0.04, nb_h=0.04, K=2),                     # sparse windows: most bins empty in a tile -> skipped, some tiles skip all
    dict(grid_size=6, nb_w=0.08, nb_h=0.08, K=2),        # sparse, 36 bins (occupancy word 1 in use)
    dict(nb_w=0.05, nb_h=0.05, K=2, mno=64, n_scenes=1, n_grids=1),   # sparse, 64-row tile
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=1, mno=64, n_scenes=1, n_grids=1, T_pred=5),   # 36 bins x 64 agents: the 64-row tile's
                                                                                            # masks no longer fit -> cluster form
])
def test_end_to_end_variants(torch_cuda, kw):
    kw = dict(kw)
    n_absent = kw.pop("n_absent", 3)
    d = small_dims(**kw)
    w = init_weights(d, 3)
    past, fut, eps, grids, gos = make_case(d, seed=4, n_absent=min(n_absent, d.mno - 1))
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    h, Y, score = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    Y0 = h.read_buffer("Y0", (d.R, d.T_pred, 2))
    assert np.abs(Y0 - ref["Y0"]).max() < TOL_Y
    _, Y2, score2 = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    assert np.abs(Y2 - ref["Y"]).max() < TOL_Y, np.abs(Y2 - ref["Y"]).max()
    assert np.abs(score2 - ref["score"]).max() < 5e-3


def test_neighbor_bins_and_scene_cells_bit_exact(torch_cuda):
    torch = torch_cuda
    from desire_amd import _lib
    from oracle import desire_oracle as O
    d = small_dims(mno=32)
    h = _lib.Handle(d)
    rng = np.random.default_rng(7)
    n_groups = 257
    pos = rng.uniform(-0.1, 1.1, (n_groups, d.mno, 2)).astype(np.float32)
    # adversarial: exact window edges, bin edges, coincident agents
    pos[0, 1] = pos[0, 0] + np.float32([d.nb_w / 2, 0])          # x_j == high  -> excluded
    pos[0, 2] = pos[0, 0] - np.float32([d.nb_w / 2, 0])          # x_j == low   -> included, cell 0
    pos[0, 3] = pos[0, 0]                                         # coincident   -> centre cell
    pos[0, 4] = pos[0, 0] + np.float32([d.nb_w / 4, d.nb_h / 4])  # bin edge
    pos[1] = np.float32(0.5) + np.float32(d.nb_w / d.grid_size) * rng.integers(-2, 3, (d.mno, 2)).astype(np.float32)
    valid = (rng.random((n_groups, d.mno)) > 0.2).astype(np.uint8)
    dev = torch.device("cuda")
    pos_t = torch.as_tensor(pos, device=dev)
    valid_t = torch.as_tensor(valid, device=dev)
    bins_t = torch.full((n_groups, d.mno, d.mno), -7, dtype=torch.int32, device=dev)
    h.neighbor_bins(pos_t.data_ptr(), valid_t.data_ptr(), bins_t.data_ptr(), n_groups)
    torch.cuda.synchronize()
    ref = O.neighbor_bins(pos, valid.astype(bool), d.nb_w, d.nb_h, d.grid_size)
    np.testing.assert_array_equal(bins_t.cpu().numpy(), ref)
    assert (ref >= 0).mean() > 0.02
    # scene cells incl. out-of-range and exact cell edges
    p = rng.uniform(-0.2, 1.2, (5000, 2)).astype(np.float32)
    p[:64, 0] = np.arange(64, dtype=np.float32) / np.float32(64)
    p[64:128, 1] = np.arange(64, dtype=np.float32) / np.float32(64)
    p_t = torch.as_tensor(p, device=dev)
    cells_t = torch.zeros((5000, 2), dtype=torch.int32, device=dev)
    h.scene_cells(p_t.data_ptr(), cells_t.data_ptr(), 5000)
    torch.cuda.synchronize()
    cy, cx = O.scene_cell(p, d.Gh, d.Gw)
    np.testing.assert_array_equal(cells_t.cpu().numpy(), np.stack([cy, cx], -1))


def test_config1_shape_properties(torch_cuda):
    """BASELINE configs[1] dims (32 agents, K=20, T=8/40, H=128) x 4 windows: size-independent checks --
    determinism (bitwise), K-permutation equivariance of the sampler, padding agents never pooled."""
    torch = torch_cuda
    d = Dims(n_scenes=4, mno=32, K=20, T_obs=8, T_pred=40, n_grids=1, nb_w=0.2, nb_h=0.2, sx=1 / 1400.0, sy=1 / 1100.0)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6)
    _, Y1, s1 = run_gpu(torch, d, w, past, fut, eps, grids, gos)
    _, Y2, s2 = run_gpu(torch, d, w, past, fut, eps, grids, gos)
    np.testing.assert_array_equal(Y1, Y2)
    np.testing.assert_array_equal(s1, s2)
    assert np.isfinite(Y1).all() and np.isfinite(s1).all()
    # permuting the K draws of eps permutes the outputs (rows of different k never interact)
    perm = np.random.default_rng(0).permutation(d.K)
    e4 = eps.reshape(d.n_scenes, d.K, d.mno, d.L)[:, perm].reshape(d.R, d.L)
    _, Y3, s3 = run_gpu(torch, d, w, past, fut, e4, grids, gos)
    Y1r = Y1.reshape(d.n_scenes, d.K, d.mno, d.T_pred, 2)[:, perm].reshape(Y1.shape)
    np.testing.assert_array_equal(Y3, Y1r)
    np.testing.assert_array_equal(s3, s1.reshape(d.n_scenes, d.K, d.mno)[:, perm].reshape(-1))
    # against the oracle on the first window only (oracle at full size takes too long for CI)
    from oracle import desire_oracle as O
    d1 = d.replace(n_scenes=1)
    r1 = d1.R
    ref = O.forward(to_oracle_layout(past[:1]), to_oracle_layout(fut[:1]), eps[:r1], grids, gos[:1], w, d1)
    _, Yw, sw = run_gpu(torch, d1, w, past[:1], fut[:1], eps[:r1], grids, gos[:1], Y_in=ref["Y0"])
    assert np.abs(Yw - ref["Y"]).max() < TOL_Y
    # one window on its own: 20 tiles run the bin-split IOC with 4 workgroups per tile, the 4-window batch above with 3 -- the partial sums
    # of e_r group differently, so the two agree to fp32 rounding (bit-identical under dims.ioc_split = 1, where both run the plain form)
    Y1w = run_gpu(torch, d1, w, past[:1], fut[:1], eps[:r1], grids, gos[:1])[1]
    assert np.abs(Y1[:r1] - Y1w).max() < 2e-6


def test_model_api_and_sample_layout(torch_cuda):
    import argparse
    from desire_amd.model import DESIREModel
    args = argparse.Namespace(rnn_size=512, num_layers=1, batch_size=2, seq_length=8, pred_length=12, d_dim=128, e_dim=256,
                              latent_size=128, max_num_obj=30, learning_rate=0.005, grad_clip=10.0, stride=1,
                              neighborhood_size=300, grid_size=4, num_samples=3, img_width=1400.0, img_height=1100.0)
    m = DESIREModel(args)
    d = small_dims(n_scenes=2, mno=30 if False else 32)
    past, fut, _, _, _ = make_case(d, seed=8)
    x = [p[:, :30].astype(np.float64) for p in past]
    y = [f[:, :30].astype(np.float64) for f in fut]
    Y, score = m.forward(x, y)
    assert tuple(Y.shape) == (2, 3, 32, 12, 2) and tuple(score.shape) == (2, 3, 32)
    assert bool(torch_cuda.isfinite(Y).all())
    assert m.cost is not None and np.isfinite(float(m.cost))
    ev = m.evaluate(Y, y)
    assert ev.shape == (64, 4) and np.isfinite(ev).all() and (ev[:, 2] <= ev[:, 0] + 1e-6).all()
    # device-side batching gives the same windows as the host path
    frames = np.zeros((40, 30, 3)); frames[:20] = np.concatenate([x[0], y[0]]); frames[20:] = np.concatenate([x[1], y[1]])
    Yv, _, pv, fv = m.forward_from_video(frames, [0, 20])
    uniq0 = np.unique(frames[:20, :, 0]); has0 = 0.0 in uniq0
    assert tuple(Yv.shape) == tuple(Y.shape) and bool(torch_cuda.isfinite(Yv).all())
    import tempfile, os as _os
    with tempfile.TemporaryDirectory() as td:
        m.save(_os.path.join(td, "w.npz"))
        m2 = DESIREModel.restore(args, _os.path.join(td, "w.npz"))
        Y2, _ = m2.forward(x, y)
        np.testing.assert_array_equal(Y2.cpu().numpy(), Y.cpu().numpy())
    out = m.sample(None, x[0], None, (1400.0, 1100.0), np.concatenate([x[0], y[0]]), num=10, mode="ioc")
    assert out.shape == (18, 30, 3)
    np.testing.assert_array_equal(out[:8], x[0])
    np.testing.assert_array_equal(out[8:, :, 0], np.broadcast_to(x[0][-1, :, 0], (10, 30)))
    assert (out[8:][:, x[0][-1, :, 0] == 0] == 0).all()


def test_cold_rows_scene_cnn_losses_temporal_pooling(torch_cuda):
    """scene CNN, losses, temporal conv O1, feature pooling O11."""
    torch = torch_cuda
    from desire_amd import _lib
    from oracle import desire_oracle as O
    d = small_dims(K=3, n_grids=2)
    w = init_weights(d, 9)
    past, fut, eps, _, gos = make_case(d, seed=10)
    rng = np.random.default_rng(11)
    image = rng.uniform(0, 1, (d.n_grids, 4 * d.Gh, 4 * d.Gw, 3)).astype(np.float32)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    h = _lib.Handle(d)
    h.set_weights(w)
    # scene CNN -> grids, used by the forward below
    image_t = t(image)
    grids_t = torch.zeros((d.n_grids, d.Gh, d.Gw, d.C), device=dev)
    h.scene_cnn(image_t.data_ptr(), 4 * d.Gh, 4 * d.Gw, grids_t.data_ptr())
    torch.cuda.synchronize()
    grids_ref = O.scene_cnn(image, w)
    assert np.abs(grids_t.cpu().numpy() - grids_ref).max() < 1e-4
    h.set_scene_grids(grids_t.data_ptr(), gos)
    past_t, fut_t, eps_t = t(past), t(fut), t(eps)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    # losses on the GPU outputs vs the oracle formulas evaluated on the SAME outputs
    kld = torch.zeros(d.A, device=dev); recon = torch.zeros(d.A, device=dev); cost = torch.zeros(2, device=dev)
    h.losses(fut_t.data_ptr(), Y.data_ptr(), kld.data_ptr(), recon.data_ptr(), cost.data_ptr())
    torch.cuda.synchronize()
    zm, zl = h.read_buffer("z_mean", (d.A, d.L)), h.read_buffer("z_log_sigma_sq", (d.A, d.L))
    futn = O.normalise(to_oracle_layout(fut), d)
    valid = to_oracle_layout(past)[d.T_obs - 1, :, 0] != 0
    k_ref, r_ref, c_ref, n_ref = O.losses(zm, zl, Y.cpu().numpy(), futn, valid, d, present=to_oracle_layout(fut)[:, :, 0] != 0)
    assert np.abs(kld.cpu().numpy() - k_ref).max() < 1e-3 * max(1.0, np.abs(k_ref).max())
    assert np.abs(recon.cpu().numpy() - r_ref).max() < 1e-5
    assert abs(float(cost[0]) - c_ref) < 1e-3 * max(1.0, abs(c_ref)) and int(cost[1]) == n_ref
    # O1 / O11
    rho = torch.zeros((d.A, 200), device=dev)
    h.temporal_conv(past_t.data_ptr(), rho.data_ptr())
    fp = torch.zeros((d.R, d.T_pred, 200), device=dev)
    h.feature_pooling(Y.data_ptr(), rho.data_ptr(), fp.data_ptr())
    torch.cuda.synchronize()
    td = to_oracle_layout(past).transpose(1, 0, 2)[None]              # [1, A, T, 3]
    rho_ref = O.temporal_conv(td, w["temporal/w"], w["temporal/b"])[0, :, 0, :]
    np.testing.assert_allclose(rho.cpu().numpy(), rho_ref, rtol=1e-5, atol=1e-3)   # inputs are raw ids/pixels
    fp_ref = O.feature_pooling(Y.cpu().numpy(), rho.cpu().numpy(), d)
    np.testing.assert_array_equal(fp.cpu().numpy(), fp_ref)


def test_next_rows_window_builder_gaussian_head_ade_fde(torch_cuda, golden_dir):
    """N1 device window+slot builder (bit-exact vs the loader, incl. the reference's
    golden windows), N3 Gaussian head, N4 ADE/FDE."""
    import os
    torch = torch_cuda
    from desire_amd import _lib
    from desire_amd.data_loader import window_to_slots
    from oracle import desire_oracle as O
    g = np.load(os.path.join(golden_dir, "loader_bookstore6_T8.npz"))
    frames = g["data0"].astype(np.float32)                       # [160, 32, 3] preprocessed by the REFERENCE loader
    d = small_dims(n_scenes=6, mno=32, K=2, T_obs=8, T_pred=12)
    h = _lib.Handle(d)
    dev = torch.device("cuda")
    fr_t = torch.as_tensor(frames, device=dev)
    starts = np.array([0, 8, 16, 40, 100, 140], np.int32)
    past = torch.full((6, d.T_obs, d.mno, 3), -1.0, device=dev)
    fut = torch.full((6, d.T_pred, d.mno, 3), -1.0, device=dev)
    h.build_windows(fr_t.data_ptr(), frames.shape[0], frames.shape[1], starts, past.data_ptr(), fut.data_ptr())
    W = d.T_obs + d.T_pred
    for i, s0 in enumerate(starts):
        src, tgt = window_to_slots(g["data0"][s0:s0 + W], W - 1, d.mno)       # reference semantics on W frames
        full = np.concatenate([src, tgt[-1:]], 0)
        np.testing.assert_array_equal(past[i].cpu().numpy(), full[:d.T_obs].astype(np.float32))
        np.testing.assert_array_equal(fut[i].cpu().numpy(), full[d.T_obs:].astype(np.float32))
    # lookahead = 1: the x of DataLoader(seq_length = W) -- slots ranked over W + 1 frames
    h.build_windows(fr_t.data_ptr(), frames.shape[0], frames.shape[1], starts, past.data_ptr(), fut.data_ptr(), lookahead=1)
    for i, s0 in enumerate(starts):
        if s0 + W >= frames.shape[0]:
            continue                                     # the video ends with the window: no look-ahead frame (and no loader window)
        src, _ = window_to_slots(g["data0"][s0:s0 + W + 1], W, d.mno)
        np.testing.assert_array_equal(past[i].cpu().numpy(), src[:d.T_obs].astype(np.float32))
        np.testing.assert_array_equal(fut[i].cpu().numpy(), src[d.T_obs:].astype(np.float32))
    # the reference's own golden batch (T=8 -> 9-frame windows, x = first 8 frames)
    d9 = small_dims(n_scenes=4, mno=32, K=2, T_obs=8, T_pred=1)
    h9 = _lib.Handle(d9)
    p9 = torch.zeros((4, 8, 32, 3), device=dev); f9 = torch.zeros((4, 1, 32, 3), device=dev)
    h9.build_windows(fr_t.data_ptr(), 160, 32, np.array([0, 8, 16, 24], np.int32), p9.data_ptr(), f9.data_ptr())
    np.testing.assert_array_equal(p9.cpu().numpy(), g["x"][0].astype(np.float32))
    np.testing.assert_array_equal(f9.cpu().numpy()[:, 0], g["y"][0][:, -1].astype(np.float32))
    # IndexError path: more unique ids than slots
    d4 = small_dims(n_scenes=1, mno=4, K=1, T_obs=8, T_pred=12)
    h4 = _lib.Handle(d4)
    with pytest.raises(_lib.DesireError, match="more unique ids"):
        h4.build_windows(fr_t.data_ptr(), 160, 32, np.array([0], np.int32), past.data_ptr(), fut.data_ptr())
    with pytest.raises(_lib.DesireError, match="out of range"):
        h.build_windows(fr_t.data_ptr(), 160, 32, np.array([150] * 6, np.int32), past.data_ptr(), fut.data_ptr())
    # N3 Gaussian head
    rng = np.random.default_rng(5)
    params = rng.normal(0, 0.7, (1000, 5)).astype(np.float32)
    normals = rng.standard_normal((1000, 2)).astype(np.float32)
    out = torch.zeros((1000, 2), device=dev)
    pt, nt = torch.as_tensor(params, device=dev), torch.as_tensor(normals, device=dev)
    h.gaussian_sample(pt.data_ptr(), nt.data_ptr(), out.data_ptr(), 1000)
    torch.cuda.synchronize()
    ref = O.gaussian_sample(params, normals)
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5 and out.max().item() <= 1.0
    # N4 ADE/FDE on a real forward
    w = init_weights(d, 2)
    h.set_weights(w)
    pst, ft, eps, grids, gos = make_case(d, seed=3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    pst_t, ft_t, eps_t, grids_t = t(pst), t(ft), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev); score = torch.zeros((d.R,), device=dev)
    h.forward(pst_t.data_ptr(), ft_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    af = torch.zeros((d.A, 4), device=dev)
    h.ade_fde(Y.data_ptr(), ft_t.data_ptr(), af.data_ptr())
    torch.cuda.synchronize()
    ref = O.ade_fde_k(Y.cpu().numpy(), O.normalise(to_oracle_layout(ft), d), d, present=to_oracle_layout(ft)[:, :, 0] != 0)
    assert np.abs(af.cpu().numpy() - ref).max() < 1e-5
    assert (af[:, 2] <= af[:, 0] + 1e-6).all() and (af[:, 3] <= af[:, 1] + 1e-6).all()


@pytest.mark.parametrize("kw", [
    dict(mno=64, n_scenes=2, K=3, n_grids=1, T_pred=9),                  # 2 workgroups per group
    dict(mno=128, n_scenes=1, K=2, n_grids=1, T_pred=6),                 # 4 workgroups per group, 128-bit masks
    dict(mno=64, H=256, n_scenes=1, K=2, n_grids=1, T_pred=6),           # BASELINE configs[3] shape: H=256, 64 agents/scene
    dict(mno=96, n_scenes=1, K=2, n_grids=1, T_pred=5, iters=2),         # 3 per group, two refinement passes
    dict(mno=64, n_scenes=2, K=3, n_grids=1, T_pred=9, nb_w=0.04, nb_h=0.04),   # sparse windows: empty bins skipped per tile
])
def test_ioc_cluster_form(torch_cuda, kw):
    """Groups larger than one workgroup tile: tpg workgroups exchange hidden states through global memory each
    step (agent-scope release/acquire hand-off).  Checked against the oracle, and -- where the single-workgroup
    64-row kernel exists -- bitwise against it (same summation order)."""
    d = small_dims(ioc_form=4, **kw)                                      # DESIRE_IOC_CLUSTER (the default form for 96 / 128 agents anyway)
    w = init_weights(d, 13)
    past, fut, eps, grids, gos = make_case(d, seed=14, n_absent=5, spread=0.3)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    assert (np.asarray(ref["Y"]) != 0).any()
    _, Y2, score2 = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    assert np.abs(Y2 - ref["Y"]).max() < TOL_Y, np.abs(Y2 - ref["Y"]).max()
    assert np.abs(score2 - ref["score"]).max() < 5e-3
    if d.mno == 64 and d.H <= 128:
        _, Y3, score3 = run_gpu(torch_cuda, d.replace(ioc_form=0), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
        np.testing.assert_array_equal(Y2, Y3)
        np.testing.assert_array_equal(score2, score3)


def test_ioc_cluster_form_under_load_is_bitwise_stable(torch_cuda):
    """800 tiles on a 256-workgroup persistent grid (every workgroup walks several groups, all CUs busy, the
    hand-off buffers are re-used and L1-warm): the cluster form must equal the single-workgroup 64-row kernel
    bit for bit, twice in a row."""
    d = Dims(n_scenes=20, mno=64, K=20, T_obs=8, T_pred=40, n_grids=1, nb_w=0.2, nb_h=0.2, sx=1 / 1400.0, sy=1 / 1100.0)
    w = init_weights(d, 21)
    past, fut, eps, grids, gos = make_case(d, seed=22, n_absent=7)
    _, Y_ref, s_ref = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    for _ in range(2):
        _, Y_cl, s_cl = run_gpu(torch_cuda, d.replace(ioc_form=4), w, past, fut, eps, grids, gos)
        np.testing.assert_array_equal(Y_cl, Y_ref)
        np.testing.assert_array_equal(s_cl, s_ref)


def test_config3_per_gpu_shard_shape(torch_cuda):
    """BASELINE configs[3] (2048 agents = 32 scenes x 64, K=50, H=256, 8 GPUs) as ONE GPU sees it after scene
    sharding: 4 scenes x 64 agents, K=50, H=256, T=8/40 -> 12 800 rows on the cluster-form IOC.  Full-size checks
    are size-independent properties; the oracle comparison runs on a K=3 cut of the first scene."""
    torch = torch_cuda
    from desire_amd.dist import shard_windows
    assert [shard_windows(32, r, 8) for r in (0, 7)] == [(0, 4), (28, 32)]
    d = Dims(n_scenes=4, mno=64, K=50, T_obs=8, T_pred=40, H=256, n_grids=1, nb_w=0.2, nb_h=0.2, sx=1 / 1400.0, sy=1 / 1100.0)
    w = init_weights(d, 31)
    past, fut, eps, grids, gos = make_case(d, seed=32, n_absent=6)
    _, Y1, s1 = run_gpu(torch, d, w, past, fut, eps, grids, gos)
    _, Y2, s2 = run_gpu(torch, d, w, past, fut, eps, grids, gos)
    np.testing.assert_array_equal(Y1, Y2)
    np.testing.assert_array_equal(s1, s2)
    assert np.isfinite(Y1).all() and np.isfinite(s1).all()
    perm = np.random.default_rng(1).permutation(d.K)
    e4 = eps.reshape(d.n_scenes, d.K, d.mno, d.L)[:, perm].reshape(d.R, d.L)
    _, Y3, s3 = run_gpu(torch, d, w, past, fut, e4, grids, gos)
    np.testing.assert_array_equal(Y3, Y1.reshape(d.n_scenes, d.K, d.mno, d.T_pred, 2)[:, perm].reshape(Y1.shape))
    # scenes are independent: scene 2 alone gives the same rows
    d1 = d.replace(n_scenes=1)
    sl = slice(2 * d.K * d.mno, 3 * d.K * d.mno)
    _, Y4, s4 = run_gpu(torch, d1, w, past[2:3], fut[2:3], eps[sl], grids, gos[:1])
    np.testing.assert_array_equal(Y4, Y1[sl])
    np.testing.assert_array_equal(s4, s1[sl])
    # oracle on a K=3 cut of scene 0
    d3 = d.replace(n_scenes=1, K=3)
    e3 = eps.reshape(d.n_scenes, d.K, d.mno, d.L)[0, :3].reshape(-1, d.L)
    ref = oracle_forward(d3, w, past[:1], fut[:1], e3, grids, gos[:1])
    h, _, _ = run_gpu(torch, d3, w, past[:1], fut[:1], e3, grids, gos[:1])
    assert np.abs(h.read_buffer("Y0", (d3.R, d3.T_pred, 2)) - ref["Y0"]).max() < TOL_Y
    _, Yo, so = run_gpu(torch, d3, w, past[:1], fut[:1], e3, grids, gos[:1], Y_in=ref["Y0"])
    assert np.abs(Yo - ref["Y"]).max() < TOL_Y


def test_abi_error_behaviour_on_device(torch_cuda):
    """State and argument errors come back as codes + desire_last_error(), never as a crash."""
    torch = torch_cuda
    from desire_amd import _lib
    d = small_dims(n_scenes=1, K=1, T_pred=4)
    h = _lib.Handle(d)
    dev = torch.device("cuda")
    z = torch.zeros(1 << 16, device=dev)
    with pytest.raises(_lib.DesireError, match="not finalized"):
        h.encode(z.data_ptr(), z.data_ptr())
    w = init_weights(d, 0)
    bad = dict(w); bad.pop("head/w")
    with pytest.raises(_lib.DesireError, match="weight not set: head/w"):
        h.set_weights(bad)
    with pytest.raises(_lib.DesireError, match="expected"):
        h.set_weights({"head/w": np.zeros(3, np.float32)})
    with pytest.raises(_lib.DesireError, match="unknown weight"):
        h.set_weights({"nope": np.zeros(3, np.float32)})
    h.set_weights(w)
    with pytest.raises(_lib.DesireError, match="needs dev_fut"):
        h.encode(z.data_ptr(), 0)
    with pytest.raises(_lib.DesireError, match="scene grids not set"):
        h.ioc_refine(z.data_ptr(), z.data_ptr())
    with pytest.raises(_lib.DesireError, match="out of range"):
        h.set_scene_grids(z.data_ptr(), [3])
    with pytest.raises(_lib.DesireError, match="unknown buffer"):
        h.read_buffer("nope", (1,))
    with pytest.raises(_lib.DesireError, match="expected"):
        h.read_buffer("Hx", (1,))
    with pytest.raises(ValueError):
        _lib.Handle(small_dims(mno=24))


def test_window_with_no_agents_and_duplicate_positions(torch_cuda):
    """Edge inputs of the loader layout: one window whose slots are ALL empty (id 0 everywhere -- the loader emits such
    windows when nothing is tracked), and a window where several agents stand on exactly the same spot (collisions: same
    scene cell, centre social bin).  The path must stay finite and match the oracle; empty slots never act as neighbours."""
    from oracle import desire_oracle as O
    d = small_dims(n_scenes=3, K=2)
    w = init_weights(d, 11)
    past, fut, eps, grids, gos = make_case(d, seed=12, n_absent=2)
    past[1] = 0.0
    fut[1] = 0.0                                                    # window 1: nobody there
    past[2, :, 1:6, 1:] = past[2, :, 0:1, 1:]
    fut[2, :, 1:6, 1:] = fut[2, :, 0:1, 1:]                          # window 2: slots 1..5 walk exactly with slot 0
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    h, Y, score = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    assert np.isfinite(Y).all() and np.isfinite(score).all()
    Y0 = h.read_buffer("Y0", (d.R, d.T_pred, 2))
    assert np.abs(Y0 - ref["Y0"]).max() < TOL_Y
    _, Y2, score2 = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    assert np.abs(Y2 - ref["Y"]).max() < TOL_Y
    assert np.abs(score2 - ref["score"]).max() < 5e-3
    # bins of the coincident agents: bit-exact against the oracle at the decoded positions
    P = ref["Y0"].reshape(d.n_scenes * d.K, d.mno, d.T_pred, 2)[:, :, 0]
    valid = np.repeat((past[:, d.T_obs - 1, :, 0] != 0)[:, None], d.K, 1).reshape(d.n_scenes * d.K, d.mno)
    torch = torch_cuda
    pos_t = torch.as_tensor(np.ascontiguousarray(P, np.float32), device="cuda")
    val_t = torch.as_tensor(np.ascontiguousarray(valid).astype(np.uint8), device="cuda")
    bins_t = torch.full((d.n_scenes * d.K, d.mno, d.mno), -7, dtype=torch.int32, device="cuda")
    h.neighbor_bins(pos_t.data_ptr(), val_t.data_ptr(), bins_t.data_ptr(), d.n_scenes * d.K)
    torch.cuda.synchronize()
    got = bins_t.cpu().numpy()
    np.testing.assert_array_equal(got, O.neighbor_bins(P.astype(np.float32), valid, d.nb_w, d.nb_h, d.grid_size))
    assert (got[d.K * 1:d.K * 2] == -1).all()                       # the empty window has no neighbours at all


@pytest.mark.parametrize("kw", [dict(H=64, T_pred=7, K=3, mno=16), dict(H=64, T_pred=7, K=3), dict(K=3, mno=8, n_scenes=5), dict(H=256, K=3, T_pred=5)])
def test_outputs_do_not_depend_on_where_a_window_sits_in_the_batch(torch_cuda, kw):
    """Reversing the order of the windows must reverse the outputs BIT-EXACTLY: a sample's arithmetic may not depend on
    the tile row / accumulator register it lands in (what scene-sharding over GPUs and the sharded-IOC equality rely on;
    an implicit fp contraction that the compiler applied to some unrolled elements only once broke this by 1 ulp)."""
    d = small_dims(**kw)
    w = init_weights(d, 31)
    past, fut, eps, grids, gos = make_case(d, seed=32, n_absent=2)
    h, Y, s = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    Y0 = h.read_buffer("Y0", (d.R, d.T_pred, 2))
    rows = d.K * d.mno
    e2 = eps.reshape(d.n_scenes, rows, d.L)[::-1].reshape(d.R, d.L).copy()
    h2, Y2, s2 = run_gpu(torch_cuda, d, w, past[::-1].copy(), fut[::-1].copy(), e2, grids, np.asarray(gos)[::-1].copy())
    Y02 = h2.read_buffer("Y0", (d.R, d.T_pred, 2))
    back = lambda x: x.reshape((d.n_scenes, rows) + x.shape[1:])[::-1].reshape(x.shape)
    np.testing.assert_array_equal(back(Y02), Y0)
    np.testing.assert_array_equal(back(Y2), Y)
    np.testing.assert_array_equal(back(s2), s)


@pytest.mark.parametrize("G", [3, 4, 6])
def test_logpolar_bins_bit_exact(torch_cuda, G):
    """dims.bin_mode = 1: rings x sectors around every agent, bit-exact against the oracle fed with the library's own
    table; random points plus adversarial ones on ring radii, on sector boundaries, coincident, and beyond r_max."""
    torch = torch_cuda
    from desire_amd import _lib
    from oracle import desire_oracle as O
    d = small_dims(mno=32, grid_size=G, bin_mode=1, nb_w=0.4, nb_h=0.05)
    h = _lib.Handle(d)
    tab = h.bin_table()
    np.testing.assert_allclose(tab, O.logpolar_table(d.nb_h, d.nb_w, G), rtol=2e-7, atol=1e-12)
    rng = np.random.default_rng(3)
    n_groups = 129
    pos = rng.uniform(0.0, 1.0, (n_groups, d.mno, 2)).astype(np.float32)
    radii = np.sqrt(tab[:G].astype(np.float64))
    for k in range(G):                                              # exactly on every ring radius, along every boundary direction
        pos[0, 1 + k] = pos[0, 0] + np.float32(radii[k]) * np.float32([tab[8 + 2 * k], tab[9 + 2 * k]])
        pos[1, 1 + k] = pos[1, 0] + np.float32(0.1) * np.float32([tab[8 + 2 * k], tab[9 + 2 * k]])
    pos[2, 1] = pos[2, 0]                                            # coincident
    pos[2, 2] = pos[2, 0] + np.float32([0.5, 0.0])                   # beyond r_max
    valid = rng.random((n_groups, d.mno)) > 0.15
    pos_t = torch.as_tensor(pos, device="cuda")
    val_t = torch.as_tensor(valid.astype(np.uint8), device="cuda")
    bins_t = torch.full((n_groups, d.mno, d.mno), -7, dtype=torch.int32, device="cuda")
    h.neighbor_bins(pos_t.data_ptr(), val_t.data_ptr(), bins_t.data_ptr(), n_groups)
    torch.cuda.synchronize()
    ref = O.neighbor_bins(pos, valid, d.nb_w, d.nb_h, G, tab)
    np.testing.assert_array_equal(bins_t.cpu().numpy(), ref)
    assert ref.max() == G * G - 1 and (ref >= 0).mean() > 0.1        # the case exercises every ring and sector


@pytest.mark.parametrize("kw", [dict(grid_size=6, K=2), dict(grid_size=4, K=3, mno=16, n_scenes=3), dict(grid_size=3, K=2, mno=64, n_scenes=1, n_grids=1),
                                dict(grid_size=6, K=2, bf16=1)])
def test_ioc_with_logpolar_pooling(torch_cuda, kw):
    """The IOC stage with the paper's log-polar social pooling (all IOC kernel forms: 32-row, 64-row / cluster, bf16)."""
    from oracle import desire_oracle as O
    kw = dict(kw)
    bf16 = kw.pop("bf16", 0)
    d = small_dims(bin_mode=1, nb_w=0.45, nb_h=0.04, **kw)
    w = init_weights(d, 17)
    past, fut, eps, grids, gos = make_case(d, seed=18, n_absent=2)
    from desire_amd import _lib
    tab = _lib.Handle(d).bin_table()
    ref0 = oracle_forward(d, w, past, fut, eps, grids, gos, bin_tab=tab)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos, bin_tab=tab, Y_override=ref0["Y0"], ioc_q=O.bf16_round if bf16 else None)
    _, Y, score = run_gpu(torch_cuda, d.replace(bf16=bf16), w, past, fut, eps, grids, gos, Y_in=ref0["Y0"])
    scale = max(1.0, float(np.abs(ref["Y"] - ref0["Y0"]).max()))
    tol = 5e-3 * scale if bf16 else TOL_Y
    assert np.abs(Y - ref["Y"]).max() < tol, np.abs(Y - ref["Y"]).max()
    assert np.abs(score - ref["score"]).max() < (2e-2 if bf16 else 5e-3) * max(1.0, float(np.abs(ref["score"]).max()))


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3, L=64), dict(posterior=0)])
def test_per_object_batch_norm_mode(torch_cuda, kw):
    """dims.bn_mode = 1: the reference's literal batch-norm (phase=train on a batch of ONE object, model/model.py:453-462,
    471-481) = per-sample, per-channel moments over each conv layer's pixels -- stagewise against the oracle's
    bn_mode="per_object"."""
    d = small_dims(bn_mode=1, **kw)
    w = init_weights(d, 41)
    rng = np.random.default_rng(5)
    for k in list(w):                                    # non-trivial affine parameters (fresh init: gamma = 1, beta = 0)
        if k.endswith("/bn/gamma"):
            w[k] = (1.0 + 0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
        if k.endswith("/bn/beta"):
            w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
    past, fut, eps, grids, gos = make_case(d, seed=42, n_absent=2)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos, bn_mode="per_object")
    h, Y, score = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    A, R = d.A, d.R
    shapes = {"z": (R, d.L), "d1": (R, 2048), "d2": (R, 4096), "d3": (R, 8192), "xhat": (R, 1024), "xz": (R, d.H), "Y0": (R, d.T_pred, 2)}
    if d.posterior:
        shapes.update({"z_mean": (A, d.L), "z_log_sigma_sq": (A, d.L)})
    report = {name: float(np.abs(h.read_buffer(name, shp) - ref[name].reshape(shp)).max()) for name, shp in shapes.items()}
    print("per-object BN, max abs err per stage:", report)
    for name, err in report.items():
        assert err < (TOL_Y if name == "Y0" else 5e-4), (name, err, report)
    # and it is a different function from the frozen-statistics default
    ref_frozen = oracle_forward(d.replace(bn_mode=0), w, past, fut, eps, grids, gos)
    assert np.abs(ref_frozen["xhat"] - ref["xhat"]).max() > 1e-2


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, K=5), dict(H=64, T_pred=7, K=3, L=64), dict(posterior=0, K=2)])
def test_whole_batch_batch_norm_mode(torch_cuda, kw):
    """dims.bn_mode = 2: prettytensor's phase=train batch-norm over everything a call batches (model/model.py:453,459-461,471)
    -- per-channel moments over all samples and pixels -- stagewise against the oracle's bn_mode="batch" (VERDICT r01 missing #5:
    this mode existed in the oracle only)."""
    d = small_dims(bn_mode=2, **kw)
    w = init_weights(d, 43)
    rng = np.random.default_rng(6)
    for k in list(w):
        if k.endswith("/bn/gamma"):
            w[k] = (1.0 + 0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
        if k.endswith("/bn/beta"):
            w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
    past, fut, eps, grids, gos = make_case(d, seed=44, n_absent=2)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos, bn_mode="batch")
    h, Y, score = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    A, R = d.A, d.R
    shapes = {"z": (R, d.L), "d1": (R, 2048), "d2": (R, 4096), "d3": (R, 8192), "xhat": (R, 1024), "xz": (R, d.H), "Y0": (R, d.T_pred, 2)}
    if d.posterior:
        shapes.update({"z_mean": (A, d.L), "z_log_sigma_sq": (A, d.L)})
    report = {name: float(np.abs(h.read_buffer(name, shp) - ref[name].reshape(shp)).max()) for name, shp in shapes.items()}
    print("whole-batch BN, max abs err per stage:", report)
    for name, err in report.items():
        assert err < (TOL_Y if name == "Y0" else 5e-4), (name, err, report)
    ref_po = oracle_forward(d.replace(bn_mode=0), w, past, fut, eps, grids, gos, bn_mode="per_object")
    assert np.abs(ref_po["xhat"] - ref["xhat"]).max() > 1e-3           # not the per-object function


def test_batch_statistics_modes_are_fp32_and_both_train(torch_cuda):
    from desire_amd import _lib
    with pytest.raises(_lib.DesireError):
        _lib.Handle(small_dims(bn_mode=1, bf16=1))
    for mode in (1, 2):                                   # per-object AND whole-batch statistics train (gradients: test_gpu_train_cluster.py)
        d = small_dims(bn_mode=mode)
        h = _lib.Handle(d)
        h.set_weights(init_weights(d, 0))
        h.set_training(True)
        h.close()


@pytest.mark.parametrize("kw", [
    dict(),                                              # dense windows, 32-agent groups
    dict(mno=16, n_scenes=3, K=5),                       # two groups per tile, ragged last tile
    dict(mno=8, n_scenes=5, K=3),
    dict(H=64, T_pred=7, K=3),
    dict(nb_w=0.04, nb_h=0.04, K=2),                     # sparse: most bins skipped, the rest hold one or two rows
    dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2),          # 4 crowded bins: more than 16 rows per bin -> two operand chunks
    dict(grid_size=6, nb_w=0.5, nb_h=0.5, K=2),          # 36 bins
    dict(grid_size=4, bin_mode=1, nb_w=0.45, nb_h=0.04, K=2),
    dict(iters=2, K=2),
    dict(mno=1, n_scenes=3, K=2, n_absent=0),
])
def test_row_compacted_pooling_form(torch_cuda, kw):
    """dims.ioc_form = 8 (DESIRE_IOC_COMPACT): the pooling contraction runs on the rows that have a neighbour in the bin only (packed into
    16-row MFMA tiles, results added back into their rows).  Same function as the default form up to fp32 summation order:
    checked against the oracle and against the default form."""
    kw = dict(kw)
    n_absent = kw.pop("n_absent", 3)
    d = small_dims(**kw)
    w = init_weights(d, 3)
    past, fut, eps, grids, gos = make_case(d, seed=4, n_absent=min(n_absent, d.mno - 1))
    from desire_amd import _lib
    tab = None
    if d.bin_mode == 1:
        tab = _lib.Handle(d).bin_table()
    ref = oracle_forward(d, w, past, fut, eps, grids, gos, **({"bin_tab": tab} if tab is not None else {}))
    _, Yd, sd = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    _, Yc, sc = run_gpu(torch_cuda, d.replace(ioc_form=8), w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    if d.iters == 1:
        assert np.abs(Yc - ref["Y"]).max() < TOL_Y, np.abs(Yc - ref["Y"]).max()
        assert np.abs(sc - ref["score"]).max() < 5e-3
        assert np.abs(Yc - Yd).max() < 2e-5 and np.abs(sc - sd).max() < 2e-4
    else:                                                # a second pass re-bins from refined positions (bin-edge caveat)
        assert np.abs(Yc - Yd).mean() < 1e-3


@pytest.mark.parametrize("kw", [dict(mno=160, n_scenes=1, K=2, n_grids=1, T_pred=6), dict(mno=256, n_scenes=2, K=1, n_grids=1, T_pred=5, H=64, L=64),
                                dict(mno=192, n_scenes=1, K=2, n_grids=1, T_pred=5, iters=2, nb_w=0.1, nb_h=0.1)])
def test_scenes_of_more_than_128_agents_run_the_step_wise_ioc(torch_cuda, kw):
    """VERDICT r03 Missing 7: max_num_obj > 128 on one GPU.  160 .. 256 agents per scene: everything before the IOC is per agent; the IOC
    pass runs as one launch per step of the agent-sharded kernel with a single rank (256-bit neighbour masks).  Stagewise against the
    oracle like every other shape; training is refused."""
    from desire_amd import _lib
    d = small_dims(**kw)
    w = init_weights(d, 17)
    past, fut, eps, grids, gos = make_case(d, seed=18, n_absent=9, spread=0.3)
    ref = oracle_forward(d, w, past, fut, eps, grids, gos)
    h, Y0, _ = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos)
    assert np.abs(h.read_buffer("Y0", (d.R, d.T_pred, 2)) - ref["Y0"]).max() < TOL_Y
    _, Y, score = run_gpu(torch_cuda, d, w, past, fut, eps, grids, gos, Y_in=ref["Y0"])
    assert (np.asarray(ref["Y"]) != np.asarray(ref["Y0"])).any()
    if d.iters == 1:
        assert np.abs(Y - ref["Y"]).max() < TOL_Y, np.abs(Y - ref["Y"]).max()
        assert np.abs(score - ref["score"]).max() < 5e-3
    else:
        assert np.abs(Y - ref["Y"]).mean() < 1e-3
    with pytest.raises(_lib.DesireError):
        h.set_training(True)
    with pytest.raises(ValueError):
        small_dims(mno=288).validate()
    with pytest.raises(ValueError):
        small_dims(mno=160, bf16=1).validate()


@pytest.mark.parametrize("G,nb_h,nb_w", [(4, 0.02, 0.3), (4, 0.125, 0.5), (3, 0.05, 0.25), (6, 0.03, 0.45), (5, 0.01, 0.2), (4, 32.0 / 1088.0, 256.0 / 1424.0)])
def test_library_logpolar_table_equals_an_independent_computation(torch_cuda, G, nb_h, nb_w):
    """the log-polar parity tests hand the oracle the LIBRARY's constant table (so that two libm's cannot move a bin edge), which left
    the table itself pinned by four hand-checked radii only.  Here the table the library built (C++ pow / cos / sin in double, rounded to fp32) is compared
    with numpy's own evaluation of the same closed forms: every one of its 3 G constants within ONE fp32 ulp, the exactly representable ones (cos / sin of
    multiples of 90 degrees, radii that are exact powers) bit-equal."""
    from desire_amd import _lib
    from oracle import desire_oracle as O
    d = small_dims(n_scenes=1, K=1, bin_mode=1, grid_size=G, nb_h=nb_h, nb_w=nb_w)
    h = _lib.Handle(d)
    lib_tab = h.bin_table()
    h.close()
    ref = O.logpolar_table(np.float32(nb_h), np.float32(nb_w), G)            # (the library receives fp32 dims)
    used = list(range(G)) + [8 + 2 * k for k in range(G)] + [9 + 2 * k for k in range(G)]
    for i in used:
        a, b = np.float32(lib_tab[i]), np.float32(ref[i])
        assert a == b or abs(float(a) - float(b)) <= float(np.spacing(np.float32(max(abs(a), abs(b), 1e-30)))), (i, a, b)
        if abs(float(b)) < 1e-7:                                               # cos / sin of a multiple of 90 degrees: |value| is rounding noise of pi, far below
            assert abs(float(a)) < 1e-7                                        # any coordinate difference it multiplies (both evaluations agree on that)
    exact = sum(1 for i in used if np.float32(lib_tab[i]) == np.float32(ref[i]))
    assert exact >= len(used) - 2, (exact, len(used))                          # at most a couple of last-bit differences between the two libm's
    assert not np.any(lib_tab[[i for i in range(20) if i not in used]])       # the rest of the 20-float table is zero
