"""Backward pass of the HIP path vs the torch-autograd training oracle (float64) on the same seeded inputs."""
import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout

pytestmark = pytest.mark.gpu


def rel_err(got, ref):
    return float(np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12))


@pytest.fixture(scope="module")
def grads_case():
    import torch
    from desire_amd import _lib
    from oracle import desire_torch as OT
    d = small_dims(n_scenes=2, mno=32, K=3, T_obs=6, T_pred=7, n_grids=1)
    w = init_weights(d, 41)
    # spread the K samples (a fresh init gives K nearly identical futures, which makes the ranking gradients vanish)
    for k in w:
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    past, fut, eps, grids, gos = make_case(d, seed=42, n_absent=4)
    vals, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev)
    score = torch.zeros((d.R,), device=dev)
    h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
    torch.cuda.synchronize()
    return d, w, h, ref, vals


DONE = ["head/w", "head/b", "dec/gates/kernel", "dec/gates/bias", "dec/candidate/kernel", "dec/candidate/bias",
        "mask_fc/w", "mask_fc/b",
        "vae_dec/deconv4/w", "vae_dec/deconv4/b", "vae_dec/deconv3/w", "vae_dec/deconv3/b", "vae_dec/deconv2/w",
        "vae_dec/deconv2/b", "vae_dec/deconv1/w", "vae_dec/deconv1/b",
        "vae_enc/fc/w", "vae_enc/fc/b", "vae_enc/conv3/w", "vae_enc/conv3/b", "vae_enc/conv2/w", "vae_enc/conv2/b",
        "vae_enc/conv1/w", "vae_enc/conv1/b", "fc_c/w", "fc_c/b",
        "enc_y/gates/kernel", "enc_y/gates/bias", "enc_y/candidate/kernel", "enc_y/candidate/bias",
        "ioc/reg/w", "ioc/reg/b", "ioc/score/w", "ioc/score/b", "ioc/gates/kernel", "ioc/gates/bias",
        "ioc/candidate/kernel", "ioc/candidate/bias", "ioc/social_fc/w", "ioc/social_fc/b", "ioc/vel_fc/w", "ioc/vel_fc/b",
        "enc_x/gates/kernel", "enc_x/gates/bias", "enc_x/candidate/kernel", "enc_x/candidate/bias"]


@pytest.mark.parametrize("name", DONE)
def test_weight_gradient_matches_autograd(grads_case, name):
    d, w, h, ref, _ = grads_case
    got = h.get_grad(name, w[name].shape)
    assert np.isfinite(got).all()
    if name == "ioc/score/b":       # softmax over K is shift invariant: the exact gradient is 0
        assert np.abs(got).max() < 1e-6
        return
    assert rel_err(got, ref[name]) < 2e-4, (name, rel_err(got, ref[name]), np.abs(ref[name]).max())


def _fresh(d, w):
    import torch
    from desire_amd import _lib
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    dev = torch.device("cuda")
    past, fut, eps, grids, gos = make_case(d, seed=42, n_absent=4)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    ten = dict(past=t(past), fut=t(fut), eps=t(eps), grids=t(grids))
    h.set_scene_grids(ten["grids"].data_ptr(), gos)
    ten["Y"] = torch.zeros((d.R, d.T_pred, 2), device=dev)
    ten["score"] = torch.zeros((d.R,), device=dev)
    return h, ten


def _fwd(h, ten):
    h.forward(ten["past"].data_ptr(), ten["fut"].data_ptr(), ten["eps"].data_ptr(), ten["Y"].data_ptr(), ten["score"].data_ptr())


def _bwd(h, ten):
    h.backward(ten["past"].data_ptr(), ten["fut"].data_ptr(), ten["eps"].data_ptr())


def test_train_loss_matches_oracle(grads_case):
    import torch
    d, w, h, ref, vals = grads_case
    past, fut, eps, grids, gos = make_case(d, seed=42, n_absent=4)
    fut_t = torch.as_tensor(np.ascontiguousarray(fut), device="cuda")
    got = h.train_loss(fut_t.data_ptr())
    v = to_oracle_layout(past)[d.T_obs - 1, :, 0] != 0
    n = max(v.sum(), 1)
    for key in ("recon", "kld", "ce", "reg"):
        want = float((vals[key] * v).sum() / n)
        assert abs(got[key] - want) <= 2e-5 * max(1.0, abs(want)), (key, got[key], want)
    assert got["n_present"] == n
    assert abs(got["loss"] - float(vals["loss"])) < 1e-4 * max(1.0, abs(float(vals["loss"])))


def test_adam_step_matches_tf_formula(grads_case):
    from oracle import desire_torch as OT
    d, w, _, _, _ = grads_case
    h, ten = _fresh(d, w)
    names = ["dec/gates/kernel", "ioc/social_fc/w", "vae_dec/deconv2/w", "vae_enc/conv1/b", "head/b", "enc_x/candidate/kernel"]
    m = {k: np.zeros(w[k].shape) for k in names}
    v = {k: np.zeros(w[k].shape) for k in names}
    cur = {k: w[k].astype(np.float64) for k in names}
    for step in (1, 2, 3):
        _fwd(h, ten); _bwd(h, ten)
        g = {k: h.get_grad(k, w[k].shape).astype(np.float64) for k in names}
        h.adam_step(0.001)
        for k in names:
            cur[k], m[k], v[k] = OT.adam_step(cur[k], g[k], m[k], v[k], step, lr=0.001)
            got = h.get_weight(k, w[k].shape)
            assert np.abs(got - cur[k]).max() < 1e-6, (k, step)
            cur[k] = got.astype(np.float64)        # follow the device weights so rounding does not accumulate
    # frozen batch-norm statistics and the gradient-free auxiliaries never move
    for k in ("vae_dec/deconv2/bn/gamma", "vae_enc/conv2/bn/moving_var", "scene_cnn/conv1/w", "temporal/w"):
        assert np.array_equal(h.get_weight(k, w[k].shape), w[k])


def test_device_repack_equals_host_pack(grads_case):
    """After Adam the packed operands are rebuilt on the device; a fresh handle packing the same weights on the host
    must produce the same forward."""
    import torch
    d, w, _, _, _ = grads_case
    h, ten = _fresh(d, w)
    for _ in range(2):
        _fwd(h, ten); _bwd(h, ten); h.adam_step(0.01)
    _fwd(h, ten)
    torch.cuda.synchronize()
    Y1, s1 = ten["Y"].cpu().numpy().copy(), ten["score"].cpu().numpy().copy()
    w2 = {k: h.get_weight(k, v.shape) for k, v in w.items()}
    assert any(not np.array_equal(w2[k], w[k]) for k in w)
    h2, ten2 = _fresh(d, w2)
    _fwd(h2, ten2)
    torch.cuda.synchronize()
    assert np.abs(ten2["Y"].cpu().numpy() - Y1).max() < 2e-6
    assert np.abs(ten2["score"].cpu().numpy() - s1).max() < 2e-5
    # and backward through the rebuilt transposed operands
    _bwd(h, ten); _bwd(h2, ten2)
    for k in ("dec/gates/kernel", "ioc/gates/kernel", "enc_x/gates/kernel", "vae_enc/conv2/w", "vae_dec/deconv3/w"):
        a, b = h.get_grad(k, w[k].shape), h2.get_grad(k, w[k].shape)
        assert rel_err(a, b) < 1e-4, k


def test_clip_by_global_norm(grads_case):
    d, w, _, _, _ = grads_case
    h, ten = _fresh(d, w)
    _fwd(h, ten); _bwd(h, ten)
    g = h.grad_tensor()
    n0 = float(g.double().norm())
    got = h.clip_grads(n0 * 10, want_norm=True)
    assert abs(got - n0) < 1e-4 * n0
    assert abs(float(g.double().norm()) - n0) < 1e-6 * n0           # under the limit: untouched
    h.clip_grads(n0 / 4)
    assert abs(float(g.double().norm()) - n0 / 4) < 1e-4 * n0


def test_loss_decreases_on_a_fixed_batch(grads_case):
    d, w, _, _, _ = grads_case
    h, ten = _fresh(d, w)
    losses = []
    for _ in range(40):
        _fwd(h, ten); _bwd(h, ten)
        losses.append(h.train_loss(ten["fut"].data_ptr())["loss"])
        h.clip_grads(10.0)
        h.adam_step(0.002)
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.7 * losses[0], losses[::8]


def test_model_train_step_and_checkpoint(tmp_path):
    from types import SimpleNamespace
    from desire_amd.model import DESIREModel
    args = SimpleNamespace(seq_length=6, pred_length=7, d_dim=64, rnn_size=512, latent_size=64, max_num_obj=8, learning_rate=0.0005,
                           grad_clip=10.0, neighborhood_size=256, grid_size=4, num_samples=3, batch_size=2)
    m = DESIREModel(args, seed=3)
    rng = np.random.default_rng(0)
    x = [np.concatenate([np.arange(1, 9, dtype=np.float32)[None, :, None].repeat(6, 0), rng.uniform(200, 1800, (6, 8, 2)).astype(np.float32)], -1) for _ in range(2)]
    y = [np.concatenate([np.arange(1, 9, dtype=np.float32)[None, :, None].repeat(7, 0), rng.uniform(200, 1800, (7, 8, 2)).astype(np.float32)], -1) for _ in range(2)]
    ls = [m.train_step(x, y, seed=1)["loss"] for _ in range(30)]
    assert ls[-1] < ls[0], ls[::5]
    p = str(tmp_path / "w.npz")
    m.save(p)
    m2 = DESIREModel.restore(args, p)
    Ya, _ = m.forward(x, y, seed=1)
    Yb, _ = m2.forward(x, y, seed=1)
    assert float((Ya - Yb).abs().max()) < 2e-6


def test_full_size_gradient_is_the_mean_of_its_halves():
    """BASELINE configs[1] shapes at 64 windows (40 960 rows: activation indices pass 2^31 elements): with every agent
    present the loss is a plain mean over windows, so the gradient of the whole batch must equal the mean of the gradients
    of its two halves -- a size-independent check of the backward path at the size the bench runs."""
    import torch
    from desire_amd import _lib
    from desire_amd.spec import Dims
    from desire_amd.synth import make_case as mk
    n = 64
    d = Dims(n_scenes=n, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4, nb_w=0.15, nb_h=0.15,
             sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1)
    w = init_weights(d, 0)
    past, fut, eps, grids, gos = mk(d, seed=1, n_absent=0)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    grids_t = t(grids)

    def grad_of(lo, hi):
        dd = d.replace(n_scenes=hi - lo)
        h = _lib.Handle(dd)
        h.set_weights(w)
        h.set_training(True)
        rows = slice(lo * d.K * d.mno, hi * d.K * d.mno)
        p, f, e = t(past[lo:hi]), t(fut[lo:hi]), t(eps[rows])
        h.set_scene_grids(grids_t.data_ptr(), gos[lo:hi])
        Y = torch.zeros((dd.R, d.T_pred, 2), device=dev); sc = torch.zeros((dd.R,), device=dev)
        h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
        h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
        torch.cuda.synchronize()
        g = h.grad_tensor().clone()
        del h
        return g

    g_all = grad_of(0, n)
    g_half = 0.5 * (grad_of(0, n // 2) + grad_of(n // 2, n))
    assert bool(torch.isfinite(g_all).all())
    scale = float(g_all.abs().max())
    err = float((g_all - g_half).abs().max())
    assert scale > 0 and err < 2e-4 * scale, (err, scale)


@pytest.mark.parametrize("kw", [dict(bin_mode=1, nb_w=0.45, nb_h=0.04),        # log-polar rings x sectors
                                dict(nb_w=0.05, nb_h=0.05, mno=32),            # sparse rectangular windows: empty bins skipped
                                dict(grid_size=6, nb_w=0.5, nb_h=0.5, mno=32, K=2),          # the paper's 36 bins (rectangular)
                                dict(grid_size=6, bin_mode=1, nb_w=0.45, nb_h=0.04, K=2),    # ... and as 6 rings x 6 sectors
                                dict(mno=64, n_scenes=1, K=2),                 # 64 agents per scene: 64-row IOC tiles forward and backward
                                dict(mno=64, n_scenes=2, K=3, H=64, T_pred=5, nb_w=0.08, nb_h=0.08)])
def test_gradients_with_logpolar_or_sparse_pooling(kw):
    """Backward through the log-polar social layout (dims.bin_mode = 1) and through windows so small that most bins of a
    tile hold nobody (forward and backward skip those bins): the pooling transpose only sees the bins through the
    neighbour / observer masks, so the IOC and encoder gradients must still match autograd."""
    import torch
    from desire_amd import _lib
    from oracle import desire_torch as OT
    d = small_dims(**{**dict(n_scenes=2, mno=16, K=3, T_obs=5, T_pred=6, n_grids=1), **kw})
    w = init_weights(d, 51)
    for k in w:
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    past, fut, eps, grids, gos = make_case(d, seed=52, n_absent=2)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    tab = h.bin_table() if d.bin_mode == 1 else None
    _, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d, bin_tab=tab)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(g.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev); sc = torch.zeros((d.R,), device=dev)
    h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
    h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
    torch.cuda.synchronize()
    for name in ("ioc/social_fc/w", "ioc/gates/kernel", "ioc/candidate/kernel", "ioc/reg/w", "ioc/score/w", "enc_x/gates/kernel", "dec/gates/kernel"):
        got = h.get_grad(name, w[name].shape)
        assert rel_err(got, ref[name]) < 2e-4, (name, rel_err(got, ref[name]))


def test_hipgraph_replay_of_forward_backward_clip(grads_case):
    """desire_graph_begin/end/launch: the stream-ordered launch sequence of a training step (forward, backward, global-norm
    clip) captured once and replayed must give the same gradients and outputs as issuing the calls directly."""
    import torch
    d, w, _, _, _ = grads_case
    h, ten = _fresh(d, w)
    side = torch.cuda.Stream()
    sp = side.cuda_stream

    def step(stream):
        h.forward(ten["past"].data_ptr(), ten["fut"].data_ptr(), ten["eps"].data_ptr(), ten["Y"].data_ptr(), ten["score"].data_ptr(), stream)
        h.backward(ten["past"].data_ptr(), ten["fut"].data_ptr(), ten["eps"].data_ptr(), stream)
        h.clip_grads(1e-3, stream=stream)

    torch.cuda.synchronize()
    step(sp)                                             # warm-up outside capture: lazy allocations and function attributes
    side.synchronize()
    g_ref = h.grad_tensor().clone()
    Y_ref = ten["Y"].clone()
    h.graph_begin(sp)
    step(sp)
    gid = h.graph_end(sp)
    for _ in range(6):                                   # repeated launches of the same exec (a memset node once broke exactly this)
        h.grad_tensor().zero_(); ten["Y"].zero_()
        torch.cuda.synchronize()
        h.graph_launch(gid, sp)
        side.synchronize()
        assert torch.equal(ten["Y"], Y_ref)
        assert torch.equal(h.grad_tensor(), g_ref), (float(h.grad_tensor().abs().max()), float(g_ref.abs().max()))
    assert float(g_ref.abs().max()) > 0
    from desire_amd import _lib
    with pytest.raises(_lib.DesireError):
        h.graph_launch(gid + 7, sp)
    with pytest.raises(_lib.DesireError):
        h.graph_begin(0)                                 # the default stream cannot be captured


def test_gradients_are_bitwise_reproducible():
    """Every reduction of the backward pass sums in a fixed order (slices, then slices of slices; no float atomics), so two
    runs on the same inputs give the same bits -- at a size where the weight-gradient GEMMs use several slices."""
    import torch
    from desire_amd import _lib
    d = small_dims(n_scenes=24, mno=32, K=5, T_obs=8, T_pred=12, n_grids=1)
    w = init_weights(d, 71)
    past, fut, eps, grids, gos = make_case(d, seed=72, n_absent=3)
    dev = torch.device("cuda")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    Y = torch.zeros((d.R, d.T_pred, 2), device=dev); sc = torch.zeros((d.R,), device=dev)
    runs = []
    for _ in range(2):
        h = _lib.Handle(d)
        h.set_weights(w)
        h.set_training(True)
        h.set_scene_grids(g.data_ptr(), gos)
        for _ in range(2):                                   # the second pass of each handle reuses warm buffers
            h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
            h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
            torch.cuda.synchronize()
            runs.append(h.grad_tensor().clone())
    assert float(runs[0].abs().max()) > 0
    for r in runs[1:]:
        assert torch.equal(runs[0], r)


def test_other_batch_sizes_see_the_trained_weights(tmp_path):
    """The optimiser updates the weights inside ONE handle (its batch size); a forward with another number of windows, or the
    prior path, must run on the current values -- also after further training steps -- and a changed training batch size is
    refused (the Adam moments would be lost)."""
    import torch
    from types import SimpleNamespace
    from desire_amd.model import DESIREModel
    args = SimpleNamespace(seq_length=4, pred_length=5, stride=1, e_dim=16, d_dim=64, rnn_size=512, num_layers=1, batch_size=2,
                           latent_size=64, max_num_obj=8, learning_rate=0.002, grad_clip=10.0, neighborhood_size=256, grid_size=4,
                           num_samples=2)
    rng = np.random.default_rng(0)
    def windows(n, T):
        w = np.zeros((n, T, 8, 3), np.float64)
        w[:, :, :5, 0] = np.arange(1, 6)
        w[:, :, :5, 1:] = rng.uniform(200, 1200, (n, 1, 5, 2)) + np.cumsum(rng.normal(0, 4, (n, T, 5, 2)), 1)
        return list(w)
    x2, y2, x3, y3 = windows(2, 4), windows(2, 5), windows(3, 4), windows(3, 5)
    m = DESIREModel(args, seed=3)
    Y_before, _ = m.forward(x3, y3, seed=1)
    Y_before = Y_before.cpu().numpy().copy()
    for _ in range(3):
        m.train_step(x2, y2, seed=2)
    Y_after, _ = m.forward(x3, y3, seed=1)
    Y_after = Y_after.cpu().numpy().copy()
    assert np.abs(Y_after - Y_before).max() > 1e-6                      # the 3-window handle was refreshed
    m.save(str(tmp_path / "w.npz"))
    Y_restored, _ = DESIREModel.restore(args, str(tmp_path / "w.npz")).forward(x3, y3, seed=1)
    np.testing.assert_array_equal(Y_after, Y_restored.cpu().numpy())
    m.train_step(x2, y2, seed=2)
    Y_again, _ = m.forward(x3, y3, seed=1)
    assert np.abs(Y_again.cpu().numpy() - Y_after).max() > 1e-7         # ... and again after the next step
    Yp, _ = m.forward(x3, None, seed=1)                                  # prior path: its own handle, current weights
    assert bool(torch.isfinite(Yp).all())
    with pytest.raises(ValueError, match="Adam moments"):
        m.train_step(x3, y3, seed=2)
    with pytest.raises(ValueError, match="object slots"):
        m.forward([np.zeros((4, 9, 3))] * 2, None)
