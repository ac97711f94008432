"""Training on crowded scenes (VERDICT r01 missing #3): (scene, k) groups of 96 / 128 agents train through the cluster forms
(k_ioc_cl with activation saves, k_ioc_bwd_cl) -- every trainable tensor against float64 autograd, then a real
deathCircle/video4 window (65 track ids per frame) and a mixed-scene loop over every committed SDD slice."""
import os

import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("kw", [dict(mno=96, n_scenes=1, K=2, H=128), dict(mno=128, n_scenes=2, K=2, H=64, L=64),
                                dict(mno=64, n_scenes=1, K=3, H=64, L=64, variant4=True)])
def test_cluster_gradients_match_autograd(kw):
    import torch
    from desire_amd import _lib
    from oracle import desire_torch as OT
    kw = dict(kw)
    form = 4 if kw.pop("variant4", False) else 0               # DESIRE_IOC_CLUSTER: 64 agents through the cluster forward (the backward stays the 64-row tile)
    d = small_dims(T_obs=5, T_pred=6, n_grids=1, ioc_form=form, **kw)
    w = init_weights(d, 61)
    for k in w:
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    past, fut, eps, grids, gos = make_case(d, seed=62, n_absent=7)
    vals, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda")
    score = torch.zeros((d.R,), device="cuda")
    h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
    torch.cuda.synchronize()
    got = h.train_loss(fut_t.data_ptr())
    assert abs(got["loss"] - float(vals["loss"])) < 1e-4 * max(1.0, abs(float(vals["loss"])))
    bad = {}
    for name in ref:
        if name not in w or name.startswith(("scene_cnn", "temporal", "gauss_head")) or "/bn/" in name:
            continue
        g = h.get_grad(name, w[name].shape)
        if name == "ioc/score/b":
            assert np.abs(g).max() < 1e-6
            continue
        e = float(np.abs(g - ref[name]).max() / (np.abs(ref[name]).max() + 1e-12))
        if not e < 3e-4:
            bad[name] = e
    assert not bad, bad


def _slice_windows(tag, starts, t_obs, t_pred, slots):
    from desire_amd.data_loader import window_to_slots
    g = np.load(os.path.join(HERE, "loader_%s.npz" % tag))
    W = t_obs + t_pred
    past, fut = [], []
    for s0 in starts:
        src, _ = window_to_slots(g["data0"][s0:s0 + W + 1], W, g["data0"].shape[1])
        if src.shape[1] > slots:
            assert not src[:, slots:].any()
            src = src[:, :slots]
        src = np.pad(src, ((0, 0), (0, slots - src.shape[1]), (0, 0)))
        past.append(src[:t_obs]); fut.append(src[t_obs:])
    return past, fut


def test_mixed_scene_training_over_every_committed_sdd_slice():
    """train.py's loop shape on windows from bookstore/video6, deathCircle/video2 AND deathCircle/video4 in one batch (the
    reference mixes all its CSVs, train.py:99-100; video4's 65 ids per frame need 96 slots): the loss goes down."""
    import argparse
    from desire_amd.model import DESIREModel
    args = argparse.Namespace(rnn_size=512, num_layers=1, batch_size=6, seq_length=8, pred_length=12, d_dim=64, e_dim=256,
                              latent_size=64, max_num_obj=96, learning_rate=0.001, grad_clip=10.0, stride=1,
                              neighborhood_size=160, grid_size=4, num_samples=3, img_width=1432.0, img_height=1948.0)
    past, fut = [], []
    for tag, starts in (("bookstore6_T8", [0, 60]), ("deathcircle2_T8", [0, 30]), ("deathcircle4_T48", [0, 25])):
        p, f = _slice_windows(tag, starts, 8, 12, 96)
        past += p; fut += f
    assert max((np.asarray(p)[-1, :, 0] != 0).sum() for p in past) >= 60            # the crowded deathCircle/video4 windows are in
    m = DESIREModel(args, seed=9)
    losses = [m.train_step(past, fut, seed=i)["loss"] for i in range(12)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    Y, score = m.forward(past, fut, seed=0)
    ev = m.evaluate(Y, fut)
    assert ev.shape == (6 * 96, 4) and np.isfinite(ev).all()


@pytest.mark.parametrize("kw", [dict(iters=2), dict(iters=3, mno=16, n_scenes=3, H=64, L=64), dict(iters=2, mno=96, n_scenes=1, K=2, H=64, L=64)])
def test_training_through_several_refinement_passes(kw):
    """iters >= 2 in training (`iters = 1` only): Y_p = Y_{p-1} + dY_p, each pass on the detached positions of the
    previous one, every pass's activations kept, BPTT per pass with accumulated weight gradients -- against autograd of the same
    graph (oracle/desire_torch.py)."""
    import torch
    from desire_amd import _lib
    from oracle import desire_torch as OT
    d = small_dims(T_obs=5, T_pred=6, n_grids=1, K=kw.pop("K", 3), **kw)
    w = init_weights(d, 63)
    for k in w:
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    w["ioc/reg/w"] = w["ioc/reg/w"] * 0.05             # small refinements: the second pass re-bins from them (no edge flips vs fp64)
    past, fut, eps, grids, gos = make_case(d, seed=64, n_absent=3)
    vals, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda")
    score = torch.zeros((d.R,), device="cuda")
    h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
    torch.cuda.synchronize()
    assert np.abs(Y.cpu().numpy() - vals["Y"]).max() < 1e-3, np.abs(Y.cpu().numpy() - vals["Y"]).max()
    got = h.train_loss(fut_t.data_ptr())
    assert abs(got["loss"] - float(vals["loss"])) < 2e-4 * max(1.0, abs(float(vals["loss"])))
    bad = {}
    for name in ref:
        if name not in w or name.startswith(("scene_cnn", "temporal", "gauss_head")) or "/bn/" in name:
            continue
        g = h.get_grad(name, w[name].shape)
        if name == "ioc/score/b":
            continue
        e = float(np.abs(g - ref[name]).max() / (np.abs(ref[name]).max() + 1e-12))
        if not e < 5e-4:
            bad[name] = e
    assert not bad, bad


@pytest.mark.parametrize("kw", [dict(), dict(mno=16, n_scenes=3, H=64, L=64), dict(H=16, K=2),
                                dict(bn_mode=2), dict(bn_mode=2, mno=16, n_scenes=3, H=64, L=64), dict(bn_mode=2, H=16, K=2)])
def test_training_through_per_object_batch_norm(kw):
    """dims.bn_mode = 1 in training (frozen statistics only): the backward goes through the reference graph's
    batch-of-one batch-norm -- per-sample, per-channel moments of every CVAE conv layer -- against autograd of the same graph.
    gamma / beta stay constants of the training spec; the conv biases cancel against the mean (their gradient is exactly 0).
    bn_mode = 2 (VERDICT r02 missing 4 / item 8): the same through WHOLE-BATCH statistics -- the reference's literal phase=train
    batch-norm with the objects batched (model/model.py:453-462): the gradient means run over every sample of the call."""
    import torch
    from desire_amd import _lib
    from oracle import desire_torch as OT
    kw = dict(kw)
    d = small_dims(T_obs=5, T_pred=6, n_grids=1, K=kw.pop("K", 3), bn_mode=kw.pop("bn_mode", 1), **kw)
    w = init_weights(d, 65)
    rng = np.random.default_rng(66)
    for k in list(w):
        if k.endswith("/bn/gamma"):
            w[k] = (1.0 + 0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
        if k.endswith("/bn/beta"):
            w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    past, fut, eps, grids, gos = make_case(d, seed=67, n_absent=3)
    vals, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda")
    score = torch.zeros((d.R,), device="cuda")
    h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
    h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
    torch.cuda.synchronize()
    got = h.train_loss(fut_t.data_ptr())
    assert abs(got["loss"] - float(vals["loss"])) < 2e-4 * max(1.0, abs(float(vals["loss"])))
    gscale = max(float(np.abs(ref[n]).max()) for n in ref if n in w and "/bn/" not in n)
    bad = {}
    for name in ref:
        if name not in w or name.startswith(("scene_cnn", "temporal", "gauss_head")) or "/bn/" in name:
            continue
        g = h.get_grad(name, w[name].shape)
        if name == "ioc/score/b":
            continue
        if (name.startswith("vae_enc/conv") or name.startswith("vae_dec/deconv")) and name.endswith("/b"):
            assert np.abs(g).max() < 1e-5 * gscale and np.abs(ref[name]).max() < 1e-9 * max(1.0, gscale), name      # cancels against the mean
            continue
        e = float(np.abs(g - ref[name]).max() / (np.abs(ref[name]).max() + 1e-12))
        if not e < 5e-4:
            bad[name] = e
    assert not bad, bad
    for _ in range(3):                                     # and the optimiser loop runs through it
        h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
        h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
        h.adam_step(0.001)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(Y).all())


def test_train_loop_over_all_eight_sdd_scenes(tmp_path):
    """BASELINE configs[4] as far as the tree allows: desire_amd.train's loop (the reference's train.py loop shape) over a loader that
    holds one slice of EVERY SDD scene (bookstore, coupa, deathCircle, gates, hyang, little, nexus, quad; the arrays the reference's
    own loader produced, tests/golden/loader_mixed8_T20.npz): batches mix windows of different videos, IOC refinement on, the loss
    goes down."""
    import random
    from desire_amd import train as T
    from desire_amd.data_loader import DataLoader
    g = np.load(os.path.join(HERE, "loader_mixed8_T20.npz"))
    frames = [g["data%d" % i] for i in range(8)]
    a = T.build_parser().parse_args(["--batch_size", "8", "--seq_length", "8", "--pred_length", "12", "--max_num_obj", "40",
                                     "--d_dim", "64", "--latent_size", "64", "--num_samples", "4", "--num_epochs", "6",
                                     "--learning_rate", "0.001", "--neighborhood_size", "160", "--save_dir", str(tmp_path / "save")])
    a.img_width, a.img_height = 2000.0, 2000.0
    dl = DataLoader(a.batch_size, a.seq_length + a.pred_length, a.max_num_obj, frames=frames)
    assert dl.num_batches >= 2
    random.seed(0)
    seen = set()
    orig = dl.next_batch_into                              # (the loop's loader thread fills pinned staging: desire_amd/prefetch.py)

    def spy(*args, **kw):
        d = orig(*args, **kw)
        seen.update(int(v) for v in d)
        return d
    dl.next_batch_into = spy
    losses = T.train(a, data_loader=dl, log=lambda l: None)
    assert len(losses) == a.num_epochs * dl.num_batches and np.isfinite(losses).all()
    assert np.mean(losses[-3:]) < np.mean(losses[:3]), losses
    assert seen == set(range(8))                           # every scene's video fed the optimiser


def test_cluster_backward_under_load_is_bitwise_stable():
    """The cluster-form BPTT's hand-off (dpre_r published per reverse step, read by the other members) with every CU busy and each
    workgroup walking several groups: 6 scenes x 128 agents x K = 8, T_pred = 12 -> 192 tiles... run three times: every gradient word
    identical (a missed release / acquire would show up as a flaky word)."""
    import torch
    from desire_amd import _lib
    d = small_dims(n_scenes=12, mno=128, K=8, T_obs=6, T_pred=12, n_grids=1, H=128, L=64)
    w = init_weights(d, 71)
    past, fut, eps, grids, gos = make_case(d, seed=72, n_absent=30)
    h = _lib.Handle(d)
    h.set_weights(w)
    h.set_training(True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past_t, fut_t, eps_t, grids_t = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(grids_t.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda")
    score = torch.zeros((d.R,), device="cuda")
    ref = None
    for rep in range(3):
        h.forward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr(), Y.data_ptr(), score.data_ptr())
        h.backward(past_t.data_ptr(), fut_t.data_ptr(), eps_t.data_ptr())
        torch.cuda.synchronize()
        g = h.grad_tensor().cpu().numpy().copy()
        assert np.isfinite(g).all()
        if ref is None:
            ref = g
            assert np.abs(ref).max() > 0
        else:
            np.testing.assert_array_equal(g, ref)
