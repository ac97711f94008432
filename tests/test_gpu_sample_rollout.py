"""DESIREModel.sample() as the reference's autoregressive rollout (model/model.py:613-688):
layout AND step semantics against oracle.rollout (a restatement of :623-681) with injected normals; sample() follows the
trained weights; an observation length other than seq_length works."""
import argparse

import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout

pytestmark = pytest.mark.gpu


def _args(**kw):
    a = argparse.Namespace(rnn_size=512, num_layers=1, batch_size=2, seq_length=8, pred_length=12, d_dim=64, e_dim=256,
                           latent_size=64, max_num_obj=16, learning_rate=0.001, grad_clip=10.0, stride=1,
                           neighborhood_size=300, grid_size=4, num_samples=3, img_width=1400.0, img_height=1100.0)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("H", [64, 16])
def test_rollout_matches_oracle_step_by_step(H):
    import torch
    from desire_amd import _lib
    from oracle import desire_oracle as O
    d = small_dims(n_scenes=3, mno=16, K=1, H=H, posterior=0, n_grids=1)
    w = init_weights(d, 71)
    w["gauss_head/b"] = np.array([0.45, 0.5, -3.0, -3.5, 0.3], np.float32)      # sigma ~ 0.05 / 0.03 of the frame
    past, _, _, _, _ = make_case(d, seed=72, n_absent=2)
    num = 9
    normals = np.random.default_rng(73).standard_normal((num, d.A, 2)).astype(np.float32)
    normals[3] += 12.0                                   # forces the clip at 1.0 for a step
    ref = O.rollout(to_oracle_layout(past), w, d, normals)
    assert (ref == 1.0).any() and (ref < 1.0).any()
    h = _lib.Handle(d)
    h.set_weights(w)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    past_t, n_t = t(past), t(normals)
    out = torch.zeros((num, d.A, 2), device="cuda")
    h.rollout(past_t.data_ptr(), n_t.data_ptr(), num, out.data_ptr())
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    # the Hx the forward path computes is untouched by a rollout (own state buffer)
    h.encode(past_t.data_ptr(), 0)
    hx0 = h.read_buffer("Hx", (d.A, d.H))
    h.rollout(past_t.data_ptr(), n_t.data_ptr(), num, out.data_ptr())
    assert np.array_equal(h.read_buffer("Hx", (d.A, d.H)), hx0)


def test_sample_layout_and_semantics_through_the_model():
    from desire_amd.model import DESIREModel
    from oracle import desire_oracle as O
    args = _args()
    m = DESIREModel(args, seed=4)
    d = small_dims(n_scenes=1, mno=16, K=3, H=64, L=64, T_obs=8, T_pred=12, posterior=0, n_grids=1, sx=1 / 1400.0, sy=1 / 1100.0)
    past, fut, _, _, _ = make_case(d, seed=74, n_absent=3)
    traj = past[0].astype(np.float64)                    # [8, 16, 3]
    truth = np.concatenate([past[0], fut[0]]).astype(np.float64)
    normals = np.random.default_rng(75).standard_normal((10, 16, 2)).astype(np.float32)
    out = m.sample(None, traj, None, (1400.0, 1100.0), truth, num=10, mode="rollout", normals=normals)
    assert out.shape == (18, 16, 3)
    np.testing.assert_array_equal(out[:8], traj)
    np.testing.assert_array_equal(out[8:, :, 0], np.broadcast_to(traj[-1, :, 0], (10, 16)))       # ids carried (:680)
    w = m.sync_weights()
    ref = O.rollout(traj[:, None].reshape(8, 16, 3), w, d, normals)                                # [10, 16, 2] normalised
    np.testing.assert_allclose(out[8:, :, 1], ref[..., 0] * 1400.0, atol=0.05)
    np.testing.assert_allclose(out[8:, :, 2], ref[..., 1] * 1100.0, atol=0.05)
    assert (out[8:, :, 1] <= 1400.0 + 1e-6).all() and (out[8:, :, 2] <= 1100.0 + 1e-6).all()     # clip at 1.0 normalised (:666-669)
    # a different observation length than seq_length (temporal/w is sized by seq_length and unused here)
    out5 = m.sample(None, traj[3:], None, (1400.0, 1100.0), truth, num=4, mode="rollout", normals=normals[:4])
    assert out5.shape == (9, 16, 3) and np.isfinite(out5).all()
    # the IOC mode keeps the round-1 behaviour: absent objects stay zero rows
    oi = m.sample(None, traj, None, (1400.0, 1100.0), truth, num=10, mode="ioc")
    assert oi.shape == (18, 16, 3) and (oi[8:][:, traj[-1, :, 0] == 0] == 0).all()


def test_sample_follows_the_trained_weights():
    """after train_step the weights live on the device; sample() must use them, not the initial host copy."""
    from desire_amd.model import DESIREModel
    args = _args()
    m = DESIREModel(args, seed=5)
    d = small_dims(n_scenes=2, mno=16, K=3, H=64, L=64, T_obs=8, T_pred=12, n_grids=1)
    past, fut, _, _, _ = make_case(d, seed=76, n_absent=3)
    x = [p.astype(np.float64) for p in past]
    y = [f.astype(np.float64) for f in fut]
    normals = np.random.default_rng(77).standard_normal((6, 16, 2)).astype(np.float32)
    truth = np.concatenate([x[0], y[0]])
    before = m.sample(None, x[0], None, (1400.0, 1100.0), truth, num=6, mode="rollout", normals=normals)
    before_ioc = m.sample(None, x[0], None, (1400.0, 1100.0), truth, num=6, mode="ioc", seed=1)
    for _ in range(3):
        m.train_step(x, y, seed=0)
    after = m.sample(None, x[0], None, (1400.0, 1100.0), truth, num=6, mode="rollout", normals=normals)
    after_ioc = m.sample(None, x[0], None, (1400.0, 1100.0), truth, num=6, mode="ioc", seed=1)
    # the rollout reads the X-encoder GRU (trained through Hx); the IOC mode reads everything
    assert np.abs(after[8:] - before[8:]).max() > 1e-3
    assert np.abs(after_ioc[8:] - before_ioc[8:]).max() > 1e-3
    # and a model restored from the saved checkpoint gives the same samples
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        m.save(os.path.join(td, "w.npz"))
        m2 = DESIREModel.restore(args, os.path.join(td, "w.npz"))
        again = m2.sample(None, x[0], None, (1400.0, 1100.0), truth, num=6, mode="rollout", normals=normals)
    np.testing.assert_allclose(again, after, atol=1e-3)


def test_checkpoint_keeps_the_optimiser_state():
    """save()/restore() carry the Adam moments and step counter, so a resumed run continues the same trajectory."""
    import os
    import tempfile
    from desire_amd.model import DESIREModel
    args = _args()
    d = small_dims(n_scenes=2, mno=16, K=3, H=64, L=64, T_obs=8, T_pred=12, n_grids=1)
    past, fut, _, _, _ = make_case(d, seed=78, n_absent=3)
    x = [p.astype(np.float64) for p in past]
    y = [f.astype(np.float64) for f in fut]
    m = DESIREModel(args, seed=6)
    for _ in range(3):
        m.train_step(x, y, seed=0)
    with tempfile.TemporaryDirectory() as td:
        m.save(os.path.join(td, "c.npz"))
        m2 = DESIREModel.restore(args, os.path.join(td, "c.npz"))
    a = m.train_step(x, y, seed=0)["loss"]
    b = m2.train_step(x, y, seed=0)["loss"]
    assert abs(a - b) < 1e-6 * max(1.0, abs(a))
    wa, wb = m.sync_weights(), m2.sync_weights()           # after the 4th step: identical only if moments and t were restored
    for k in ("dec/gates/kernel", "ioc/social_fc/w", "vae_dec/deconv2/w", "enc_x/candidate/bias"):
        assert np.abs(wa[k] - wb[k]).max() < 1e-7, k
    assert m2._trained.opt_state()["t"][0] == 4


def test_default_mode_is_the_trained_path_and_old_checkpoints_load():
    """with the default loss train_step never touches gauss_head/*, so the DEFAULT sample() of a model whose head was
    neither supplied nor trained is the IOC path (one warning says so; the mode is never inferred from `normals`); an explicit rollout
    on such a head warns.  (low) an archive written before
    gauss_head/* existed still restores (the head is filled with its initial values, the optimiser state is dropped)."""
    import os
    import tempfile
    import warnings
    from desire_amd.formats import load_weights, save_weights
    from desire_amd.model import DESIREModel
    args = _args()
    m = DESIREModel(args, seed=7)
    d = small_dims(n_scenes=1, mno=16, K=3, H=64, L=64, T_obs=8, T_pred=12, posterior=0, n_grids=1)
    past, fut, _, _, _ = make_case(d, seed=79, n_absent=3)
    traj = past[0].astype(np.float64)
    truth = np.concatenate([past[0], fut[0]]).astype(np.float64)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        a = m.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, seed=3)
        m.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, seed=3)
    assert sum("mode='ioc'" in str(r.message) for r in rec) == 1          # said once
    b = m.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="ioc", seed=3)
    np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        m.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, normals=np.zeros((6, 16, 2), np.float32))
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        m.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="rollout")
    assert any("gauss_head" in str(r.message) for r in rec)
    with tempfile.TemporaryDirectory() as td:
        m.save(os.path.join(td, "new.npz"))
        blob = load_weights(os.path.join(td, "new.npz"))
        old = {k: v for k, v in blob.items() if not k.startswith("gauss_head/")}
        save_weights(os.path.join(td, "old.npz"), old)
        m2 = DESIREModel.restore(args, os.path.join(td, "old.npz"))
        c = m2.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="ioc", seed=3)
        np.testing.assert_allclose(c, a, atol=1e-3)
        # save() always writes gauss_head/*, so an archive of an UNTRAINED head must not flip sample()'s default to the rollout:
        # the archive carries an explicit marker (meta/head_trained), and this model never trained or received a head
        assert float(blob["meta/head_trained"][0]) == 0.0
        m3 = DESIREModel.restore(args, os.path.join(td, "new.npz"))
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            e = m3.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, seed=3)
            m3.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="rollout")
        assert any("mode='ioc'" in str(r.message) for r in rec) and any("random initial values" in str(r.message) for r in rec)
        np.testing.assert_allclose(e, a, atol=1e-3)                           # the same default as before the save / restore round trip
        # a head that CAME WITH the weights (caller-supplied): the marker is set, survives save -> restore, and the reference-compatible rollout is the default
        given = {k: v for k, v in blob.items() if not k.startswith(("opt/", "meta/"))}
        m4 = DESIREModel(args, weights=given)
        m4.save(os.path.join(td, "given.npz"))
        assert float(load_weights(os.path.join(td, "given.npz"))["meta/head_trained"][0]) == 1.0
        m5 = DESIREModel.restore(args, os.path.join(td, "given.npz"))
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            m5.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="rollout")
        assert not any("gauss_head" in str(r.message) for r in rec)
        # an archive of rounds 2-4 (head inside, no marker): the rule it was written under still applies -- the head counts as given
        save_weights(os.path.join(td, "legacy.npz"), {k: v for k, v in blob.items() if not k.startswith("meta/")})
        m6 = DESIREModel.restore(args, os.path.join(td, "legacy.npz"))
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            m6.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="rollout")
        assert m6._head_given and not any("gauss_head" in str(r.message) for r in rec)
        nrm = np.random.default_rng(1).standard_normal((6, 16, 2)).astype(np.float32)
        np.testing.assert_array_equal(m5.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, normals=nrm),
                                      m5.sample(None, traj, None, (1400.0, 1100.0), truth, num=6, mode="rollout", normals=nrm))


def test_ref_compat_handle_with_its_own_prediction_length():
    """forward_ref_compat forces T_pred = T_obs; the model's own pred_length may differ and either call may
    come first -- the T_pred-sized IOC regression head is re-fitted per shape instead of failing in set_weights."""
    import torch
    from desire_amd.model import DESIREModel
    args = _args()
    args.d_dim, args.seq_length, args.pred_length = 16, 8, 12
    for first in ("ref", "normal"):
        m = DESIREModel(args, seed=8)
        d = small_dims(n_scenes=2, mno=16, K=3, H=16, L=64, T_obs=8, T_pred=12, n_grids=1)
        past, fut, _, _, _ = make_case(d, seed=80, n_absent=3)
        x = [p.astype(np.float64) for p in past]
        y8 = [f[:8].astype(np.float64) for f in fut]
        y12 = [f.astype(np.float64) for f in fut]
        if first == "ref":
            out = m.forward_ref_compat(x, y8, seed=1)
            Y, _ = m.forward(x, y12, seed=1)
        else:
            Y, _ = m.forward(x, y12, seed=1)
            out = m.forward_ref_compat(x, y8, seed=1)
        assert Y.shape[-2] == 12 and bool(torch.isfinite(Y).all())
        assert out["output_states"].shape[-2] == 8 and bool(torch.isfinite(out["output_states"]).all())
        assert m._weights["ioc/reg/w"].shape == (16, 24)
