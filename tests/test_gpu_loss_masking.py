"""Objects that leave during the prediction window (the reference drops an object from the cost when it does not
exist in the target, model/model.py:351-366).  A target frame whose id is 0 has no ground truth: it is skipped in the
reconstruction / ranking-target / regression terms, their gradients and ADE/FDE; an object in no target frame does not count."""
import numpy as np
import pytest

from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims, to_oracle_layout


def _leaving_case(d, seed):
    past, fut, eps, grids, gos = make_case(d, seed=seed, n_absent=3)
    fut = fut.copy()
    fut[0, 3:, 1] = 0          # scene 0 slot 1 leaves after 3 target frames
    fut[0, 1:, 4] = 0          # slot 4 after one frame
    fut[0, :, 3] = 0           # slot 3 is never in the target: must not count (but still pools: present at the last observed frame)
    fut[1, 2:5, 2] = 0         # scene 1 slot 2 is missing in the middle (track gap)
    fut[1, d.T_pred - 1:, 0] = 0
    return past, fut, eps, grids, gos


def test_oracle_masks_by_hand():
    """CPU: closed-form check of the masked oracle terms on a 1-agent, K=1 case."""
    from oracle import desire_oracle as O
    d = small_dims(n_scenes=1, mno=1, K=1, T_pred=4, n_grids=1)
    Y = np.array([[[1, 0], [2, 0], [3, 0], [4, 0]]], np.float32)
    futn = np.zeros((4, 1, 2), np.float32)
    present = np.array([[True], [True], [False], [False]])
    zm = np.zeros((1, d.L), np.float32)
    kld, recon, cost, n = O.losses(zm, zm, Y, futn, np.array([True]), d, present=present)
    assert n == 1 and abs(recon[0] - 1.5) < 1e-7 and abs(cost - 1.5) < 1e-7
    af = O.ade_fde_k(Y, futn, d, present=present)
    np.testing.assert_allclose(af[0], [1.5, 2.0, 1.5, 2.0])
    _, recon, cost, n = O.losses(zm, zm, Y, futn, np.array([True]), d, present=np.zeros((4, 1), bool))
    assert n == 0 and recon[0] == 0 and cost == 0
    assert np.all(O.ade_fde_k(Y, futn, d, present=np.zeros((4, 1), bool)) == 0)


@pytest.mark.gpu
def test_losses_ade_fde_and_gradients_with_leaving_objects():
    import torch
    from desire_amd import _lib
    from oracle import desire_oracle as O
    from oracle import desire_torch as OT
    d = small_dims(n_scenes=2, mno=8, K=3, T_obs=6, T_pred=7, n_grids=1, H=64)
    w = init_weights(d, 51)
    for k in w:
        if k.startswith("vae_dec/") and k.endswith("/w"):
            w[k] = w[k] * 3
    w["mask_fc/w"] = w["mask_fc/w"] * 20
    w["head/w"] = w["head/w"] * 4
    w["ioc/score/w"] = w["ioc/score/w"] * 3
    past, fut, eps, grids, gos = _leaving_case(d, 52)
    po, fo = to_oracle_layout(past), to_oracle_layout(fut)
    vals, ref = OT.loss_and_grads(po, fo, eps, grids, gos, w, d)
    present = fo[:, :, 0] != 0
    valid = po[d.T_obs - 1, :, 0] != 0
    counted = valid & present.any(0)
    assert counted.sum() == valid.sum() - 1            # slot 3 of scene 0 dropped
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
    n = counted.sum()
    assert got["n_present"] == n
    for key in ("recon", "kld", "ce", "reg"):
        want = float((vals[key] * counted).sum() / n)
        assert abs(got[key] - want) <= 3e-5 * max(1.0, abs(want)), (key, got[key], want)
    assert abs(got["loss"] - float(vals["loss"])) < 1e-4 * max(1.0, abs(float(vals["loss"])))
    for name in ("head/w", "dec/gates/kernel", "ioc/reg/w", "ioc/score/w", "ioc/gates/kernel", "vae_enc/fc/w", "fc_c/w",
                 "enc_x/candidate/kernel", "enc_y/gates/kernel", "vae_dec/deconv2/w", "ioc/social_fc/w"):
        g = h.get_grad(name, w[name].shape)
        e = float(np.abs(g - ref[name]).max() / (np.abs(ref[name]).max() + 1e-12))
        assert e < 3e-4, (name, e)
    # inference-side cost and the ADE / FDE harness
    kld = torch.zeros(d.A, device="cuda"); recon = torch.zeros(d.A, device="cuda"); cost = torch.zeros(2, device="cuda")
    Y0 = t(h.read_buffer("Y0", (d.R, d.T_pred, 2)))
    h.losses(fut_t.data_ptr(), Y0.data_ptr(), kld.data_ptr(), recon.data_ptr(), cost.data_ptr())
    futn = O.normalise(fo, d)
    zm, zl = h.read_buffer("z_mean", (d.A, d.L)), h.read_buffer("z_log_sigma_sq", (d.A, d.L))
    k_ref, r_ref, c_ref, n_ref = O.losses(zm, zl, Y0.cpu().numpy(), futn, valid, d, present=present)
    assert np.abs(recon.cpu().numpy() - r_ref).max() < 1e-5
    assert abs(float(cost[0]) - c_ref) < 1e-4 * max(1.0, abs(c_ref)) and int(cost[1]) == n_ref == n
    af = torch.zeros((d.A, 4), device="cuda")
    h.ade_fde(Y.data_ptr(), fut_t.data_ptr(), af.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_allclose(af.cpu().numpy(), O.ade_fde_k(Y.cpu().numpy(), futn, d, present=present), atol=1e-5)
