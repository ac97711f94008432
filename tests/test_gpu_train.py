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
