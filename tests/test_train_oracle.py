"""Training oracle (torch autograd, CPU): its forward must equal the numpy oracle; its gradients are checked by
finite differences on a few weights; Adam follows the TF formula."""
import numpy as np
import pytest

from desire_amd.spec import init_weights
from oracle import desire_oracle as O
from oracle import desire_torch as OT
from tests.helpers import make_case, small_dims, to_oracle_layout


@pytest.fixture(scope="module")
def case():
    d = small_dims(n_scenes=1, mno=8, K=3, T_obs=4, T_pred=5, n_grids=1, nb_w=0.5, nb_h=0.5)
    w = init_weights(d, 5)
    past, fut, eps, grids, gos = make_case(d, seed=6, n_absent=2)
    return d, w, to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos


def test_torch_forward_equals_numpy_oracle(case):
    d, w, past, fut, eps, grids, gos = case
    ref = O.forward(past, fut, eps, grids, gos, w, d, dt=np.float64)
    vals, grads = OT.loss_and_grads(past, fut, eps, grids, gos, w, d)
    for k in ("Hx", "Hy", "z_mean", "z", "xhat", "xz", "Y0", "Y", "score"):
        assert np.abs(vals[k] - ref[k]).max() < 1e-9, k
    valid = past[d.T_obs - 1, :, 0] != 0
    kld, recon, cost, n = O.losses(ref["z_mean"], ref["z_log_sigma_sq"], ref["Y0"], O.normalise(fut, d, np.float64), valid, d, np.float64)
    assert abs(float(vals["L_sgm"]) - cost) < 1e-9
    assert set(grads) == set(w) and all(np.isfinite(g).all() for g in grads.values())
    # the IOC module only reaches the sample-generation weights through Hx; the decoder gets no IOC gradient
    assert np.abs(grads["ioc/reg/w"]).max() > 0 and np.abs(grads["dec/gates/kernel"]).max() > 0


def test_autograd_matches_finite_differences(case):
    d, w, past, fut, eps, grids, gos = case
    vals, grads = OT.loss_and_grads(past, fut, eps, grids, gos, w, d)
    rng = np.random.default_rng(0)
    fixed = {"Yd": vals["Yd"], "dmax": vals["dmax"]}           # the stop-gradient quantities stay put
    def loss_of(wn):
        wt = {k: OT._t(v) for k, v in wn.items()}
        return float(OT.forward_loss(past, fut, eps, grids, gos, wt, d, fixed=fixed)["loss"])
    for name in ("head/w", "dec/candidate/kernel", "vae_dec/deconv2/w", "vae_enc/conv2/w", "enc_x/gates/kernel",
                 "ioc/social_fc/w", "ioc/gates/kernel", "ioc/vel_fc/w", "mask_fc/w"):
        idx = tuple(rng.integers(0, s) for s in w[name].shape)
        h = 1e-5
        wp = {k: v.astype(np.float64).copy() for k, v in w.items()}
        wm = {k: v.astype(np.float64).copy() for k, v in w.items()}
        wp[name][idx] += h; wm[name][idx] -= h
        fd = (loss_of(wp) - loss_of(wm)) / (2 * h)
        assert abs(fd - grads[name][idx]) < 1e-6 + 1e-4 * abs(fd), (name, idx, fd, grads[name][idx])


def test_adam_step_tf_formula():
    w, g = np.array([1.0, -2.0]), np.array([0.5, -0.25])
    m = v = np.zeros(2)
    w1, m1, v1 = OT.adam_step(w, g, m, v, 1, lr=0.005)
    np.testing.assert_allclose(w1, w - 0.005 * np.sign(g) * (1 / (1 + 1e-8 / np.sqrt(1 - 0.999) / np.abs(g))), rtol=1e-6)
    assert np.allclose(m1, 0.1 * g) and np.allclose(v1, 0.001 * g * g)


def test_head_nll_torch_graph_equals_the_reference_formula():
    """oracle.head_nll evaluates model/model.py:494-550 literally (pdf, max(pdf, 1e-20), -log, sum) on the X encoder's observed steps;
    desire_torch.head_nll is the same quantity in log form (what the kernels compute) -- equal where the pdf does not underflow."""
    import torch
    from oracle import desire_oracle as O, desire_torch as OT
    from desire_amd.spec import init_weights
    from tests.helpers import make_case, small_dims, to_oracle_layout
    d = small_dims(n_scenes=2, mno=8, K=2, T_obs=5, T_pred=6, H=64, L=64)
    w = init_weights(d, 3)
    w["gauss_head/b"] = np.array([0.3, 0.6, -0.5, -0.7, 0.2], np.float32)
    past, fut, _, _, _ = make_case(d, seed=4, n_absent=3)
    p, f = to_oracle_layout(past), to_oracle_layout(fut)
    a, n = O.head_nll(p, f, w, d)
    b, nb = OT.head_nll(p, f, {k: torch.as_tensor(np.asarray(v), dtype=torch.float64) for k, v in w.items()}, d)
    assert n == nb > 0 and abs(a - float(b)) < 1e-10 * max(1.0, abs(a))
    # hand check of one pair against scipy's bivariate normal
    from scipy.stats import multivariate_normal
    mux, muy, sx, sy, rho = 0.2, -0.1, 0.5, 0.8, 0.3
    cov = [[sx * sx, rho * sx * sy], [rho * sx * sy, sy * sy]]
    want = -np.log(multivariate_normal.pdf([0.4, 0.3], [mux, muy], cov))
    got = O.reconstr_loss(np.array([mux]), np.array([muy]), np.array([sx]), np.array([sy]), np.array([rho]), np.array([0.4]), np.array([0.3]))
    assert abs(got - want) < 1e-12
