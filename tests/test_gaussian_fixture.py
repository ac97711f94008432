"""The bivariate-Gaussian head of sample() against the REFERENCE's own function: tests/golden/gaussian_head.npz holds
outputs of model/model.py:595-611 `sample_gaussian_2d`, lifted from the file's syntax tree and run in the build container
(tests/golden/make_gaussian_golden.py).  The reference factorises its covariance (:606) by SVD inside
np.random.multivariate_normal, this repo by Cholesky, so for the SAME distribution the same normals give different points;
what is pinned exactly is the distribution and the plumbing:
  * n = L^-1 (point - mean) with L from OUR parameterisation has |n| == |z| for every draw  <=>  L L^T == the reference's
    covariance (an orthogonal change of normals);
  * the sampler fed those n returns min(point, 1.0) (clip of :666-669)."""
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load():
    g = np.load(os.path.join(HERE, "gaussian_head.npz"))
    raw, z, point = g["raw"], g["z"], g["point"]
    mux, muy = raw[:, 0].astype(np.float64), raw[:, 1].astype(np.float64)
    # model/model.py:661-663, float32 like the fetched TF outputs
    sx, sy, rho = (np.exp(raw[:, 2]).astype(np.float64), np.exp(raw[:, 3]).astype(np.float64), np.tanh(raw[:, 4]).astype(np.float64))
    n0 = (point[:, 0] - mux) / sx
    n1 = ((point[:, 1] - muy) / sy - rho * n0) / np.sqrt(1 - rho * rho)
    return raw, z, point, np.stack([n0, n1], -1), rho


def test_parameterisation_has_the_references_covariance():
    raw, z, point, n, rho = _load()
    ratio = np.hypot(n[:, 0], n[:, 1]) / np.hypot(z[:, 0], z[:, 1])
    assert np.abs(ratio - 1).max() < 1e-4, np.abs(ratio - 1).max()
    assert (point.max(axis=0) > 1.0).all()            # the fixture does exercise the clip


def test_oracle_sampler_reproduces_reference_points():
    from oracle import desire_oracle as O
    raw, z, point, n, rho = _load()
    got = O.gaussian_sample(raw, n.astype(np.float32))
    ok = np.abs(rho) < 0.999                           # nearly singular covariances amplify the fp32 rounding of n
    assert np.abs(got - np.minimum(point, 1.0))[ok].max() < 2e-5
    assert np.abs(got - np.minimum(point, 1.0)).max() < 1e-3


@pytest.mark.gpu
def test_device_sampler_reproduces_reference_points():
    import torch
    from desire_amd import _lib
    from tests.helpers import small_dims
    raw, z, point, n, rho = _load()
    h = _lib.Handle(small_dims())
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device="cuda")
    p_t, n_t = t(raw), t(n)
    out = torch.empty((raw.shape[0], 2), device="cuda")
    h.gaussian_sample(p_t.data_ptr(), n_t.data_ptr(), out.data_ptr(), raw.shape[0])
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    ok = np.abs(rho) < 0.999
    assert np.abs(got - np.minimum(point, 1.0))[ok].max() < 2e-5
    assert np.abs(got - np.minimum(point, 1.0)).max() < 1e-3
