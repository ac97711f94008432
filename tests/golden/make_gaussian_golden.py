#!/usr/bin/env python3
"""Generate the N3 fixture by RUNNING the reference's own `DESIREModel.sample_gaussian_2d`
(/root/reference/model/model.py:595-611) -- the one piece of the model file that is plain numpy.

The module itself cannot be imported (tensorflow / prettytensor / ipdb at :16-21 are not installable here), so the
function is lifted out of the file's syntax tree at generation time (ast.parse -> the FunctionDef node -> compile) and
called with self = None; nothing of the source text is kept.  Runs only in the build container.

    python tests/golden/make_gaussian_golden.py      -> tests/golden/gaussian_head.npz

Per draw i the fixture holds
  raw[i]    the 5 raw head outputs (mux, muy, log sx, log sy, atanh-ish corr) before :661-663's exp / exp / tanh
  z[i]      the two N(0,1) numbers np.random.multivariate_normal consumed for that draw (same seed, same call shape (1, 2))
  point[i]  what the reference function returned for np.random.seed(seed[i])   (NOT clipped: the clip is :666-669)
The reference factorises the covariance by SVD, this repo's sampler by Cholesky, so the same z gives different points of
the same distribution; tests/test_gaussian_fixture.py checks the distribution exactly: with L the Cholesky factor of
the covariance OUR parameterisation builds, n = L^-1 (point - mean) must have |n| == |z| for every draw (<=> L L^T equals
the reference's covariance at :606), and the sampler fed n must return min(point, 1).
"""
import ast
import os

import numpy as np

REF = "/root/reference/model/model.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def lift(name="sample_gaussian_2d"):
    tree = ast.parse(open(REF).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef) and node.name == name:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = {"np": np}
            exec(compile(mod, REF, "exec"), ns)
            return ns[name]
    raise SystemExit("function not found: " + name)


def main():
    fn = lift()
    rng = np.random.default_rng(20260928)
    n = 512
    raw = np.zeros((n, 5), np.float32)
    raw[:, 0:2] = rng.uniform(-0.2, 1.2, (n, 2))             # means around the normalised frame (some beyond 1.0: clip path)
    raw[:, 2:4] = rng.uniform(-4.0, 0.5, (n, 2))             # log std devs
    raw[:, 4] = rng.uniform(-2.5, 2.5, n)                    # pre-tanh correlation (|rho| up to 0.987)
    raw[:8, 4] = [0.0, 0.0, 4.0, -4.0, 1e-3, -1e-3, 3.0, -3.0]   # uncorrelated and nearly degenerate cases
    seeds = (1000 + np.arange(n)).astype(np.int64)
    z = np.zeros((n, 2), np.float64)
    point = np.zeros((n, 2), np.float64)
    for i in range(n):
        # model/model.py:661-663: mux, muy, exp(sx), exp(sy), tanh(corr) -- evaluated in float32 like the fetched TF outputs
        mux, muy = raw[i, 0], raw[i, 1]
        sx, sy, rho = np.exp(raw[i, 2]), np.exp(raw[i, 3]), np.tanh(raw[i, 4])
        np.random.seed(int(seeds[i]))
        point[i] = fn(None, mux, muy, sx, sy, rho)
        np.random.seed(int(seeds[i]))
        z[i] = np.random.standard_normal((1, 2))[0]
    out = os.path.join(HERE, "gaussian_head.npz")
    np.savez_compressed(out, raw=raw, z=z, point=point, seed=seeds)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
