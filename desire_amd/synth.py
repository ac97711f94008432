"""Seeded synthetic inputs in the loader layout (SDD-like tracks); used by bench.py and the tests."""
import numpy as np

from .spec import Dims


def make_case(d: Dims, seed: int = 0, n_absent: int = 3, img=(1400.0, 1100.0), spread=0.35):
    """Loader-layout windows: past [n_scenes, T_obs, mno, 3], fut [n_scenes, T_pred, mno, 3] in PIXELS,
    eps [R, L], scene grids, grid_of_scene.  Agents random-walk around a scene centre so that the
    social windows are populated; `n_absent` slots per scene are padding (id 0, zero rows)."""
    rng = np.random.default_rng(seed)
    W, Hh = img
    T = d.T_obs + d.T_pred
    centre = rng.uniform(0.35, 0.65, (d.n_scenes, 1, 1, 2))
    start = centre + rng.uniform(-spread, spread, (d.n_scenes, 1, d.mno, 2)) * 0.5
    vel = rng.normal(0, 0.004, (d.n_scenes, 1, d.mno, 2))
    steps = np.arange(T).reshape(1, T, 1, 1)
    pos = start + vel * steps + rng.normal(0, 0.0015, (d.n_scenes, T, d.mno, 2)).cumsum(axis=1)
    pos = np.clip(pos, 0.01, 0.99)
    px = np.round(pos * np.array([W, Hh]) * 2) / 2            # SDD centres are multiples of 0.5
    frames = np.zeros((d.n_scenes, T, d.mno, 3), np.float32)
    frames[..., 0] = np.arange(1, d.mno + 1)
    frames[..., 1:] = px
    if n_absent:
        frames[:, :, d.mno - n_absent:, :] = 0
    past = np.ascontiguousarray(frames[:, :d.T_obs])
    fut = np.ascontiguousarray(frames[:, d.T_obs:])
    eps = rng.standard_normal((d.R, d.L)).astype(np.float32)
    grids = rng.uniform(-1, 1, (d.n_grids, d.Gh, d.Gw, d.C)).astype(np.float32)
    gos = (np.arange(d.n_scenes) % d.n_grids).astype(np.int32)
    return past, fut, eps, grids, gos
