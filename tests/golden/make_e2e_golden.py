#!/usr/bin/env python3
"""Build-oracle end-to-end goldens on REAL SDD inputs.

Inputs: the bookstore/video6 CSV slice already committed inside loader_bookstore6_T8.npz (frames
0..159; produced by make_loader_golden.py from the reference's data) run through THIS repo's
DataLoader.  Expected outputs: oracle/desire_oracle.py with seeded weights/eps.  These pin the HIP
path against committed numbers (not only against an oracle recomputed at test time); they are
labelled build-oracle goldens because the reference's model path cannot produce any.

    python tests/golden/make_e2e_golden.py        # writes tests/golden/e2e_cfg0.npz, e2e_cfg1.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from desire_amd.data_loader import DataLoader, frames_from_csv  # noqa: E402
from desire_amd.spec import Dims, init_weights  # noqa: E402
from oracle import desire_oracle as O  # noqa: E402

IMG_W, IMG_H = 1424.0, 1088.0          # SDD bookstore reference frame size (pixels)


def sdd_windows(mno, T_obs, T_pred, n_windows, keep_first=None):
    g = np.load(os.path.join(HERE, "loader_bookstore6_T8.npz"))
    frames, _, _ = frames_from_csv(g["csv"].astype(np.float64), 32)
    dl = DataLoader(batch_size=n_windows, seq_length=T_obs + T_pred - 1, max_num_obj=32, frames=[frames])
    x, y, _ = dl.next_batch(random_update=False)          # x: frames idx..idx+T-1, y: shifted by one
    wins = []
    for xi, yi in zip(x, y):
        full = np.concatenate([xi, yi[-1:]], axis=0)       # T_obs+T_pred consecutive frames, slot-assigned
        if keep_first is not None:                         # configs[0]: first `keep_first` non-zero ids
            present = np.where((full[:, :, 0] != 0).any(axis=0))[0][:keep_first]
            sel = np.zeros((full.shape[0], mno, 3))
            sel[:, :len(present)] = full[:, present]
            full = sel
        else:
            full = full[:, :mno]
        wins.append(full)
    w = np.stack(wins).astype(np.float32)                  # [n, T, mno, 3]
    return np.ascontiguousarray(w[:, :T_obs]), np.ascontiguousarray(w[:, T_obs:])


def make(tag, d, seed, keep_first=None):
    past, fut = sdd_windows(d.mno, d.T_obs, d.T_pred, d.n_scenes, keep_first)
    rng = np.random.default_rng(seed)
    eps = rng.standard_normal((d.R, d.L)).astype(np.float32)
    grids = rng.uniform(-1, 1, (d.n_grids, d.Gh, d.Gw, d.C)).astype(np.float32)
    gos = np.zeros(d.n_scenes, np.int32)
    w = init_weights(d, seed)
    tr = lambda x: np.ascontiguousarray(x.transpose(1, 0, 2, 3).reshape(x.shape[1], -1, 3))
    ref = O.forward(tr(past), tr(fut), eps, grids, gos, w, d)
    pos = ref["Y0"].reshape(d.n_scenes * d.K, d.mno, d.T_pred, 2).transpose(0, 2, 1, 3)
    valid = np.broadcast_to((past[:, -1, :, 0] != 0)[:, None, None, :], (d.n_scenes, d.K, d.T_pred, d.mno))
    margin = O.bin_margin(pos, d.nb_w, d.nb_h, d.grid_size, valid.reshape(pos.shape[:-1]))
    # eps and grids are regenerated from `seed` by the test (np.random.default_rng(seed), same call order)
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}.npz"), past=past, fut=fut, seed=np.int64(seed),
                        dims=np.array([d.n_scenes, d.mno, d.K, d.T_obs, d.T_pred, d.H, d.L]),
                        scale=np.array([d.sx, d.sy, d.nb_w, d.nb_h], np.float64),
                        Hx=ref["Hx"], z_mean=ref["z_mean"], Y0=ref["Y0"],
                        Y=ref["Y"], score=ref["score"], bin_margin=np.float64(margin))
    print(tag, "agents present:", int((past[:, -1, :, 0] != 0).sum()), "of", d.A, "| bin margin %.2e" % margin,
          "| |Y-Y0| max %.3f" % np.abs(ref["Y"] - ref["Y0"]).max())


if __name__ == "__main__":
    nb = 96.0                                              # px: 3x the reference's neighborhood_size=32 so bins are populated
    common = dict(sx=1 / IMG_W, sy=1 / IMG_H, nb_w=nb / IMG_W, nb_h=nb / IMG_H, n_grids=1)
    make("cfg0", Dims(n_scenes=1, mno=4, K=1, T_obs=8, T_pred=12, **common), seed=11, keep_first=4)
    make("cfg1", Dims(n_scenes=1, mno=32, K=20, T_obs=8, T_pred=40, **common), seed=12)
