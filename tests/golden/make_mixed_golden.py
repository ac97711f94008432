#!/usr/bin/env python3
"""A MIXED-scene loader golden (BASELINE configs[4]: "Full SDD mixed-scene training loop"): 30-frame slices of one video from
EACH of the eight SDD scenes, loaded and batched by the reference's own DataLoader in one go (train.py:99-100 loads several CSVs
and next_batch walks over them, utils/data_loader.py:88-92,185-258).  Runs only in the build container.

    python tests/golden/make_mixed_golden.py   -> tests/golden/loader_mixed8_T20.npz

Stored: per video (in the ORDER the reference loaded them -- it walks directories in raw os.walk order) the scene name, the CSV
column subset and the preprocessed array, plus the batches next_batch(random_update=False) returned.
"""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_loader_golden import HERE, REF, load_ref_loader  # noqa: E402

VIDEOS = [("bookstore", "video4"), ("coupa", "video0"), ("deathCircle", "video2"), ("gates", "video0"),
          ("hyang", "video1"), ("little", "video0"), ("nexus", "video0"), ("quad", "video0")]
FRAMES, KW = 30, dict(batch_size=8, seq_length=20, max_num_obj=40)


def main():
    mod = load_ref_loader()
    tmp = tempfile.mkdtemp(prefix="desire_golden_mixed_")
    cwd = os.getcwd()
    subs = {}
    try:
        for scene, vid in VIDEOS:
            raw = np.genfromtxt(os.path.join(REF, "data", scene, vid, "annotations_processed.csv"), delimiter=",")
            f0 = raw[0].min()
            sub = raw[:, raw[0] < f0 + FRAMES]
            os.makedirs(os.path.join(tmp, "data", scene, vid))
            np.savetxt(os.path.join(tmp, "data", scene, vid, "annotations_processed.csv"), sub, delimiter=",", fmt="%.1f")
            subs[scene] = sub
        os.chdir(tmp)
        with contextlib.redirect_stdout(io.StringIO()):
            dl = mod.DataLoader(leave_dataset=len(VIDEOS), preprocess=True, **KW)
            xs, ys, ds = [], [], []
            for _ in range(dl.num_batches):
                x, y, dv = dl.next_batch(random_update=False)
                xs.append(np.stack(x)); ys.append(np.stack(y)); ds.append(np.asarray(dv))
        # which scene is video i of the reference's list?  match the preprocessed arrays against each CSV's own first frame
        order = []
        for i in range(len(dl.data)):
            ids0 = set(dl.data[i][0, :, 0][dl.data[i][0, :, 0] != 0].tolist()) | {0.0}
            x0 = dl.data[i][0, 0, 1]
            hit = [s for s, sub in subs.items() if abs(sub[2][sub[0] == sub[0].min()][0] - x0) < 1e-9 and dl.data[i].shape[0] == np.unique(sub[0]).size
                   and set(sub[1][sub[0] == sub[0].min()].tolist()) | {0.0} == ids0]
            assert len(hit) == 1, (i, hit)
            order.append(hit[0])
        out = dict(num_batches=np.int64(dl.num_batches), x=np.stack(xs), y=np.stack(ys), d=np.stack(ds),
                   kw=np.array([KW["batch_size"], KW["seq_length"], KW["max_num_obj"]]), order=np.array(order))
        for i, s in enumerate(order):
            out["csv%d" % i] = subs[s].astype(np.float32)
            out["data%d" % i] = dl.data[i]
        path = os.path.join(HERE, "loader_mixed8_T20.npz")
        np.savez_compressed(path, **out)
        print("order", order, "num_batches", dl.num_batches, "x", out["x"].shape, os.path.getsize(path), "bytes")
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
