#!/usr/bin/env python3
"""Generate loader golden vectors by IMPORTING the reference's utils/data_loader.py.

Runs only in the build container (needs /root/reference).  Nothing of the reference's source
travels: the outputs are (a) small column subsets of two SDD annotation CSVs (data files, the
loader's input format, scripts/preprocess.py:30-34) and (b) the arrays the reference's
DataLoader produced from exactly those subsets.

    python tests/golden/make_loader_golden.py

writes tests/golden/loader_<tag>.npz for each case below.
"""
import contextlib
import importlib.util
import io
import os
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# tag, csv (relative to REF/data), frame range kept, DataLoader kwargs, batches to draw
CASES = [
    ("bookstore6_T8", "bookstore/video6", (0, 160), dict(batch_size=4, seq_length=8, max_num_obj=32), 3),
    ("bookstore6_T48", "bookstore/video6", (0, 160), dict(batch_size=2, seq_length=48, max_num_obj=32), 1),
    ("deathcircle2_T8", "deathCircle/video2", (0, 60), dict(batch_size=2, seq_length=8, max_num_obj=70), 2),
    # BASELINE configs[2]: a deathCircle window of T_obs + T_pred = 48 frames; video4 holds 65 track ids in every frame
    ("deathcircle4_T48", "deathCircle/video4", (0, 50), dict(batch_size=1, seq_length=48, max_num_obj=70), 1),
]
ONLY = set(sys.argv[1:])            # optional: tags to (re)generate; default all


def load_ref_loader():
    spec = importlib.util.spec_from_file_location("ref_data_loader", os.path.join(REF, "utils/data_loader.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    mod = load_ref_loader()
    for tag, rel, (f0, f1), kw, nb in CASES:
        if ONLY and tag not in ONLY:
            continue
        raw = np.genfromtxt(os.path.join(REF, "data", rel, "annotations_processed.csv"), delimiter=",")
        keep = (raw[0] >= f0) & (raw[0] < f1)
        sub = raw[:, keep]
        tmp = tempfile.mkdtemp(prefix="desire_golden_")
        cwd = os.getcwd()
        try:
            os.makedirs(os.path.join(tmp, "data", rel))
            np.savetxt(os.path.join(tmp, "data", rel, "annotations_processed.csv"), sub, delimiter=",", fmt="%.1f")
            os.chdir(tmp)
            with contextlib.redirect_stdout(io.StringIO()):
                dl = mod.DataLoader(leave_dataset=1, preprocess=True, **kw)
                xs, ys, ds = [], [], []
                for _ in range(nb):
                    x, y, dv = dl.next_batch(random_update=False)
                    xs.append(np.stack(x)); ys.append(np.stack(y)); ds.append(np.asarray(dv))
            np.savez_compressed(
                os.path.join(HERE, f"loader_{tag}.npz"),
                csv=sub.astype(np.float32),               # halves are exact in fp32
                data0=dl.data[0], frame_list0=np.asarray(dl.frame_list[0]),
                num_obj0=np.asarray(dl.num_obj_list[0]), num_batches=np.int64(dl.num_batches),
                x=np.stack(xs), y=np.stack(ys), d=np.stack(ds),
                frame_pointer=np.int64(dl.frame_pointer), dataset_pointer=np.int64(dl.dataset_pointer),
                kw=np.array([kw["batch_size"], kw["seq_length"], kw["max_num_obj"]]),
            )
            print(tag, "csv cols", sub.shape[1], "data", dl.data[0].shape, "num_batches", dl.num_batches)
        finally:
            os.chdir(cwd)
            shutil.rmtree(tmp)


if __name__ == "__main__":
    sys.exit(main())
