#!/usr/bin/env python3
"""Randomised comparison of desire_amd.data_loader with the reference's utils/data_loader.py, IMPORTED from /root/reference
(build container only; nothing of the reference travels).  Random annotation tables -- id 0 present, ids entering and
leaving, duplicate ids in a frame, frames with too many objects, windows with too many unique ids -- go through both loaders;
arrays must be equal bit for bit and exceptions must have the same type.

    python tests/golden/fuzz_loader_vs_reference.py [N] [SEED]
"""
import contextlib
import io
import os
import random
import shutil
import signal
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.golden.make_loader_golden import load_ref_loader  # noqa: E402


def random_table(rng):
    n_frames = int(rng.integers(6, 40))
    n_ids = int(rng.integers(1, 14))
    ids = rng.choice(np.arange(0, 30), n_ids, replace=False)
    cols = []
    for f in range(n_frames):
        for i in ids:
            if rng.random() < 0.7:
                cols.append((f, i, np.round(rng.uniform(3, 1900) * 2) / 2, np.round(rng.uniform(3, 1900) * 2) / 2))
                if rng.random() < 0.003:                      # a duplicated id inside one frame
                    cols.append((f, i, np.round(rng.uniform(3, 1900) * 2) / 2, np.round(rng.uniform(3, 1900) * 2) / 2))
    if not cols:
        cols.append((0, 1, 10.0, 10.0))
    return np.asarray(cols, np.float64).T, n_ids


class Hang(Exception):
    pass


def _alarm(sig, frm):
    raise Hang()


def run(make, kw, nb, tmp, seed):
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            dl = make(**kw)
            out = [np.asarray(dl.data[0]), np.asarray(dl.frame_list[0]), np.asarray(dl.num_obj_list[0]), np.int64(dl.num_batches)]
            for _ in range(nb):
                x, y, d = dl.next_batch(random_update=bool(seed & 1))
                out += [np.stack(x), np.stack(y), np.asarray(d)]
            out += [np.int64(dl.frame_pointer), np.int64(dl.dataset_pointer)]
        return out
    finally:
        os.chdir(cwd)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    ref = load_ref_loader()
    from desire_amd.data_loader import DataLoader
    same = raised = hung = 0
    for it in range(n):
        tab, n_ids = random_table(rng)
        mno = int(rng.integers(2, 16)) if rng.random() < 0.25 else n_ids + 1 + int(rng.integers(0, 6))   # mostly roomy enough
        kw = dict(batch_size=int(rng.integers(1, 5)), seq_length=int(rng.integers(1, 9)), max_num_obj=mno,
                  leave_dataset=1, preprocess=True)
        nb = int(rng.integers(1, 4))
        res = []
        for which in ("ref", "ours"):
            tmp = tempfile.mkdtemp(prefix="desire_fuzz_")
            try:
                os.makedirs(os.path.join(tmp, "data", "scene", "video0"))
                np.savetxt(os.path.join(tmp, "data", "scene", "video0", "annotations_processed.csv"), tab, delimiter=",", fmt="%.1f")
                signal.signal(signal.SIGALRM, _alarm)
                signal.alarm(10)
                try:
                    if which == "ref":
                        res.append(("ok", run(ref.DataLoader, kw, nb, tmp, it)))
                    else:
                        res.append(("ok", run(lambda **k: DataLoader(data_dir=os.path.join(tmp, "data") + "/", **k), kw, nb, tmp, it)))
                except Hang:                                  # the reference spins forever when no window of seq_length+1 frames exists
                    res.append(("hang", which))
                except Exception as ex:                       # noqa: BLE001
                    res.append(("raise", type(ex).__name__))
                finally:
                    signal.alarm(0)
            finally:
                shutil.rmtree(tmp)
        (ka, a), (kb, b) = res
        if ka == "hang":                                      # nothing to compare against; ours must not hang too
            assert kb != "hang", (it, kw)
            hung += 1
            continue
        if ka != kb:
            print("case %d: reference %s / ours %s  kw=%s" % (it, (ka, a if ka == "raise" else ""), (kb, b if kb == "raise" else ""), kw))
            return 1
        if ka == "raise":
            assert a == b, (it, a, b, kw)
            raised += 1
            continue
        assert len(a) == len(b)
        for u, v in zip(a, b):
            assert u.dtype == v.dtype and u.shape == v.shape and np.array_equal(u, v), (it, kw, u.shape, v.shape)
        same += 1
    # on-disk interchange: the pickle desire_amd.formats.write_cpkl produces, read by the REFERENCE's load_preprocessed / next_batch
    from desire_amd.formats import write_cpkl
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "loader_bookstore6_T8.npz"))
    tmp = tempfile.mkdtemp(prefix="desire_fuzz_")
    try:
        pk = os.path.join(tmp, "trajectories.cpkl")
        write_cpkl(pk, [g["data0"]], [g["frame_list0"].tolist()], [g["num_obj0"].tolist()])
        rdl = object.__new__(ref.DataLoader)                  # (its __init__ always re-preprocesses and overwrites the pickle)
        rdl.batch_size, rdl.seq_length, rdl.max_num_obj = 4, 8, 32
        with contextlib.redirect_stdout(io.StringIO()):
            rdl.load_preprocessed(pk)
            rdl.reset_batch_pointer()
            x, y, _ = rdl.next_batch(random_update=False)
        assert np.array_equal(np.stack(x), g["x"][0]) and np.array_equal(np.stack(y), g["y"][0]) and rdl.num_batches == int(g["num_batches"])
        print("reference loader reads the pickle written by desire_amd.formats.write_cpkl: batches identical")
    finally:
        shutil.rmtree(tmp)
    print("loader fuzz: %d cases identical, %d raised the same exception type in both, %d where the reference never returns "
          "(video shorter than one window) and ours does" % (same, raised, hung))
    return 0


if __name__ == "__main__":
    sys.exit(main())
