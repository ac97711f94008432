"""N2: on-disk formats (reference CSV / cpkl kept, DSRTRJ1 memory-mappable container added) and N1 extras."""
import os
import pickle
import time

import numpy as np
import pytest

from desire_amd.data_loader import DataLoader, frames_from_csv
from desire_amd.formats import (cpkl_to_bin, load_weights, read_cpkl, read_traj_bin, save_weights, write_cpkl,
                                write_traj_bin)
from desire_amd.spec import Dims, init_weights


def _golden(golden_dir, tag="bookstore6_T8"):
    return np.load(os.path.join(golden_dir, f"loader_{tag}.npz"))


def test_bin_roundtrip_is_lossless_and_mmapped(golden_dir, tmp_path):
    g = _golden(golden_dir)
    vids = [g["data0"], g["data0"][:37]]
    p = str(tmp_path / "traj.bin")
    write_traj_bin(p, vids)
    back = read_traj_bin(p)
    assert all(isinstance(b, np.memmap) for b in back)
    for a, b in zip(vids, back):
        np.testing.assert_array_equal(np.asarray(b, np.float64), a)
    with pytest.raises(ValueError):
        write_traj_bin(p, [np.full((2, 4, 3), 0.1)])          # 0.1 is not exact in fp32
    with pytest.raises(ValueError):
        read_traj_bin(__file__)


def test_cpkl_is_the_reference_pickle_and_converts(golden_dir, tmp_path):
    g = _golden(golden_dir)
    pk, pb = str(tmp_path / "trajectories.cpkl"), str(tmp_path / "traj.bin")
    write_cpkl(pk, [g["data0"]], [g["frame_list0"].tolist()], [g["num_obj0"].tolist()])
    with open(pk, "rb") as fh:
        raw = pickle.load(fh)                                   # plain protocol-2 pickle, as the reference reads it
    assert isinstance(raw, tuple) and len(raw) == 3
    np.testing.assert_array_equal(read_cpkl(pk)[0][0], g["data0"])
    cpkl_to_bin(pk, pb)
    np.testing.assert_array_equal(np.asarray(read_traj_bin(pb)[0], np.float64), g["data0"])


def test_loader_on_bin_matches_reference_batches(golden_dir, tmp_path):
    g = _golden(golden_dir)
    bs, T, mno = (int(v) for v in g["kw"])
    p = str(tmp_path / "traj.bin")
    write_traj_bin(p, [g["data0"]])
    dl = DataLoader(batch_size=bs, seq_length=T, max_num_obj=mno, traj_bin=p)
    assert dl.num_batches == int(g["num_batches"])
    for b in range(g["x"].shape[0]):
        x, y, d = dl.next_batch(random_update=False)
        np.testing.assert_array_equal(np.stack(x), g["x"][b])
        np.testing.assert_array_equal(np.stack(y), g["y"][b])
        assert x[0].dtype == np.float64


def test_fix_id0_keeps_the_dropped_track():
    # every SDD video has a track id 0, which the reference treats as padding and drops (utils/data_loader.py:221-222)
    fr = np.arange(12.0)
    csv = np.concatenate([np.stack([fr, np.zeros(12), 10 + fr, 20 + fr]), np.stack([fr, np.full(12, 5.0), 30 + fr, 40 + fr])], axis=1)
    ref, _, _ = frames_from_csv(csv, 4)
    fix, _, _ = frames_from_csv(csv, 4, fix_id0=True)
    assert (fix[:, :, 0] == 6).sum() == 12 and (ref[:, :, 0] == 6).sum() == 0
    a = DataLoader(batch_size=1, seq_length=8, max_num_obj=4, frames=[ref]).next_batch(False)[0][0]
    b = DataLoader(batch_size=1, seq_length=8, max_num_obj=4, frames=[fix]).next_batch(False)[0][0]
    assert (a[:, :, 0] != 0).sum() == 8 and (b[:, :, 0] != 0).sum() == 16      # the id-0 track survives only with the flag


def test_preprocess_speed_vs_reference_figure(golden_dir):
    """The reference needs 5.4 s for bookstore/video6 (108 886 columns); the vectorised pass must stay
    interactive.  Synthetic CSV of that size."""
    rng = np.random.default_rng(0)
    n_tracks, n_frames = 133, 14558
    cols = []
    for tid in range(n_tracks):
        f0 = rng.integers(0, n_frames - 900)
        fr = np.arange(f0, f0 + rng.integers(200, 900))
        cols.append(np.stack([fr, np.full(fr.size, tid), rng.integers(0, 2800, fr.size) / 2, rng.integers(0, 2200, fr.size) / 2]))
    data = np.concatenate(cols, axis=1).astype(np.float64)
    mno = int(np.bincount(data[0].astype(int)).max())
    t0 = time.perf_counter()
    arr, fl, no = frames_from_csv(data, mno)
    dt = time.perf_counter() - t0
    assert arr.shape[1] == mno and max(no) == mno and dt < 2.0, dt


def test_weight_checkpoint_roundtrip(tmp_path):
    w = init_weights(Dims(T_pred=12), 3)
    p = str(tmp_path / "ckpt.npz")
    save_weights(p, w)
    back = load_weights(p)
    assert set(back) == set(w)
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])
