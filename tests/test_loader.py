"""DataLoader vs golden vectors produced by the reference's own utils/data_loader.py
(tests/golden/make_loader_golden.py).  Bit-exact: the layout is integer/copy work."""
import os

import numpy as np
import pytest

from desire_amd.data_loader import DataLoader, frames_from_csv, window_to_slots

TAGS = ["bookstore6_T8", "bookstore6_T48", "deathcircle2_T8", "deathcircle4_T48"]


@pytest.mark.parametrize("tag", TAGS)
def test_preprocess_and_batches_match_reference(tag, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, f"loader_{tag}.npz"))
    bs, T, mno = (int(v) for v in g["kw"])
    vid = tmp_path / "data" / "scene" / "video0"
    vid.mkdir(parents=True)
    np.savetxt(vid / "annotations_processed.csv", g["csv"].astype(np.float64), delimiter=",", fmt="%.1f")
    dl = DataLoader(batch_size=bs, seq_length=T, max_num_obj=mno, leave_dataset=1, preprocess=True,
                    data_dir=str(tmp_path / "data") + "/")
    assert dl.data[0].dtype == np.float64
    np.testing.assert_array_equal(dl.data[0], g["data0"])
    np.testing.assert_array_equal(np.asarray(dl.frame_list[0]), g["frame_list0"])
    np.testing.assert_array_equal(np.asarray(dl.num_obj_list[0]), g["num_obj0"])
    assert dl.num_batches == int(g["num_batches"])
    for b in range(g["x"].shape[0]):
        x, y, d = dl.next_batch(random_update=False)
        assert len(x) == bs and x[0].shape == (T, mno, 3) and x[0].dtype == np.float64
        np.testing.assert_array_equal(np.stack(x), g["x"][b])
        np.testing.assert_array_equal(np.stack(y), g["y"][b])
        np.testing.assert_array_equal(np.asarray(d), g["d"][b])
    assert dl.frame_pointer == int(g["frame_pointer"])
    assert dl.dataset_pointer == int(g["dataset_pointer"])
    # the pickle the reference writes (protocol 2 tuple) is reproduced
    import pickle
    with open(tmp_path / "data" / "trajectories.cpkl", "rb") as fh:
        raw = pickle.load(fh)
    assert isinstance(raw, tuple) and len(raw) == 3
    np.testing.assert_array_equal(raw[0][0], g["data0"])


def test_too_many_objects_raises_like_reference():
    data = np.array([[0, 0, 0], [1, 2, 3], [1.0, 2.0, 3.0], [4.0, 5.0, 6.0]])
    with pytest.raises(ValueError):
        frames_from_csv(data, 2)


def test_too_many_unique_ids_raises_like_reference():
    win = np.zeros((3, 2, 3))
    win[0, :, 0] = [1, 2]
    win[1, :, 0] = [3, 4]
    with pytest.raises(IndexError):
        window_to_slots(win, 2, 2)


def test_duplicate_id_in_a_window_frame_raises_like_reference():
    # utils/data_loader.py:224-229: a (2,3) block cannot be assigned to one (3,) slot -> ValueError (found by
    # tests/golden/fuzz_loader_vs_reference.py, which runs both loaders on random tables in the build container)
    win = np.zeros((3, 3, 3))
    win[:, 0] = [4, 1.0, 2.0]
    win[1, 1] = [4, 5.0, 6.0]
    with pytest.raises(ValueError):
        window_to_slots(win, 2, 3)
    win[1, 1] = [9, 5.0, 6.0]
    src, tgt = window_to_slots(win, 2, 3)
    assert src[1, 2, 0] == 9 and tgt[0, 2, 0] == 9 and src[1, 1, 0] == 4       # slots = rank among the window's unique ids (0 first)


def test_id_zero_track_is_dropped_like_reference():
    # SDD track id 0 is indistinguishable from padding (utils/data_loader.py:221-222)
    data = np.array([[0, 0, 1, 1], [0, 5, 0, 5], [1.0, 2.0, 3.0, 4.0], [1.0, 2.0, 3.0, 4.0]])
    arr, _, _ = frames_from_csv(data, 4)
    src, tgt = window_to_slots(arr, 1, 4)
    assert (src[:, :, 0] != 0).sum() == 1 and src[0, 1, 0] == 5


def test_duplicate_id_takes_first_xy():
    data = np.array([[0, 0], [7, 7], [1.0, 9.0], [2.0, 8.0]])
    arr, _, n = frames_from_csv(data, 4)
    assert n == [2]
    np.testing.assert_array_equal(arr[0, :2], [[7, 1, 2], [7, 1, 2]])


def test_frames_kwarg_and_random_update_deterministic_with_seed():
    import random
    rng = np.random.default_rng(0)
    fr = np.zeros((50, 8, 3))
    fr[:, :5, 0] = np.arange(1, 6)
    fr[:, :5, 1:] = rng.random((50, 5, 2))
    a = DataLoader(batch_size=3, seq_length=4, max_num_obj=8, frames=[fr])
    b = DataLoader(batch_size=3, seq_length=4, max_num_obj=8, frames=[fr])
    random.seed(1); xa, _, _ = a.next_batch()
    random.seed(1); xb, _, _ = b.next_batch()
    np.testing.assert_array_equal(np.stack(xa), np.stack(xb))
    assert a.num_batches == 2 * ((50 // 6) // 3)


def test_mixed_scene_batches_match_reference(golden_dir):
    """Eight SDD scenes loaded and batched together by the REFERENCE loader (tests/golden/make_mixed_golden.py; BASELINE configs[4]'s
    "mixed-scene" input): per-video preprocessing and the cross-video next_batch walk, bit-exact."""
    g = np.load(os.path.join(golden_dir, "loader_mixed8_T20.npz"))
    bs, T, mno = (int(v) for v in g["kw"])
    n = len(g["order"])
    assert n == 8 and len(set(g["order"].tolist())) == 8
    frames = []
    for i in range(n):
        arr, _, _ = frames_from_csv(g["csv%d" % i].astype(np.float64), mno)
        np.testing.assert_array_equal(arr, g["data%d" % i])
        frames.append(arr)
    dl = DataLoader(batch_size=bs, seq_length=T, max_num_obj=mno, frames=frames)      # same video order as the reference walked
    assert dl.num_batches == int(g["num_batches"])
    for b in range(g["x"].shape[0]):
        x, y, d = dl.next_batch(random_update=False)
        np.testing.assert_array_equal(np.stack(x), g["x"][b])
        np.testing.assert_array_equal(np.stack(y), g["y"][b])
        np.testing.assert_array_equal(np.asarray(d), g["d"][b])
    assert len(set(g["d"][0].tolist())) > 1                                        # one batch really spans several videos
