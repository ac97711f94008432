"""Loader in the loop, CPU side: the batched window builder against the per-window one (the path checked against the reference's goldens),
and the background feeder against the serial loop -- same batches, same order, same pointer / random-state trajectory."""
import random

import numpy as np
import pytest

from desire_amd.data_loader import DataLoader, window_to_slots, windows_to_slots_batch
from desire_amd.prefetch import WindowFeeder, serial_batches


def _video(rng, frames=120, mno=12, ids=7, p=0.7):
    v = np.zeros((frames, mno, 3))
    for f in range(frames):
        here = [i for i in range(1, ids + 1) if rng.random() < p]
        rng.shuffle(here)
        for s, i in enumerate(here[:mno]):
            v[f, s] = (i + (f // 40) * 3, rng.random() * 1400, rng.random() * 1100)     # ids drift, so windows hold different id sets
    return v


@pytest.mark.parametrize("seed", range(6))
def test_batched_builder_equals_the_per_window_builder(seed):
    rng = random.Random(seed)
    T, mno = 7, 12
    wins = np.stack([_video(rng, frames=T + 1, mno=mno, ids=rng.randint(1, 11)) for _ in range(21)])
    out = windows_to_slots_batch(wins, T, mno)
    assert out is not None
    for i in range(wins.shape[0]):
        s, t = window_to_slots(wins[i], T, mno)
        np.testing.assert_array_equal(out[0][i], s)
        np.testing.assert_array_equal(out[1][i], t)
    # caller-owned float32 outputs (the pinned staging of the feeder), target skipped
    buf = np.full((21, T, mno, 3), 7.0, np.float32)
    assert windows_to_slots_batch(wins, T, mno, buf, False)[1] is None
    np.testing.assert_array_equal(buf, out[0].astype(np.float32))


def test_batched_builder_declines_where_the_reference_raises():
    T, mno = 3, 2
    w = np.zeros((2, T + 1, mno, 3))
    w[0, 0, :, 0] = [1, 2]; w[0, 1, :, 0] = [3, 4]                     # more unique ids than slots: IndexError in the reference
    assert windows_to_slots_batch(w, T, mno) is None
    w[:] = 0
    w[1, 2, :, 0] = [5, 5]                                               # an id twice in a frame: ValueError in the reference
    assert windows_to_slots_batch(w, T, mno) is None
    w[:] = 0
    w[0, 0, 0, 0] = 2.5                                                  # not an integer id: the per-window path decides
    assert windows_to_slots_batch(w, T, mno) is None


def test_next_batch_raises_like_the_per_window_walk_and_leaves_the_same_state():
    rng = random.Random(3)
    good = _video(rng, frames=60, mno=8, ids=4)
    bad = good.copy()
    bad[33, :2, 0] = 77                                                  # duplicate id in one frame -> ValueError at the window that holds it
    for frames in ([good, bad], [bad]):
        a = DataLoader(5, 6, 8, frames=frames)
        b = DataLoader(5, 6, 8, frames=frames)
        for dl, fn in ((a, a.next_batch), (b, b._next_batch_scalar)):
            random.seed(11)
            dl.err = None
            try:
                for _ in range(6):
                    fn()
            except (ValueError, IndexError) as e:
                dl.err = (type(e).__name__, str(e))
            dl.state = (dl.dataset_pointer, dl.frame_pointer, random.random())
        assert a.err is not None and a.err == b.err and a.state == b.state


@pytest.mark.parametrize("shard", [(0, 1), (1, 2)])
def test_feeder_returns_the_serial_loop_s_batches(shard):
    rng = random.Random(5)
    frames = [_video(rng, frames=150, mno=12), _video(rng, frames=90, mno=12)]
    t_obs, t_pred = 3, 4
    a = DataLoader(4, t_obs + t_pred, 12, frames=frames)
    b = DataLoader(4, t_obs + t_pred, 12, frames=frames)
    assert a.num_batches > 3
    random.seed(21)
    want = list(serial_batches(a, t_obs, num_epochs=2, shard=shard))
    end_serial = (a.dataset_pointer, a.frame_pointer, random.random())
    random.seed(21)
    got = []
    feeder = WindowFeeder(b, t_obs, t_pred, device=None, depth=2, num_epochs=2, shard=shard, mno=16)
    for bt in feeder:
        got.append((bt.past.numpy().copy(), bt.fut.numpy().copy(), list(bt.d), bt.epoch, bt.index))
        bt.release()
    feeder.close()
    assert (b.dataset_pointer, b.frame_pointer, random.random()) == end_serial
    assert len(got) == len(want) == 2 * a.num_batches
    for (p, f, d, e, i), (wp, wf, wd, we, wi) in zip(got, want):
        assert (e, i, d) == (we, wi, list(wd))
        np.testing.assert_array_equal(p[:, :, :12], wp)
        np.testing.assert_array_equal(f[:, :, :12], wf)
        assert not p[:, :, 12:].any() and not f[:, :, 12:].any()         # padded slots = absent


def test_feeder_stops_early_and_surfaces_loader_errors():
    rng = random.Random(9)
    good = _video(rng, frames=200, mno=12, ids=4)
    dl = DataLoader(3, 5, 12, frames=[good])
    f = WindowFeeder(dl, 2, 3, depth=2, num_epochs=5, max_batches=4)
    assert sum(1 for bt in f if bt.release() is None) == 4
    f.close()
    bad = good.copy()
    bad[40, :2, 0] = 9
    dl = DataLoader(3, 5, 12, frames=[bad])
    f = WindowFeeder(dl, 2, 3, depth=2, num_epochs=1, random_update=False)
    with pytest.raises(ValueError):
        for bt in f:
            bt.release()
    f.close()
