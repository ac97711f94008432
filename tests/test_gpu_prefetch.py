"""Loader in the loop on the device (desire_amd/prefetch.py): the feeders deliver the serial loop's batches through pinned staging / the
copy stream (host loader) or the device window builder, and the overlapped training loop (loader thread + loss read one step late) takes
exactly the optimiser steps of the serial one."""
import random

import numpy as np
import pytest

import desire_amd.train as T
from tests.test_prefetch import _video

pytestmark = pytest.mark.gpu


def _frames(seed=5):
    rng = random.Random(seed)
    return [_video(rng, frames=150, mno=12), _video(rng, frames=90, mno=12)]


@pytest.mark.parametrize("depth", [1, 3])
def test_host_feeder_on_the_device_equals_the_serial_loop(depth):
    import torch
    from desire_amd.data_loader import DataLoader
    from desire_amd.prefetch import WindowFeeder, serial_batches
    t_obs, t_pred = 3, 4
    a = DataLoader(4, t_obs + t_pred, 12, frames=_frames())
    b = DataLoader(4, t_obs + t_pred, 12, frames=_frames())
    random.seed(21)
    want = list(serial_batches(a, t_obs, num_epochs=2))
    random.seed(21)
    feeder = WindowFeeder(b, t_obs, t_pred, device=torch.device("cuda"), depth=depth, num_epochs=2, mno=16)
    got = []
    for bt in feeder:
        bt.wait()
        got.append((bt.past.cpu().numpy(), bt.fut.cpu().numpy(), list(bt.d)))      # (.cpu() synchronises the current stream, which waited)
        bt.release()
    feeder.close()
    assert len(got) == len(want)
    for (p, f, d), (wp, wf, wd, _, _) in zip(got, want):
        assert d == list(wd)
        np.testing.assert_array_equal(p[:, :, :12], wp)
        np.testing.assert_array_equal(f[:, :, :12], wf)
        assert not p[:, :, 12:].any() and not f[:, :, 12:].any()


def test_device_feeder_equals_the_serial_loop():
    import torch
    from desire_amd import _lib
    from desire_amd.data_loader import DataLoader
    from desire_amd.prefetch import DeviceWindowFeeder, serial_batches
    from desire_amd.spec import init_weights
    from tests.helpers import small_dims
    t_obs, t_pred = 3, 4
    d = small_dims(n_scenes=4, mno=16, T_obs=t_obs, T_pred=t_pred, K=2, n_grids=1)
    h = _lib.Handle(d)
    h.set_weights(init_weights(d, 0))
    a = DataLoader(4, t_obs + t_pred, 12, frames=_frames(7))
    b = DataLoader(4, t_obs + t_pred, 12, frames=_frames(7))
    random.seed(4)
    want = list(serial_batches(a, t_obs, num_epochs=2))
    random.seed(4)
    feeder = DeviceWindowFeeder(b, h, torch.device("cuda"), depth=2, num_epochs=2)
    n = 0
    for bt, (wp, wf, wd, _, _) in zip(feeder, want):
        bt.wait()
        p, f = bt.past.cpu().numpy(), bt.fut.cpu().numpy()
        bt.release()
        assert list(bt.d) == list(wd)
        np.testing.assert_array_equal(p[:, :, :12], wp)
        np.testing.assert_array_equal(f[:, :, :12], wf)
        n += 1
    feeder.close()
    assert n == len(want) > 4


def test_overlapped_training_takes_the_serial_loop_s_steps(tmp_path):
    """--prefetch 2 (loader thread, copy stream, loss read one step late) against --prefetch 0 (the reference's serial order): same
    batches, same seeds, same optimiser -- the per-step losses are the same numbers and so are the final weights."""
    from desire_amd.data_loader import DataLoader
    from desire_amd.model import DESIREModel
    from tests.test_train_loop import _synthetic_video
    out = {}
    for pf in (0, 2):
        rng = np.random.default_rng(0)
        frames = [_synthetic_video(120, 8, 6, rng), _synthetic_video(90, 8, 5, rng)]
        a = T.build_parser().parse_args(["--batch_size", "4", "--seq_length", "4", "--pred_length", "6", "--max_num_obj", "8",
                                         "--d_dim", "64", "--latent_size", "64", "--num_samples", "3", "--num_epochs", "2",
                                         "--save_every", "7", "--learning_rate", "0.0005", "--neighborhood_size", "256",
                                         "--prefetch", str(pf), "--save_dir", str(tmp_path / ("save%d" % pf))])
        dl = DataLoader(a.batch_size, a.seq_length + a.pred_length, a.max_num_obj, frames=frames)
        model = DESIREModel(a, seed=3)
        lines = []
        losses = T.train(a, data_loader=dl, model=model, log=lines.append)
        w = model.sync_weights()
        out[pf] = (losses, {k: np.array(v) for k, v in w.items()}, [l for l in lines if "train_loss" in l], [l for l in lines if "saved" in l])
    l0, l2 = out[0][0], out[2][0]
    assert len(l0) == len(l2) > 6
    np.testing.assert_allclose(l2, l0, rtol=1e-6, atol=1e-7)
    for k in out[0][1]:
        np.testing.assert_allclose(out[2][1][k], out[0][1][k], rtol=1e-5, atol=1e-7, err_msg=k)
    assert len(out[0][2]) == len(out[2][2]) and len(out[0][3]) == len(out[2][3]) >= 1
    assert [l.split(",")[0] for l in out[0][2]] == [l.split(",")[0] for l in out[2][2]]       # the same "i/N (epoch e)" sequence
