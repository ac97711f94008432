"""Randomised comparison of the device window / slot builder (desire_build_windows) with the CPU loader's window_to_slots
(itself checked against the imported reference loader by tests/golden/fuzz_loader_vs_reference.py).  GPU box:
`python -m tests.fuzz_windows N SEED`."""
import sys

import numpy as np


def main():
    import torch
    from desire_amd import _lib
    from desire_amd.data_loader import window_to_slots
    from tests.helpers import small_dims
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        mno_in = int(rng.integers(2, 40))
        mno = int(rng.choice([4, 8, 16, 32, 64]))
        T_obs, T_pred = int(rng.integers(1, 9)), int(rng.integers(1, 13))
        W = T_obs + T_pred
        n_frames = W + int(rng.integers(0, 30))
        n_ids = int(rng.integers(1, mno_in + 1))
        ids = rng.choice(np.arange(0, 60), n_ids, replace=False).astype(np.float64)
        frames = np.zeros((n_frames, mno_in, 3))
        for f in range(n_frames):
            present = ids[rng.random(n_ids) < 0.7]
            rng.shuffle(present)
            k = len(present)
            frames[f, :k, 0] = present
            frames[f, :k, 1:] = np.round(rng.uniform(3, 1900, (k, 2)) * 2) / 2
            frames[f, :k][present == 0] = 0                     # id 0 = padding, exactly as the loader sees it
            if k >= 2 and rng.random() < 0.02:
                frames[f, 1, 0] = frames[f, 0, 0]               # a duplicated id
        n_win = int(rng.integers(1, 5))
        starts = rng.integers(0, n_frames - W + 1, n_win).astype(np.int32)
        d = small_dims(n_scenes=n_win, mno=mno, K=1, T_obs=T_obs, T_pred=T_pred, n_grids=1)
        h = _lib.Handle(d)
        fr = torch.as_tensor(frames.astype(np.float32), device="cuda")
        past = torch.full((n_win, T_obs, mno, 3), -1.0, device="cuda")
        fut = torch.full((n_win, T_pred, mno, 3), -1.0, device="cuda")
        look = it % 2                                            # odd rounds: lookahead = 1 = the x of DataLoader(seq_length = W)
        exp, exp_err = [], None
        for s0 in starts:
            try:
                if look and s0 + W < n_frames:
                    src, _ = window_to_slots(frames[s0:s0 + W + 1], W, mno)
                    exp.append(src.astype(np.float32))
                else:
                    src, tgt = window_to_slots(frames[s0:s0 + W], W - 1, mno)
                    exp.append(np.concatenate([src, tgt[-1:]], 0).astype(np.float32))
            except (IndexError, ValueError) as ex:
                exp_err = type(ex).__name__
                break
        try:
            h.build_windows(fr.data_ptr(), n_frames, mno_in, starts, past.data_ptr(), fut.data_ptr(), lookahead=look)
            got_err = None
        except _lib.DesireError as ex:
            got_err = str(ex)
        if (exp_err is None) != (got_err is None):
            print("%3d mno_in=%d mno=%d W=%d: loader %s / device %s" % (it, mno_in, mno, W, exp_err, got_err))
            bad += 1
            continue
        if exp_err is None:
            full = np.stack(exp)
            ok = np.array_equal(past.cpu().numpy(), full[:, :T_obs]) and np.array_equal(fut.cpu().numpy(), full[:, T_obs:])
            if not ok:
                print("%3d mno_in=%d mno=%d W=%d: arrays differ" % (it, mno_in, mno, W))
                bad += 1
    print("bad =", bad, "of", n)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
