"""Randomised sweep of the agent-sharded IOC (virtual ranks on one GPU vs the unsharded kernels): `python -m tests.fuzz_sharded N SEED`
on a GPU box.  Reuses the comparison of tests/test_gpu_sharded_ioc.py with random group sizes, rank counts, widths and bins."""
import sys
import traceback

import numpy as np


def main():
    from tests.test_gpu_sharded_ioc import test_virtual_ranks_reproduce_the_unsharded_ioc as check
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        mno = int(rng.choice([16, 32, 32, 64, 128]))
        nranks = int(rng.choice([2, 4]))
        H = int(rng.choice([64, 128, 128, 256]))
        kw = dict(mno=mno, H=H, K=int(rng.integers(1, 4)), T_pred=int(rng.integers(1, 10)), n_scenes=1 if mno > 32 else int(rng.integers(1, 3)),
                  grid_size=int(rng.integers(1, 5)), nb_w=float(rng.choice([0.05, 0.25, 0.5])), nb_h=float(rng.choice([0.05, 0.3])),
                  n_grids=1, iters=int(rng.choice([1, 1, 1, 2])))
        try:
            check(kw, nranks)
            print("%3d %s x%d ok" % (it, kw, nranks), flush=True)
        except Exception as ex:                              # noqa: BLE001
            print("%3d %s x%d FAILED: %s" % (it, kw, nranks, str(ex)[:200]), flush=True)
            traceback.print_exc()
            bad += 1
    print("bad =", bad, "of", n)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
