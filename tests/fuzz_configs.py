"""Randomised configuration sweep (not collected by pytest: run by hand on a GPU box, `python -m tests.fuzz_configs N SEED`).
Draws dims from the whole supported range, runs the HIP path through the C ABI and compares it with the oracle."""
import sys
import traceback

import numpy as np


def main():
    import torch
    from desire_amd import _lib
    from desire_amd.spec import init_weights
    from oracle import desire_oracle as O
    from tests.helpers import make_case, small_dims, to_oracle_layout
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        mno = int(rng.choice([1, 2, 4, 8, 16, 32, 32, 64, 96, 128, 160, 256]))      # 160 / 256: the step-wise IOC (round 4)
        H = int(rng.choice([16, 32, 64, 128, 128, 256]))
        kw = dict(mno=mno, H=H, K=int(rng.integers(1, 6)), T_pred=int(rng.integers(1, 14)), T_obs=int(rng.integers(2, 9)),
                  n_scenes=int(rng.integers(1, 4)) if mno <= 32 else 1, grid_size=int(rng.integers(1, 7)),
                  posterior=int(rng.integers(0, 2)), iters=int(rng.choice([1, 1, 2])), L=int(rng.choice([64, 128])),
                  nb_w=float(rng.choice([0.05, 0.2, 0.5])), nb_h=float(rng.choice([0.05, 0.25, 0.5])))
        kw["n_grids"] = int(rng.integers(1, kw["n_scenes"] + 1))
        if kw["grid_size"] > 4 and H > 128:
            kw["grid_size"] = 4
        if rng.random() < 0.25:
            kw.update(bin_mode=1, nb_w=0.45, nb_h=0.04, grid_size=max(2, kw["grid_size"]))
        bf16 = int(rng.random() < 0.3 and mno <= 128)         # mno 96 / 128: the bf16 cluster form (160+ agents: fp32 / split operands only)
        split = (not bf16) and rng.random() < 0.45            # dims.bf16 = 2 / 3: split operands (fp32 kernels where no split form exists)
        six = split and rng.random() < 0.5                    # three pieces, six products: the fp32 kernels' accuracy class (sample generation too)
        bn_mode = int(rng.choice([1, 2])) if (not bf16) and rng.random() < 0.25 else 0
        bn_per_object = bn_mode != 0
        if bn_per_object:
            kw["bn_mode"] = bn_mode
        try:
            d = small_dims(**kw)
            w = init_weights(d, 100 + it)
            if bn_per_object:                                # non-trivial affine parameters
                for k in list(w):
                    if k.endswith("/bn/gamma"):
                        w[k] = (1.0 + 0.3 * rng.standard_normal(w[k].shape)).astype(np.float32)
                    if k.endswith("/bn/beta"):
                        w[k] = (0.2 * rng.standard_normal(w[k].shape)).astype(np.float32)
            past, fut, eps, grids, gos = make_case(d, seed=200 + it, n_absent=min(int(rng.integers(0, 4)), d.mno - 1))
            tab = None
            h = _lib.Handle(d.replace(bf16=(3 if six else 2) if split else bf16))
            h.set_weights(w)
            if d.bin_mode == 1:
                tab = h.bin_table()
            q = O.bf16_round if bf16 else None
            fo = to_oracle_layout(fut) if d.posterior else None
            okw = dict(bn_mode={1: "per_object", 2: "batch"}[bn_mode]) if bn_per_object else {}
            ref = O.forward(to_oracle_layout(past), fo, eps, grids, gos, w, d, bin_tab=tab, **okw)
            if bf16:                                         # IOC stage against the oracle with the kernels' operand rounding
                r16 = O.forward(to_oracle_layout(past), fo, eps, grids, gos, w, d, bin_tab=tab, Y_override=ref["Y0"], ioc_q=q)
                ref = dict(ref, Y=r16["Y"])
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
            p, f, e, g = t(past), t(fut), t(eps), t(grids)
            h.set_scene_grids(g.data_ptr(), gos)
            Y = t(ref["Y0"].astype(np.float32)).clone(); sc = torch.zeros((d.R,), device="cuda")
            h.encode(p.data_ptr(), f.data_ptr() if d.posterior else 0)
            Ys = torch.zeros_like(Y)
            h.sample(e.data_ptr(), Ys.data_ptr())
            torch.cuda.synchronize()
            e0 = float(np.abs(Ys.cpu().numpy() - ref["Y0"]).max())
            h.ioc_refine(Y.data_ptr(), sc.data_ptr())
            torch.cuda.synchronize()
            if d.iters == 1 and not bf16:
                e1 = float(np.abs(Y.cpu().numpy() - ref["Y"]).max())
                ok = e0 < (2e-5 if six else 1e-3) and e1 < ((2e-5 if six else 1e-4) if split else 1e-3)
            else:                                            # re-binning after a refinement pass / bf16 operands: looser
                e1 = float(np.abs(Y.cpu().numpy() - ref["Y"]).mean())
                # (bf16 operands AND a second pass that re-bins from positions that already differ by ~1e-2: mean error up to a few 1e-2)
                ok = e0 < (2e-2 if bf16 else 1e-3) and e1 < (5e-2 if bf16 and d.iters > 1 else 2e-2) and bool(np.isfinite(Y.cpu().numpy()).all())
            # a refinement that moves points by more than a frame width means the random-weight model is in its chaotic regime for
            # this draw (one bin pooling ~30 neighbours: rounding differences grow ~3x every few steps, fp32 kernel vs oracle
            # included) -- no two implementations agree there, so the draw is reported, not judged
            wild = float(np.abs(ref["Y"] - ref["Y0"]).max()) > 1.0
            print("%3d %s bf16=%d  Y0 err %.2e  Y err %.2e  %s" % (it, kw, (3 if six else 2) if split else bf16, e0, e1,
                                                                   "ok" if ok else "ill-conditioned draw (|dY| > 1): not judged" if wild else "MISMATCH"), flush=True)
            bad += 0 if (ok or wild) else 1
        except Exception as ex:                              # noqa: BLE001
            msg = str(ex)
            refused = isinstance(ex, _lib.DesireError)
            print("%3d %s bf16=%d  %s: %s" % (it, kw, bf16, "refused" if refused else "EXCEPTION", msg[:160]), flush=True)
            if not refused:
                traceback.print_exc()
                bad += 1
    print("bad =", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
