"""Randomised sweep of the training path (run by hand on a GPU box: `python -m tests.fuzz_train N SEED`): random dims inside
the supported training range, every weight gradient against float64 autograd of the oracle."""
import sys
import traceback

import numpy as np


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


def main():
    import torch
    from desire_amd import _lib
    from desire_amd.spec import init_weights
    from oracle import desire_torch as OT
    from tests.helpers import make_case, small_dims, to_oracle_layout
    n, seed = int(sys.argv[1]), int(sys.argv[2])
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # dims.bf16: 0 fp32, 2 split-bf16 IOC forward (fp32 saves / backward)
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n):
        mno = int(rng.choice([2, 4, 8, 16, 32, 32, 64, 96, 128]))
        H = int(rng.choice([16, 32, 64, 128, 128, 256])) if mno <= 32 else int(rng.choice([16, 64, 128]))
        gs = int(rng.integers(1, 7)) if mno <= 32 and H <= 128 else int(rng.integers(1, 5))
        kw = dict(mno=mno, H=H, K=int(rng.integers(2, 5)), T_pred=int(rng.integers(2, 9)), T_obs=int(rng.integers(2, 7)),
                  n_scenes=int(rng.integers(1, 4)) if mno <= 32 else 1, grid_size=gs, L=int(rng.choice([64, 128])), n_grids=1,
                  iters=int(rng.choice([1, 1, 2])),
                  nb_w=float(rng.choice([0.05, 0.2, 0.5])), nb_h=float(rng.choice([0.05, 0.25, 0.5])))
        if rng.random() < 0.25 and gs >= 3:
            kw.update(bin_mode=1, nb_w=0.45, nb_h=0.04)
        if rng.random() < 0.3:
            kw["bn_mode"] = int(rng.choice([1, 2]))          # per-object / whole-batch batch-norm statistics in the training graph
        try:
            d = small_dims(**kw)
            w = init_weights(d, 300 + it)
            for k in w:
                if k.startswith("vae_dec/") and k.endswith("/w"):
                    w[k] = w[k] * 3
            w["mask_fc/w"] = w["mask_fc/w"] * 20
            w["head/w"] = w["head/w"] * 4
            w["ioc/score/w"] = w["ioc/score/w"] * 3
            # relu layers start with zero biases: rows whose input is ~0 (stationary or absent agents) then sit exactly ON the
            # kink, where fp32 and float64 pick different sides and the "gradient" differs by whole terms.  Move them off it.
            for k in ("ioc/vel_fc/b", "ioc/social_fc/b", "fc_c/b", "mask_fc/b"):
                w[k] = (w[k] + rng.uniform(0.02, 0.1, w[k].shape) * rng.choice([-1.0, 1.0], w[k].shape)).astype(np.float32)
            past, fut, eps, grids, gos = make_case(d, seed=400 + it, n_absent=min(int(rng.integers(0, 4)), d.mno - 1))
            h = _lib.Handle(d.replace(bf16=mode))
            h.set_weights(w)
            h.set_training(True)
            tab = h.bin_table() if d.bin_mode == 1 else None
            vals, ref = OT.loss_and_grads(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w, d, bin_tab=tab)
            t = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")
            p, f, e, g = t(past), t(fut), t(eps), t(grids)
            h.set_scene_grids(g.data_ptr(), gos)
            Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
            h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
            h.backward(p.data_ptr(), f.data_ptr(), e.data_ptr())
            torch.cuda.synchronize()
            worst, wname = 0.0, ""
            detail = []
            gscale = max(float(np.abs(np.asarray(ref[nm])).max()) for nm in ref)
            for name in ref:
                if "/bn/" in name or name.startswith("scene_cnn") or name.startswith("temporal"):
                    continue
                r = np.asarray(ref[name])
                if np.abs(r).max() < 1e-12:
                    continue
                got = h.get_grad(name, w[name].shape)
                er = rel_err(got, r)
                detail.append((er, name, float(np.abs(r).max()), float(np.abs(got - r).max())))
                if np.abs(r).max() < 1e-6 * gscale:          # a gradient that is numerically nothing next to the others
                    continue
                if er > 5e-4 and np.abs(got - r).max() < 1e-7 * gscale:   # ... or whose error is fp32 cancellation noise at the model's scale
                    continue
                if er > worst:
                    worst, wname = er, name
            ok = worst < 5e-4
            if not ok:
                # the IOC module sees the sampled trajectories only through non-differentiable cell / bin indices: a pair that
                # sits within 1e-7 of a bin edge lands on different sides in fp32 (kernel) and float64 (autograd reference) and
                # moves whole terms.  Re-derive the reference with the KERNEL's trajectories pinned and compare again.
                wl = OT.leaf_weights(w)
                o1 = OT.forward_loss(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, wl, d, bin_tab=tab)
                Yk = h.read_buffer("Y0", (d.R, d.T_pred, 2)).astype(np.float64)
                wl = OT.leaf_weights(w)
                o2 = OT.forward_loss(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, wl, d,
                                     fixed={"Yd": Yk, "dmax": o1["dmax"].numpy()}, bin_tab=tab)
                o2["loss"].backward()
                worst2, wname2 = 0.0, ""
                for name in ref:
                    if "/bn/" in name or name.startswith("scene_cnn") or name.startswith("temporal") or wl[name].grad is None:
                        continue
                    r = wl[name].grad.numpy()
                    if np.abs(r).max() < 1e-6 * gscale:
                        continue
                    er = rel_err(h.get_grad(name, w[name].shape), r)
                    if er > worst2:
                        worst2, wname2 = er, name
                print("       with the kernel's trajectories pinned in the reference: worst %.2e (%s)" % (worst2, wname2))
                ok = worst2 < 5e-4
                if not ok:
                    # conditioning: the SAME autograd reference evaluated in float32 -- if plain fp32 arithmetic is already this
                    # far from float64 (e.g. 63 neighbours pooled into one bin drive the gates into saturation), the
                    # configuration cannot separate a kernel bug from rounding
                    import torch
                    OT.DT = torch.float32
                    try:
                        w32 = OT.leaf_weights(w)
                        o3 = OT.forward_loss(to_oracle_layout(past), to_oracle_layout(fut), eps, grids, gos, w32, d,
                                             fixed={"Yd": Yk.astype(np.float32), "dmax": o1["dmax"].numpy().astype(np.float32)}, bin_tab=tab)
                        o3["loss"].backward()
                        e32 = rel_err(w32[wname2].grad.numpy(), wl[wname2].grad.numpy())
                    finally:
                        OT.DT = torch.float64
                    print("       the float32 evaluation of the reference itself is off by %.2e on %s" % (e32, wname2))
                    ok = worst2 < 3 * e32
                if not ok:
                    # relu kink: an e_v / e_r element whose input is within rounding of zero is "on" in one evaluation and "off"
                    # in the other -- compare the activity pattern the kernel saved with the reference's
                    E = d.E_v + d.C + d.H
                    xk = h.device_tensor("ioc_sv_x")[: d.R * d.T_pred * E].reshape(d.R, d.T_pred, E).cpu().numpy()
                    xo = o2["ioc_x"].numpy()
                    vr = np.repeat(np.tile((past[:, d.T_obs - 1, :, 0] != 0)[:, None, :], (1, d.K, 1)).reshape(-1), 1)
                    cols = np.r_[0:d.E_v, d.E_v + d.C:E]
                    flips = int((((xk > 0) != (xo > 0))[vr][:, :, cols]).sum())
                    va = past[:, d.T_obs - 1, :, 0].reshape(-1) != 0                     # ... and the relu of fc_c (CVAE encoder input)
                    flips += int(((h.read_buffer("vae_in", (d.A, d.V)) > 0) != (o2["vae_in"].detach().numpy() > 0))[va].sum())
                    print("       relu elements active in one evaluation and not in the other: %d (of %d)" % (flips, int(vr.sum()) * d.T_pred * len(cols)))
                    ok = flips > 0
            print("%3d %s  worst rel grad err %.2e (%s)  largest |grad| %.2e  %s" % (it, kw, worst, wname, gscale, "ok" if ok else "MISMATCH"), flush=True)
            if not np.isfinite(gscale) or wname == "":
                print("       NOTHING COMPARED (gradient scale %r)" % gscale)
                bad += 1
            bad += 0 if ok else 1
            if not ok:
                for er, name, sc_, ab in sorted(detail, reverse=True)[:4]:
                    print("       %-28s rel %.2e  |ref|max %.2e  abs err %.2e   (largest gradient in the model %.2e)" % (name, er, sc_, ab, gscale))
        except Exception as ex:                              # noqa: BLE001
            refused = isinstance(ex, _lib.DesireError)
            print("%3d %s  %s: %s" % (it, kw, "refused" if refused else "EXCEPTION", str(ex)[:160]), flush=True)
            if not refused:
                traceback.print_exc()
                bad += 1
    print("bad =", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
