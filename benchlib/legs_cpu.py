"""bench.py: the `cpu_baseline` object -- the CPU restatement (oracle/) timed on this host on a bounded sample; the ONLY bench module that imports oracle/."""
import json
import os
import sys
import time

import numpy as np

from .common import (BF16_MFMA_PEAK_TFLOPS, FP32_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, ROOT, committed_traffic, ioc_flops_per_row,  # noqa: F401
                     sdd_windows)


def cpu_baseline(d_full, seed):
    """The CPU restatement (oracle, NOT TF1 -- the reference cannot run), timed on this host on a bounded sample.  `value` is
    the faster of the two restatements: oracle/desire_torch.py (torch fp32, batched GEMMs on every host thread -- the fair
    one); the numpy oracle the parity tests check against is reported next to it."""
    import torch
    from oracle import desire_oracle as O                      # cpu_baseline leg: allowed importer
    from oracle import desire_torch as OT
    from desire_amd.spec import init_weights
    from desire_amd.synth import make_case
    tr = lambda x: np.ascontiguousarray(x.transpose(1, 0, 2, 3).reshape(x.shape[1], -1, 3))
    def torch_run(n_windows):
        d = d_full.replace(n_scenes=n_windows, n_grids=1)
        w = init_weights(d, seed)
        past, fut, eps, grids, gos = make_case(d, seed=seed + 1)
        OT.DT = torch.float32
        wt = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in w.items()}
        with torch.no_grad():
            t0 = time.perf_counter()
            OT.forward_loss(tr(past), tr(fut), eps, grids, gos, wt, d)
            return d.R, time.perf_counter() - t0
    nthr0 = torch.get_num_threads()
    try:
        torch_run(1)                                           # thread pools / allocator warm
        best = None
        for nt in sorted({min(nthr0, c) for c in (8, 16, 32, 64, 128)}):      # many small GEMMs: the widest pool is not the fastest
            torch.set_num_threads(nt)
            r8, t8 = torch_run(8)
            if best is None or t8 < best[2]:
                best = (nt, r8, t8)
        threads, r8, t8 = best
        torch.set_num_threads(threads)
        n = int(min(64, max(8, 8 * 12.0 / max(t8, 1e-3))))     # about 12 s of CPU work, at most 64 windows (host memory)
        rt, tt = torch_run(n) if n > 8 else (r8, t8)
    finally:
        OT.DT = torch.float64
        torch.set_num_threads(nthr0)
    d = d_full.replace(n_scenes=4, n_grids=1)
    w = init_weights(d, seed)
    past, fut, eps, grids, gos = make_case(d, seed=seed + 1)
    t0 = time.perf_counter()
    ref = O.forward(tr(past), tr(fut), eps, grids, gos, w, d)
    dt = time.perf_counter() - t0
    # accuracy gate of the metric: the HIP path on the SAME 4 windows against that oracle run.  Sample
    # generation end to end; the IOC pass from the oracle's own Y0, so that a neighbour sitting within 1e-7 of a bin edge
    # cannot land in different bins on the two sides.
    from desire_amd import _lib
    dev = torch.device("cuda", torch.cuda.current_device())
    tt_ = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    hh = _lib.Handle(d)
    hh.set_weights(w)
    p_t, f_t, e_t, g_t = tt_(past), tt_(fut), tt_(eps), tt_(grids)
    hh.set_scene_grids(g_t.data_ptr(), gos)
    Yg = torch.zeros((d.R, d.T_pred, 2), device=dev); sg = torch.zeros((d.R,), device=dev)
    hh.encode(p_t.data_ptr(), f_t.data_ptr())
    hh.sample(e_t.data_ptr(), Yg.data_ptr())
    torch.cuda.synchronize()
    Y0f = Yg.cpu().numpy()
    e_y0 = float(np.abs(Y0f - ref["Y0"]).max())
    Yg.copy_(tt_(ref["Y0"].astype(np.float32)))
    hh.ioc_refine(Yg.data_ptr(), sg.data_ptr())
    torch.cuda.synchronize()
    Yf = Yg.cpu().numpy()
    dY = Yf - ref["Y"]
    accuracy = {"max_abs_err_Y0": e_y0, "max_abs_err_Y": float(np.abs(dY).max()), "ade_vs_oracle": float(np.sqrt((dY ** 2).sum(-1)).mean()),
                "gate": 1e-3, "units": "normalised frame coordinates", "sample": "%d samples (4 windows), HIP path vs oracle/desire_oracle.py" % d.R}
    hh.close()
    if d.bf16 == 0 and d.mno <= 32 and d.H in (64, 128):
        # the same check through the six-product forms (dims.bf16 = 3: decoder, deconv2, deconv3, IOC on the bf16 matrix pipe with three
        # exact pieces per operand): its distance from the ORACLE next to the fp32 kernels' own (VERDICT r02 item 5's acceptance)
        h6 = _lib.Handle(d.replace(bf16=3))
        h6.set_weights(w)
        h6.set_scene_grids(g_t.data_ptr(), gos)
        h6.encode(p_t.data_ptr(), f_t.data_ptr())
        h6.sample(e_t.data_ptr(), Yg.data_ptr())
        torch.cuda.synchronize()
        Y06 = Yg.cpu().numpy()
        accuracy["x6_max_abs_err_Y0"] = float(np.abs(Y06 - ref["Y0"]).max())
        Yg.copy_(tt_(ref["Y0"].astype(np.float32)))
        h6.ioc_refine(Yg.data_ptr(), sg.data_ptr())
        torch.cuda.synchronize()
        Y6 = Yg.cpu().numpy()
        accuracy["x6_max_abs_err_Y"] = float(np.abs(Y6 - ref["Y"]).max())
        h6.close()
        # ... and all three fp32 roundings (numpy oracle, fp32 MFMA kernels, six-product kernels) against the oracle evaluated in
        # float64 on the same inputs: which of them is closer to exact arithmetic (IOC pass from the same fp32 Y0 everywhere)
        ref64 = O.forward(tr(past), tr(fut), eps, grids, gos, w, d, dt=np.float64, Y_override=ref["Y0"])
        e64 = lambda Y, key: np.abs(np.asarray(Y, np.float64) - ref64[key])
        st = lambda e: {"max": float(e.max()), "rms": float(np.sqrt((e ** 2).mean()))}
        accuracy["vs_float64_oracle"] = {
            "Y0": {"six_products": st(e64(Y06, "Y0")), "fp32_kernels": st(e64(Y0f, "Y0")), "fp32_numpy_oracle": st(e64(ref["Y0"], "Y0"))},
            "Y": {"six_products": st(e64(Y6, "Y")), "fp32_kernels": st(e64(Yf, "Y")), "fp32_numpy_oracle": st(e64(ref["Y"], "Y"))}}
    # the reference's own structure (model/model.py:211): one object at a time, batch dimension 1, for the
    # sample-generation stages (the IOC stage needs the whole group and stays batched above)
    d1 = d.replace(n_scenes=1, mno=1, iters=1)
    n_obj = 16
    t1 = time.perf_counter()
    for a_ in range(n_obj):
        e1 = eps.reshape(d.n_scenes, d.K, d.mno, d.L)[0, :, a_].reshape(d1.R, d.L)
        pn = O.normalise(tr(past)[:, a_:a_ + 1], d1)
        fn = O.normalise(tr(fut)[:, a_:a_ + 1], d1)
        Hx = O.gru_encode(pn, w, "enc_x"); Hy = O.gru_encode(fn, w, "enc_y")
        vin = O.relu(np.concatenate([Hx, Hy], -1) @ w["fc_c/w"] + w["fc_c/b"])
        mu, ls = O.vae_encoder(vin, w, d.L)
        z = O.rows_from_agents(mu, d1) + np.sqrt(np.exp(O.rows_from_agents(ls, d1))) * e1
        xh = O.vae_decoder(z, w)
        Hr = O.rows_from_agents(Hx, d1)
        xz = O.softmax(O.relu(xh @ w["mask_fc/w"] + w["mask_fc/b"])) * Hr
        O.decode(xz, Hr, O.rows_from_agents(pn[-1], d1), w, d1)
    dt1 = (time.perf_counter() - t1) / n_obj
    return {"accuracy": accuracy, "value": rt / tt, "unit": "agent-trajectory-samples/s", "cores": int(threads), "threads": int(threads),
            "host_cores": int(os.cpu_count() or 0), "kind": "port",
            "sample": "oracle/desire_torch.py forward (torch fp32, batched, %d threads: the fastest of 8..128) on %d windows = %d samples, %.1f s; "
                      "CPU restatement, not TF1 (reference graph does not build)" % (threads, rt // (d.K * d.mno), rt, tt),
            "numpy_oracle": {"value": d.R / dt, "unit": "agent-trajectory-samples/s",
                             "note": "oracle/desire_oracle.py (the parity checker) on 4 windows = %d samples, %.1f s" % (d.R, dt)},
            "per_object_loop": {"value": d.K / dt1, "unit": "agent-trajectory-samples/s",
                                "note": "sample-generation stages only, one object at a time like model/model.py:211 "
                                        "(%d objects x K=%d, %.2f s each)" % (n_obj, d.K, dt1)}}
