"""VERDICT r02 item 5's acceptance, literally: on ALL 327 680 rows of the default bench batch (512 windows x K = 20 x 32 slots), the IOC pass
of the fp32 kernels, of the six-product kernels (dims.bf16 = 3) and of the CPU oracle from ONE shared Y0 (the fp32 kernels' decoder
output), i.e. the same cells and bins everywhere.  The numpy oracle takes ~10 minutes for this batch on the GPU box's host: this is a
one-off evidence script (python -m tests.x6_full_batch_check out.json), not part of bench.py or the test suite.  The oracle is used here as
the checker only, as in tests/."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from desire_amd import _lib                                   # noqa: E402
from desire_amd.spec import Dims, init_weights                # noqa: E402
from desire_amd.synth import make_case                        # noqa: E402
from oracle import desire_oracle as O                         # noqa: E402


def main():
    windows = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    d = Dims(n_scenes=windows, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4, nb_w=0.15, nb_h=0.15,
             sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1)
    w = init_weights(d, 0)
    past, fut, eps, grids, gos = make_case(d, seed=1, n_absent=0)
    dev = torch.device("cuda", 0)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device=dev)
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    out = {}
    Y0 = None
    for mode in (0, 3):
        h = _lib.Handle(d.replace(bf16=mode)); h.set_weights(w); h.set_scene_grids(g.data_ptr(), gos)
        h.encode(p.data_ptr(), f.data_ptr())
        if Y0 is None:
            Y0 = torch.zeros((d.R, d.T_pred, 2), device=dev)
            h.sample(e.data_ptr(), Y0.data_ptr())
            Hx = h.read_buffer("Hx", (d.A, d.H)); p_last = h.read_buffer("p_last", (d.A, 2))
        Y = Y0.clone(); sc = torch.zeros((d.R,), device=dev)
        h.ioc_refine(Y.data_ptr(), sc.data_ptr())
        torch.cuda.synchronize()
        out[mode] = (Y.cpu().numpy(), sc.cpu().numpy())
        h.close()
    tr = lambda x: np.ascontiguousarray(x.transpose(1, 0, 2, 3).reshape(x.shape[1], -1, 3))
    valid = tr(past)[d.T_obs - 1, :, 0] != 0
    y0 = Y0.cpu().numpy()
    t0 = time.perf_counter()
    score, dY = O.ioc_pass(y0, O.rows_from_agents(Hx, d), O.rows_from_agents(p_last, d), O.rows_from_agents(valid, d), grids, gos, w, d)
    yo = (y0 + dY).astype(np.float32)
    secs = time.perf_counter() - t0
    res = {"rows": int(d.R), "windows": windows, "oracle_seconds": secs,
           "max_abs_Y_fp32_kernel_vs_oracle": float(np.abs(out[0][0] - yo).max()),
           "max_abs_Y_six_products_vs_oracle": float(np.abs(out[3][0] - yo).max()),
           "max_abs_Y_six_products_vs_fp32_kernel": float(np.abs(out[3][0] - out[0][0]).max()),
           "rms_Y_fp32_kernel_vs_oracle": float(np.sqrt(((out[0][0] - yo).astype(np.float64) ** 2).mean())),
           "rms_Y_six_products_vs_oracle": float(np.sqrt(((out[3][0] - yo).astype(np.float64) ** 2).mean())),
           "max_abs_score_fp32_kernel_vs_oracle": float(np.abs(out[0][1] - score).max()),
           "max_abs_score_six_products_vs_oracle": float(np.abs(out[3][1] - score).max()),
           "note": "IOC pass from the fp32 kernels' Y0 on every row of the default bench batch; oracle = oracle/desire_oracle.py ioc_pass (numpy fp32), "
                   "fed the kernels' own encoder state Hx and last observed positions"}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 1:
        json.dump(res, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
