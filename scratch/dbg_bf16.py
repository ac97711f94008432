import numpy as np, torch, sys
sys.path.insert(0, '.')
from desire_amd.spec import init_weights
from tests.helpers import make_case, small_dims
from tests.test_gpu_parity import oracle_forward, run_gpu
from oracle import desire_oracle as O
def run(tag, soc_scale=1.0, n_absent=3, **kw):
    d32 = small_dims(**kw); d16 = d32.replace(bf16=1)
    w = init_weights(d32, 3); w["ioc/social_fc/w"] = w["ioc/social_fc/w"] * soc_scale
    past, fut, eps, grids, gos = make_case(d32, seed=4, n_absent=min(n_absent, d32.mno-1))
    ref32 = oracle_forward(d32, w, past, fut, eps, grids, gos)
    ref16 = oracle_forward(d32, w, past, fut, eps, grids, gos, Y_override=ref32["Y0"], ioc_q=O.bf16_round)
    _, Y, score = run_gpu(torch, d16, w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    _, Yf, scoref = run_gpu(torch, d32, w, past, fut, eps, grids, gos, Y_in=ref32["Y0"])
    print(tag, "bf16 vs rounding-oracle %.2e | bf16 vs fp32-oracle %.2e | rounding-oracle vs fp32-oracle %.2e | fp32 kernel vs fp32 oracle %.2e | dY max %.2e | score err %.2e (|s| %.1f)" % (
        np.abs(Y - ref16["Y"]).max(), np.abs(Y - ref32["Y"]).max(), np.abs(ref16["Y"] - ref32["Y"]).max(), np.abs(Yf - ref32["Y"]).max(),
        np.abs(ref32["Y"] - ref32["Y0"]).max(), np.abs(score - ref16["score"]).max(), np.abs(ref16["score"]).max()))
for kw in [dict(), dict(mno=16, n_scenes=3, K=5), dict(mno=64, n_scenes=1, K=2, n_grids=1), dict(H=64, T_pred=7, K=3), dict(H=256, K=3, n_scenes=1, n_grids=1, T_pred=10),
           dict(grid_size=2, nb_w=0.6, nb_h=0.6, K=2), dict(iters=2, K=2), dict(T_pred=40, K=2), dict(soc_scale=2.0), dict(soc_scale=4.0)]:
    kw = dict(kw); sc = kw.pop("soc_scale", 1.0)
    run(str(kw) + " x%g" % sc, soc_scale=sc, **kw)
