import sys, numpy as np, torch
sys.path.insert(0, '.')
from desire_amd import _lib
from desire_amd.spec import Dims, init_weights
from desire_amd.synth import make_case
out = sys.argv[1]
res = {}
for tag, kw in (("c3", dict(n_scenes=2, mno=64, K=5, H=256)), ("m160", dict(n_scenes=1, mno=160, K=2, H=128)), ("h64", dict(n_scenes=1, mno=192, K=2, H=64))):
    d = Dims(T_obs=8, T_pred=12, L=128, n_grids=1, grid_size=4, nb_w=0.15, nb_h=0.15, sx=1/1400., sy=1/1100., iters=1, posterior=1, bf16=2, **kw)
    w = init_weights(d, 0)
    past, fut, eps, grids, gos = make_case(d, seed=1, n_absent=3)
    h = _lib.Handle(d); h.set_weights(w)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), device="cuda")
    p, f, e, g = t(past), t(fut), t(eps), t(grids)
    h.set_scene_grids(g.data_ptr(), gos)
    Y = torch.zeros((d.R, d.T_pred, 2), device="cuda"); sc = torch.zeros((d.R,), device="cuda")
    h.forward(p.data_ptr(), f.data_ptr(), e.data_ptr(), Y.data_ptr(), sc.data_ptr())
    torch.cuda.synchronize()
    res[tag + "_Y"] = Y.cpu().numpy(); res[tag + "_s"] = sc.cpu().numpy()
np.savez(out, **res)
