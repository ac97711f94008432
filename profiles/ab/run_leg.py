"""Run ONE alt leg of bench.py by name (A/B helper for gpurun): python profiles/ab/run_leg.py training_step [--steps 6]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from desire_amd.spec import Dims, init_weights

ap = argparse.ArgumentParser(); ap.add_argument("leg"); ap.add_argument("--steps", type=int, default=6); ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda", 0)
d = Dims(n_scenes=512, mno=32, K=20, T_obs=8, T_pred=40, H=128, L=128, n_grids=1, grid_size=4, nb_w=0.1, nb_h=0.1, sx=1.0 / 1400.0, sy=1.0 / 1100.0, iters=1, posterior=1)
if a.leg == "training_step":
    out = bench.training_step_leg(d, a.seed, dev, a.steps)
elif a.leg == "config3_shape":
    out = bench.config3_shape_leg(a.seed, dev, a.steps)
elif a.leg == "few_windows":
    out = bench.few_windows_leg(d, a.seed, dev)
elif a.leg == "with_loader":
    out = bench.with_loader_leg(d, init_weights(d, a.seed), a.seed, dev, a.steps)
elif a.leg == "bf16_config2":
    out = bench.bf16_config2_leg(d, init_weights(d, a.seed), a.seed, dev, a.steps, with_accuracy=False)
else:
    raise SystemExit("unknown leg")
print(json.dumps(out, indent=1))
