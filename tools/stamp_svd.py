import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2, simulate
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2(); N, T = 512, 200
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.as_tensor(simulate(mat, p, N, seed=1), device="cuda:0")
eng = Engine(0)
out = eng.svd_filter(mat, p, y)
st = out["status"][:8].cpu().numpy()
print("sweeps per step: time update", st[1] / T, " measurement update", st[2] / T)
print("cycles per step: Jacobi", st[3], " whole step", st[4])
