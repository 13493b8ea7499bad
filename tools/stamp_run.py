import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2, simulate
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2(); N, T = 10000, 1000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.as_tensor(simulate(mat, p, N, seed=1), device="cuda:0")
eng = Engine(0)
for _ in range(3):
    out = eng.filter_smooth(mat, p, y)
st = out["status"][:9].cpu().numpy()
names = ["0 top+K exchange", "1 MFMA chain 1 (+b2)", "2 chain-2 issue", "3 PK exchange + M", "4 congruence + q", "5 output (x2, stores)", "6 -", "7 loop head"]
print("cycles per step, series 0 (s_memtime ticks):")
for k in range(8): print(f"  {names[k]:28s} {st[1+k]}")
print("  total", st[1:9].sum(), " timing", eng.last_timing())
