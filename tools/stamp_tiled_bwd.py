import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine
mod = Dlm.polynomial(2)
for _ in range(19): mod = mod * Dlm.polynomial(2)
N, T = 512, 200
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
rng = np.random.default_rng(40); A = rng.standard_normal((40, 40))
p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
y = torch.as_tensor(rng.standard_normal((N, T, 20)).cumsum(axis=1), device="cuda:0")
eng = Engine(0)
for _ in range(2): out = eng.filter_smooth(mat, p, y)
st = out["status"][:20].cpu().numpy()
names = ["0 loop-top barrier", "1 record -> LDS, prefetch, y", "2 (mask inverse)", "3 C F | a", "4 K | e", "5 F^T K", "6 Qm^-1",
         "7 P C, P K | u, C q", "8 C(PC), X | u-K^T q", "9 output stores + barrier", "10 F X | r", "11 M (rank update)",
         "12 M G | q", "13 G^T (M G)"]
for k in range(14): print(f"  {names[k]:34s} {st[1+k]}")
print("  total", st[1:15].sum())
