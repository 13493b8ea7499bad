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
for _ in range(2): out = eng.filter(mat, p, y)
st = out["status"][:10].cpu().numpy()
print("NS iterations total", st[6], "direct fallbacks", st[7], "far breaks", st[9], "of", T, "steps")
names = ["0 advance (G C, T G^T, +W, a)", "1 forecast (RF, f, Q, +V, y)", "2 Q inverse", "3 K, m, C update", "4 output stores", "5", "6", "7 loop head"]
for k in range(8): print(f"  {names[k]:34s} {st[1+k]}")
print("  total", st[1:9].sum())
