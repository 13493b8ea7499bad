"""Diagnostic build (-DDLM_STAMP): steps of the d <= 15 forward / backward kernels that took the steady-state path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2(); N, T = 256, 1000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
eng = Engine(0)
y = eng.simulate(mat, p, N, seed=1, device=True, want_x=False)["y"]
out = eng.filter_smooth(mat, p, y)
st = out["status"].cpu().numpy().astype(np.int64) & 0xffffffff
print("forward steady steps:", ((st >> 8) & 0xfff)[:6], " backward steady steps:", ((st >> 20) & 0xfff)[:6], "of", T)
f = out["filt"].cpu().numpy()[0]
dC = [np.abs(f[t, 13:] - f[t - 1, 13:]).max() for t in range(1, T + 1)]
print("max |C_t - C_{t-1}| at t = 50, 100, 150, 200, 300, 500:", [dC[t - 1] for t in (50, 100, 150, 200, 300, 500)])
