"""Diagnostic build (-DDLM_STAMP): how many steps of the register-tile sampler (16 <= d <= 48) took the steady-state path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import multivariate_c4
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = multivariate_c4(); N, T = 512, int(os.environ.get("T", 400))
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
eng = Engine(0)
y = eng.simulate(mat, p, N, seed=1, device=True, want_x=False)["y"]
out = eng.ffbs(mat, p, y, seed=3, want_theta=False, want_stats=True, flags=_lib.OPT_STATS_OUTER)
st = out["status"].cpu().numpy()
print("variant", eng.last_variant, "steady-state steps per series (of %d):" % T, (st >> 8)[:8], "status bits", (st & 255)[:8])
f = out["filt"].cpu().numpy()[0]
dC = [np.abs(f[t, 40:] - f[t - 1, 40:]).max() / np.abs(f[t, 40:]).max() for t in range(1, T + 1)]
print("relative change of C_t:", ["%.1e" % x for x in dC[::25]])
