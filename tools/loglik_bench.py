"""dlm_loglik_batch at C2 (10 000 x 1000, d = 13): the forward recursion with one reduction per series and no record output."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2(); N, T = 10000, 1000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
eng = Engine(0)
y = eng.simulate(mat, p, N, seed=1, device=True, want_x=False)["y"]
for fl in (0, _lib.OPT_NO_STEADY):
    for _ in range(3): eng.loglik(mat, p, y, flags=fl)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): eng.loglik(mat, p, y, flags=fl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("flags", fl, "ms", round(dt * 1e3, 3), "series*steps/s", round(N * T / dt / 1e9, 2), "e9")
