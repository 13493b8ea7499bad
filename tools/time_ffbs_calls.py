"""On the GPU box: wall time of consecutive dlm_ffbs_batch calls (device-resident inputs), C4 Gibbs shape."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from bench import multivariate_c4
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
eng = Engine(0)
mod, p = multivariate_c4()
T, N = 1000, 2000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.randn(N, T, 20, dtype=torch.float64, device="cuda").cumsum(1) * 0.1
for flags in (_lib.OPT_STATS_OUTER, _lib.OPT_STATS_OUTER | _lib.OPT_SAMPLER_PER_SERIES):
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = eng.ffbs(mat, p, y, seed=i, flags=flags, want_theta=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(eng.last_variant, i, "%.2f ms" % ((t1 - t0) * 1e3), flush=True)
