"""On the GPU box: where the host time of one pooled Gibbs iteration (C3) goes."""
import sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, ".")
import torch
from bench import seasonal_c2, simulate
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
from bayesian_dlms_amd.gibbs import GibbsSampling, InverseGamma
eng = Engine(0)
mod, p = seasonal_c2()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
y = torch.randn(N, 1000, 1, dtype=torch.float64, device="cuda").cumsum(1) * 0.1
chain = GibbsSampling.sample(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p, mat.times, y, eng, n_iter=12, seed=7, pooled=True)
for _ in range(4):
    next(chain)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(6):
    next(chain)
torch.cuda.synchronize()
pr.disable()
print("ms per iteration", (time.perf_counter() - t0) / 6 * 1e3, "kernels fwd+bwd", eng.last_timing())
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
