"""On the GPU box: wall time per pooled Inverse-Wishart Gibbs iteration on the C4 model, shared factors then own factors."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from bench import multivariate_c4
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
from bayesian_dlms_amd.gibbs import GibbsWishart, InverseGamma, InverseWishart
eng = Engine(0)
mod, p = multivariate_c4()
T, N = 1000, 2000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.randn(N, T, 20, dtype=torch.float64, device="cuda").cumsum(1) * 0.1
for name, extra in (("shared", 0), ("own", _lib.OPT_SAMPLER_PER_SERIES), ("shared", 0), ("own", _lib.OPT_SAMPLER_PER_SERIES)):
    ffbs = (lambda *a_, flags=0, **k_: eng.ffbs(*a_, flags=flags | extra, **k_))
    chain = GibbsWishart.sample(mod, InverseGamma(5.0, 4.0), InverseWishart(42.0, np.eye(40)), p, mat.times, y, eng, n_iter=5, seed=7, pooled=True, ffbs=ffbs)
    for i in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = next(chain)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(name, eng.last_variant, i, "%.2f ms" % ((t1 - t0) * 1e3), "kernels %.2f + %.2f" % eng.last_timing(), flush=True)
    torch.cuda.empty_cache()

