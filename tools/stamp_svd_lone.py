"""One wave alone on the device running k_svd_filter (the series of zeros of the shared-factor SVD path): time per step and, in the
DLM_STAMP build (tools: build_tu_variant('stamp_svd', 'dlm_svd.hip', ['DLM_STAMP'])), sweeps and clock ticks inside the decompositions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2()
eng = Engine(0)
for T in (100, 400, 1000):
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    y = torch.zeros((8, T, 1), dtype=torch.float64, device="cuda:0")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = eng.svd_filter(mat, p, y, flags=_lib.OPT_SVD_PER_SERIES | _lib.OPT_COUNT_STEPS)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = out["status"].cpu().numpy()
    cnt = eng.last_counters()
    print(f"T={T}: {dt * 1e3:.3f} ms, {dt * 1e6 / T:.2f} us per step overall; steady steps {cnt[0]}; status words {st.tolist()}")
