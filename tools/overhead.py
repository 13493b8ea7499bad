"""Host-side overhead of one dlm_filter_smooth_batch call (tiny workload), host-staged vs device-resident model."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from bench import seasonal_c2, simulate
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2()
T, N = 10, 4
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.as_tensor(simulate(mat, p, N, seed=1), device="cuda")
eng = Engine(0)
rec = mat.d + mat.d ** 2
out = {"filt": torch.empty((N, T + 1, rec), dtype=torch.float64, device="cuda"),
       "smooth": torch.empty((N, T + 1, rec), dtype=torch.float64, device="cuda"),
       "status": torch.empty((N,), dtype=torch.int32, device="cuda")}
for _ in range(5): eng.filter_smooth(mat, p, y, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): eng.filter_smooth(mat, p, y, out=out)
torch.cuda.synchronize(); print("per call us:", (time.perf_counter() - t0) / 200 * 1e6, eng.last_timing())
