"""Where the per-call overhead of the fused C2 call goes (tools only)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2, simulate
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine, _Device
mod, p = seasonal_c2(); N, T = 10000, 1000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.as_tensor(simulate(mat, p, N, seed=1), device="cuda")
eng = Engine(0)
rec = 13 + 169
out = {"filt": torch.empty((N, T + 1, rec), dtype=torch.float64, device="cuda"), "smooth": torch.empty((N, T + 1, rec), dtype=torch.float64, device="cuda"),
       "status": torch.empty((N,), dtype=torch.int32, device="cuda")}
for _ in range(3): eng.filter_smooth(mat, p, y, out=out)
torch.cuda.synchronize()
be = _Device(eng.device)
t0 = time.perf_counter()
for _ in range(20): eng.prepare(mat, p, N, be, 0)
torch.cuda.synchronize(); t_prep = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20): eng.filter_smooth(mat, p, y, out=out)
torch.cuda.synchronize(); t_call = (time.perf_counter() - t0) / 20
f, b = eng.last_timing()
print(json.dumps({"prepare_ms": t_prep * 1e3, "call_ms": t_call * 1e3, "kernels_ms": f + b, "overhead_ms": t_call * 1e3 - f - b}))
