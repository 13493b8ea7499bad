"""Does a forward pass of one half of the batch overlap usefully with the backward pass of the other half? (tools only)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import seasonal_c2, simulate
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = seasonal_c2(); N, T = 10000, 1000
mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
y = torch.as_tensor(simulate(mat, p, N, seed=1), device="cuda")
rec = 13 + 169
def outs(n): return {"filt": torch.empty((n, T + 1, rec), dtype=torch.float64, device="cuda"), "smooth": torch.empty((n, T + 1, rec), dtype=torch.float64, device="cuda"), "status": torch.empty((n,), dtype=torch.int32, device="cuda")}
e0 = Engine(0); o0 = outs(N)
def base(): e0.filter_smooth(mat, p, y, out=o0)
for _ in range(3): base()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): base()
torch.cuda.synchronize(); tb = (time.perf_counter() - t0) / 10
res = {"single_stream_ms": tb * 1e3}
for chunks in (2, 4):
    engs = [Engine(0) for _ in range(chunks)]
    streams = [torch.cuda.Stream() for _ in range(chunks)]
    for e, s in zip(engs, streams): e.set_stream(s.cuda_stream)
    n = N // chunks
    ys = [y[i * n:(i + 1) * n].contiguous() for i in range(chunks)]
    os_ = [outs(n) for _ in range(chunks)]
    def run():
        for e, yy, oo in zip(engs, ys, os_): e.filter_smooth(mat, p, yy, out=oo, flags=_lib.OPT_ASYNC)
        torch.cuda.synchronize()
    for _ in range(3): run()
    t0 = time.perf_counter()
    for _ in range(10): run()
    res[f"{chunks}_streams_ms"] = (time.perf_counter() - t0) / 10 * 1e3
print(json.dumps(res))
