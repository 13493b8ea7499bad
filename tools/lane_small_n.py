"""Lane-per-series vs wavefront-per-series at small batch sizes (d = 2, T = 1000); tools only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine
from bayesian_dlms_amd import _lib
eng = Engine(0); dev = "cuda:0"
mat = materialise(Dlm.polynomial(2), np.arange(1, 1001, dtype=np.float64))
p = DlmParameters([[2.0]], np.eye(2) * 0.5, np.zeros(2), np.eye(2) * 10.0)
for N in (1, 64, 1024, 8192):
    y = torch.randn((N, 1000, 1), device=dev, dtype=torch.float64).cumsum(dim=1)
    res = {}
    for name, fl in (("lane", 0), ("sparse16", _lib.OPT_NO_LANE)):
        eng.filter_smooth(mat, p, y, flags=fl); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): eng.filter_smooth(mat, p, y, flags=fl)
        e1.record(); torch.cuda.synchronize()
        res[name] = (eng.last_variant, e0.elapsed_time(e1) / 5)
    print(json.dumps({"N": N, **{k: {"variant": v[0], "ms": round(v[1], 3)} for k, v in res.items()}}))
