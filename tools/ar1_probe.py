"""AR(1) lane-kernel probe: filter-only and FFBS timings at N = 1e5, T = 1e3 (tools, not part of the product)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bayesian_dlms_amd.engine import Engine

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

eng = Engine(0); dev = "cuda:0"
N, T = int(os.environ.get("N", 100000)), int(os.environ.get("T", 1000))
rng = np.random.default_rng(3)
y = torch.as_tensor(rng.standard_normal((N, T)).cumsum(axis=1) * 0.1, device=dev)
v = torch.as_tensor(rng.uniform(0.2, 2.0, (N, T)), device=dev)
sv = torch.as_tensor(np.stack([rng.uniform(0.5, 0.95, N), rng.standard_normal(N), rng.uniform(0.1, 0.5, N)], axis=1), device=dev)
z = torch.randn((N, T + 1), device=dev, dtype=torch.float64)
for name, fn in [("filter only", lambda: eng.ar1_ffbs(y, v, sv, want_theta=False)),
                 ("ffbs philox", lambda: eng.ar1_ffbs(y, v, sv, seed=1, want_filt=False)),
                 ("ffbs given z", lambda: eng.ar1_ffbs(y, v, sv, z=z, want_filt=False))]:
    print(json.dumps({"case": name, "N": N, "T": T, "ms": timeit(fn)}))
