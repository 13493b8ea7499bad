"""Backward-sampler kernel time at C4 (d = 40, p = 20, 2000 x T) for the statistics variants; tools only."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import multivariate_c4
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import materialise
from bayesian_dlms_amd.engine import Engine
mod, p = multivariate_c4(); N = int(os.environ.get("N", 2000))
eng = Engine(0)
for T in (250, 1000):
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    y = eng.simulate(mat, p, N, seed=1, device=True, want_x=False)["y"]
    for name, kw in (("outer stats", dict(flags=_lib.OPT_STATS_OUTER, want_stats=True, want_theta=False)),
                     ("diag stats", dict(flags=0, want_stats=True, want_theta=False)),
                     ("no stats, theta", dict(flags=0, want_stats=False, want_theta=True))):
        eng.ffbs(mat, p, y, seed=3, **kw)
        ts = []
        for _ in range(2):
            eng.ffbs(mat, p, y, seed=3, **kw); ts.append(eng.last_timing())
        print(json.dumps({"T": T, "case": name, "variant": eng.last_variant, "fwd_ms": round(min(t[0] for t in ts), 2), "bwd_ms": round(min(t[1] for t in ts), 2)}))
