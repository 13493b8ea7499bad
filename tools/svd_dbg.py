import sys, numpy as np
sys.path.insert(0, ".")
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine
e = Engine(0)
rng = np.random.default_rng(7)
for d in (2, 7, 16):
    T = 90
    Gm = 0.8 * np.eye(d) + 0.15 * np.eye(d, k=1)
    Fv = rng.standard_normal((d, 1))
    mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
    A = rng.standard_normal((d, d))
    p = DlmParameters([[0.9]], A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2)
    y = rng.standard_normal((6, T, 1)).cumsum(axis=1)
    sh = e.svd_filter(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    print(d, e.last_counters(), sh["status"])
    ps = e.svd_filter(mat, p, y, flags=_lib.OPT_SVD_PER_SERIES)
    df = np.abs(sh["svd"] - ps["svd"])
    bad = np.argwhere(df > 0)
    print("   max diff", df.max(), "count", len(bad), "first", bad[:4].tolist(), "t range", (bad[:,1].min(), bad[:,1].max()) if len(bad) else None, "cols", np.unique(bad[:,2])[:20] if len(bad) else None)
