import sys, numpy as np
sys.path.insert(0, ".")
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine
e = Engine(0)
T, N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, 8
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
W = np.array([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
mat = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 6), np.arange(1, T + 1, dtype=np.float64))
p = DlmParameters([[1.0]], np.diag(W), np.zeros(13), np.eye(13))
rng = np.random.default_rng(1)
y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
sh = e.filter_smooth(mat, p, y, flags=flags | _lib.OPT_COUNT_STEPS)
print("counters", e.last_counters())
ps = e.filter_smooth(mat, p, y, flags=flags | _lib.OPT_NO_SHARED_COV)
for key in ("filt", "smooth"):
    a, b = sh[key], ps[key]
    for part, sl in (("mean", slice(0, 13)), ("cov", slice(13, None))):
        df = np.abs(a[..., sl] - b[..., sl])
        bad = np.argwhere(df > 0)
        print(key, part, "max diff", df.max(), "first mismatches (n, t, j):", bad[:5].tolist(), "count", len(bad), "of", df.size)
        if len(bad):
            ts = np.unique(bad[:, 1])
            print("   t range", ts.min(), ts.max(), "distinct t", len(ts), " rel", (df / (np.abs(b[..., sl]) + 1e-300)).max())
