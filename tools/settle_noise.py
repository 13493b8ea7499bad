"""Rounding-noise level of the one-step covariance changes of the every-step kernels (what settle_test's `noise` stands for)."""
import sys, numpy as np
sys.path.insert(0, ".")
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine
e = Engine(0)
def run(name, mod, p, T, flags, svd=False):
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    rng = np.random.default_rng(1)
    y = rng.standard_normal((2, T, q)).cumsum(axis=1)
    out = e.filter_smooth(mat, p, y, flags=flags | _lib.OPT_NO_STEADY)
    for key in ("filt", "smooth"):
        C = out[key][0][:, d:]
        dl = np.abs(np.diff(C, axis=0)).max(axis=1) / np.abs(C[1:]).max(axis=1)
        print(f"{name:28s} {e.last_variant:10s} {key:6s} d={d:2d} one-step change / max|C| at the end: median {np.median(dl[-100:]):.2e} max {dl[-100:].max():.2e}   (t=20: {dl[20]:.2e})")
W_C2 = np.array([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
run("C2", Dlm.polynomial(1) + Dlm.seasonal(24, 6), DlmParameters([[1.0]], np.diag(W_C2), np.zeros(13), np.eye(13)), 1500, 0)
for k in (4, 8, 12, 17, 20, 24):
    mod = Dlm.polynomial(2)
    for _ in range(k - 1):
        mod = mod * Dlm.polynomial(2)
    d = 2 * k
    A = np.random.default_rng(90 + k).standard_normal((d, d))
    p = DlmParameters(np.eye(k) * 0.8, A @ A.T / d + 0.2 * np.eye(d), np.zeros(d), np.eye(d))
    run(f"|*| of {k} polynomial(2)", mod, p, 400, _lib.OPT_FORCE_WAVE)
