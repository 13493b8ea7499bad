"""How far is the shared-factor RTS route (k_smoother_rts16's tables + k_mean_rts16) from the information-form per-series kernel and from the
oracle, on the C2 model at T = 1000?  (Under tests/: the oracle is test infrastructure.)  python tests/rts_accuracy_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import oracle
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine

W_C2 = np.array([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
T, N = 1000, 16
mat = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 6), np.arange(1, T + 1, dtype=np.float64))
p = DlmParameters([[1.0]], np.diag(W_C2), np.zeros(13), np.eye(13))
y = np.random.default_rng(0).standard_normal((N, T, 1)).cumsum(axis=1) * 0.3
eng = Engine(0)
om = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[0])
s = oracle.smoother(om, f)
runs = {"tables": _lib.OPT_NO_SMALL_BATCH, "per-series (information form)": _lib.OPT_SMOOTHER_PER_SERIES,
        "per-series, every step in full": _lib.OPT_NO_STEADY}
out = {k: eng.filter_smooth(mat, p, y, flags=v) for k, v in runs.items()}
for k in out:
    sm = np.array(out[k]["smooth"])[0]
    print(f"{k:34s} ({eng.last_variant if k == list(out)[-1] else ''}) vs oracle: means {np.abs(sm[:, :13] - s['s']).max():.2e}, covariances {np.abs(sm[:, 13:] - s['S']).max():.2e}"
          f"  (largest |S| {np.abs(s['S']).max():.3g}, largest |s| {np.abs(s['s']).max():.3g})")
a, b = np.array(out["tables"]["smooth"]), np.array(out["per-series (information form)"]["smooth"])
print("tables vs information form: means", np.abs(a[..., :13] - b[..., :13]).max(), "covariances", np.abs(a[..., 13:] - b[..., 13:]).max())
