"""Does the literal-Q1 smoother's S_t reach a fixed point bit for bit (what lets the table run k_smoother_rts16<EXP> skip the products of its
reused steps)?  Counts the steps with S_t == S_{t+1} in the C2 model's smoothed records.  python tools/rts_table_probe.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import Engine

W_C2 = np.array([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
T, N = 1000, 8
mat = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 6), np.arange(1, T + 1, dtype=np.float64))
p = DlmParameters([[1.0]], np.diag(W_C2), np.zeros(13), np.eye(13))
y = np.random.default_rng(0).standard_normal((N, T, 1)).cumsum(axis=1)
eng = Engine(0)
for name, fl in (("literal", _lib.OPT_SMOOTHER_COMPAT_Q1),):
    out = eng.filter_smooth(mat, p, y, flags=fl)
    S = np.array(out["smooth"])[0][:, 13:]
    C = np.array(out["filt"])[0][:, 13:]
    fixed = [t for t in range(T - 1, -1, -1) if np.array_equal(S[t], S[t + 1])]
    cfix = [t for t in range(T - 1, -1, -1) if np.array_equal(C[t], C[t + 1])]
    print(name, "steps with S_t == S_{t+1} bit for bit:", len(fixed), "highest", fixed[:3], "lowest", fixed[-3:])
    print(name, "steps with C_t == C_{t+1} bit for bit:", len(cfix), "highest", cfix[:3], "lowest", cfix[-3:])
    d = [float(np.abs(S[t] - S[t + 1]).max()) for t in (990, 950, 900, 800, 600, 450, 400)]
    print("max |S_t - S_{t+1}| at t = 990, 950, 900, 800, 600, 450, 400:", d)
