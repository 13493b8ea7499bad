"""The README's usage example, runnable on an MI355X (tools only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, Data
from bayesian_dlms_amd.engine import Engine
from bayesian_dlms_amd.api import KalmanFilter, Smoothing, simulate
from bayesian_dlms_amd.gibbs import GibbsSampling, InverseGamma

eng = Engine(0)                                               # one engine handle per GPU
mod = Dlm.polynomial(1) + Dlm.seasonal(24, 3)                 # |+| and |*| compose models as in the reference
p = DlmParameters(v=[[2.0]], w=np.eye(7) * 0.1, m0=np.zeros(7), c0=np.eye(7) * 10.0)
times = np.arange(1.0, 201.0)
x, y = simulate(mod, times, p, eng, n_series=64, seed=1)       # Dlm.simulateRegular on the device, 64 realisations
y[:, 50:55] = np.nan                                           # missing observations are NaN (None in the reference)
ys = [[Data(t, y[n, i]) for i, t in enumerate(times)] for n in range(64)]
filtered = KalmanFilter.filter_dlm(mod, ys, p, eng)            # KfState per series and time, as KalmanFilter.filterDlm
smoothed = Smoothing.filter_smooth(mod, ys, p, eng)            # fused filter + RTS smoother
ll = KalmanFilter.log_likelihood(mod, ys, p, eng)              # one number per series
chain = list(GibbsSampling.sample(mod, InverseGamma(3.0, 3.0), InverseGamma(3.0, 0.3), p, times, y, eng, n_iter=20, seed=7))
print(len(filtered), len(filtered[0]), filtered[0][-1].mt[:2], float(ll[0]), chain[-1].p[0].v)
