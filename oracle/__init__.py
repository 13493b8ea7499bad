"""ctypes bindings for the CPU oracle (oracle/dlm_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (bayesian_dlms_amd) never
imports this module.

All matrices are column-major (Breeze DenseMatrix.data order); numpy arrays are
passed as flat fp64 buffers.  Helper `cm(A)` flattens a 2-D numpy array
column-major, `from_cm(buf, r, c)` goes back.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DLM_ORACLE_LIB selects another build of the same file (the AddressSanitizer build of `make -C oracle asan`)
_LIB_PATH = os.environ.get("DLM_ORACLE_LIB") or os.path.join(_HERE, "libdlm_oracle.so")


def build(force=False):
    """Compile the C restatement with gcc (no GPU needed)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "dlm_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "libdlm_oracle.so"])
    return _LIB_PATH


_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_normal.restype = ctypes.c_double
        _lib.oracle_normal.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
    return _lib


def cm(a):
    """Column-major flat fp64 copy of a 1-D/2-D array."""
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(a.T).reshape(-1) if a.ndim == 2 else np.ascontiguousarray(a).reshape(-1)


def from_cm(buf, r, c):
    return np.asarray(buf, dtype=np.float64).reshape(c, r).T.copy()


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


class Model:
    """Host-materialised model: F [nF][d*p], G [nG][d*d] (column-major each),
    g_index [T] (or None), dt [T] (or None)."""

    def __init__(self, d, p, T, F, G, g_index=None, dt=None, f_stride=0):
        self.d, self.p, self.T = int(d), int(p), int(T)
        self.F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1)
        self.G = np.ascontiguousarray(G, dtype=np.float64).reshape(-1)
        self.g_index = None if g_index is None else np.ascontiguousarray(g_index, dtype=np.int32)
        self.dt = None if dt is None else np.ascontiguousarray(dt, dtype=np.float64)
        self.f_stride = int(f_stride)


def _tv(A, n):
    """A single n x n matrix, or a [T] stream of them -> (flat column-major array, per-time stride)."""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 3:
        return np.ascontiguousarray(np.transpose(A, (0, 2, 1))).reshape(-1), n * n
    return cm(A), 0


def kf_filter(M, V, W, m0, C0, y):
    """KalmanFilter(...).filter: returns dict of m,C,a,R,f,Q with T+1 records
    (matrices column-major flattened in the last axis).  V / W may be [T] streams of matrices."""
    d, p, T = M.d, M.p, M.T
    if np.ndim(V) == 3 or np.ndim(W) == 3:
        y = np.ascontiguousarray(y, dtype=np.float64).reshape(T, p)
        out = {"m": np.empty((T + 1, d)), "C": np.empty((T + 1, d * d)), "a": np.empty((T + 1, d)),
               "R": np.empty((T + 1, d * d)), "f": np.empty((T + 1, p)), "Q": np.empty((T + 1, p * p))}
        (Vf, vts), (Wf, wts) = _tv(V, p), _tv(W, d)
        m0, C0 = cm(m0), cm(C0)
        out["rc"] = lib().oracle_kf_filter_tv(
            d, p, T, _p(M.F), ctypes.c_long(M.f_stride), _p(M.G), _pi(M.g_index), _p(M.dt),
            _p(Vf), ctypes.c_long(vts), _p(Wf), ctypes.c_long(wts), _p(m0), _p(C0), _p(y),
            _p(out["m"]), _p(out["C"]), _p(out["a"]), _p(out["R"]), _p(out["f"]), _p(out["Q"]))
        return out
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(T, p)
    out = {
        "m": np.empty((T + 1, d)), "C": np.empty((T + 1, d * d)),
        "a": np.empty((T + 1, d)), "R": np.empty((T + 1, d * d)),
        "f": np.empty((T + 1, p)), "Q": np.empty((T + 1, p * p)),
    }
    V, W, m0, C0 = cm(V), cm(W), cm(m0), cm(C0)
    rc = lib().oracle_kf_filter(
        d, p, T, _p(M.F), ctypes.c_long(M.f_stride), _p(M.G), _pi(M.g_index), _p(M.dt),
        _p(V), _p(W), _p(m0), _p(C0), _p(y),
        _p(out["m"]), _p(out["C"]), _p(out["a"]), _p(out["R"]), _p(out["f"]), _p(out["Q"]))
    out["rc"] = rc
    return out


def loglik(M, filt, y):
    """Sum of KalmanFilter.conditionalLikelihood over the series, from the forecasts of kf_filter."""
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(M.T, M.p)
    fn = lib().oracle_loglik
    fn.restype = ctypes.c_double
    return float(fn(M.p, M.T, _p(filt["f"]), _p(filt["Q"]), _p(y)))


def likelihood_q7(M, filt, W):
    """KalmanFilter.likelihood (KalmanFilter.scala:299-306): the transition density of the filtered means,
    sum_t log N(m_t; g(dt_t) m_{t-1}, W dt_t) (KalmanFilter.logLikelihood :175-183) -- what MetropolisHastings.dlm evaluates."""
    out = ctypes.c_double()
    lib().oracle_loglik_q7(M.d, M.T, _p(M.G), _pi(M.g_index), _p(M.dt), _p(cm(W)), _p(np.ascontiguousarray(filt["m"])), ctypes.byref(out))
    return float(out.value)


def ar1_filter(y, v, phi, mu, sigma_eta):
    """FilterAr.filterUnivariate: T+1 records of (m, c, a, r)."""
    y = np.ascontiguousarray(y, dtype=np.float64); T = y.size
    v = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (T,)))
    out = {k: np.empty(T + 1) for k in ("m", "c", "a", "r")}
    lib().oracle_ar1_filter(T, _p(y), _p(v), ctypes.c_double(phi), ctypes.c_double(mu), ctypes.c_double(sigma_eta),
                            _p(out["m"]), _p(out["c"]), _p(out["a"]), _p(out["r"]))
    return out


def ar1_backward_sample(filt, phi, z):
    """FilterAr.univariateSample with injected normals z [T+1]."""
    T = filt["m"].size - 1
    z = np.ascontiguousarray(z, dtype=np.float64)
    theta = np.empty(T + 1)
    lib().oracle_ar1_backward_sample(T, ctypes.c_double(phi), _p(filt["m"]), _p(filt["c"]), _p(filt["a"]), _p(filt["r"]),
                                     _p(z), _p(theta))
    return theta


def ou_filter(times, y, v, phi, mu, sigma_eta):
    """FilterOu.filterUnivariate (literal c0 and initial time): T+1 records of (m, c, a, r)."""
    y = np.ascontiguousarray(y, dtype=np.float64); T = y.size
    times = np.ascontiguousarray(times, dtype=np.float64)
    v = np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (T,)))
    out = {k: np.empty(T + 1) for k in ("m", "c", "a", "r")}
    lib().oracle_ou_filter(T, _p(times), _p(y), _p(v), ctypes.c_double(phi), ctypes.c_double(mu), ctypes.c_double(sigma_eta),
                           _p(out["m"]), _p(out["c"]), _p(out["a"]), _p(out["r"]))
    return out


def ou_backward_sample(times, filt, phi, z):
    T = filt["m"].size - 1
    times = np.ascontiguousarray(times, dtype=np.float64)
    z = np.ascontiguousarray(z, dtype=np.float64)
    theta = np.empty(T + 1)
    lib().oracle_ou_backward_sample(T, _p(times), ctypes.c_double(phi), _p(filt["m"]), _p(filt["c"]), _p(filt["a"]),
                                    _p(filt["r"]), _p(z), _p(theta))
    return theta


def smoother(M, filt, compat_q1=False):
    d, T = M.d, M.T
    s = np.empty((T + 1, d)); S = np.empty((T + 1, d * d))
    rc = lib().oracle_smoother(d, T, _p(M.G), _pi(M.g_index), _p(filt["m"]), _p(filt["C"]),
                               _p(filt["a"]), _p(filt["R"]), int(bool(compat_q1)), _p(s), _p(S))
    return {"s": s, "S": S, "rc": rc}


def backward_sample(M, W, filt, z, factor="eig"):
    d, T = M.d, M.T
    z = np.ascontiguousarray(z, dtype=np.float64).reshape(T + 1, d)
    theta = np.empty((T + 1, d)); h = np.empty((T + 1, d)); H = np.empty((T + 1, d * d))
    W, wts = _tv(W, d)
    rc = lib().oracle_backward_sample_tv(d, T, _p(M.G), _pi(M.g_index), _p(M.dt), _p(W), ctypes.c_long(wts),
                                         _p(filt["m"]), _p(filt["C"]), _p(filt["a"]), _p(filt["R"]),
                                         _p(z), 0 if factor == "eig" else 1, _p(theta), _p(h), _p(H))
    return {"theta": theta, "h": h, "H": H, "rc": rc}


def gibbs_stats(M, y, theta, want_outer=False):
    d, p, T = M.d, M.p, M.T
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(T, p)
    theta = np.ascontiguousarray(theta, dtype=np.float64).reshape(T + 1, d)
    ssy = np.empty(p); n = np.empty(p); ss = np.empty(d)
    outer = np.empty(d * d) if want_outer else None
    lib().oracle_gibbs_stats(d, p, T, _p(M.F), ctypes.c_long(M.f_stride), _p(M.G), _pi(M.g_index),
                             _p(M.dt), _p(y), _p(theta), _p(ssy), _p(n), _p(ss), _p(outer))
    return {"ssy": ssy, "n": n, "ss": ss, "outer": outer}


def sqrt_svd(Mx, inverse=False):
    Mx = np.asarray(Mx, dtype=np.float64)
    n = Mx.shape[0]
    out = np.empty(n * n)
    flat = cm(Mx)
    lib().oracle_sqrt_svd(n, _p(flat), int(bool(inverse)), _p(out))
    return from_cm(out, n, n)


def svd_filter(M, V, W, m0, C0, y, raw_w_q2=False):
    """SvdFilter (SvdFilter.scala:38-95, :183-236).  V / W may be [T] streams of matrices (DlmFsv.ffbsSvd, DlmFsvSystem.ffbsSvd)."""
    d, p, T = M.d, M.p, M.T
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(T, p)
    out = {"m": np.empty((T + 1, d)), "dc": np.empty((T + 1, d)), "uc": np.empty((T + 1, d * d)),
           "a": np.empty((T + 1, d)), "dr": np.empty((T + 1, d)), "ur": np.empty((T + 1, d * d))}
    (Vf, vts), (Wf, wts) = _tv(V, p), _tv(W, d)
    m0, C0 = cm(m0), cm(C0)
    lib().oracle_svd_filter_tv(d, p, T, _p(M.F), ctypes.c_long(M.f_stride), _p(M.G), _pi(M.g_index),
                               _p(M.dt), _p(Vf), ctypes.c_long(vts), _p(Wf), ctypes.c_long(wts), _p(m0), _p(C0), _p(y), int(bool(raw_w_q2)),
                               _p(out["m"]), _p(out["dc"]), _p(out["uc"]), _p(out["a"]),
                               _p(out["dr"]), _p(out["ur"]))
    return out


def svd_backward_sample(M, W, sf, z, literal_q9=True):
    """SvdSampler (SvdSampler.scala:15-60).  W may be a [T] stream (the step from record t uses W_t)."""
    d, T = M.d, M.T
    z = np.ascontiguousarray(z, dtype=np.float64).reshape(T + 1, d)
    theta = np.empty((T + 1, d)); h = np.empty((T + 1, d))
    dh = np.empty((T + 1, d)); uh = np.empty((T + 1, d * d))
    Wf, wts = _tv(W, d)
    lib().oracle_svd_backward_sample_tv(d, T, _p(M.G), _pi(M.g_index), _p(Wf), ctypes.c_long(wts), _p(sf["m"]), _p(sf["dc"]),
                                        _p(sf["uc"]), _p(sf["a"]), _p(z), int(bool(literal_q9)), _p(theta), _p(h), _p(dh), _p(uh))
    return {"theta": theta, "h": h, "dh": dh, "uh": uh}


def dinvgamma_step(d, p, stats, av, bv, aw, bw, seed, series, iteration):
    """The engine's device d-Inverse-Gamma step for one series, restated on the CPU: (diag V, diag W)."""
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    v = np.empty(p); w = np.empty(d)
    lib().oracle_dinvgamma_step(d, p, _p(stats), ctypes.c_double(av), ctypes.c_double(bv), ctypes.c_double(aw),
                                ctypes.c_double(bw), ctypes.c_uint64(seed), ctypes.c_uint64(series),
                                ctypes.c_uint64(iteration), _p(v), _p(w))
    return v, w


def simulate(M, V, W, m0, C0, seed, series):
    """Dlm.simulateRegular on the engine's Philox convention: (x [T+1][d], y [T][p])."""
    d, p, T = M.d, M.p, M.T
    x = np.empty((T + 1, d)); y = np.empty((T, p))
    V, W, m0, C0 = cm(V), cm(W), cm(m0), cm(C0)
    lib().oracle_simulate(d, p, T, _p(M.F), ctypes.c_long(M.f_stride), _p(M.G), _pi(M.g_index), _p(M.dt),
                          _p(V), _p(W), _p(m0), _p(C0), ctypes.c_uint64(seed), ctypes.c_uint64(series), _p(x), _p(y))
    return x, y


def normals(seed, series, T1, d):
    z = np.empty((T1, d))
    lib().oracle_normals(ctypes.c_uint64(seed), ctypes.c_uint64(series), T1, d, _p(z))
    return z


def filter_smooth_batch(N, M, V, W, m0, C0, y, want_out=True):
    """Batched CPU baseline (OpenMP over series). y [N][T][p]."""
    d, p, T = M.d, M.p, M.T
    y = np.ascontiguousarray(y, dtype=np.float64).reshape(N, T, p)
    rec = d + d * d
    filt = np.empty((N, T + 1, rec)) if want_out else None
    sm = np.empty((N, T + 1, rec)) if want_out else None
    V, W, m0, C0 = cm(V), cm(W), cm(m0), cm(C0)
    bad = lib().oracle_filter_smooth_batch(N, d, p, T, _p(M.F), _p(M.G), _p(V), _p(W), _p(m0), _p(C0),
                                           _p(y), _p(filt), _p(sm))
    return filt, sm, bad


def set_threads(n=0):
    """OpenMP threads of filter_smooth_batch (0: leave as is); returns the number in force."""
    return int(lib().oracle_set_threads(int(n)))
