"""The steady-state shortcut in the slow regime ("what's weak" 1).

The reference recomputes C_t, K_t, Q_t and the smoother's matrices at every step (KalmanFilter.scala:64-107,
Smoothing.scala:31-47).  The kernels stop once the recursion has converged -- judged by settle_test (dlm_internal.h), which
bounds the geometric tail of the step-to-step changes by DLM_SETTLE_TOL = 1e-12 of the largest entry instead of looking at one
step's change.  These tests run models that contract slowly (small W / V, large V; long series, the reference's own AR example
is T = 5000) on every kernel that freezes, against DLM_OPT_NO_STEADY (or the kernel's every-step switch) AND against the oracle,
with the covariances compared relative to their own scale (max |C_t| of the record), not to the means.

Tolerances (fp64): shortcut against the every-step recursion 5e-12 of max|C_t| (filtered covariances), 5e-11 of max|S_t|
(smoothed), means 1e-10 of the largest mean; against the oracle the tolerances DESIGN.md section 2 states."""
import numpy as np
import pytest

import oracle
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise

pytestmark = pytest.mark.gpu


W_C2 = np.array([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])


@pytest.fixture(scope="module")
def eng():
    from bayesian_dlms_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def omodel(mat):
    return oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)


def c2(T, wscale=1.0, v=1.0):
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    return mat, DlmParameters([[v]], np.diag(W_C2 * wscale), np.zeros(13), np.eye(13))


def rel_cov_err(a, b, d):
    """largest |a - b| over the covariance part of each record, relative to that record's largest covariance entry"""
    Ca, Cb = a[..., d:], b[..., d:]
    scale = np.abs(Cb).max(axis=-1, keepdims=True)
    return float((np.abs(Ca - Cb) / scale).max())


def rel_mean_err(a, b, d):
    return float(np.abs(a[..., :d] - b[..., :d]).max() / np.abs(b[..., :d]).max())


SLOW = [  # (W scale, V, T, may the forward pass freeze at all within T?)
    (1.0, 1.0, 1000, True),          # the bench model: freezes near t = 383 (one-step rule of round 2: 355)
    (1e-2, 1.0, 5000, True),         # freezes near t = 1363 (round 2: 1127, 9e-11 off)
    (1e-3, 1.0, 5000, False),        # round 2 froze at t = 3123, 7e-10 off; now recomputed to the end
    (1e-4, 1.0, 10000, False),       # round 2 froze at t = 8731, 6e-9 off
    (1.0, 1e4, 10000, False),        # large observation variance with the bench's W: round 2 froze at t = 8631, 6e-9 off
]


@pytest.mark.parametrize("wscale,v,T,freezes", SLOW)
def test_structured_kernels_slow_regime(eng, wscale, v, T, freezes):
    """sparse16 forward + backward kernels (the C2 path), the log-likelihood instantiation and the packed records."""
    mat, p = c2(T, wscale, v)
    rng = np.random.default_rng(5)
    N = 4
    y = (rng.standard_normal((N, T, 1)) * np.sqrt(v)).cumsum(axis=1) * 0.05 + rng.standard_normal((N, T, 1)) * np.sqrt(v)
    fast = eng.filter_smooth(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "sparse16" and np.all(fast["status"] == 0)
    nf, nb, _, _ = eng.last_counters()
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[:2] == (0, 0)
    if freezes:
        assert nf > 0 and nb > 0, "the shortcut never engaged on a model that settles well inside the series"
    else:
        assert nf == 0 and nb == 0, f"froze ({nf}, {nb} steps) although the tail bound cannot be met within T"
    assert rel_cov_err(fast["filt"], full["filt"], 13) <= 5e-12
    assert rel_cov_err(fast["smooth"], full["smooth"], 13) <= 5e-11
    assert rel_mean_err(fast["filt"], full["filt"], 13) <= 1e-10
    assert rel_mean_err(fast["smooth"], full["smooth"], 13) <= 1e-10
    # against the oracle (one series, the whole length): DESIGN.md section 2 tolerances, covariances relative to their scale
    om = omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[0])
    s = oracle.smoother(om, f, compat_q1=False)
    assert rel_cov_err(fast["filt"][0], np.concatenate([f["m"], f["C"]], axis=1), 13) <= 1e-9
    assert rel_cov_err(fast["smooth"][0], np.concatenate([s["s"], s["S"]], axis=1), 13) <= 1e-8
    np.testing.assert_allclose(fast["filt"][0][:, :13], f["m"], rtol=1e-9, atol=1e-9 * np.abs(f["m"]).max())
    np.testing.assert_allclose(fast["smooth"][0][:, :13], s["s"], rtol=1e-8, atol=1e-8 * np.abs(s["s"]).max())
    # the log-likelihood instantiation reuses log Q of the settled forecast variance
    ll = eng.loglik(mat, p, y)["loglik"]
    ll0 = eng.loglik(mat, p, y, flags=_lib.OPT_NO_STEADY)["loglik"]
    np.testing.assert_allclose(ll, ll0, rtol=1e-11)
    np.testing.assert_allclose(ll[0], oracle.loglik(om, f, y[0]), rtol=1e-9)


def test_freeze_gap_refreeze_late(eng):
    """A series that freezes, meets a gap and freezes again late; one that is disturbed every few hundred steps; packed records."""
    T = 6000
    mat, p = c2(T, 1e-2, 1.0)
    rng = np.random.default_rng(6)
    y = rng.standard_normal((3, T, 1)).cumsum(axis=1) * 0.1 + rng.standard_normal((3, T, 1))
    y[1, 3000:3004, 0] = np.nan
    y[2, 700::900, 0] = np.nan
    fast = eng.filter_smooth(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    nf, nb, _, _ = eng.last_counters()
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    assert nf > 3 * 2000 and nb > 0
    assert rel_cov_err(fast["filt"], full["filt"], 13) <= 5e-12
    assert rel_cov_err(fast["smooth"], full["smooth"], 13) <= 5e-11
    assert rel_mean_err(fast["smooth"], full["smooth"], 13) <= 1e-10
    C1 = fast["filt"][1][:, 13:]
    assert np.array_equal(C1[2999], C1[2900])            # frozen before the gap
    assert not np.array_equal(C1[3010], C1[2999])        # moving after it
    assert np.array_equal(C1[5999], C1[5900])            # frozen again late
    pk = eng.filter_smooth(mat, p, y, flags=_lib.OPT_PACKED_SYM)
    assert rel_cov_err(eng.unpack_records(13, pk["smooth"]), full["smooth"], 13) <= 5e-11


@pytest.mark.parametrize("N", [7, 3300])
def test_series_with_many_gaps_take_every_step_in_full(eng, N):
    """A series that misses more than T / 256 of its observations is marked from its data alone (k_count_gaps) and never tests its
    covariance recursion for convergence: forward kernel without the test, backward kernel in the instantiation without the
    shortcut's machinery -- bit for bit the DLM_OPT_NO_STEADY call for that series, whatever the batch (both batch-size variants
    of the backward kernel); the other series keep the shortcut."""
    T = 700
    mat, p = c2(T)
    rng = np.random.default_rng(N)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    gaps = rng.random((T,)) < 0.05
    y[1, gaps, 0] = np.nan                                  # ~35 gaps: marked
    y[3, 450, 0] = np.nan                                   # one gap: not marked
    fast = eng.filter_smooth(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    nf = eng.last_counters()[0]
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    assert nf > 0                                           # (the others do take their short steps)
    for name in ("filt", "smooth"):
        assert np.array_equal(fast[name][1], full[name][1], equal_nan=True), name       # the marked series: the every-step path
        assert rel_cov_err(fast[name], full[name], 13) <= 5e-11
    C3 = fast["filt"][3][:, 13:]
    assert np.array_equal(C3[449], C3[430])                 # the series with one gap had settled before it
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[1])
    s_ = oracle.smoother(omodel(mat), f)
    np.testing.assert_allclose(fast["smooth"][1][:, :13], s_["s"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(fast["smooth"][1][:, 13:], s_["S"], rtol=1e-8, atol=1e-9)
    alone = eng.filter_smooth(mat, p, y[1:2])               # the same series in a batch of one: the same bits
    assert np.array_equal(alone["smooth"][0], fast["smooth"][1], equal_nan=True)


@pytest.mark.parametrize("wscale,T", [(1.0, 300), (1e-3, 3000)])
def test_per_wave_kernels_slow_regime(eng, wscale, T):
    """k_filter_w48 / k_smoother_w48 (16 <= d <= 48): the C4-type model with its W scaled down."""
    rng = np.random.default_rng(7)
    mod = Dlm.polynomial(2)
    for _ in range(9):
        mod = mod * Dlm.polynomial(2)              # d = 20, p = 10
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A = rng.standard_normal((d, d))
    p = DlmParameters(np.eye(q), (A @ A.T / d + 0.1 * np.eye(d)) * wscale, np.zeros(d), np.eye(d))
    y = rng.standard_normal((3, T, q)).cumsum(axis=1) * 0.1 + rng.standard_normal((3, T, q))
    y[1, T // 2, :] = np.nan
    fast = eng.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE | _lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "wave-mfma" and np.all(fast["status"] == 0)
    nf, nb, _, _ = eng.last_counters()
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE | _lib.OPT_NO_STEADY)
    if wscale == 1.0:
        assert nf > 0 and nb > 0
    assert rel_cov_err(fast["filt"], full["filt"], d) <= 5e-12
    assert rel_cov_err(fast["smooth"], full["smooth"], d) <= 5e-11
    assert rel_mean_err(fast["filt"], full["filt"], d) <= 1e-10
    assert rel_mean_err(fast["smooth"], full["smooth"], d) <= 1e-10
    om = omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[0])
    s = oracle.smoother(om, f, compat_q1=False)
    assert rel_cov_err(fast["filt"][0], np.concatenate([f["m"], f["C"]], axis=1), d) <= 1e-8
    assert rel_cov_err(fast["smooth"][0], np.concatenate([s["s"], s["S"]], axis=1), d) <= 1e-7


@pytest.mark.parametrize("wscale,T", [(1.0, 600), (1.0, 1400), (1e-2, 2500)])   # (T = 1400: the smoothed covariance itself settles in the middle of the series)
def test_sampler_and_rts_kernels_slow_regime(eng, wscale, T):
    """sparse16-sampler (reference-form backward sampler) and sparse16-rts (dlm_smooth_batch, literal Q1): they reuse J, H and
    the factor while THIS step's filtered covariance is within DLM_SETTLE_TOL max|C| of the one they were computed from -- a
    bound on the distance itself.  Draws against the oracle under injected normals (1e-7), the RTS smoother
    against the fused information-form pass run with DLM_OPT_NO_STEADY and against the oracle."""
    mat, p = c2(T, wscale, 1.0)
    rng = np.random.default_rng(8)
    N = 3
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.1 + rng.standard_normal((N, T, 1))
    z = rng.standard_normal((N, T + 1, 13))
    out = eng.ffbs(mat, p, y, z=z, want_cond=True)
    assert eng.last_variant == "sparse16-sampler" and np.all(out["status"] == 0)
    om = omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[0])
    o = oracle.backward_sample(om, p.w, f, z[0], factor="chol")
    scale = np.abs(o["theta"]).max()
    assert np.abs(out["theta"][0] - o["theta"]).max() <= 1e-7 * scale
    exact = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    sm = eng.smooth(mat, p, exact["filt"])
    assert eng.last_variant == "sparse16-rts"
    assert rel_cov_err(sm["smooth"], exact["smooth"], 13) <= 1e-9
    assert rel_mean_err(sm["smooth"], exact["smooth"], 13) <= 1e-9
    q1 = eng.filter_smooth(mat, p, y[:1], flags=_lib.OPT_SMOOTHER_COMPAT_Q1)
    s1 = oracle.smoother(om, f, compat_q1=True)
    assert rel_cov_err(q1["smooth"][0], np.concatenate([s1["s"], s1["S"]], axis=1), 13) <= 1e-8


@pytest.mark.parametrize("wscale,T", [(1.0, 700), (1e-2, 2500)])
def test_svd_filter_slow_regime(eng, wscale, T):
    """svd-jacobi: the factors count as settled by the same tail bound; DLM_OPT_FORCE_GENERIC recomputes both decompositions
    at every step.  Covariances U D^2 U^T relative to their scale: 1e-9 against the every-step path, 1e-7 against the oracle's
    Kalman filter (the tolerance of this path)."""
    mat, p = c2(T, wscale, 1.0)
    rng = np.random.default_rng(9)
    y = rng.standard_normal((3, T, 1)).cumsum(axis=1) * 0.1 + rng.standard_normal((3, T, 1))
    y[1, T // 2, 0] = np.nan
    fast = eng.svd_filter(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    ns = eng.last_counters()[0]
    assert eng.last_variant == "svd-jacobi" and np.all(fast["status"] == 0)
    full = eng.svd_filter(mat, p, y, flags=_lib.OPT_FORCE_GENERIC)
    if wscale == 1.0:
        assert ns > 0, "the SVD filter never reused its decompositions on a model that settles within ~400 steps"

    def cov(rec):
        m = rec[:, :13]; dc = rec[:, 13:26]; U = rec[:, 26:].reshape(-1, 13, 13).transpose(0, 2, 1)
        return m, np.einsum("tij,tj,tkj->tik", U, dc * dc, U)

    om = omodel(mat)
    for n in range(3):
        m, C = cov(fast["svd"][n]); m2, C2 = cov(full["svd"][n])
        sc = np.abs(C2).max(axis=(1, 2), keepdims=True)
        assert (np.abs(C - C2) / sc).max() <= 1e-9
        assert np.abs(m - m2).max() <= 1e-9 * np.abs(m2).max()
        kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        Co = kf["C"].reshape(T + 1, 13, 13).transpose(0, 2, 1)
        assert (np.abs(C - Co) / np.abs(Co).max(axis=(1, 2), keepdims=True)).max() <= 1e-7
        assert np.abs(m - kf["m"]).max() <= 1e-7 * np.abs(kf["m"]).max()


def test_model_unchanged_promise_is_verified(eng):
    """DLM_OPT_MODEL_UNCHANGED from a device-memory caller is checked with a device checksum of F, G and the time grid: a broken
    promise fails the call instead of running the kernels on another model's structure tables."""
    import ctypes
    import torch
    from bayesian_dlms_amd.engine import EngineError
    T = 60
    rng = np.random.default_rng(10)
    m1 = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 6), np.arange(1, T + 1, dtype=np.float64))
    m2 = materialise(Dlm.polynomial(1) + Dlm.seasonal(12, 6), np.arange(1, T + 1, dtype=np.float64))   # same shape, another G
    p = DlmParameters([[1.0]], np.diag(W_C2), np.zeros(13), np.eye(13))
    y = torch.as_tensor(rng.standard_normal((4, T, 1)), device="cuda:0")
    ref1 = eng.filter_smooth(m1, p, y.cpu().numpy())
    ref2 = eng.filter_smooth(m2, p, y.cpu().numpy())

    def raw_call(mat, flags):
        dev = lambda a, dt=np.float64: torch.as_tensor(np.ascontiguousarray(a, dtype=dt), device="cuda:0")
        F, G = dev(mat.F), dev(mat.G)
        V, W, m0, C0 = dev(p.v), dev(p.w), dev(p.m0), dev(p.c0)
        md = _lib.ModelDesc(13, 1, T, 4, F.data_ptr(), 0, G.data_ptr(), 1, None, None)
        pd = _lib.ParamsDesc(V.data_ptr(), 0, W.data_ptr(), 0, m0.data_ptr(), 0, C0.data_ptr(), 0, 0, 0)
        filt = torch.empty((4, T + 1, 182), dtype=torch.float64, device="cuda:0")
        sm = torch.empty_like(filt)
        st = torch.zeros(4, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        rc = eng.lib.dlm_filter_smooth_batch(eng.h, md, pd, ctypes.c_void_p(y.data_ptr()), _lib.Options(flags, _lib.DLM_MEM_DEVICE, 0, 0),
                                             ctypes.c_void_p(filt.data_ptr()), ctypes.c_void_p(sm.data_ptr()), ctypes.c_void_p(st.data_ptr()))
        return rc, sm.cpu().numpy()

    rc, sm = raw_call(m1, 0)
    assert rc == 0
    np.testing.assert_allclose(sm, ref1["smooth"], rtol=1e-12, atol=1e-12)
    rc, sm = raw_call(m1, _lib.OPT_MODEL_UNCHANGED)                      # the promise holds
    assert rc == 0
    np.testing.assert_allclose(sm, ref1["smooth"], rtol=1e-12, atol=1e-12)
    rc, _ = raw_call(m2, _lib.OPT_MODEL_UNCHANGED)                       # broken: another G in the same shape
    assert rc == -1 and b"DLM_OPT_MODEL_UNCHANGED" in eng.lib.dlm_last_error(eng.h)
    rc, sm = raw_call(m2, 0)                                             # the engine analyses afresh afterwards
    assert rc == 0
    np.testing.assert_allclose(sm, ref2["smooth"], rtol=1e-12, atol=1e-12)
    assert isinstance(EngineError("x"), RuntimeError)


# ------------------------------------------------------------------------------------------------------------------------
# KalmanFilter.likelihood literally (DLM_OPT_LOGLIK_LITERAL_Q7; VERDICT round 2, missing 1 / SURVEY 8f-2)
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["c1_golden", "sparse16_d13", "per_series_bank", "irregular_missing_d3", "lane_d2", "wave_d20_p10", "generic_dense_d9", "generic_d64"])
def test_likelihood_literal_q7(eng, golden_dir, case):
    """dlm_loglik_batch with DLM_OPT_LOGLIK_LITERAL_Q7 = KalmanFilter.likelihood as the reference writes it
    (KalmanFilter.scala:299-306, :175-183; what MetropolisHastings.dlm calls, MetropolisHastings.scala:134, :205):
    sum_t log N(m_t; g(dt_t) m_{t-1}, W dt_t) over the filtered means, on every forward kernel family, shared and per-series
    parameters, an irregular grid, missing data -- against the oracle's restatement (1e-9 relative) and, for config C1, against
    the filtered means the reference itself wrote (first_order_dlm_filtered.csv)."""
    import csv
    import os
    rng = np.random.default_rng({"c1_golden": 1, "sparse16_d13": 2, "per_series_bank": 3, "irregular_missing_d3": 4, "lane_d2": 5,
                                 "wave_d20_p10": 6, "generic_dense_d9": 7, "generic_d64": 8}[case])
    N = 3
    params = None
    if case == "c1_golden":
        rows = list(csv.reader(open(os.path.join(golden_dir, "first_order_dlm.csv"))))[1:]
        times = np.array([float(r[0]) for r in rows]); y = np.array([float(r[1]) for r in rows]).reshape(1, -1, 1)
        mat = materialise(Dlm.polynomial(1), times)
        p = DlmParameters([[2.0]], [[3.0]], [0.0], [[10.0]])
        N = 1
    elif case in ("sparse16_d13", "per_series_bank"):
        mat, p = c2(120)
        A = rng.standard_normal((13, 13))
        p = DlmParameters(p.v, A @ A.T / 13 + np.diag(W_C2), p.m0, p.c0)     # a dense W
        if case == "per_series_bank":
            params = [DlmParameters(p.v * s, p.w * s, p.m0, p.c0) for s in (0.5, 1.0, 2.0)]
    elif case == "irregular_missing_d3":
        mat = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 1), np.cumsum(np.array([1, 2, 1, 0.5, 3] * 12, dtype=np.float64)))
        A = rng.standard_normal((3, 3))
        p = DlmParameters([[0.7]], A @ A.T + 0.3 * np.eye(3), rng.standard_normal(3), np.eye(3))
    elif case == "lane_d2":
        mat = materialise(Dlm.polynomial(2), np.arange(1, 201, dtype=np.float64))
        p = DlmParameters([[1.3]], np.array([[0.5, 0.1], [0.1, 0.2]]), np.zeros(2), np.eye(2))
    elif case == "wave_d20_p10":
        mod = Dlm.polynomial(2)
        for _ in range(9):
            mod = mod * Dlm.polynomial(2)
        mat = materialise(mod, np.arange(1, 61, dtype=np.float64))
        B = rng.standard_normal((10, 10)); A2 = rng.standard_normal((20, 20))
        p = DlmParameters(B @ B.T / 10 + 0.5 * np.eye(10), A2 @ A2.T / 20 + 0.1 * np.eye(20), rng.standard_normal(20), np.eye(20))
    elif case == "generic_d64":     # the largest state the engine takes: W^-1 needs 66 560 bytes of LDS (more than a kernel gets without asking)
        mod = Dlm.polynomial(4)
        for _ in range(15):
            mod = mod + Dlm.polynomial(4)         # d = 64, p = 1
        mat = materialise(mod, np.arange(1, 13, dtype=np.float64))
        A2 = rng.standard_normal((64, 64))
        p = DlmParameters([[0.8]], A2 @ A2.T / 64 + 0.2 * np.eye(64), np.zeros(64), np.eye(64))
        N = 2
    else:
        A = rng.standard_normal((9, 9)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((9, 2))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 51, dtype=np.float64))
        A2 = rng.standard_normal((9, 9))
        p = DlmParameters(np.eye(2) * 0.8, A2 @ A2.T / 9 + 0.2 * np.eye(9), np.zeros(9), np.eye(9))
    if case != "c1_golden":
        y = rng.standard_normal((N, mat.T, mat.p)).cumsum(axis=1)
        if case in ("irregular_missing_d3", "wave_d20_p10", "sparse16_d13"):
            y[rng.random(y.shape) < 0.15] = np.nan
            y[:, 7, :] = np.nan
    out = eng.loglik(mat, params if params is not None else p, y, flags=_lib.OPT_LOGLIK_LITERAL_Q7)
    assert np.all(np.asarray(out["status"]) == 0)
    om = omodel(mat)
    for n in range(N):
        q = params[n] if params is not None else p
        f = oracle.kf_filter(om, q.v, q.w, q.m0, q.c0, y[n])
        want = oracle.likelihood_q7(om, f, q.w)
        assert out["loglik"][n] == pytest.approx(want, rel=1e-9), (case, n)
        assert abs(want - oracle.loglik(om, f, y[n])) > 1e-3 * abs(want)       # not the prediction-error likelihood
    if case == "c1_golden":
        from scipy.stats import norm
        fr = list(csv.reader(open(os.path.join(golden_dir, "first_order_dlm_filtered.csv"))))[1:]
        m_ref = np.array([float(r[1]) for r in fr])
        assert out["loglik"][0] == pytest.approx(norm(m_ref[:-1], np.sqrt(3.0)).logpdf(m_ref[1:]).sum(), rel=1e-11)
    # device-resident call gives the same numbers; a W that is not positive definite is flagged, not silently used
    import torch
    dev = eng.loglik(mat, params if params is not None else p, torch.as_tensor(y, device="cuda:0"), flags=_lib.OPT_LOGLIK_LITERAL_Q7)
    np.testing.assert_allclose(dev["loglik"].cpu().numpy(), out["loglik"], rtol=1e-13)
    if case == "generic_d64":     # d = 64 with p = 32 does not fit one CU's LDS on the general filter kernel: refused with a message, not a raw HIP error
        wide = Dlm.polynomial(2)
        for _ in range(31):
            wide = wide * Dlm.polynomial(2)
        mw = materialise(wide, np.arange(1, 6, dtype=np.float64))
        with pytest.raises(Exception, match="160 KB"):
            eng.filter(mw, DlmParameters(np.eye(32), np.eye(64), np.zeros(64), np.eye(64)), np.zeros((1, 5, 32)))
    if case == "lane_d2":
        bad = DlmParameters(p.v, np.array([[0.5, 0.6], [0.6, 0.2]]), p.m0, p.c0)
        ob = eng.loglik(mat, bad, y, flags=_lib.OPT_LOGLIK_LITERAL_Q7)
        assert np.all(np.isnan(ob["loglik"])) and np.all(np.asarray(ob["status"]) & _lib.ST_NOT_PD)
        with pytest.raises(Exception):
            eng.loglik(mat, DlmParameters(p.v, np.stack([p.w] * mat.T), p.m0, p.c0), y, flags=_lib.OPT_LOGLIK_LITERAL_Q7)   # a W_t stream
