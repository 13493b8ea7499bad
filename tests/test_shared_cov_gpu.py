"""Shared covariance sequence ("next round" 4b).

With V, W and C0 shared by the batch and no missing observation the covariances C_t, K_t, Q_t, P_t, S_t do not depend on the
data.  The engine then runs the covariance recursions ONCE per call (one wave: the per-series kernels' own code on a series of
zeros) and every series only its mean recursions; a series with a missing observation is routed to the per-series kernels.
The arithmetic of every output element is that of the per-series kernels, operation for operation -- so the parity bar here
is EQUALITY OF BITS with the default kernels (the path is opt-in: DLM_OPT_SHARED_COV) (np.array_equal on whole outputs; -0.0 == 0.0), with and without the steady-state
shortcut, plus the usual tolerance against the oracle (the reference's per-series semantics: KalmanFilter.scala:64-107,
Smoothing.scala:31-64)."""
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


def c2(T, wscale=1.0):
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    return mat, DlmParameters([[1.0]], np.diag(W_C2 * wscale), np.zeros(13), np.eye(13))


def both(eng, mat, p, y, flags=0):
    sh = eng.filter_smooth(mat, p, y, flags=flags | _lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS)
    cnt = eng.last_counters()
    assert eng.last_variant == "sparse16"
    ps = eng.filter_smooth(mat, p, y, flags=flags | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (0, 0)          # the default: every series its own covariance recursion
    return sh, ps, cnt


@pytest.mark.parametrize("flags", [0, _lib.OPT_NO_STEADY])
@pytest.mark.parametrize("T,N", [(1000, 37), (1, 5), (2, 3), (40, 9), (63, 4), (64, 4), (65, 4), (129, 300)])
def test_bit_for_bit_with_the_per_series_kernels(eng, T, N, flags):
    mat, p = c2(T)
    rng = np.random.default_rng(T + N)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    sh, ps, cnt = both(eng, mat, p, y, flags)
    assert cnt[2] == N and cnt[3] == 0, cnt
    assert np.array_equal(sh["filt"], ps["filt"])
    assert np.array_equal(sh["smooth"], ps["smooth"])
    assert np.all(sh["status"] == 0) and np.all(ps["status"] == 0)
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[N // 2])
    s = oracle.smoother(omodel(mat), f)
    np.testing.assert_allclose(sh["filt"][N // 2][:, :13], f["m"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(sh["filt"][N // 2][:, 13:], f["C"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(sh["smooth"][N // 2][:, :13], s["s"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sh["smooth"][N // 2][:, 13:], s["S"], rtol=1e-8, atol=1e-9)


def test_series_with_missing_observations_are_routed_to_their_own_recursion(eng):
    """Gaps at the first step, the last step, in the middle, every step: those series run k_filter_sp16 / k_smoother_sp16 (the
    counters say how many), the others the mean-only kernels -- every series bit for bit what the per-series path gives."""
    T, N = 300, 64
    mat, p = c2(T)
    rng = np.random.default_rng(3)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    y[1, 0, 0] = np.nan
    y[5, T - 1, 0] = np.nan
    y[17, 100:110, 0] = np.nan
    y[40, :, 0] = np.nan
    y[63, 64, 0] = np.nan
    for flags in (0, _lib.OPT_NO_STEADY):
        sh, ps, cnt = both(eng, mat, p, y, flags)
        assert cnt[2] == N - 5 and cnt[3] == 5, cnt
        assert np.array_equal(sh["filt"], ps["filt"], equal_nan=True)
        assert np.array_equal(sh["smooth"], ps["smooth"], equal_nan=True)
    for n in (1, 17, 40, 2):
        f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
        s = oracle.smoother(omodel(mat), f)
        np.testing.assert_allclose(sh["filt"][n][:, 13:], f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sh["smooth"][n][:, :13], s["s"], rtol=1e-8, atol=1e-9)


def test_per_series_prior_means_share_the_covariances_other_parameters_do_not(eng):
    T, N = 200, 6
    mat, p = c2(T)
    rng = np.random.default_rng(4)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    # per-series m0 only: still eligible (the covariances do not see the mean)
    import torch
    from bayesian_dlms_amd.engine import pack_params
    V, vs, W, ws, m0, ms, C0, cs, vts, wts = pack_params(p, N)
    m0s = rng.standard_normal((N, 13))
    packed = (V, 0, W, 0, m0s.reshape(-1), 13, C0, 0, 0, 0)
    sh = eng.filter_smooth(mat, packed, y, flags=_lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2] == N
    ps = eng.filter_smooth(mat, packed, y)
    assert np.array_equal(sh["filt"], ps["filt"]) and np.array_equal(sh["smooth"], ps["smooth"])
    np.testing.assert_array_equal(sh["filt"][:, 0, :13], m0s)
    # per-series V: every series has its own covariances -- the per-series kernels, untouched
    plist = [DlmParameters(p.v * (1 + 0.1 * n), p.w, p.m0, p.c0) for n in range(N)]
    out = eng.filter_smooth(mat, plist, y, flags=_lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (0, 0)
    f = oracle.kf_filter(omodel(mat), plist[3].v, p.w, p.m0, p.c0, y[3])
    np.testing.assert_allclose(out["filt"][3][:, 13:], f["C"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("knz", [1, 3, 4])
def test_every_sparsity_instantiation(eng, knz):
    """1, 3 and 4 nonzeros per row / column of G (the C2 model has 2), d = 8, a gap in one series, both shortcut settings."""
    rng = np.random.default_rng(80 + knz)
    d, T = 8, 400
    Gm = np.zeros((d, d))
    coef = {1: [0.9], 3: [0.6, 0.25, -0.2], 4: [0.5, 0.3, -0.2, 0.15]}[knz]
    for i in range(d):
        for s_, cf in enumerate(coef):
            Gm[i, (i + s_) % d] = cf
    Fv = np.array([1.0, 0.0, 1.0, 0.5, 0.0, 1.0, 0.0, -0.5]).reshape(-1, 1)
    mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
    p = DlmParameters([[0.7]], np.diag(rng.uniform(0.1, 0.5, d)), rng.standard_normal(d), np.eye(d))
    y = rng.standard_normal((5, T, 1))
    y[1, 200:203, 0] = np.nan
    for flags in (0, _lib.OPT_NO_STEADY):
        sh, ps, cnt = both(eng, mat, p, y, flags)
        assert cnt[2] == 4 and cnt[3] == 1
        assert np.array_equal(sh["filt"], ps["filt"]) and np.array_equal(sh["smooth"], ps["smooth"])
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0])
    s = oracle.smoother(omodel(mat), f)
    np.testing.assert_allclose(sh["smooth"][0][:, d:], s["S"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sh["smooth"][0][:, :d], s["s"], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("d", [6, 7, 12, 14, 15])
def test_other_state_dimensions_even_and_odd(eng, d):
    """The record is written as 16-byte pieces with the mean patched into the first d doubles: d even and odd, up to 15."""
    rng = np.random.default_rng(d)
    T = 150
    Gm = 0.8 * np.eye(d) + 0.15 * np.eye(d, k=1)
    Fv = rng.standard_normal((d, 1))
    mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
    A = rng.standard_normal((d, d))
    p = DlmParameters([[0.9]], A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2)
    y = rng.standard_normal((7, T, 1)).cumsum(axis=1)
    sh, ps, cnt = both(eng, mat, p, y)
    assert cnt[2] == 7
    assert np.array_equal(sh["filt"], ps["filt"]) and np.array_equal(sh["smooth"], ps["smooth"])
    fo = eng.filter(mat, p, y, flags=_lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS)          # dlm_filter_batch alone takes the forward half
    assert eng.last_counters()[2] == 7
    assert np.array_equal(fo["filt"], ps["filt"])
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[2])
    np.testing.assert_allclose(sh["filt"][2][:, :d], f["m"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(sh["filt"][2][:, d:], f["C"], rtol=1e-9, atol=1e-10)


def test_device_resident_call_and_what_falls_back(eng):
    import torch
    T, N = 120, 50
    mat, p = c2(T)
    rng = np.random.default_rng(9)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    yd = torch.as_tensor(y, device="cuda:0")
    ref = eng.filter_smooth(mat, p, y)
    dev = eng.filter_smooth(mat, p, yd, flags=_lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2] == N
    assert np.array_equal(dev["smooth"].cpu().numpy(), ref["smooth"]) and np.array_equal(dev["filt"].cpu().numpy(), ref["filt"])
    for kw in (dict(flags=_lib.OPT_SHARED_COV | _lib.OPT_PACKED_SYM | _lib.OPT_COUNT_STEPS), dict(flags=_lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS, want_filt=False)):
        eng.filter_smooth(mat, p, y, **kw)                          # packed records / engine-internal filtered records: per-series kernels
        assert eng.last_counters()[2] == 0
    mi = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 6), np.cumsum(np.array([1.0, 2.0, 1.0] * 40)))
    out = eng.filter_smooth(mi, p, y, flags=_lib.OPT_SHARED_COV | _lib.OPT_COUNT_STEPS)   # an irregular grid: per-series kernels (for now)
    assert eng.last_counters()[2] == 0 and np.all(out["status"] == 0)


# ------------------------------------------------------------------------------------------------------------------------
# SVD filter: shared factors (on by default where the batch shares V, W, C0; DLM_OPT_SVD_PER_SERIES turns it off)
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,N", [(600, 9), (1, 3), (2, 5), (65, 4), (130, 70)])
def test_svd_filter_shared_factors_bit_for_bit(eng, T, N):
    """The decompositions of the SVD filter once per call (k_svd_filter on a series of zeros) and a mean-only kernel per series:
    the records [m | dc | uc] bit for bit those of k_svd_filter run per series (SvdFilter.scala:38-95, :183-202), series with a
    missing observation routed to it; the covariance U D^2 U^T against the oracle's Kalman filter (1e-7, the path's tolerance)."""
    mat, p = c2(T)
    rng = np.random.default_rng(100 + T)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    if N > 3:
        y[1, T // 2, 0] = np.nan
        y[3, 0, 0] = np.nan
    sh = eng.svd_filter(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    cnt = eng.last_counters()
    ps = eng.svd_filter(mat, p, y, flags=_lib.OPT_SVD_PER_SERIES | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (0, 0)
    ngap = 2 if N > 3 else 0
    assert cnt[2] == N - ngap and cnt[3] == ngap, cnt
    assert np.array_equal(sh["svd"], ps["svd"], equal_nan=True)
    assert np.array_equal(sh["status"], ps["status"]) and np.all(sh["status"] == 0)
    n = N - 1
    kf = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
    rec = sh["svd"][n]
    m, dc, U = rec[:, :13], rec[:, 13:26], rec[:, 26:].reshape(-1, 13, 13).transpose(0, 2, 1)
    C = np.einsum("tij,tj,tkj->tik", U, dc * dc, U)
    Co = kf["C"].reshape(T + 1, 13, 13).transpose(0, 2, 1)
    assert (np.abs(C - Co) / np.abs(Co).max(axis=(1, 2), keepdims=True)).max() <= 1e-7
    assert np.abs(m - kf["m"]).max() <= 1e-7 * max(1.0, np.abs(kf["m"]).max())


def test_svd_shared_factors_other_shapes_and_fallbacks(eng):
    rng = np.random.default_rng(7)
    for d in (2, 7, 16):
        T = 90
        Gm = 0.8 * np.eye(d) + 0.15 * np.eye(d, k=1)
        Fv = rng.standard_normal((d, 1))
        mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
        A = rng.standard_normal((d, d))
        p = DlmParameters([[0.9]], A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2)
        y = rng.standard_normal((6, T, 1)).cumsum(axis=1)
        sh = eng.svd_filter(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
        assert eng.last_counters()[2] == 6
        ps = eng.svd_filter(mat, p, y, flags=_lib.OPT_SVD_PER_SERIES)
        assert np.array_equal(sh["svd"], ps["svd"])
    # per-series parameters, or the literal raw-W quirk with shared ones: the first falls back, the second shares
    mat, p = c2(50)
    y = rng.standard_normal((4, 50, 1))
    eng.svd_filter(mat, [p] * 4, y, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2] == 0
    a = eng.svd_filter(mat, p, y, flags=_lib.OPT_SVD_RAW_W_Q2 | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2] == 4
    b = eng.svd_filter(mat, p, y, flags=_lib.OPT_SVD_RAW_W_Q2 | _lib.OPT_SVD_PER_SERIES)
    assert np.array_equal(a["svd"], b["svd"])
    # the FFBS entry point (its forward half is the per-series kernel: unchanged)
    z = rng.standard_normal((4, 51, 13))
    f1 = eng.svd_ffbs(mat, p, y, z=z)
    assert np.all(np.isfinite(f1["theta"]))
