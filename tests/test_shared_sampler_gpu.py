"""Shared factors of the reference-form backward sampler ("next round" 7).

Smoothing.sampleDlm / backSampleStep (Smoothing.scala:74-122): J_t = C_t G^T R_{t+1}^-1, H_t and its factor depend on the filtered
COVARIANCES alone.  With V, W, C0 shared by the batch on a regular grid (the pooled Gibbs samplers, GibbsSampling.sample,
Gibbs.scala:134-180) the engine computes them ONCE per call -- k_sampler_sp16 itself on the filter records of a series of zeros,
one wave -- and every series without a missing observation draws against that table with a mean-only kernel (four series per
wave): a+ = G m_t, h = m_t + J_t (theta_{t+1} - a+), theta_t = h + L_t z_t and the Gibbs sums, in the per-series kernel's
operations one for one.  The bar is therefore EQUALITY OF BITS with the per-series kernel (DLM_OPT_SAMPLER_PER_SERIES) on whole
outputs -- draws, statistics, status -- plus the usual tolerance against the oracle with injected normals."""
import numpy as np
import pytest

import oracle
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise

pytestmark = pytest.mark.gpu

W_C2 = np.array([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
SMALL = 0


@pytest.fixture(scope="module")
def eng():
    from bayesian_dlms_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def omodel(mat):
    return oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)


def c2(T):
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    return mat, DlmParameters([[1.0]], np.diag(W_C2), np.zeros(13), np.eye(13))


def both(eng, mat, p, y, flags=0, **kw):
    sh = eng.ffbs(mat, p, y, flags=flags | SMALL | _lib.OPT_COUNT_STEPS, **kw)
    cnt = eng.last_counters()
    assert eng.last_variant == "sparse16-sampler-shared"
    ps = eng.ffbs(mat, p, y, flags=flags | SMALL | _lib.OPT_SAMPLER_PER_SERIES | _lib.OPT_COUNT_STEPS, **kw)
    assert eng.last_variant == "sparse16-sampler"
    assert eng.last_counters()[2:] == (0, 0)
    return sh, ps, cnt


def _where(a, b):
    """Where two outputs differ, for the assertion message: count, leading indices, the values, the largest difference."""
    a, b = np.asarray(a), np.asarray(b)
    ne = np.argwhere(~((a == b) | ((a != a) & (b != b))))
    if len(ne) == 0:
        return "equal"
    lead = [np.unique(ne[:, ax]).tolist()[:12] for ax in range(ne.shape[1])]
    vals = [(a[tuple(i)].item(), b[tuple(i)].item()) for i in ne[:4]]
    return f"{len(ne)} of {a.size} values differ; indices per axis {lead}; first {ne[:4].tolist()} = {vals}; max |diff| {np.nanmax(np.abs(a.astype(float) - b.astype(float))):.3e}"


def same(sh, ps):
    for k in ("theta", "stats", "filt", "status"):
        if sh[k] is not None:
            assert np.array_equal(sh[k], ps[k], equal_nan=True), (k, _where(sh[k], ps[k]))


@pytest.mark.parametrize("T,N", [(1000, 37), (1, 5), (2, 3), (3, 4), (40, 9), (63, 4), (64, 4), (65, 4), (129, 300)])
def test_draw_for_draw_the_per_series_kernel(eng, T, N):
    mat, p = c2(T)
    rng = np.random.default_rng(T + N)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    sh, ps, cnt = both(eng, mat, p, y, seed=11 + T, series_offset=5)
    assert cnt[2] == N and cnt[3] == 0, cnt
    same(sh, ps)
    assert np.all(sh["status"] == 0)
    assert np.isfinite(sh["theta"]).all() and np.isfinite(sh["stats"]).all()


def test_injected_normals_against_the_oracle(eng):
    """Smoothing.sampleDlm with the engine's canonical factor: the draws from injected normals against the oracle's sampler."""
    T, N = 200, 6
    mat, p = c2(T)
    rng = np.random.default_rng(1)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    z = rng.standard_normal((N, T + 1, 13))
    sh, ps, cnt = both(eng, mat, p, y, z=z)
    assert cnt[2] == N
    same(sh, ps)
    n = 4
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
    o = oracle.backward_sample(omodel(mat), p.w, f, z[n], factor="chol")
    np.testing.assert_allclose(sh["theta"][n], o["theta"], rtol=1e-7, atol=1e-8)


def test_series_with_missing_observations_compute_their_own_factors(eng):
    T, N = 300, 64
    mat, p = c2(T)
    rng = np.random.default_rng(3)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    y[1, 0, 0] = np.nan
    y[5, T - 1, 0] = np.nan
    y[17, 100:110, 0] = np.nan
    y[40, :, 0] = np.nan
    y[63, 64, 0] = np.nan
    sh, ps, cnt = both(eng, mat, p, y, seed=4)
    assert cnt[2] == N - 5 and cnt[3] == 5, cnt
    same(sh, ps)
    assert np.all(sh["status"] == 0)


@pytest.mark.parametrize("knz", [1, 3, 4])
def test_every_sparsity_instantiation(eng, knz):
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
    y = rng.standard_normal((9, T, 1))
    y[1, 200:203, 0] = np.nan
    sh, ps, cnt = both(eng, mat, p, y, seed=knz)
    assert cnt[2] == 8 and cnt[3] == 1
    same(sh, ps)


@pytest.mark.parametrize("d", [2, 3, 4, 6, 7, 11, 12, 14, 15])
def test_other_state_dimensions(eng, d):
    """Table rows of 17 d sixteen-byte pieces: one to four DMA instructions per row."""
    rng = np.random.default_rng(d)
    T = 150
    Gm = 0.8 * np.eye(d) + 0.15 * np.eye(d, k=1)
    Fv = rng.standard_normal((d, 1))
    mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
    A = rng.standard_normal((d, d))
    p = DlmParameters([[0.9]], A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2)
    y = rng.standard_normal((7, T, 1)).cumsum(axis=1)
    sh, ps, cnt = both(eng, mat, p, y, flags=_lib.OPT_NO_LANE, seed=d)
    assert cnt[2] == 7
    same(sh, ps)


def test_without_statistics_without_draws_and_what_falls_back(eng):
    T, N = 120, 50
    mat, p = c2(T)
    rng = np.random.default_rng(9)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    sh, ps, _ = both(eng, mat, p, y, want_stats=False)
    same(sh, ps)
    sh, ps, _ = both(eng, mat, p, y, want_theta=False)
    same(sh, ps)
    eng.ffbs(mat, p, y, flags=_lib.OPT_COUNT_STEPS)                      # the default at every batch size (tools/sweep_c3.sh)
    assert eng.last_variant == "sparse16-sampler-shared" and eng.last_counters()[2] == N
    eng.ffbs(mat, p, y, flags=SMALL | _lib.OPT_COUNT_STEPS, want_cond=True)      # conditional-moment records: H_t is not in the table
    assert eng.last_variant == "sparse16-sampler"
    plist = [DlmParameters(p.v * (1 + 0.1 * n), p.w, p.m0, p.c0) for n in range(N)]
    eng.ffbs(mat, plist, y, flags=SMALL | _lib.OPT_COUNT_STEPS)             # per-series V: per-series covariances
    assert eng.last_variant == "sparse16-sampler"
    mi = materialise(Dlm.polynomial(1) + Dlm.seasonal(24, 6), np.cumsum(np.array([1.0, 2.0, 1.0] * 40)))
    out = eng.ffbs(mi, p, y, flags=SMALL)                                   # an irregular grid
    assert eng.last_variant == "sparse16-sampler" and np.all(out["status"] == 0)


def test_device_resident_large_batch_is_shared_by_default(eng):
    import torch
    T, N = 60, 2600
    mat, p = c2(T)
    rng = np.random.default_rng(10)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    y[77, 30, 0] = np.nan
    yd = torch.as_tensor(y, device="cuda:0")
    dev = eng.ffbs(mat, p, yd, seed=3, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "sparse16-sampler-shared" and eng.last_counters()[2:] == (N - 1, 1)
    ref = eng.ffbs(mat, p, y, seed=3, flags=_lib.OPT_SAMPLER_PER_SERIES)
    assert np.array_equal(dev["theta"].cpu().numpy(), ref["theta"], equal_nan=True)
    assert np.array_equal(dev["stats"].cpu().numpy(), ref["stats"], equal_nan=True)


# ------------------------------------------------------------------------------------------------------------------------
# 16 <= d <= 48 (dlm_wave48.hip): the multivariate models of the Inverse-Wishart Gibbs sampler (BASELINE configs[3])
# ------------------------------------------------------------------------------------------------------------------------
def blocks(nblk, T, seed, per=2):
    mod = Dlm.polynomial(per)
    for _ in range(nblk - 1):
        mod = mod * Dlm.polynomial(per)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A = np.random.default_rng(seed).standard_normal((d, d))
    return mat, DlmParameters(np.eye(q) * 1.1, A @ A.T / d + 0.1 * np.eye(d), np.zeros(d), np.eye(d))


def both_big(eng, mat, p, y, flags=0, **kw):
    sh = eng.ffbs(mat, p, y, flags=flags | _lib.OPT_COUNT_STEPS, **kw)
    cnt = eng.last_counters()
    assert eng.last_variant == "wave-sampler-shared"
    ps = eng.ffbs(mat, p, y, flags=flags | _lib.OPT_SAMPLER_PER_SERIES | _lib.OPT_COUNT_STEPS, **kw)
    assert eng.last_variant == "wave-sampler"
    assert eng.last_counters()[2:] == (0, 0)
    return sh, ps, cnt


@pytest.mark.parametrize("nblk,T,N,flags", [(20, 130, 6, _lib.OPT_STATS_OUTER), (20, 70, 5, 0), (10, 65, 9, _lib.OPT_STATS_OUTER), (16, 33, 4, _lib.OPT_STATS_OUTER),
                                            (8, 1, 3, 0), (8, 2, 3, _lib.OPT_STATS_OUTER), (24, 40, 3, _lib.OPT_STATS_OUTER)])
def test_multivariate_draw_for_draw_the_per_series_kernel(eng, nblk, T, N, flags):
    """d = 2 nblk, p = nblk (d = 40, p = 20 is C4): two and three tiles per dimension, diagonal and outer-product statistics, one
    series with a missing component (its own factors, k_sampler_w48)."""
    mat, p = blocks(nblk, T, seed=nblk)
    rng = np.random.default_rng(nblk + T)
    y = rng.standard_normal((N, T, mat.p)).cumsum(axis=1) * 0.5 + rng.standard_normal((N, T, mat.p))
    if N > 3:
        y[2, T // 2, 3] = np.nan
    sh, ps, cnt = both_big(eng, mat, p, y, flags=flags, seed=5, series_offset=3)
    ngap = 1 if N > 3 else 0
    assert cnt[2] == N - ngap and cnt[3] == ngap, cnt
    if not np.array_equal(sh["filt"], ps["filt"], equal_nan=True):      # which of the two calls left the plain filter's records?
        ref = eng.filter(mat, p, y)["filt"]
        raise AssertionError(("filt", "shared call vs dlm_filter_batch: " + _where(sh["filt"], ref), "per-series call vs dlm_filter_batch: " + _where(ps["filt"], ref)))
    same(sh, ps)
    assert np.all(sh["status"] == 0) and np.isfinite(sh["stats"]).all()


def test_multivariate_injected_normals_against_the_oracle(eng):
    T, N = 60, 3
    mat, p = blocks(10, T, seed=2)
    rng = np.random.default_rng(2)
    y = rng.standard_normal((N, T, mat.p)).cumsum(axis=1)
    z = rng.standard_normal((N, T + 1, mat.d))
    sh, ps, cnt = both_big(eng, mat, p, y, z=z, flags=_lib.OPT_STATS_OUTER)
    assert cnt[2] == N
    same(sh, ps)
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[1])
    o = oracle.backward_sample(omodel(mat), p.w, f, z[1], factor="chol")
    np.testing.assert_allclose(sh["theta"][1], o["theta"], rtol=1e-7, atol=1e-8)
    st = oracle.gibbs_stats(omodel(mat), y[1], sh["theta"][1], want_outer=True)
    q = mat.p
    np.testing.assert_allclose(sh["stats"][1][:q], st["ssy"], rtol=1e-9)
    np.testing.assert_allclose(sh["stats"][1][2 * q:-1], st["outer"], rtol=1e-8, atol=1e-10)


# ------------------------------------------------------------------------------------------------------------------------
# dlm_ffbs_batch that does not want the filter's records (filt_ws = NULL): with shared factors at d <= 15 the forward pass is the
# mean-only kernel against one covariance table (no record is produced at all); otherwise an engine workspace takes them
# ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,N", [(1000, 37), (1, 5), (2, 3), (65, 4), (129, 300)])
def test_without_filter_records_the_same_draws(eng, T, N):
    mat, p = c2(T)
    rng = np.random.default_rng(200 + T + N)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    if N > 4:
        y[1, T // 2, 0] = np.nan
        y[4, 0, 0] = np.nan
    ngap = 2 if N > 4 else 0
    ref = eng.ffbs(mat, p, y, seed=21, series_offset=9, flags=_lib.OPT_SAMPLER_PER_SERIES)
    out = eng.ffbs(mat, p, y, seed=21, series_offset=9, want_filt=False, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "sparse16-sampler-shared" and out["filt"] is None
    cnt = eng.last_counters()
    assert cnt[2] >= N - ngap and cnt[3] >= ngap, cnt        # (the mean-only forward kernel counts its series too)
    for k in ("theta", "stats", "status"):
        assert np.array_equal(out[k], ref[k], equal_nan=True), k


def test_without_filter_records_other_shapes_and_fallbacks(eng):
    rng = np.random.default_rng(77)
    for d in (6, 11, 15):
        T = 140
        Gm = 0.8 * np.eye(d) + 0.15 * np.eye(d, k=1)
        Fv = rng.standard_normal((d, 1))
        mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
        A = rng.standard_normal((d, d))
        p = DlmParameters([[0.9]], A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2)
        y = rng.standard_normal((9, T, 1)).cumsum(axis=1)
        y[3, 70:73, 0] = np.nan
        ref = eng.ffbs(mat, p, y, seed=d, flags=_lib.OPT_SAMPLER_PER_SERIES)
        out = eng.ffbs(mat, p, y, seed=d, want_filt=False)
        assert eng.last_variant == "sparse16-sampler-shared"
        for k in ("theta", "stats", "status"):
            assert np.array_equal(out[k], ref[k], equal_nan=True), (d, k)
    # per-series parameters, the multivariate kernels, the simulation smoother, conditional records: an engine workspace takes the records
    mat, p = c2(60)
    y = rng.standard_normal((6, 60, 1)).cumsum(axis=1)
    plist = [DlmParameters(p.v * (1 + 0.1 * n), p.w, p.m0, p.c0) for n in range(6)]
    for kw in (dict(params=plist), dict(params=p, flags=_lib.OPT_FFBS_SIMSMOOTH), dict(params=p, want_cond=True)):
        prm = kw.pop("params")
        ref = eng.ffbs(mat, prm, y, seed=3, **kw)
        out = eng.ffbs(mat, prm, y, seed=3, want_filt=False, **kw)
        for k in ("theta", "stats", "cond"):
            if ref[k] is not None:
                assert np.array_equal(out[k], ref[k], equal_nan=True), (kw, k)
    mb, pb = blocks(10, 50, seed=4)
    yb = rng.standard_normal((5, 50, mb.p)).cumsum(axis=1)
    ref = eng.ffbs(mb, pb, yb, seed=8, flags=_lib.OPT_STATS_OUTER)
    out = eng.ffbs(mb, pb, yb, seed=8, flags=_lib.OPT_STATS_OUTER, want_filt=False)
    assert eng.last_variant == "wave-sampler-shared"
    assert np.array_equal(out["theta"], ref["theta"]) and np.array_equal(out["stats"], ref["stats"])


def test_without_filter_records_injected_normals_and_per_series_prior_means(eng):
    T, N = 90, 10
    mat, p = c2(T)
    rng = np.random.default_rng(5)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    y[7, 40, 0] = np.nan
    z = rng.standard_normal((N, T + 1, 13))
    ref = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_SAMPLER_PER_SERIES)
    out = eng.ffbs(mat, p, y, z=z, want_filt=False)
    assert eng.last_variant == "sparse16-sampler-shared"
    for k in ("theta", "stats", "status"):
        assert np.array_equal(out[k], ref[k], equal_nan=True), k
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[2])
    o = oracle.backward_sample(omodel(mat), p.w, f, z[2], factor="chol")
    np.testing.assert_allclose(out["theta"][2], o["theta"], rtol=1e-7, atol=1e-8)
    from bayesian_dlms_amd.engine import pack_params
    V, vs, W, ws, m0, ms, C0, cs, vts, wts = pack_params(p, N)
    m0s = rng.standard_normal((N, 13))
    packed = (V, 0, W, 0, m0s.reshape(-1), 13, C0, 0, 0, 0)     # per-series prior means share the covariances
    ref = eng.ffbs(mat, packed, y, seed=2, flags=_lib.OPT_SAMPLER_PER_SERIES)
    out = eng.ffbs(mat, packed, y, seed=2, want_filt=False)
    assert eng.last_variant == "sparse16-sampler-shared"
    for k in ("theta", "stats", "status"):
        assert np.array_equal(out[k], ref[k], equal_nan=True), k


@pytest.mark.parametrize("nblk,T,N", [(20, 130, 6), (10, 400, 5), (16, 33, 4)])
def test_multivariate_without_filter_records(eng, nblk, T, N):
    """16 <= d <= 48 with filt_ws = NULL: the series without a gap store the mean alone once their covariance recursion has settled
    (KArgs::keep_cov), the series with a gap whole records for their own sampler -- draws and statistics unchanged, bit for bit."""
    mat, p = blocks(nblk, T, seed=nblk + 1)
    rng = np.random.default_rng(nblk * T)
    y = rng.standard_normal((N, T, mat.p)).cumsum(axis=1) * 0.5 + rng.standard_normal((N, T, mat.p))
    y[2, T // 2, 1] = np.nan
    y[3, T - 1, 0] = np.nan
    ref = eng.ffbs(mat, p, y, seed=6, flags=_lib.OPT_STATS_OUTER | _lib.OPT_SAMPLER_PER_SERIES)
    out = eng.ffbs(mat, p, y, seed=6, flags=_lib.OPT_STATS_OUTER | _lib.OPT_COUNT_STEPS, want_filt=False)
    assert eng.last_variant == "wave-sampler-shared" and eng.last_counters()[2:] == (N - 2, 2)
    for k in ("theta", "stats", "status"):
        assert np.array_equal(out[k], ref[k], equal_nan=True), k


def test_alternating_calls_share_the_engine_workspaces(eng):
    """The tables, the normals, the gap marks and the side streams are engine state reused from call to call: calls of different
    shapes and kinds follow each other on one engine (growing and shrinking workspaces, d <= 15 and d >= 16 tables in the same
    buffer, records kept and not kept) and each equals its own per-series reference."""
    rng = np.random.default_rng(123)
    cases = []
    for T, N in ((129, 300), (40, 9), (700, 21)):
        mat, p = c2(T)
        y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
        y[1, T // 3, 0] = np.nan
        cases.append((mat, p, y, 0))
    for nblk, T, N in ((10, 90, 5), (20, 60, 4)):
        mat, p = blocks(nblk, T, seed=nblk)
        y = rng.standard_normal((N, T, mat.p)).cumsum(axis=1)
        cases.append((mat, p, y, _lib.OPT_STATS_OUTER))
    refs = [eng.ffbs(m, p, y, seed=i, flags=f | _lib.OPT_SAMPLER_PER_SERIES) for i, (m, p, y, f) in enumerate(cases)]
    order = [0, 3, 1, 4, 2, 0, 4, 3, 2, 1]
    for rnd in range(2):
        for i in order:
            m, p, y, f = cases[i]
            out = eng.ffbs(m, p, y, seed=i, flags=f, want_filt=bool((i + rnd) & 1))
            assert eng.last_variant.endswith("-shared")
            for k in ("theta", "stats", "status"):
                assert np.array_equal(out[k], refs[i][k], equal_nan=True), (rnd, i, k)
            if i == 0:       # a filter + smoother call in between (its own workspaces, the same streams)
                fs = eng.filter_smooth(m, p, y)
                assert np.all(fs["status"] == 0)


# ------------------------------------------------------------------------------------------------------------------------
# The engine's auxiliary streams (tables on cov_stream, normals on rng_stream): rules A and B of dlm_engine.hip (aux_join / drain_all)
# ------------------------------------------------------------------------------------------------------------------------
def test_an_error_exit_with_the_auxiliary_streams_busy_leaves_the_engine_usable(eng):
    """DLM_OPT_TEST_FAIL_AFTER_TABLES makes dlm_ffbs_batch return an error right after the shared-factor tables and the normals were
    started on the second and third stream (round 3 left such an exit with kernels in flight that nothing waited for: they read the
    staging arena and the table workspace of the NEXT call).  Every shape of shared-factor call: d <= 15 with and without records,
    16 <= d <= 48 with and without records; host-memory calls, so that the arena is re-used at once; growing workspaces after the
    failed call, so that the re-size path frees what the failed call's kernels use."""
    from bayesian_dlms_amd.engine import EngineError
    rng = np.random.default_rng(77)
    cases = []
    for T, N, want_filt in ((200, 64, True), (200, 64, False)):
        mat, p = c2(T)
        cases.append((mat, p, rng.standard_normal((N, T, 1)).cumsum(axis=1), 0, want_filt))
    for T, N, want_filt in ((70, 6, True), (70, 6, False)):
        mat, p = blocks(20, T, seed=20)
        cases.append((mat, p, rng.standard_normal((N, T, mat.p)).cumsum(axis=1), _lib.OPT_STATS_OUTER, want_filt))
    for i, (mat, p, y, f, wf) in enumerate(cases):
        ref = eng.ffbs(mat, p, y, seed=40 + i, flags=f | _lib.OPT_SAMPLER_PER_SERIES, want_filt=wf)
        with pytest.raises(EngineError, match="DLM_OPT_TEST_FAIL_AFTER_TABLES"):
            eng.ffbs(mat, p, y, seed=40 + i, flags=f | _lib.OPT_TEST_FAIL_AFTER_TABLES, want_filt=wf)
        # a larger call straight after (every workspace grows: the re-size path must wait for the failed call's kernels) ...
        yb = np.concatenate([y, y[:, ::-1]], axis=0)
        big = eng.ffbs(mat, p, yb, seed=40 + i, flags=f, want_filt=wf)
        assert eng.last_variant.endswith("-shared") and np.all(big["status"] == 0)
        # ... and the call itself again: the per-series reference, bit for bit
        out = eng.ffbs(mat, p, y, seed=40 + i, flags=f, want_filt=wf)
        for k in ("theta", "stats", "status"):
            assert np.array_equal(out[k], ref[k], equal_nan=True), (i, k)
        assert np.array_equal(big["theta"][: y.shape[0]], ref["theta"]), i    # (draws are keyed by the series index: the first half is the same batch)


def test_an_engine_destroyed_right_after_an_asynchronous_shared_factor_call():
    """dlm_engine_destroy drains all three streams before it frees the workspaces and destroys the streams (rule B): a
    DLM_OPT_ASYNC call that is still running when the engine goes away must neither fault nor hang."""
    import torch
    from bayesian_dlms_amd.engine import Engine
    rng = np.random.default_rng(5)
    for mk, N, T, f in ((lambda T: c2(T), 4000, 300, 0), (lambda T: blocks(20, T, seed=20), 600, 120, _lib.OPT_STATS_OUTER)):
        mat, p = mk(T)
        yd = torch.as_tensor(rng.standard_normal((N, T, mat.p)).cumsum(axis=1), device="cuda:0")
        for want_filt in (True, False):
            e2 = Engine(0)
            out = e2.ffbs(mat, p, yd, seed=9, flags=f | _lib.OPT_ASYNC, want_filt=want_filt)
            assert e2.last_variant.endswith("-shared")
            e2.close()                                    # kernels of the call are in flight on up to three streams
            torch.cuda.synchronize()
            theta = out["theta"].cpu().numpy()
            assert np.isfinite(theta).all() and int(out["status"].abs().sum().item()) == 0
    e3 = Engine(0)                                        # the device is fine afterwards
    mat, p = c2(50)
    y = rng.standard_normal((8, 50, 1))
    a = e3.ffbs(mat, p, y, seed=1)
    b = e3.ffbs(mat, p, y, seed=1, flags=_lib.OPT_SAMPLER_PER_SERIES)
    assert np.array_equal(a["theta"], b["theta"])
    e3.close()
