"""Parity at BASELINE.json's full sizes through size-independent properties (the oracle would need hours for a whole
batch): (i) the batch is a set of independent series -- any series run on its own gives bit-identical records;
(ii) the backward pass starts from the last filtered state; (iii) smoothing never increases a variance and every
covariance stays symmetric with a positive diagonal; (iv) a few series are compared with the oracle at full length."""
import os

import numpy as np
import pytest

import oracle
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from bayesian_dlms_amd.engine import Engine
    return Engine(0)


@pytest.fixture(autouse=True)
def _release_cached_device_memory():
    """The outputs of a full-size call are 29 GB apiece: hand torch's cached blocks back after every test, or the engine's own hipMalloc
    (workspaces of the next configuration) finds the device full."""
    yield
    import torch
    torch.cuda.empty_cache()


def _om(mat):
    return oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)


def _properties(eng, mat, p, y, pick, expect_variant, tol_f, tol_s, sub_flags=_lib.OPT_FORCE_WAVE):
    import torch
    d = mat.d
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == expect_variant
    assert int((out["status"] != 0).sum().item()) == 0
    filt, sm = out["filt"], out["smooth"]
    N, T1, rec = filt.shape
    # (ii) s_T = m_T, S_T = C_T
    assert torch.equal(sm[:, -1], filt[:, -1])
    # (iii) variances: positive, smoothing does not increase them; symmetry of S (on a slab of series, all records)
    diag = torch.arange(d, device=filt.device) * (d + 1) + d
    cd, sd = filt[:, :, diag], sm[:, :, diag]
    assert bool((cd > 0).all()) and bool((sd > 0).all())
    assert bool((sd <= cd * (1 + 1e-9) + 1e-12).all())
    S = sm[:64, :, d:].reshape(64, T1, d, d)
    assert float((S - S.transpose(-1, -2)).abs().max()) <= 1e-9 * float(S.abs().max())
    # (i) independence: the picked series on their own, bit for bit
    # the same kernels as the full batch: a handful of series alone would take the workgroup-per-series kernels at d >= 16, and at d <= 15
    # their own covariance recursions instead of the call's shared tables (DLM_OPT_NO_SMALL_BATCH: the large batch's route)
    sub = eng.filter_smooth(mat, p, y[pick], flags=sub_flags)
    assert torch.equal(sub["filt"], filt[pick]) and torch.equal(sub["smooth"], sm[pick])
    # (iv) oracle at full length
    for n in pick[:2]:
        f = oracle.kf_filter(_om(mat), p.v, p.w, p.m0, p.c0, y[n].cpu().numpy())
        s = oracle.smoother(_om(mat), f)
        fr, sr = filt[n].cpu().numpy(), sm[n].cpu().numpy()
        np.testing.assert_allclose(fr[:, :d], f["m"], rtol=tol_f, atol=tol_f)
        np.testing.assert_allclose(fr[:, d:], f["C"], rtol=tol_f, atol=tol_f)
        np.testing.assert_allclose(sr[:, :d], s["s"], rtol=tol_s, atol=tol_s)
        np.testing.assert_allclose(sr[:, d:], s["S"], rtol=tol_s, atol=tol_s)


def test_c2_full_size_properties(eng):
    """BASELINE configs[1]: seasonal d = 13, 10 000 series x T = 1000 (the bench workload, with 3 % missing values)."""
    import torch
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    yh = simulate(mat, p, 10000, seed=20261004)
    yh[np.random.default_rng(1).random(yh.shape) < 0.03] = np.nan
    y = torch.as_tensor(yh, device="cuda")
    # (every series has a gap: the call decides on the device to make no tables and runs the per-series kernels)
    _properties(eng, mat, p, y, [0, 4999, 9999, 1234, 7777], "sparse16-rts-shared", 1e-8, 1e-7, _lib.OPT_NO_SMALL_BATCH)
    eng.filter_smooth(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (0, 10000)


def test_c2_full_size_steady_state_path(eng):
    """The bench workload itself (no missing values): 64 % of the forward and 19 % of the backward steps take the steady-state
    path and the backward pass fetches most records as their mean alone.  The same properties, the oracle at full length, and the
    whole batch against the run with DLM_OPT_NO_STEADY (every step the full recursion)."""
    import torch
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    y = torch.as_tensor(simulate(mat, p, 10000, seed=20261005), device="cuda")
    _properties(eng, mat, p, y, [0, 4999, 9999, 4321], "sparse16-rts-shared", 1e-8, 1e-7, _lib.OPT_NO_SMALL_BATCH)   # J_t, S_t from the call's tables
    own = eng.filter_smooth(mat, p, y, flags=_lib.OPT_SMOOTHER_PER_SERIES | _lib.OPT_COUNT_STEPS)                   # every series its own (information form)
    assert eng.last_variant == "sparse16" and eng.last_counters()[2:] == (0, 0)
    a = eng.filter_smooth(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (10000, 0)
    assert torch.equal(a["filt"], own["filt"])
    assert float((a["smooth"] - own["smooth"]).abs().max()) <= 1e-10 * float(own["smooth"].abs().max())
    del own
    torch.cuda.empty_cache()
    C = a["filt"][:, :, 13:]
    assert torch.equal(C[:, 900], C[:, 600])                       # frozen covariances on the settled stretch
    b = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    for k in ("filt", "smooth"):
        diff = float((a[k] - b[k]).abs().max())
        scale = float(b[k].abs().max())
        assert diff <= 1e-10 * scale, (k, diff, scale)


def test_c2_full_size_literal_q1_on_shared_factors(eng):
    """The bench workload with Smoothing.scala:44 as written (DLM_OPT_SMOOTHER_COMPAT_Q1) at full size: the tables serve all 10 000 series; a handful of
    series alone through the same route give the same bits (independence); the per-series kernel gives the same bits (DLM_OPT_SMOOTHER_PER_SERIES);
    two series against the oracle's literal smoother at full length."""
    import torch
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    y = torch.as_tensor(simulate(mat, p, 10000, seed=20261006), device="cuda")
    Q1 = _lib.OPT_SMOOTHER_COMPAT_Q1
    a = eng.filter_smooth(mat, p, y, flags=Q1 | _lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "sparse16-rts-shared" and eng.last_counters()[2:] == (10000, 0)
    assert int((a["status"] != 0).sum().item()) == 0
    pick = [0, 4999, 9999, 777]
    sub = eng.filter_smooth(mat, p, y[pick], flags=Q1 | _lib.OPT_NO_SMALL_BATCH)
    assert torch.equal(sub["filt"], a["filt"][pick]) and torch.equal(sub["smooth"], a["smooth"][pick])
    own = eng.filter_smooth(mat, p, y[:2048], flags=Q1 | _lib.OPT_SMOOTHER_PER_SERIES)
    assert torch.equal(own["smooth"], a["smooth"][:2048])
    del own
    for n in pick[:2]:
        f = oracle.kf_filter(_om(mat), p.v, p.w, p.m0, p.c0, y[n].cpu().numpy())
        s_ = oracle.smoother(_om(mat), f, compat_q1=True)
        sr = a["smooth"][n].cpu().numpy()
        np.testing.assert_allclose(sr[:, :13], s_["s"], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(sr[:, 13:], s_["S"], rtol=1e-7, atol=1e-7)


def test_c4_full_size_steady_state_path(eng):
    """BASELINE configs[3] without missing components: the covariance recursion settles within 30 steps and both per-wave passes
    stream records from then on.  Properties, oracle at full length, and the batch against DLM_OPT_NO_STEADY."""
    import torch
    mod = Dlm.polynomial(2)
    for _ in range(19):
        mod = mod * Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    rng = np.random.default_rng(41); A = np.random.default_rng(40).standard_normal((40, 40))
    p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
    y = torch.as_tensor(rng.standard_normal((2000, 1000, 20)).cumsum(axis=1), device="cuda")
    _properties(eng, mat, p, y, [0, 1999, 1024], "wave-mfma", 1e-7, 1e-6)
    a = eng.filter_smooth(mat, p, y)
    b = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    for k in ("filt", "smooth"):
        diff = float((a[k] - b[k]).abs().max())
        scale = float(b[k].abs().max())
        assert diff <= 1e-9 * scale, (k, diff, scale)


def test_c4_full_size_properties(eng):
    """BASELINE configs[3]: |*| of 20 polynomial(2), d = 40, p = 20, 2000 series x T = 1000 (SURVEY 8d inputs)."""
    import torch
    mod = Dlm.polynomial(2)
    for _ in range(19):
        mod = mod * Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    rng = np.random.default_rng(40); A = rng.standard_normal((40, 40))
    p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
    yh = rng.standard_normal((2000, 1000, 20)).cumsum(axis=1)
    yh[rng.random(yh.shape) < 0.02] = np.nan
    y = torch.as_tensor(yh, device="cuda")
    _properties(eng, mat, p, y, [0, 1999, 777], "wave-mfma", 1e-7, 1e-6)


def test_c5_full_size_svd_agrees_with_standard_filter(eng):
    """BASELINE configs[4]: SVD filter, d = 13, 10 000 series: U D^2 U^T and the means equal the standard filter's
    (the reference's own check, core/src/test/scala/SvdFilter.scala:102-158), on the whole batch."""
    import torch
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    T = 200
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    y = torch.as_tensor(simulate(mat, p, 10000, seed=5), device="cuda")
    kf = eng.filter(mat, p, y)["filt"]
    sv = eng.svd_filter(mat, p, y)
    assert int((sv["status"] != 0).sum().item()) == 0
    rec = sv["svd"]
    d = 13
    m, dc, U = rec[..., :d], rec[..., d:2 * d], rec[..., 2 * d:].reshape(10000, T + 1, d, d).transpose(-1, -2)   # column-major
    C = (U * (dc * dc).unsqueeze(-2)) @ U.transpose(-1, -2)
    Ck = kf[..., d:].reshape(10000, T + 1, d, d)
    assert float((m - kf[..., :d]).abs().max()) < 1e-7
    assert float((C - Ck).abs().max()) < 1e-7


def test_c5_full_size_T1000_shared_factors_and_own_decompositions(eng):
    """BASELINE configs[4] at its full length, 10 000 x T = 1000 -- past the ~390-step transient of the shared SVD factors, so that the
    reuse branch of the table kernel and the mean kernel's long steady stretch are what is tested: the
    shared-factor records equal the records of series that run their own decompositions (DLM_OPT_SVD_PER_SERIES) bit for bit, a
    series with a gap is routed, and U D^2 U^T / the means equal the standard filter's on the whole batch."""
    import torch
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    T, N, d = 1000, 10000, 13
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    yh = simulate(mat, p, N, seed=6)
    yh[4321, 700, 0] = np.nan
    y = torch.as_tensor(yh, device="cuda")
    sv = eng.svd_filter(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (N - 1, 1)
    assert int((sv["status"] != 0).sum().item()) == 0
    rec = sv["svd"]
    pick = [0, 4321, 4322, 9999]
    own = eng.svd_filter(mat, p, y[pick], flags=_lib.OPT_SVD_PER_SERIES)["svd"]
    assert torch.equal(own, rec[pick])
    kf = eng.filter(mat, p, y)["filt"]
    worst_m = worst_c = 0.0
    for lo in range(0, N, 1000):                       # (in blocks: the reconstruction makes 13.5 GB temporaries per 10 000 series)
        r = rec[lo:lo + 1000]
        m, dc, U = r[..., :d], r[..., d:2 * d], r[..., 2 * d:].reshape(-1, T + 1, d, d).transpose(-1, -2)
        C = (U * (dc * dc).unsqueeze(-2)) @ U.transpose(-1, -2)
        k = kf[lo:lo + 1000]
        worst_m = max(worst_m, float((m - k[..., :d]).abs().max()))
        worst_c = max(worst_c, float((C - k[..., d:].reshape(-1, T + 1, d, d)).abs().max()))
    assert worst_m < 1e-7 and worst_c < 1e-7, (worst_m, worst_c)


def test_c4g_full_size_inside_inverse_wishart_gibbs(eng):
    """BASELINE configs[3] as the pooled Inverse-Wishart Gibbs sampler calls it (GibbsWishart.scala:40-80): 2000 series x T = 1000,
    d = 40, p = 20, outer-product statistics, filt_ws = NULL -- the records-free call: every series resident in the (64, 4) draw kernel,
    the series without a gap leaving k_filter_w48 where the recursion settles (k_steady_filter_w48), gaps routed to k_sampler_w48.
    The same seed reproduces; a shard with series_offset reproduces its block; a few series equal the generic kernel's draws (the
    reference's operation sequence in LDS, Smoothing.scala:74-103) at full length; the series with gaps equal their
    DLM_OPT_SAMPLER_PER_SERIES results bit for bit; the statistics are those of the states."""
    import torch
    from bench import multivariate_c4
    mod, p = multivariate_c4()
    T, N, d, q = 1000, 2000, 40, 20
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    y = eng.simulate(mat, p, N, seed=0xC46, series_offset=0, device=True, want_x=False)["y"]
    gaps = [17, 1024, 1999]
    for i, n in enumerate(gaps):
        y[n, 100 + 300 * i, i] = float("nan")
    fl = _lib.OPT_STATS_OUTER
    a = eng.ffbs(mat, p, y, seed=13, flags=fl | _lib.OPT_COUNT_STEPS, want_filt=False)
    assert eng.last_variant == "wave-sampler-shared" and eng.last_counters()[2:] == (N - len(gaps), len(gaps))
    assert int((a["status"] != 0).sum().item()) == 0
    b = eng.ffbs(mat, p, y, seed=13, flags=fl, want_filt=False)
    assert torch.equal(a["theta"], b["theta"]) and torch.equal(a["stats"], b["stats"])
    lo, hi = 1000, 1250                                    # rank 4 of 8
    c = eng.ffbs(mat, p, y[lo:hi], seed=13, series_offset=lo, flags=fl, want_filt=False)
    assert torch.equal(c["theta"], a["theta"][lo:hi]) and torch.equal(c["stats"], a["stats"][lo:hi])
    for n in gaps:                                         # their own factors, whatever the route
        g = eng.ffbs(mat, p, y[n:n + 1], seed=13, series_offset=n, flags=fl | _lib.OPT_SAMPLER_PER_SERIES)
        assert eng.last_variant == "wave-sampler"
        assert torch.equal(g["theta"][0], a["theta"][n]) and torch.equal(g["stats"][0], a["stats"][n])
    om = _om(mat)
    for n in (0, 1024, 1998):
        g = eng.ffbs(mat, p, y[n:n + 1], seed=13, series_offset=n, flags=fl | _lib.OPT_FORCE_GENERIC)
        assert eng.last_variant == "generic"
        np.testing.assert_allclose(a["theta"][n].cpu().numpy(), g["theta"][0].cpu().numpy(), rtol=1e-6, atol=1e-6)
        st = oracle.gibbs_stats(om, y[n].cpu().numpy(), a["theta"][n].cpu().numpy(), want_outer=True)
        got = a["stats"][n].cpu().numpy()
        np.testing.assert_allclose(got[:q], st["ssy"], rtol=1e-9)
        np.testing.assert_array_equal(got[q:2 * q], st["n"])
        np.testing.assert_allclose(got[2 * q:2 * q + d * d], np.asarray(st["outer"]).reshape(-1, order="F"), rtol=1e-8, atol=1e-8)
        assert got[-1] == T


def test_c3_full_size_ffbs_properties(eng):
    """BASELINE configs[2] (one rank's FFBS pass): 10 000 series x T = 1000, simulation smoother + statistics.  Properties:
    the same seed reproduces the draw bit for bit; a shard [lo, hi) with series_offset = lo draws exactly the same states
    (what makes an 8-GPU run reproduce the 1-GPU run); the statistics equal those recomputed from the states by the
    oracle; the draws scatter around the smoothed means with the smoothed variances."""
    import torch
    from bayesian_dlms_amd import _lib
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    y = torch.as_tensor(simulate(mat, p, 10000, seed=77), device="cuda")
    fl = _lib.OPT_FFBS_SIMSMOOTH
    a = eng.ffbs(mat, p, y, seed=9, flags=fl)
    assert eng.last_variant == "sparse16-simsmooth" and int((a["status"] != 0).sum().item()) == 0
    b = eng.ffbs(mat, p, y, seed=9, flags=fl)
    assert torch.equal(a["theta"], b["theta"]) and torch.equal(a["stats"], b["stats"])
    lo, hi = 2500, 3750                                    # rank 2 of 8
    c = eng.ffbs(mat, p, y[lo:hi], seed=9, series_offset=lo, flags=fl)
    assert torch.equal(c["theta"], a["theta"][lo:hi]) and torch.equal(c["stats"], a["stats"][lo:hi])
    for n in (0, 9999):
        st = oracle.gibbs_stats(_om(mat), y[n].cpu().numpy(), a["theta"][n].cpu().numpy())
        got = a["stats"][n].cpu().numpy()
        np.testing.assert_allclose(got[0], st["ssy"][0], rtol=1e-9)
        np.testing.assert_allclose(got[2:15], st["ss"], rtol=1e-9)
        assert got[1] == st["n"][0] and got[-1] == 1000
    sm = eng.filter_smooth(mat, p, y)["smooth"]
    zsc = (a["theta"] - sm[..., :13]) / torch.sqrt(sm[..., 13:].reshape(10000, 1001, 13, 13).diagonal(dim1=-2, dim2=-1))
    assert abs(float(zsc.mean())) < 2e-3 and abs(float(zsc.var()) - 1.0) < 5e-3   # 1.3e8 standardised draws


def test_c3_full_size_reference_form_sampler(eng):
    """BASELINE configs[2] with the reference's backward sampler (Smoothing.sampleDlm, the default of dlm_ffbs_batch): 10 000
    series x T = 1000 on the shared-factor path (the table of one wave per stretch, the mean-only draw kernel).  The same seed reproduces the
    draw, a shard with series_offset draws the same states, a few series equal the generic kernel's draws (the reference's
    operation sequence in LDS) at full length, and the statistics are those of the states."""
    import torch
    from bench import seasonal_c2, simulate
    mod, p = seasonal_c2()
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    y = torch.as_tensor(simulate(mat, p, 10000, seed=78), device="cuda")
    a = eng.ffbs(mat, p, y, seed=11)
    assert eng.last_variant == "sparse16-sampler-shared" and int((a["status"] != 0).sum().item()) == 0   # J_t, H_t, chol(H_t) once per call
    b = eng.ffbs(mat, p, y, seed=11)
    assert torch.equal(a["theta"], b["theta"]) and torch.equal(a["stats"], b["stats"])
    b = eng.ffbs(mat, p, y, seed=11, flags=_lib.OPT_SAMPLER_PER_SERIES)     # every series its own factors: the same draws, bit for bit
    assert eng.last_variant == "sparse16-sampler"
    assert torch.equal(a["theta"], b["theta"]) and torch.equal(a["stats"], b["stats"])
    lo, hi = 6250, 7500                                    # rank 5 of 8
    c = eng.ffbs(mat, p, y[lo:hi], seed=11, series_offset=lo)
    assert torch.equal(c["theta"], a["theta"][lo:hi]) and torch.equal(c["stats"], a["stats"][lo:hi])
    pick = [0, 5000, 9999]
    for n in pick:
        g = eng.ffbs(mat, p, y[n:n + 1], seed=11, series_offset=n, flags=_lib.OPT_NO_SAMPLER16)
        assert eng.last_variant == "generic"
        np.testing.assert_allclose(a["theta"][n].cpu().numpy(), g["theta"][0].cpu().numpy(), rtol=1e-7, atol=1e-7)
        st = oracle.gibbs_stats(_om(mat), y[n].cpu().numpy(), a["theta"][n].cpu().numpy())
        got = a["stats"][n].cpu().numpy()
        np.testing.assert_allclose(got[0], st["ssy"][0], rtol=1e-9)
        np.testing.assert_allclose(got[2:15], st["ss"], rtol=1e-9)
        assert got[1] == st["n"][0] and got[-1] == 1000


def test_c1_shape_at_scale_lane_kernels(eng):
    """BASELINE configs[0]'s model (first-order DLM, T = 1000) over 200 000 series, and a linear-growth model over 100 000
    with an irregular grid: one lane per series (dlm_lane.hip) through the same properties."""
    import torch
    rng = np.random.default_rng(101)
    mat = materialise(Dlm.polynomial(1), np.arange(1, 1001, dtype=np.float64))
    p = DlmParameters([[2.0]], [[3.0]], [0.0], [[10.0]])
    yh = rng.standard_normal((200000, 1000, 1)).cumsum(axis=1)
    yh[rng.random(yh.shape) < 0.03] = np.nan
    _properties(eng, mat, p, torch.as_tensor(yh, device="cuda"), [0, 199999, 63, 64, 123457], "lane", 1e-10, 1e-9)
    gaps = np.tile(np.array([1.0, 2.0, 1.0, 0.5, 3.0, 1.0, 1.0, 0.5]), 125)
    mat2 = materialise(Dlm.polynomial(2), np.cumsum(gaps))
    p2 = DlmParameters([[1.5]], np.array([[0.4, 0.1], [0.1, 0.2]]), [0.0, 0.0], np.eye(2) * 5.0)
    y2 = rng.standard_normal((100000, 1000, 1)).cumsum(axis=1)
    y2[rng.random(y2.shape) < 0.03] = np.nan
    _properties(eng, mat2, p2, torch.as_tensor(y2, device="cuda"), [0, 99999, 31415], "lane", 1e-9, 1e-8)


def test_small_multivariate_full_size(eng):
    """d = 8, p = 4 (four linear-growth components), 10 000 series x T = 1000 on the per-wave kernels with one tile per dimension."""
    import torch
    rng = np.random.default_rng(84)
    mod = Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, 1001, dtype=np.float64))
    B = rng.standard_normal((4, 4)); A = rng.standard_normal((8, 8))
    p = DlmParameters(B @ B.T / 4 + 0.5 * np.eye(4), A @ A.T / 8 * 0.2 + 0.05 * np.eye(8), np.zeros(8), np.eye(8) * 4.0)
    yh = rng.standard_normal((10000, 1000, 4)).cumsum(axis=1)
    yh[rng.random(yh.shape) < 0.03] = np.nan
    _properties(eng, mat, p, torch.as_tensor(yh, device="cuda"), [0, 9999, 4242], "wave-mfma", 1e-8, 1e-7)


@pytest.mark.parametrize("config", ["c2", "c3"])
def test_bench_multi_rank_control_flow_rehearsal(config, tmp_path):
    """bench.py with two ranks (the driver's launch shape: RANK / WORLD_SIZE / MASTER_* in the environment) on the one GPU
    of this box, torch.distributed over gloo (two RCCL ranks cannot share a device): strong scaling shards the job's
    series, rank 0 prints ONE JSON line with the whole-job value; for c3 the pooled statistics are summed over the ranks and
    every rank makes the same draw.  The timings of such a run mean nothing -- this is the control flow only."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
            "--config", config, "--series", "1200", "--T", "120"]
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561" if config == "c2" else "29562",
               DLM_BENCH_BACKEND="gloo", DLM_BENCH_DEVICE="0")
    procs = [subprocess.Popen(args, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not [l for l in outs[1][0].splitlines() if l.startswith("{")]     # rank 0 alone reports
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0 and j["status_nonzero_series"] == 0
    assert j["config"]["series_total"] == 1200 and j["config"]["series_per_gpu"] == 600
    if config == "c3":
        assert "comm world 2" in j["config"]["parallelism"] and np.isfinite(j["config"]["pooled_V"])
