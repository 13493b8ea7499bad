"""Shared factors of the RTS smoother, literal Q1 and textbook ("next round" 9).

`DLM_OPT_SMOOTHER_COMPAT_Q1` selects Smoothing.scala:44 as written (S_t = C_t + J_t (S_{t+1} - R_{t+1}) J_t, no transpose).  J_t and S_t
depend on V, W, C0 and the grid, not on the data: where the batch shares them, `k_smoother_rts16` runs once per call on the filter
records of a series of zeros and every series without a missing observation runs only its mean recursion against the tables
(`k_mean_rts16`); a series with a gap is routed to the per-series kernel.  The arithmetic of every output element is the per-series
kernel's, operation for operation: the parity bar is EQUALITY OF BITS with `DLM_OPT_SMOOTHER_PER_SERIES`, plus the usual tolerance
against the oracle's literal smoother (Smoothing.scala:31-64)."""
import numpy as np
import pytest

import oracle
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise

pytestmark = pytest.mark.gpu

Q1 = _lib.OPT_SMOOTHER_COMPAT_Q1 | _lib.OPT_NO_SMALL_BATCH    # (the shared route starts at 2048 series; the flag takes small batches there too)
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
    sh = eng.filter_smooth(mat, p, y, flags=flags | Q1 | _lib.OPT_COUNT_STEPS)
    cnt = eng.last_counters()
    assert eng.last_variant == "sparse16-rts-shared"
    ps = eng.filter_smooth(mat, p, y, flags=flags | Q1 | _lib.OPT_SMOOTHER_PER_SERIES | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (0, 0)
    return sh, ps, cnt


def same(a, b, what):
    if np.array_equal(a, b, equal_nan=True):
        return
    ne = np.argwhere(~((a == b) | (np.isnan(a) & np.isnan(b))))
    raise AssertionError(f"{what}: {len(ne)} values differ; first at {ne[0].tolist()}: {a[tuple(ne[0])]!r} vs {b[tuple(ne[0])]!r}; "
                         f"series {np.unique(ne[:, 0])[:8].tolist()}, t in [{ne[:, 1].min()}, {ne[:, 1].max()}], entries {np.unique(ne[:, 2])[:8].tolist()}")


@pytest.mark.parametrize("flags", [0])
@pytest.mark.parametrize("T,N", [(1000, 37), (1, 5), (2, 3), (3, 4), (40, 9), (63, 4), (64, 1), (65, 6), (129, 300), (700, 2)])
def test_bit_for_bit_with_the_per_series_kernel(eng, T, N, flags):
    mat, p = c2(T)
    rng = np.random.default_rng(T + N)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    sh, ps, cnt = both(eng, mat, p, y, flags)
    assert cnt[2] == N and cnt[3] == 0, cnt
    same(sh["filt"], ps["filt"], "filtered records")
    same(sh["smooth"], ps["smooth"], "smoothed records")
    assert np.all(sh["status"] == 0) and np.all(ps["status"] == 0)
    n = N // 2
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
    s = oracle.smoother(omodel(mat), f, compat_q1=True)
    np.testing.assert_allclose(sh["smooth"][n][:, :13], s["s"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sh["smooth"][n][:, 13:], s["S"], rtol=1e-8, atol=1e-9)
    if T > 2:   # the literal form is not the textbook one (and not symmetric)
        st = oracle.smoother(omodel(mat), f)
        assert np.abs(s["S"] - st["S"]).max() > 1e-3


def test_series_with_missing_observations_are_routed_to_their_own_recursion(eng):
    T, N = 300, 67
    mat, p = c2(T)
    rng = np.random.default_rng(3)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    y[1, 0, 0] = np.nan
    y[5, T - 1, 0] = np.nan
    y[17, 100:110, 0] = np.nan
    y[40, :, 0] = np.nan
    y[66, 64, 0] = np.nan
    sh, ps, cnt = both(eng, mat, p, y)
    assert cnt[2] == N - 5 and cnt[3] == 5, cnt
    same(sh["filt"], ps["filt"], "filtered records")
    same(sh["smooth"], ps["smooth"], "smoothed records")
    for n in (1, 17, 40, 2, 66):
        f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
        s = oracle.smoother(omodel(mat), f, compat_q1=True)
        np.testing.assert_allclose(sh["smooth"][n][:, :13], s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sh["smooth"][n][:, 13:], s["S"], rtol=1e-8, atol=1e-9)


def test_per_series_prior_means_share_the_factors_other_parameters_do_not(eng):
    T, N = 200, 6
    mat, p = c2(T)
    rng = np.random.default_rng(4)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    from bayesian_dlms_amd.engine import pack_params
    V, vs, W, ws, m0, ms, C0, cs, vts, wts = pack_params(p, N)
    m0s = rng.standard_normal((N, 13))
    packed = (V, 0, W, 0, m0s.reshape(-1), 13, C0, 0, 0, 0)
    sh = eng.filter_smooth(mat, packed, y, flags=Q1 | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2] == N
    ps = eng.filter_smooth(mat, packed, y, flags=Q1 | _lib.OPT_SMOOTHER_PER_SERIES)
    same(sh["smooth"], ps["smooth"], "smoothed records")
    small = eng.filter_smooth(mat, packed, y, flags=_lib.OPT_SMOOTHER_COMPAT_Q1 | _lib.OPT_COUNT_STEPS)   # six series: below the threshold, every series its own
    assert eng.last_counters()[2:] == (0, 0)
    same(small["smooth"], ps["smooth"], "smoothed records of the small batch")
    plist = [DlmParameters(p.v * (1 + 0.1 * n), p.w, p.m0, p.c0) for n in range(N)]
    out = eng.filter_smooth(mat, plist, y, flags=Q1 | _lib.OPT_COUNT_STEPS)
    assert eng.last_counters()[2:] == (0, 0)
    f = oracle.kf_filter(omodel(mat), plist[3].v, p.w, p.m0, p.c0, y[3])
    s = oracle.smoother(omodel(mat), f, compat_q1=True)
    np.testing.assert_allclose(out["smooth"][3][:, 13:], s["S"], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("d,knz", [(1, 1), (2, 1), (3, 2), (6, 3), (7, 1), (8, 4), (10, 2), (11, 3), (12, 4), (14, 2), (15, 3)])
def test_every_state_dimension_and_sparsity_instantiation(eng, d, knz):
    """d = 1 .. 15 (the five store / DMA instantiations of k_mean_rts16) and 1 .. 4 nonzeros per row of G, a gap in one series, batches that
    do not fill their last wave.  d <= 5 would take the lane kernels' forward pass: DLM_OPT_NO_LANE keeps the structured path."""
    rng = np.random.default_rng(80 + 16 * knz + d)
    T, N = 150, 7
    knz = min(knz, d)
    Gm = np.zeros((d, d))
    coef = {1: [0.9], 2: [0.7, 0.25], 3: [0.6, 0.25, -0.2], 4: [0.5, 0.3, -0.2, 0.15]}[knz]
    for i in range(d):
        for s_, cf in enumerate(coef):
            Gm[i, (i + s_) % d] += cf
    Fv = rng.choice([1.0, 0.0, 0.5, -0.5], size=d).reshape(-1, 1)
    Fv[0, 0] = 1.0
    mat = materialise(Dlm(lambda t: Fv, lambda dt: Gm), np.arange(1, T + 1, dtype=np.float64))
    p = DlmParameters([[0.7]], np.diag(rng.uniform(0.1, 0.5, d)), rng.standard_normal(d), np.eye(d))
    y = rng.standard_normal((N, T, 1))
    y[1, 70:73, 0] = np.nan
    sh, ps, cnt = both(eng, mat, p, y, _lib.OPT_NO_LANE)
    assert cnt[2] == N - 1 and cnt[3] == 1, cnt
    same(sh["filt"], ps["filt"], "filtered records")
    same(sh["smooth"], ps["smooth"], "smoothed records")
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0])
    s = oracle.smoother(omodel(mat), f, compat_q1=True)
    np.testing.assert_allclose(sh["smooth"][0][:, d:], s["S"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sh["smooth"][0][:, :d], s["s"], rtol=1e-8, atol=1e-9)


def test_smoothed_moments_only_and_repeated_calls_of_changing_size(eng):
    """filt = NULL (the records stay in an engine workspace), then calls of other sizes on the same engine: the tables' workspace is re-carved."""
    rng = np.random.default_rng(9)
    ref = None
    for T, N in ((120, 5), (400, 33), (50, 2), (400, 33)):
        mat, p = c2(T)
        y = np.random.default_rng(T * 7 + N).standard_normal((N, T, 1)).cumsum(axis=1) * 0.2
        sh = eng.filter_smooth(mat, p, y, flags=Q1, want_filt=False)
        ps = eng.filter_smooth(mat, p, y, flags=Q1 | _lib.OPT_SMOOTHER_PER_SERIES)
        same(sh["smooth"], ps["smooth"], f"smoothed records T={T} N={N}")
        if (T, N) == (400, 33):
            if ref is None:
                ref = np.array(sh["smooth"])
            else:
                same(np.array(sh["smooth"]), ref, "the same call again")


def test_a_batch_above_the_threshold_takes_the_shared_route_by_default(eng):
    """2100 series, no flag: the tables serve every series but the two with a gap."""
    T, N = 60, 2100
    mat, p = c2(T)
    rng = np.random.default_rng(12)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3
    y[7, 30, 0] = np.nan
    y[2099, 0, 0] = np.nan
    sh = eng.filter_smooth(mat, p, y, flags=_lib.OPT_SMOOTHER_COMPAT_Q1 | _lib.OPT_COUNT_STEPS)
    cnt = eng.last_counters()
    assert cnt[2] == N - 2 and cnt[3] == 2, cnt
    ps = eng.filter_smooth(mat, p, y, flags=_lib.OPT_SMOOTHER_COMPAT_Q1 | _lib.OPT_SMOOTHER_PER_SERIES)
    same(sh["filt"], ps["filt"], "filtered records")
    same(sh["smooth"], ps["smooth"], "smoothed records")


def test_every_step_in_full_means_every_series_its_own(eng):
    """DLM_OPT_NO_STEADY asks for the reference's work, every series' own recursion at every step: the tables are not used."""
    T, N = 80, 1100
    mat, p = c2(T)
    y = np.random.default_rng(13).standard_normal((N, T, 1)).cumsum(axis=1) * 0.3
    for sem in (_lib.OPT_SMOOTHER_COMPAT_Q1, 0):
        eng.filter_smooth(mat, p, y, flags=sem | _lib.OPT_NO_SMALL_BATCH | _lib.OPT_NO_STEADY | _lib.OPT_COUNT_STEPS)
        assert eng.last_counters()[2:] == (0, 0)
        assert eng.last_variant in ("sparse16", "sparse16-rts")


# ---- the textbook covariance (the default semantics) through the same tables ----------------------------------------------------------

TB = _lib.OPT_NO_SMALL_BATCH


@pytest.mark.parametrize("T,N", [(1000, 37), (1, 5), (2, 3), (65, 6), (129, 300)])
def test_textbook_through_the_tables_is_the_rts_kernel_bit_for_bit_and_the_default_kernel_to_rounding(eng, T, N):
    """From 6144 series (DLM_OPT_NO_SMALL_BATCH here) the DEFAULT call takes the tables: S = C - J (R+ - S+) J^T from k_smoother_rts16's table
    run, the means from k_mean_rts16.  Equal bit for bit to the per-series RTS kernel on the same filter records (dlm_smooth_batch); equal to
    the default per-series kernel (information form, k_smoother_sp16) and to the oracle within rounding."""
    mat, p = c2(T)
    rng = np.random.default_rng(T + N)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3 + rng.standard_normal((N, T, 1))
    sh = eng.filter_smooth(mat, p, y, flags=TB | _lib.OPT_COUNT_STEPS)
    cnt = eng.last_counters()
    assert eng.last_variant == "sparse16-rts-shared" and cnt[2] == N and cnt[3] == 0, (eng.last_variant, cnt)
    own = eng.filter_smooth(mat, p, y, flags=TB | _lib.OPT_SMOOTHER_PER_SERIES | _lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "sparse16" and eng.last_counters()[2:] == (0, 0)
    same(sh["filt"], own["filt"], "filtered records")
    rts = eng.smooth(mat, p, sh["filt"])
    assert eng.last_variant == "sparse16-rts"
    same(sh["smooth"], rts["smooth"], "smoothed records against the per-series RTS kernel")
    scale = np.abs(own["smooth"]).max()
    assert np.abs(sh["smooth"] - own["smooth"]).max() <= 1e-9 * scale
    n = N // 2
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
    s = oracle.smoother(omodel(mat), f)
    np.testing.assert_allclose(sh["smooth"][n][:, :13], s["s"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sh["smooth"][n][:, 13:], s["S"], rtol=1e-8, atol=1e-9)


def test_textbook_routes_gaps_and_keeps_the_default_below_the_threshold(eng):
    T, N = 200, 41
    mat, p = c2(T)
    rng = np.random.default_rng(21)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3
    y[3, 50:60, 0] = np.nan
    y[40, T - 1, 0] = np.nan
    sh = eng.filter_smooth(mat, p, y, flags=TB | _lib.OPT_COUNT_STEPS)
    cnt = eng.last_counters()
    assert cnt[2] == N - 2 and cnt[3] == 2, cnt
    rts = eng.smooth(mat, p, sh["filt"])
    own = eng.filter_smooth(mat, p, y, flags=TB | _lib.OPT_SMOOTHER_PER_SERIES)
    gap = np.zeros(N, bool); gap[[3, 40]] = True
    same(sh["smooth"][~gap], rts["smooth"][~gap], "smoothed records against the per-series RTS kernel")
    same(sh["smooth"][gap], own["smooth"][gap], "series with a gap: the information-form kernel's records, as without the tables")
    same(sh["filt"], own["filt"], "filtered records")
    for n in (3, 40, 0):
        f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
        s = oracle.smoother(omodel(mat), f)
        np.testing.assert_allclose(sh["smooth"][n][:, :13], s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sh["smooth"][n][:, 13:], s["S"], rtol=1e-8, atol=1e-9)
    # no flag, 41 series: the per-series information-form kernel as before
    eng.filter_smooth(mat, p, y, flags=_lib.OPT_COUNT_STEPS)
    assert eng.last_variant == "sparse16" and eng.last_counters()[2:] == (0, 0)
    # the smoothed moments alone (filt = NULL): the same route, the filtered records in the engine's workspace -- the same smoothed records
    only = eng.filter_smooth(mat, p, y, flags=TB | _lib.OPT_COUNT_STEPS, want_filt=False)
    assert eng.last_variant == "sparse16-rts-shared" and eng.last_counters()[2:] == (N - 2, 2)
    same(only["smooth"], sh["smooth"], "smoothed records without the filtered ones")


@pytest.mark.parametrize("sem", [0, _lib.OPT_SMOOTHER_COMPAT_Q1])
def test_a_batch_whose_series_mostly_have_gaps_makes_no_tables(eng, sem):
    """More than half of the series with a missing observation: decided on the device -- the table kernels return at once and every series takes
    the per-series kernel (the counters say so); exactly half: the tables serve the other half."""
    T, N = 120, 20
    mat, p = c2(T)
    rng = np.random.default_rng(31)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3
    own = None
    for ngap in (11, 10):
        yy = y.copy()
        yy[:ngap, 40, 0] = np.nan
        sh = eng.filter_smooth(mat, p, yy, flags=sem | TB | _lib.OPT_COUNT_STEPS)
        cnt = eng.last_counters()
        assert (cnt[2], cnt[3]) == ((0, N) if ngap == 11 else (N - ngap, ngap)), (ngap, cnt)
        own = eng.filter_smooth(mat, p, yy, flags=sem | TB | _lib.OPT_SMOOTHER_PER_SERIES)
        if ngap == 11 or sem:
            same(sh["smooth"], own["smooth"], "smoothed records")
        else:
            same(sh["smooth"][:ngap], own["smooth"][:ngap], "smoothed records of the series with a gap")
            assert np.abs(sh["smooth"] - own["smooth"]).max() <= 1e-9 * np.abs(own["smooth"]).max()
        same(sh["filt"], own["filt"], "filtered records")


def test_asynchronous_calls_back_to_back_and_an_engine_destroyed_with_the_tables_in_flight():
    """Rules A and B of DESIGN.md 4.12 on this route: DLM_OPT_ASYNC calls of different sizes one behind the other on one engine (the second re-carves
    the tables' workspace while the first call's kernels may still run on two streams), then an engine closed right after such a call."""
    import torch
    from bayesian_dlms_amd.engine import Engine
    rng = np.random.default_rng(17)
    e2 = Engine(0)
    calls = []
    for T, N, sem in ((300, 3000, 0), (500, 1500, _lib.OPT_SMOOTHER_COMPAT_Q1), (200, 6000, 0), (500, 1500, 0)):
        mat, p = c2(T)
        yd = torch.as_tensor(rng.standard_normal((N, T, 1)).cumsum(axis=1) * 0.3, device="cuda:0")
        out = e2.filter_smooth(mat, p, yd, flags=sem | TB | _lib.OPT_ASYNC)
        assert e2.last_variant == "sparse16-rts-shared"
        calls.append((mat, p, yd, sem, out))
    e2.sync()
    for mat, p, yd, sem, out in calls:
        ref = e2.filter_smooth(mat, p, yd, flags=sem | TB)
        assert torch.equal(out["smooth"], ref["smooth"]) and torch.equal(out["filt"], ref["filt"])
        assert int(out["status"].abs().sum().item()) == 0
    mat, p, yd, sem, _ = calls[2]
    out = e2.filter_smooth(mat, p, yd, flags=TB | _lib.OPT_ASYNC)
    e2.close()                                            # the table run and the batch's kernels are in flight
    torch.cuda.synchronize()
    assert torch.equal(out["smooth"], calls[2][4]["smooth"])
    e3 = Engine(0)                                        # the device is fine afterwards
    ref = e3.filter_smooth(mat, p, yd, flags=TB)
    assert torch.equal(ref["smooth"], out["smooth"])
    e3.close()


def test_an_error_exit_with_the_table_run_in_flight_leaves_the_engine_usable(eng):
    """DLM_OPT_TEST_FAIL_AFTER_TABLES: the call fails right after it has started the covariance filter and the table run.  The next calls -- a larger
    one that re-sizes the tables' workspace, then the call itself -- are bit for bit what a fresh engine gives (rule A: the exit joined the stream)."""
    from bayesian_dlms_amd.engine import EngineError
    rng = np.random.default_rng(23)
    for sem in (0, _lib.OPT_SMOOTHER_COMPAT_Q1):
        mat, p = c2(400)
        y = rng.standard_normal((600, 400, 1)).cumsum(axis=1) * 0.3
        ref = eng.filter_smooth(mat, p, y, flags=sem | TB)
        with pytest.raises(EngineError):
            eng.filter_smooth(mat, p, y, flags=sem | TB | _lib.OPT_TEST_FAIL_AFTER_TABLES)
        mat2, p2 = c2(700)
        big = eng.filter_smooth(mat2, p2, np.concatenate([y, y[:, ::-1]], axis=1)[:, :700], flags=sem | TB)
        assert np.all(big["status"] == 0)
        out = eng.filter_smooth(mat, p, y, flags=sem | TB)
        same(out["smooth"], ref["smooth"], "smoothed records after the failed call")
        same(out["filt"], ref["filt"], "filtered records after the failed call")


def test_accuracy_of_the_tables_against_the_oracle_at_full_length(eng):
    """C2 at T = 1000: the tables' records, the per-series information-form kernel's and the oracle's (what tests/rts_accuracy_probe.py prints): both
    routes stay within the steady-state shortcut's budget of the exact recursion -- a few 1e-11 absolute on records whose largest entries are 3.5 and 12 --
    and the every-step kernels within 1e-13."""
    T, N = 1000, 16
    mat, p = c2(T)
    y = np.random.default_rng(0).standard_normal((N, T, 1)).cumsum(axis=1) * 0.3
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0])
    s = oracle.smoother(omodel(mat), f)
    for flags, tol in ((TB, 2e-10), (_lib.OPT_SMOOTHER_PER_SERIES, 2e-10), (_lib.OPT_NO_STEADY, 1e-13)):
        sm = np.array(eng.filter_smooth(mat, p, y, flags=flags)["smooth"])[0]
        assert np.abs(sm[:, :13] - s["s"]).max() <= tol and np.abs(sm[:, 13:] - s["S"]).max() <= tol, (flags, np.abs(sm[:, :13] - s["s"]).max(), np.abs(sm[:, 13:] - s["S"]).max())
