"""GPU parity tests: the HIP engine (through the C ABI) against the CPU oracle and the
reference's golden vectors.  Tolerances are fp64 and stated per test."""
import csv
import json
import os

import numpy as np
import pytest

import oracle
from bayesian_dlms_amd import _lib
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise
from bayesian_dlms_amd.engine import EngineError

pytestmark = pytest.mark.gpu

VARIANTS = [0, _lib.OPT_FORCE_GENERIC]


class _Eng:
    """The engine on one of its two kernel families for 16 <= d <= 48: "workgroup" (DLM_OPT_NO_WAVE: the workgroup-per-series
    kernels, which production takes for up to 256 series of a model that is not time-invariant) or "per-wave"
    (DLM_OPT_FORCE_WAVE: the wave-per-series kernels, which production takes otherwise).  Every test of this module runs under
    both; test_dispatch_rule_of_the_multivariate_kernels checks which one production picks.

    Since round 3 the "per-wave" run also turns on the opt-in shared-covariance kernels (DLM_OPT_SHARED_COV) for d <= 15, so that
    every test of this module exercises them where they are eligible; tests/test_shared_cov_gpu.py compares them with the default
    kernels bit for bit."""

    def __init__(self, engine, rule):
        self._e, self.rule, self._d = engine, rule, 0

    def __getattr__(self, name):
        attr = getattr(self._e, name)
        if name not in ("filter", "loglik", "smooth", "filter_smooth", "ffbs", "svd_filter", "svd_ffbs"):
            return attr

        def call(mat, *a, **kw):
            self._d = mat.d
            if self.rule == "per-wave":
                kw["flags"] = kw.get("flags", 0) | _lib.OPT_FORCE_WAVE
            elif mat.d >= 16 and not (kw.get("flags", 0) & _lib.OPT_FORCE_WAVE):
                kw["flags"] = kw.get("flags", 0) | _lib.OPT_NO_WAVE
            if self.rule == "per-wave" and mat.d <= 15:
                kw["flags"] = kw.get("flags", 0) | _lib.OPT_SHARED_COV
            return attr(mat, *a, **kw)
        return call

    def expect(self, variant):
        """Name of the variant the last call must have used, given the rule: a wave-* kernel of the d >= 16 range is
        its workgroup counterpart for the small batches of these tests under the production rule."""
        if self.rule == "workgroup" and self._d >= 16:
            return "generic" if variant == "wave-sampler" else variant.replace("wave-", "tiled-")   # (no workgroup form of the sampler)
        return variant


@pytest.fixture(scope="module", params=["workgroup", "per-wave"])
def eng(request):
    from bayesian_dlms_amd.engine import Engine
    e = Engine(0)
    yield _Eng(e, request.param)
    e.close()


def omodel(mat):
    return oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)


def split(rec, d):
    """[.., d+d*d] records -> (mean [.., d], cov column-major flat [.., d*d])"""
    return rec[..., :d], rec[..., d:]


def seasonal_model(T):
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    w = np.diag([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
    return mod, mat, DlmParameters([[1.0]], w, np.zeros(13), np.eye(13))


def simulate(mat, p, N, seed, missing=0.0):
    rng = np.random.default_rng(seed)
    d, q, T = mat.d, mat.p, mat.T
    G = oracle.from_cm(mat.G[: d * d], d, d); F = oracle.from_cm(mat.F[: d * q], d, q)
    Lw = np.linalg.cholesky(p.w + 1e-300 * np.eye(d)); Lv = np.linalg.cholesky(p.v)
    y = np.empty((N, T, q))
    for n in range(N):
        x = p.m0 + np.linalg.cholesky(p.c0) @ rng.standard_normal(d)
        for t in range(T):
            x = G @ x + Lw @ rng.standard_normal(d)
            y[n, t] = F.T @ x + Lv @ rng.standard_normal(q)
    if missing > 0:
        y[rng.random(y.shape) < missing] = np.nan
    return y


def oracle_filter_smooth(mat, p, y, compat=False):
    om = omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    s = oracle.smoother(om, f, compat_q1=compat)
    return f, s


# ------------------------------------------------------------------------------------------
# golden vectors straight from the reference
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags", VARIANTS)
def test_first_order_golden_csv(eng, golden_dir, flags):
    rows = list(csv.reader(open(os.path.join(golden_dir, "first_order_dlm.csv"))))[1:]
    times = np.array([float(r[0]) for r in rows]); y = np.array([float(r[1]) for r in rows]).reshape(1, -1, 1)
    mat = materialise(Dlm.polynomial(1), times)
    p = DlmParameters([[2.0]], [[3.0]], [0.0], [[10.0]])
    out = eng.filter_smooth(mat, p, y, flags=flags)
    fr = list(csv.reader(open(os.path.join(golden_dir, "first_order_dlm_filtered.csv"))))[1:]
    sr = list(csv.reader(open(os.path.join(golden_dir, "first_order_dlm_smoothed.csv"))))[1:]
    # tolerance: 1e-12 relative (the CSVs hold the reference's full-precision doubles)
    np.testing.assert_allclose(out["filt"][0, :, 0], [float(r[1]) for r in fr], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["filt"][0, :, 1], [float(r[2]) for r in fr], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out["smooth"][0, :, 0], [float(r[1]) for r in sr], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(out["smooth"][0, :, 1], [float(r[2]) for r in sr], rtol=1e-11, atol=1e-11)
    assert int(out["status"][0]) == 0
    fq = eng.filter(mat, p, y, want_fq=True, flags=flags)["fq"]
    np.testing.assert_allclose(fq[0, 1:, 0], [float(r[3]) for r in fr[1:]], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(fq[0, 1:, 1], [float(r[4]) for r in fr[1:]], rtol=1e-12, atol=1e-12)


def test_kalman_filter_test_table(eng, golden_dir):
    """core/src/test/scala/KalmanFilter.scala:78-189 through the engine (generic kernel: p = 2)."""
    g = json.load(open(os.path.join(golden_dir, "kalman_filter_test.json")))
    mod = Dlm.polynomial(1) * Dlm.polynomial(1)
    p = DlmParameters(np.diag(g["v"]), np.diag(g["w"]), np.array(g["m0"]), np.diag(g["c0"]))
    y = np.array([[np.nan if v is None else v for v in row] for row in g["obs"]]).reshape(1, 6, 2)
    mat = materialise(mod, g["times"])
    out = eng.filter(mat, p, y, want_prior=True, want_fq=True)
    m, C = split(out["filt"][0], 2); a, R = split(out["prior"][0], 2)
    f, Q = out["fq"][0][:, :2], out["fq"][0][:, 2:]
    got = {"m": m, "a": a, "f": f}
    gotm = {"C": C, "R": R, "Q": Q}
    for k, exp in g["steps"].items():
        t = int(k)
        for name in ("a", "f", "m"):
            if name in exp:
                np.testing.assert_allclose(got[name][t], exp[name], atol=g["tol"], rtol=0)
        for name in ("R", "Q", "C"):
            if name in exp:
                np.testing.assert_allclose(oracle.from_cm(gotm[name][t], 2, 2), np.diag(exp[name]), atol=g["tol"], rtol=0)
    assert abs(m[6][0] - g["steps"]["6"]["m_first_commented"]) < 1e-4
    assert np.all(np.isnan(out["fq"][0][0]))


# ------------------------------------------------------------------------------------------
# engine vs oracle on seeded inputs
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags", VARIANTS)
@pytest.mark.parametrize("missing", [0.0, 0.15])
def test_seasonal_d13_filter_smooth(eng, flags, missing):
    mod, mat, p = seasonal_model(T=200)
    N = 5
    y = simulate(mat, p, N, seed=42, missing=missing)
    out = eng.filter_smooth(mat, p, y, flags=flags)
    assert np.all(out["status"] == 0)
    for n in range(N):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], 13); sm, S = split(out["smooth"][n], 13)
        # tolerance: 1e-9 relative + 1e-10 absolute over T = 200 steps of d = 13 recursions
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)


def _custom_model(G, F):
    return Dlm(lambda t: F.reshape(-1, 1), lambda dt: G)


@pytest.mark.parametrize("kind,expect", [("dense", "mfma16"), ("band3", "sparse16"), ("band4", "sparse16"),
                                         ("identity", "sparse16"), ("d15", "sparse16"), ("dense15", "mfma16")])
def test_fast_path_variants_by_g_structure(eng, kind, expect):
    """The engine picks its kernel from the structure of G; every variant must match the oracle."""
    rng = np.random.default_rng({"dense": 1, "band3": 2, "band4": 3, "identity": 4, "d15": 5, "dense15": 6}[kind])
    d = 15 if kind.endswith("15") else 9
    if kind.startswith("dense"):
        A = rng.standard_normal((d, d)); G = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
    elif kind == "band3":
        G = 0.5 * np.eye(d) + 0.3 * np.eye(d, k=1) - 0.2 * np.eye(d, k=-1)
    elif kind == "band4":
        G = 0.4 * np.eye(d) + 0.3 * np.eye(d, k=1) - 0.2 * np.eye(d, k=-1) + 0.1 * np.eye(d, k=3)
    elif kind == "identity":
        G = np.eye(d)
    else:
        G = 0.7 * np.eye(d) + 0.4 * np.eye(d, k=2)
    F = rng.standard_normal(d)
    mod = _custom_model(G, F)
    mat = materialise(mod, np.arange(1, 121, dtype=np.float64))
    A = rng.standard_normal((d, d))
    p = DlmParameters([[0.7]], A @ A.T / d + 0.05 * np.eye(d), rng.standard_normal(d), np.eye(d) * 1.5)
    y = simulate(mat, p, 6, seed=3, missing=0.1)
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == eng.expect(expect)
    assert np.all(out["status"] == 0)
    for n in range(6):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); sm, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
    fq = eng.filter(mat, p, y, want_fq=True)["fq"]
    f0 = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0])
    np.testing.assert_allclose(fq[0, 1:, 0], f0["f"][1:, 0], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(fq[0, 1:, 1], f0["Q"][1:, 0], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("case", ["seasonal_d13", "seasonal_d13_missing_irregular", "poly2_d2_dt0", "d8_p4_multivariate", "dense_g_d6"])
def test_smoother_q1_compat_switch(eng, case):
    """DLM_OPT_SMOOTHER_COMPAT_Q1 reproduces the literal Smoothing.scala:44 form (J X J, no transpose) -- on the register-tile
    RTS kernel (dlm_sampler16.hip: k_smoother_rts16) wherever G is structured and d <= 15, on the generic kernel elsewhere --
    and dlm_smooth_batch on bare filter records runs the same kernel in its textbook form."""
    rng = np.random.default_rng(4)
    expect = "sparse16-rts"
    if case.startswith("seasonal"):
        mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
        times = np.cumsum(np.array([1, 1, 2, 1, 3, 0.5] * 10, dtype=np.float64)) if "irregular" in case else np.arange(1, 121, dtype=np.float64)
        p = DlmParameters([[1.0]], np.diag([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4]), np.zeros(13), np.eye(13))
    elif case == "poly2_d2_dt0":
        mod, times = Dlm.polynomial(2), np.array([1, 2, 2, 3, 4, 4, 5, 7, 8, 9, 10, 10, 11], dtype=np.float64)   # repeated times: dt = 0
        p = DlmParameters([[2.0]], np.array([[0.5, 0.1], [0.1, 0.3]]), np.zeros(2), np.eye(2) * 5.0)
    elif case == "d8_p4_multivariate":
        mod = Dlm.polynomial(2)
        for _ in range(3):
            mod = mod * Dlm.polynomial(2)
        times = np.arange(1, 41, dtype=np.float64)
        A = rng.standard_normal((8, 8))
        p = DlmParameters(np.diag([1.0, 2.0, 0.5, 1.5]), A @ A.T / 8 + 0.1 * np.eye(8), rng.standard_normal(8), np.eye(8) * 2)
    else:
        Gd = 0.9 * np.eye(6) + 0.05 * rng.standard_normal((6, 6))
        mod, times = Dlm(lambda t: np.ones((6, 1)), lambda dt: Gd), np.arange(1, 41, dtype=np.float64)
        p = DlmParameters([[1.0]], np.eye(6) * 0.3, np.zeros(6), np.eye(6))
        expect = "generic"
    mat = materialise(mod, times)
    d = mat.d
    y = rng.standard_normal((3, mat.T, mat.p)).cumsum(axis=1)
    if "missing" in case:
        y[rng.random(y.shape) < 0.15] = np.nan
    out = eng.filter_smooth(mat, p, y, flags=_lib.OPT_SMOOTHER_COMPAT_Q1)
    assert eng.last_variant == expect and np.all(out["status"] == 0)
    tb = eng.smooth(mat, p, out["filt"])                       # textbook RTS from the records alone
    assert eng.last_variant == (expect if d > 5 or mat.p > 1 else "lane")
    for n in range(3):
        f, s = oracle_filter_smooth(mat, p, y[n], compat=True)
        sm, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
        s2 = oracle.smoother(omodel(mat), f, compat_q1=False)
        sm2, S2 = split(tb["smooth"][n], d)
        np.testing.assert_allclose(sm2, s2["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S2, s2["S"], rtol=1e-8, atol=1e-9)
    if d == 13:   # the literal covariance really is another matrix, the means are not
        assert np.abs(out["smooth"][..., 13:] - tb["smooth"][..., 13:]).max() > 1e-3
        np.testing.assert_allclose(out["smooth"][..., :13], tb["smooth"][..., :13], rtol=1e-8, atol=1e-9)
    # a long stationary stretch: the steady-state reuse of J must not change a digit that matters
    if case == "seasonal_d13":
        mat2 = materialise(mod, np.arange(1, 601, dtype=np.float64))
        y2 = rng.standard_normal((2, 600, 1)).cumsum(axis=1)
        o2 = eng.filter_smooth(mat2, p, y2, flags=_lib.OPT_SMOOTHER_COMPAT_Q1)
        g2 = eng.filter_smooth(mat2, p, y2, flags=_lib.OPT_SMOOTHER_COMPAT_Q1 | _lib.OPT_FORCE_GENERIC)
        assert eng.last_variant == "generic"
        np.testing.assert_allclose(o2["smooth"], g2["smooth"], rtol=1e-8, atol=1e-9)


def test_multivariate_d8_p4_missing_irregular(eng):
    """|*| of four polynomial(2) models: p = 4 observations, partial missingness, irregular dt."""
    mod = Dlm.polynomial(2)
    for _ in range(3):
        mod = mod * Dlm.polynomial(2)
    times = np.cumsum(np.array([1, 1, 2, 1, 1, 3, 1, 1, 1, 2] * 5, dtype=np.float64))
    mat = materialise(mod, times)
    assert mat.d == 8 and mat.p == 4 and mat.dt is not None
    rng = np.random.default_rng(9)
    A = rng.standard_normal((8, 8))
    p = DlmParameters(np.diag([1.0, 2.0, 0.5, 1.5]), A @ A.T / 8 + 0.1 * np.eye(8), rng.standard_normal(8), np.eye(8) * 2)
    y = simulate(mat, p, 3, seed=5, missing=0.25)
    out = eng.filter_smooth(mat, p, y)
    pr = eng.filter(mat, p, y, want_prior=True, want_fq=True)
    for n in range(3):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], 8); sm, S = split(out["smooth"][n], 8)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
        a, R = split(pr["prior"][n], 8)
        np.testing.assert_allclose(a, f["a"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(R, f["R"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(pr["fq"][n][1:, :4], f["f"][1:], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(pr["fq"][n][1:, 4:], f["Q"][1:], rtol=1e-9, atol=1e-10)


def test_per_series_parameters(eng):
    mod, mat, p = seasonal_model(T=50)
    rng = np.random.default_rng(2)
    ps = [DlmParameters(p.v * rng.uniform(0.5, 2), p.w * rng.uniform(0.5, 2), rng.standard_normal(13), np.eye(13) * rng.uniform(0.5, 3))
          for _ in range(4)]
    y = simulate(mat, p, 4, seed=8)
    for flags in VARIANTS:
        out = eng.filter_smooth(mat, ps, y, flags=flags)
        for n in range(4):
            f, s = oracle_filter_smooth(mat, ps[n], y[n])
            m, C = split(out["filt"][n], 13); sm, S = split(out["smooth"][n], 13)
            np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)


def test_d40_p20_config4_small(eng):
    """Config C4 shape (|*| of 20 polynomial(2)), small N and T."""
    mod = Dlm.polynomial(2)
    for _ in range(19):
        mod = mod * Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, 21, dtype=np.float64))
    assert mat.d == 40 and mat.p == 20
    rng = np.random.default_rng(40)
    A = rng.standard_normal((40, 40))
    p = DlmParameters(np.eye(20), A @ A.T / 40 + 0.1 * np.eye(40), np.zeros(40), np.eye(40))
    y = simulate(mat, p, 2, seed=40, missing=0.05)
    out = eng.filter_smooth(mat, p, y)
    for n in range(2):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], 40); sm, S = split(out["smooth"][n], 40)
        np.testing.assert_allclose(m, f["m"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(C, f["C"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(S, s["S"], rtol=1e-7, atol=1e-8)


# ------------------------------------------------------------------------------------------
# FFBS: conditional moments, draws under a shared RNG, sufficient statistics
# ------------------------------------------------------------------------------------------
def test_ffbs_moments_draws_and_stats(eng):
    mod, mat, p = seasonal_model(T=80)
    N = 3
    y = simulate(mat, p, N, seed=77, missing=0.1)
    seed = 20240611
    out = eng.ffbs(mat, p, y, seed=seed, series_offset=5, want_cond=True)
    assert eng.last_variant == "sparse16-sampler" and np.all(out["status"] == 0)   # register-tile literal sampler (dlm_sampler16.hip)
    gen = eng.ffbs(mat, p, y, seed=seed, series_offset=5, want_cond=True, flags=_lib.OPT_NO_SAMPLER16)
    assert eng.last_variant == "generic"
    np.testing.assert_allclose(out["theta"], gen["theta"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(out["cond"], gen["cond"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(out["stats"], gen["stats"], rtol=1e-8, atol=1e-9)
    om = omodel(mat)
    for n in range(N):
        f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        z = oracle.normals(seed, 5 + n, 81, 13)
        o = oracle.backward_sample(om, p.w, f, z, factor="chol")
        # (ii) draws equal the oracle's under the shared Philox stream and the Cholesky factor
        np.testing.assert_allclose(out["theta"][n], o["theta"], rtol=1e-7, atol=1e-8)
        # (i) conditional moments (h_t, H_t) given the same theta_{t+1}
        h, H = split(out["cond"][n], 13)
        np.testing.assert_allclose(h, o["h"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(H, o["H"], rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], o["theta"])
        L = out["stats"].shape[1]
        assert L == 1 + 1 + 13 + 1
        np.testing.assert_allclose(out["stats"][n, 0], st["ssy"][0], rtol=1e-7)
        assert out["stats"][n, 1] == st["n"][0]
        np.testing.assert_allclose(out["stats"][n, 2:15], st["ss"], rtol=1e-7)
        assert out["stats"][n, 15] == 80
    # injected normals path + outer-product statistics (GibbsWishart)
    zin = np.random.default_rng(3).standard_normal((N, 81, 13))
    out2 = eng.ffbs(mat, p, y, z=zin, flags=_lib.OPT_STATS_OUTER)
    for n in range(N):
        f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        o = oracle.backward_sample(om, p.w, f, zin[n], factor="chol")
        np.testing.assert_allclose(out2["theta"][n], o["theta"], rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], o["theta"], want_outer=True)
        np.testing.assert_allclose(out2["stats"][n, 2:2 + 169], st["outer"], rtol=1e-6, atol=1e-8)
    pooled = eng.stats_pool(out2["stats"])
    np.testing.assert_allclose(pooled, out2["stats"].sum(axis=0), rtol=1e-12)


@pytest.mark.parametrize("shape", ["seasonal_d13", "blocks_d20_p10", "local_level", "dense_d9_p2"])
def test_ffbs_with_the_references_eigen_factor(eng, shape):
    """DLM_OPT_DRAW_EIG: the draw factor of the reference itself -- MultivariateGaussianSvd (MultivariateGaussianSvd.scala:13-22):
    eigSym(H), theta = h + E sqrt(Lambda) z -- on the device (the general kernel's backward pass; eigenvalues ascending, the sign LAPACK
    leaves open fixed as in the oracle).  Against the oracle's factor = "eig" branch under injected normals: draws, conditional moments
    and statistics (1e-7); the draws differ from the Cholesky-factor draws of the same normals, the conditional moments do not."""
    rng = np.random.default_rng({"seasonal_d13": 1, "blocks_d20_p10": 2, "local_level": 3, "dense_d9_p2": 4}[shape])
    if shape == "seasonal_d13":
        mod, mat, p = seasonal_model(T=40)
        A = rng.standard_normal((13, 13))
        p = DlmParameters(p.v, A @ A.T / 13 + p.w, rng.standard_normal(13), np.diag(np.linspace(0.5, 2.0, 13)))
    elif shape == "blocks_d20_p10":
        mat, p = _block_model(10, 25, 2, seed=3)
    elif shape == "local_level":
        mat = materialise(Dlm.polynomial(1), np.arange(1, 31, dtype=np.float64))
        p = DlmParameters([[2.0]], [[3.0]], [0.0], [[10.0]])
    else:
        A = rng.standard_normal((9, 9)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((9, 2))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 31, dtype=np.float64))
        A2 = rng.standard_normal((9, 9))
        p = DlmParameters(np.eye(2) * 0.8, A2 @ A2.T / 9 + 0.2 * np.eye(9), np.zeros(9), np.eye(9) + 0.1 * A2 @ A2.T)
    d, q, T = mat.d, mat.p, mat.T
    N = 3
    y = rng.standard_normal((N, T, q)).cumsum(axis=1)
    y[1, T // 3, 0] = np.nan
    z = rng.standard_normal((N, T + 1, d))
    out = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_DRAW_EIG, want_cond=True)
    assert eng.last_variant == "generic-eig" and np.all(out["status"] == 0)
    chol = eng.ffbs(mat, p, y, z=z, want_cond=True)
    om = omodel(mat)
    for n in range(N):
        f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        o = oracle.backward_sample(om, p.w, f, z[n], factor="eig")
        np.testing.assert_allclose(out["theta"][n], o["theta"], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(out["cond"][n][:, :d], o["h"], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(out["cond"][n][:, d:], o["H"], rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], o["theta"])
        np.testing.assert_allclose(out["stats"][n, :q], st["ssy"], rtol=1e-6)
        np.testing.assert_allclose(out["stats"][n, 2 * q:2 * q + d], st["ss"], rtol=1e-6)
    if d > 1:
        assert np.abs(out["theta"] - chol["theta"]).max() > 1e-3          # another factor, other draws ...
    np.testing.assert_allclose(out["cond"][:, T], chol["cond"][:, T], rtol=1e-9, atol=1e-10)   # ... of the same conditional law (record T: m_T, C_T)
    with pytest.raises(EngineError):
        eng.ffbs(mat, p, y, flags=_lib.OPT_DRAW_EIG | _lib.OPT_FFBS_SIMSMOOTH)


def test_ffbs_distribution_matches_rts(eng):
    """(iii) many draws of one series: sample mean/cov of theta_t match the RTS smoother."""
    mod = Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, 31, dtype=np.float64))
    p = DlmParameters([[3.0]], np.diag([1.0, 0.3]), [0.0, 0.0], np.eye(2) * 0.5)
    y1 = simulate(mat, p, 1, seed=4)
    N = 4000
    y = np.repeat(y1, N, axis=0)
    out = eng.ffbs(mat, p, y, seed=99)
    f, s = oracle_filter_smooth(mat, p, y1[0])
    th = out["theta"]
    for t in (0, 10, 30):
        mean = th[:, t].mean(axis=0)
        cov = np.cov(th[:, t].T)
        S = oracle.from_cm(s["S"][t], 2, 2)
        se = np.sqrt(np.diag(S) / N)
        assert np.all(np.abs(mean - s["s"][t]) < 5 * se)
        np.testing.assert_allclose(cov, S, rtol=0.15, atol=0.02)


def test_argument_errors(eng):
    mod, mat, p = seasonal_model(T=10)
    from bayesian_dlms_amd.engine import EngineError
    y = np.zeros((1, 10, 1))
    bad = materialise(mod, np.arange(1, 11, dtype=np.float64))
    bad.T = 0
    with pytest.raises(EngineError):
        eng.filter(bad, p, np.zeros((1, 0, 1)))


def test_device_pointer_mode_matches_host_mode(eng):
    import torch
    mod, mat, p = seasonal_model(T=64)
    y = simulate(mat, p, 8, seed=12)
    host = eng.filter_smooth(mat, p, y)
    dev = eng.filter_smooth(mat, p, torch.as_tensor(y, device="cuda:0"))
    np.testing.assert_array_equal(dev["filt"].cpu().numpy(), host["filt"])
    np.testing.assert_array_equal(dev["smooth"].cpu().numpy(), host["smooth"])


# ------------------------------------------------------------------------------------------
# SVD (square-root) filter and sampler: config C5
# ------------------------------------------------------------------------------------------
def _svd_cov(rec, d):
    """[T+1, 2d+d*d] SVD records -> (m [T+1, d], C [T+1, d, d]) with C = uc diag(dc^2) uc^T"""
    m, dc = rec[:, :d], rec[:, d:2 * d]
    uc = rec[:, 2 * d:].reshape(-1, d, d).transpose(0, 2, 1)
    return m, np.einsum("tij,tj,tkj->tik", uc, dc ** 2, uc)


def test_svd_filter_reference_fixture(eng, golden_dir):
    """core/src/test/scala/SvdFilter.scala:102-158: SVD filter == Kalman filter on the bivariate fixture."""
    g = json.load(open(os.path.join(golden_dir, "kalman_filter_test.json")))
    mod = Dlm.polynomial(1) * Dlm.polynomial(1)
    p = DlmParameters(np.diag(g["v"]), np.diag(g["w"]), np.array(g["m0"]), np.diag(g["c0"]))
    y = np.array([[np.nan if v is None else v for v in row] for row in g["obs"]]).reshape(1, 6, 2)
    mat = materialise(mod, g["times"])
    kf = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0])
    for flags in (0, _lib.OPT_SVD_RAW_W_Q2):     # W = I: Q2 is invisible, exactly as in the reference test
        out = eng.svd_filter(mat, p, y, flags=flags)
        assert int(out["status"][0]) == 0
        m, C = _svd_cov(out["svd"][0], 2)
        np.testing.assert_allclose(m, kf["m"], atol=1e-9)
        for t in range(7):
            np.testing.assert_allclose(C[t], oracle.from_cm(kf["C"][t], 2, 2), atol=1e-9)


@pytest.mark.parametrize("literal_q2", [False, True])
def test_svd_filter_d13_vs_oracle(eng, literal_q2):
    mod, mat, p = seasonal_model(T=60)
    y = simulate(mat, p, 3, seed=21, missing=0.1)
    out = eng.svd_filter(mat, p, y, flags=_lib.OPT_SVD_RAW_W_Q2 if literal_q2 else 0)
    assert eng.last_variant == "svd-jacobi" and np.all(out["status"] == 0)
    om = omodel(mat)
    for n in range(3):
        o = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y[n], raw_w_q2=literal_q2)
        m, C = _svd_cov(out["svd"][n], 13)
        om_, oC = _svd_cov(np.concatenate([o["m"], o["dc"], o["uc"]], axis=1), 13)
        # tolerance 1e-7: two Jacobi SVDs per step, factors compared through U D^2 U^T
        np.testing.assert_allclose(m, om_, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(C, oC, rtol=1e-7, atol=1e-8)
    if not literal_q2:  # the consistent form is the Kalman filter (stability check vs. the standard filter)
        kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[0])
        m, C = _svd_cov(out["svd"][0], 13)
        np.testing.assert_allclose(m, kf["m"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(C.transpose(0, 2, 1).reshape(61, 169), kf["C"], rtol=1e-6, atol=1e-7)


def _block_model(nblk, T, per=2, seed=0):
    mod = Dlm.polynomial(per)
    for _ in range(nblk - 1):
        mod = mod * Dlm.polynomial(per)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d)); B = rng.standard_normal((q, q))
    return mat, DlmParameters(B @ B.T / q + 0.5 * np.eye(q), A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d) * 0.1,
                              np.diag(np.linspace(0.5, 2.0, d)))


@pytest.mark.parametrize("nblk,per,T", [(20, 2, 50), (10, 2, 40), (17, 1, 30), (12, 4, 25)])
def test_svd_filter_and_sampler_beyond_sixteen_states(eng, nblk, per, T):
    """SvdFilter.filterDlm / SvdSampler.ffbsDlm take any model in the reference (SvdFilter.scala:38-68, 183-202; SvdSampler.scala:15-36):
    the NM = 48 instantiation of dlm_svd.hip serves 16 < d <= 48 or 16 < p <= 32 -- the C4 model (d = 40, p = 20) among them, d = 20 /
    p = 10, d = p = 17 (p alone beyond sixteen), d = 48 / p = 12 -- with dense V and W, partially and fully missing observations:
    U D^2 U^T and the means against the oracle's SVD filter (1e-7) and its Kalman filter (1e-6), the draws under injected normals
    against the oracle's SVD sampler (1e-6), the statistics against those of the draws."""
    mat, p = _block_model(nblk, T, per, seed=nblk)
    d, q = mat.d, mat.p
    rng = np.random.default_rng(100 + nblk)
    N = 3
    y = rng.standard_normal((N, T, q)).cumsum(axis=1) * 0.5 + rng.standard_normal((N, T, q))
    y[2, 3, :] = np.nan
    om = omodel(mat)
    # a partially missing observation selects rows and columns of sqrt(V)^-1 = diag(sigma^-1/2) V^T (SvdFilter.scala:51):
    # that means something only when the decomposition leaves the components in the model's order -- a diagonal V whose entries
    # descend (LAPACK sorts the singular values, the oracle restates that; the engine's Jacobi order leaves a diagonal matrix alone).
    # That case runs with such a V; the dense V with whole observations present or missing.
    pdiag = DlmParameters(np.diag(np.linspace(1.8, 0.6, q)), p.w, p.m0, p.c0)
    ymiss = y[:1].copy()
    ymiss[0, T // 2, : q // 2] = np.nan
    ymiss[0, T // 2 + 1, 1::2] = np.nan
    for pp, yy in ((p, y), (pdiag, ymiss)):
        out = eng.svd_filter(mat, pp, yy)
        assert eng.last_variant == "svd-jacobi" and np.all(out["status"] == 0)
        for n in range(yy.shape[0]):
            o = oracle.svd_filter(om, pp.v, pp.w, pp.m0, pp.c0, yy[n])
            m, C = _svd_cov(out["svd"][n], d)
            om_, oC = _svd_cov(np.concatenate([o["m"], o["dc"], o["uc"]], axis=1), d)
            np.testing.assert_allclose(m, om_, rtol=1e-7, atol=1e-8)
            np.testing.assert_allclose(C, oC, rtol=1e-7, atol=1e-8)
    out = eng.svd_filter(mat, p, y)
    kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[0])
    m, C = _svd_cov(out["svd"][0], d)
    np.testing.assert_allclose(m, kf["m"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(C.transpose(0, 2, 1).reshape(T + 1, d * d), kf["C"], rtol=1e-6, atol=1e-7)
    z = rng.standard_normal((N, T + 1, d))
    fb = eng.svd_ffbs(mat, p, y, z=z, flags=_lib.OPT_STATS_OUTER)
    assert np.all(fb["status"] == 0)
    for n in range(N):
        sf = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        o = oracle.svd_backward_sample(om, p.w, sf, z[n], literal_q9=False)
        np.testing.assert_allclose(fb["theta"][n], o["theta"], rtol=1e-6, atol=1e-6)
        st = oracle.gibbs_stats(om, y[n], fb["theta"][n], want_outer=True)
        got = fb["stats"][n]
        np.testing.assert_allclose(got[:q], st["ssy"], rtol=1e-9)
        np.testing.assert_array_equal(got[q:2 * q], st["n"])
        np.testing.assert_allclose(got[2 * q:2 * q + d * d], st["outer"], rtol=1e-8, atol=1e-9)
    with pytest.raises(Exception, match="d <= 48"):
        big, pb = _block_model(25, 5, 2)
        eng.svd_filter(big, pb, np.zeros((1, 5, big.p)))


def test_svd_filter_ill_conditioned_stays_psd(eng):
    """Stability check: tiny observation noise and a wide prior; U D^2 U^T is PSD by construction."""
    mod = Dlm.polynomial(3)
    mat = materialise(mod, np.arange(1, 201, dtype=np.float64))
    p = DlmParameters([[1e-10]], np.diag([1e-8, 1e-8, 1e-8]), np.zeros(3), np.eye(3) * 1e8)
    y = simulate(mat, DlmParameters([[1e-2]], np.diag([1e-4] * 3), np.zeros(3), np.eye(3)), 4, seed=2)
    out = eng.svd_filter(mat, p, y)
    for n in range(4):
        m, C = _svd_cov(out["svd"][n], 3)
        assert np.all(np.isfinite(C))
        assert min(np.linalg.eigvalsh((c + c.T) / 2).min() for c in C) >= -1e-18


@pytest.mark.parametrize("literal", [False, True])
def test_svd_ffbs_draws_and_stats(eng, literal):
    mod, mat, p = seasonal_model(T=40)
    # C0 = I makes the record-0 factor fully degenerate (any orthogonal basis is an SVD), which
    # leaves the draw theta_0 = h + U D z undefined; use distinct prior variances instead
    p = DlmParameters(p.v, p.w, p.m0, np.diag(np.linspace(0.5, 2.0, 13)))
    N = 2
    y = simulate(mat, p, N, seed=31, missing=0.1)
    flags = (_lib.OPT_SVD_RAW_W_Q2 | _lib.OPT_SVD_SAMPLER_Q9) if literal else 0
    out = eng.svd_ffbs(mat, p, y, seed=77, series_offset=3, flags=flags)
    assert np.all(out["status"] == 0)
    om = omodel(mat)
    for n in range(N):
        sf = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y[n], raw_w_q2=literal)
        z = oracle.normals(77, 3 + n, 41, 13)
        o = oracle.svd_backward_sample(om, p.w, sf, z, literal_q9=literal)
        np.testing.assert_allclose(out["theta"][n], o["theta"], rtol=1e-6, atol=1e-7)
        st = oracle.gibbs_stats(om, y[n], o["theta"])
        np.testing.assert_allclose(out["stats"][n, 0], st["ssy"][0], rtol=1e-6)
        np.testing.assert_allclose(out["stats"][n, 2:15], st["ss"], rtol=1e-6)
    if not literal:  # zero noise => the SVD sampler reproduces the RTS mean
        z0 = np.zeros((N, 41, 13))
        o0 = eng.svd_ffbs(mat, p, y, z=z0)
        f, s = oracle_filter_smooth(mat, p, y[0])
        np.testing.assert_allclose(o0["theta"][0], s["s"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------
# FFBS by the Durbin-Koopman simulation smoother (DLM_OPT_FFBS_SIMSMOOTH)
# ------------------------------------------------------------------------------------------
def dk_reference_draw(mat, p, y, z):
    """theta = E[x | y - y+] + x+ with (x+, y+) simulated from z [T+1][d+1]; smoothing by the oracle.  p.w / p.v may be [T] streams
    (W_t drives the transition into record t, V_t observation t)."""
    d, T = mat.d, mat.T
    G = oracle.from_cm(mat.G[: d * d], d, d); F = mat.F[:d]
    Lc = np.linalg.cholesky(p.c0)
    Lws = [np.linalg.cholesky(w) for w in p.w] if p.w.ndim == 3 else [np.linalg.cholesky(p.w)] * T
    svs = [np.sqrt(v[0, 0]) for v in p.v] if p.v.ndim == 3 else [np.sqrt(p.v[0, 0])] * T
    x = p.m0 + Lc @ z[0, :d]
    xs, yp = [x], np.empty((T, 1))
    for t in range(1, T + 1):
        x = G @ x + Lws[t - 1] @ z[t, :d]
        xs.append(x)
        yp[t - 1, 0] = F @ x + svs[t - 1] * z[t, d]
    om = omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, np.zeros(d), p.c0, y - yp)      # zero prior mean
    s = oracle.smoother(om, f, compat_q1=False)
    return s["s"] + np.array(xs)


@pytest.mark.parametrize("dense_w", [False, True])
def test_ffbs_simulation_smoother_matches_reference_construction(eng, dense_w):
    mod, mat, p = seasonal_model(T=90)
    if dense_w:
        rng = np.random.default_rng(17)
        A = rng.standard_normal((13, 13))
        p = DlmParameters(p.v, A @ A.T / 13 + 0.05 * np.eye(13), rng.standard_normal(13), A.T @ A / 13 + 0.5 * np.eye(13))
    N = 4
    y = simulate(mat, p, N, seed=61, missing=0.1)
    z = np.random.default_rng(9).standard_normal((N, 91, 14))
    flags = _lib.OPT_FFBS_SIMSMOOTH
    out = eng.ffbs(mat, p, y, z=z, flags=flags)
    assert eng.last_variant == "sparse16-simsmooth" and np.all(out["status"] == 0)
    om = omodel(mat)
    for n in range(N):
        ref = dk_reference_draw(mat, p, y[n], z[n])
        # tolerance 1e-7: a filter and a mean smoother on top of simulated states
        np.testing.assert_allclose(out["theta"][n], ref, rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], ref)
        np.testing.assert_allclose(out["stats"][n, 0], st["ssy"][0], rtol=1e-7)
        assert out["stats"][n, 1] == st["n"][0] and out["stats"][n, -1] == 90
        np.testing.assert_allclose(out["stats"][n, 2:15], st["ss"], rtol=1e-7)
    # outer-product statistics (GibbsWishart) and the Philox stream (same normals as the oracle's)
    out2 = eng.ffbs(mat, p, y, z=z, flags=flags | _lib.OPT_STATS_OUTER)
    st = oracle.gibbs_stats(om, y[0], dk_reference_draw(mat, p, y[0], z[0]), want_outer=True)
    np.testing.assert_allclose(out2["stats"][0, 2:2 + 169], st["outer"], rtol=1e-6, atol=1e-8)
    out3 = eng.ffbs(mat, p, y, seed=5, series_offset=2, flags=flags)
    zz = oracle.normals(5, 2 + 1, 91, 14)
    np.testing.assert_allclose(out3["theta"][1], dk_reference_draw(mat, p, y[1], zz), rtol=1e-7, atol=1e-8)


def test_ffbs_simulation_smoother_with_variance_streams(eng):
    """DlmFsvSystem.ffbs feeds one W per step (DlmFsvSystem.scala:137-208), StudentT.filter one V per step (StudentTGibbs.scala:100-136):
    the simulation smoother on the structured d <= 15, p = 1 path takes both streams -- the Cholesky factor of W_t at every step of the
    simulation, W_t / V_t in the filter of y - y+ -- against the same construction from the oracle's filter and smoother (injected normals);
    elsewhere the streams are refused with a message."""
    mod, mat, p0 = seasonal_model(T=70)
    d, T = 13, 70
    rng = np.random.default_rng(23)
    Ws = np.empty((T, d, d)); Vs = np.empty((T, 1, 1))
    for t in range(T):
        A = rng.standard_normal((d, d)) * 0.3
        Ws[t] = A @ A.T + np.diag(rng.uniform(0.05, 0.5, d))
        Vs[t, 0, 0] = rng.uniform(0.5, 2.0)
    N = 3
    y = simulate(mat, p0, N, seed=62, missing=0.1)
    z = rng.standard_normal((N, T + 1, d + 1))
    for p in (DlmParameters(p0.v, Ws, p0.m0, p0.c0), DlmParameters(Vs, Ws, rng.standard_normal(d), np.eye(d) * 2.0)):
        out = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH)
        assert eng.last_variant == "sparse16-simsmooth" and np.all(out["status"] == 0)
        for n in range(N):
            np.testing.assert_allclose(out["theta"][n], dk_reference_draw(mat, p, y[n], z[n]), rtol=1e-7, atol=1e-8)
    # 16 <= d <= 48 and small multivariate models (the per-wave kernels): W_t streams at any batch size -- the factor of W_t per step in the
    # simulation prologue; a V_t stream (it would enter the backward gain) and a dense G are refused with a message
    for nblk, per, Tb in ((10, 2, 40), (20, 2, 30), (3, 2, 50)):
        big, pb = _block_model(nblk, Tb, per, seed=5 + nblk)
        db, qb = big.d, big.p
        Wb = np.empty((Tb, db, db))
        for t in range(Tb):
            A = rng.standard_normal((db, db)) * 0.2
            Wb[t] = A @ A.T + np.diag(rng.uniform(0.05, 0.4, db))
        pw = DlmParameters(pb.v, Wb, pb.m0, pb.c0)
        yb = rng.standard_normal((3, Tb, qb)).cumsum(axis=1)
        yb[1, Tb // 2, 0] = np.nan
        zb = rng.standard_normal((3, Tb + 1, db + qb))
        ob = eng.ffbs(big, pw, yb, z=zb, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_FORCE_WAVE)     # (the module's "workgroup" run adds DLM_OPT_NO_WAVE otherwise)
        assert eng.last_variant == "wave-simsmooth" and np.all(ob["status"] == 0)
        for n in range(3):
            np.testing.assert_allclose(ob["theta"][n], dk_reference_draw_mv(big, pw, yb[n], zb[n]), rtol=1e-6, atol=1e-7)
    with pytest.raises(Exception, match="V_t stream only"):
        eng.ffbs(big, DlmParameters(np.stack([pb.v] * Tb), pb.w, pb.m0, pb.c0), yb, flags=_lib.OPT_FFBS_SIMSMOOTH)
    with pytest.raises(Exception, match="structured G"):      # the workgroup-per-series kernels factor W once
        eng.ffbs(big, pw, yb, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_NO_WAVE)


def test_ffbs_simulation_smoother_distribution(eng):
    """Many draws of one series: mean and covariance of theta_t equal the RTS smoothing moments."""
    mod = Dlm.polynomial(2)
    mat = materialise(mod, np.arange(1, 31, dtype=np.float64))
    p = DlmParameters([[3.0]], np.diag([1.0, 0.3]), [0.5, -0.2], np.eye(2) * 0.5)
    y1 = simulate(mat, p, 1, seed=4)
    N = 6000
    out = eng.ffbs(mat, p, np.repeat(y1, N, axis=0), seed=123, flags=_lib.OPT_FFBS_SIMSMOOTH)
    assert eng.last_variant == "lane-simsmooth"   # d = 2, p = 1: one lane per series
    f, s = oracle_filter_smooth(mat, p, y1[0])
    for t in (0, 12, 30):
        th = out["theta"][:, t]
        S = oracle.from_cm(s["S"][t], 2, 2)
        assert np.all(np.abs(th.mean(axis=0) - s["s"][t]) < 5 * np.sqrt(np.diag(S) / N))
        np.testing.assert_allclose(np.cov(th.T), S, rtol=0.12, atol=0.02)


# ------------------------------------------------------------------------------------------
# multivariate tiled-MFMA path (16 <= d <= 48, p <= 32): config C4 and friends
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["c4", "d17_p3_irregular", "d24_p5_timevarying_f", "d20_p10_structured_irregular",
                                  "d48_p32_dense", "d20_p6_dense_f", "d36_p18_tridiag_g_two_per_column_f"])
def test_tiled_mfma_path(eng, case):
    rng = np.random.default_rng({"c4": 40, "d17_p3_irregular": 17, "d24_p5_timevarying_f": 24,
                                 "d20_p10_structured_irregular": 20, "d48_p32_dense": 48, "d20_p6_dense_f": 206,
                                 "d36_p18_tridiag_g_two_per_column_f": 36}[case])
    if case == "d20_p10_structured_irregular":
        # |*| of ten polynomial(2) blocks on an irregular grid with a repeated time: several structured G tables
        # (the gather congruence), W dt, and the dt = 0 identity advance
        mod = Dlm.polynomial(2)
        for _ in range(9):
            mod = mod * Dlm.polynomial(2)
        times = np.cumsum(np.array([1, 1, 2, 0, 3, 1, 1, 2] * 4, dtype=np.float64)) + 1.0
        d, p_ = 20, 10
        V = np.diag(rng.uniform(0.5, 2.0, p_))
    elif case == "d48_p32_dense":
        # the largest shape of the tiled path: G read from global memory in the backward pass, full 2 x 2 tile inverse
        d, p_ = 48, 32
        A = rng.standard_normal((d, d)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((d, p_))
        mod = Dlm(lambda t: F, lambda dt: G1)
        times = np.arange(1, 13, dtype=np.float64)
        B = rng.standard_normal((p_, p_)); V = B @ B.T / p_ + 0.5 * np.eye(p_)
    elif case == "d20_p6_dense_f":
        # structured G (bidiagonal) with a dense time-invariant F: the per-wave kernels with MFMA products for F
        d, p_ = 20, 6
        G1 = 0.85 * np.eye(d) + 0.1 * np.eye(d, k=1)
        F = rng.standard_normal((d, p_))
        mod = Dlm(lambda t: F, lambda dt: G1)
        times = np.arange(1, 31, dtype=np.float64)
        B = rng.standard_normal((p_, p_)); V = B @ B.T / p_ + 0.5 * np.eye(p_)
    elif case == "d36_p18_tridiag_g_two_per_column_f":
        # three nonzeros per row of G and two per column of F: the wider gather tables (K = 4, KF = 4 instantiations)
        d, p_ = 36, 18
        G1 = 0.7 * np.eye(d) + 0.12 * np.eye(d, k=1) - 0.1 * np.eye(d, k=-1)
        F = np.zeros((d, p_))
        for j in range(p_):
            F[2 * j, j] = 1.0; F[(2 * j + 5) % d, j] = 0.5
        mod = Dlm(lambda t: F, lambda dt: G1)
        times = np.arange(1, 31, dtype=np.float64)
        V = np.diag(rng.uniform(0.5, 2.0, p_))
    elif case == "c4":
        mod = Dlm.polynomial(2)
        for _ in range(19):
            mod = mod * Dlm.polynomial(2)
        times = np.arange(1, 41, dtype=np.float64)
        d, p_ = 40, 20
        V = np.eye(20)
    elif case == "d17_p3_irregular":
        d, p_ = 17, 3
        A = rng.standard_normal((d, d)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((d, p_))
        mod = Dlm(lambda t: F, lambda dt: np.linalg.matrix_power(G1, int(dt)) if dt > 0 else np.eye(d))
        times = np.cumsum(np.array([1, 1, 2, 1, 3, 1, 1, 2] * 4, dtype=np.float64))
        B = rng.standard_normal((p_, p_)); V = B @ B.T / p_ + 0.5 * np.eye(p_)   # correlated observation noise
    else:
        d, p_ = 24, 5
        G1 = 0.8 * np.eye(d) + 0.1 * np.eye(d, k=1)
        Fs = rng.standard_normal((30, d, p_))
        mod = Dlm(lambda t: Fs[int(t) - 1], lambda dt: G1)
        times = np.arange(1, 31, dtype=np.float64)
        V = np.diag(rng.uniform(0.5, 2.0, p_))
    mat = materialise(mod, times)
    assert mat.d == d and mat.p == p_
    A = rng.standard_normal((d, d))
    p = DlmParameters(V, A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 1.5)
    N = 3
    y = rng.standard_normal((N, mat.T, p_)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.2] = np.nan
    y[:, 5, :] = np.nan                                  # a fully missing record
    out = eng.filter_smooth(mat, p, y)
    # a structured G runs one wavefront per series with register-resident tiles (dlm_wave48.hip), a dense G the
    # workgroup-per-series kernels (dlm_tiled.hip)
    structured = case in ("c4", "d20_p10_structured_irregular", "d24_p5_timevarying_f", "d20_p6_dense_f",
                          "d36_p18_tridiag_g_two_per_column_f")
    assert eng.last_variant == eng.expect("wave-mfma" if structured else "tiled-mfma") and np.all(out["status"] == 0)
    if structured:   # the two implementations agree far inside the oracle tolerance
        ref = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_WAVE)
        assert eng.last_variant == "tiled-mfma"
        np.testing.assert_allclose(out["filt"], ref["filt"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(out["smooth"], ref["smooth"], rtol=1e-8, atol=1e-9)
    fq = eng.filter(mat, p, y, want_fq=True)
    assert eng.last_variant == eng.expect("wave-mfma" if structured else "tiled-mfma")
    for n in range(N):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); sm, S = split(out["smooth"][n], d)
        # tolerance: 1e-8 (filter) / 1e-7 (smoother): 40-dimensional products, p x p inverses
        np.testing.assert_allclose(m, f["m"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(C, f["C"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(S, s["S"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(fq["fq"][n][1:, :p_], f["f"][1:], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(fq["fq"][n][1:, p_:], f["Q"][1:], rtol=1e-8, atol=1e-9)


# ------------------------------------------------------------------------------------------
# unit-root models: the information-form backward pass must not amplify rounding asymmetry
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["sparse16_poly2_blocks", "mfma16_dense_unit_root", "tiled_poly2_blocks", "sparse16_poly3"])
def test_unit_root_models_long_series(eng, kind):
    """polynomial trends have G with unit eigenvalues; with a dense W an expanded P update that assumes
    exact symmetry blows up within ~100 steps (found in round 1).  T = 400 here."""
    rng = np.random.default_rng({"sparse16_poly2_blocks": 1, "mfma16_dense_unit_root": 2, "tiled_poly2_blocks": 3,
                                 "sparse16_poly3": 4}[kind])
    T = 400
    if kind == "sparse16_poly2_blocks":
        mod = Dlm.polynomial(2)
        for _ in range(5):
            mod = mod + Dlm.polynomial(2)             # d = 12, p = 1, block bidiagonal G
        expect = "sparse16"
    elif kind == "sparse16_poly3":
        mod = Dlm.polynomial(3) + Dlm.seasonal(12, 2)  # d = 7
        expect = "sparse16"
    elif kind == "mfma16_dense_unit_root":
        Q_, _ = np.linalg.qr(rng.standard_normal((10, 10)))
        Gd = Q_ @ np.kron(np.eye(5), np.array([[1.0, 1.0], [0.0, 1.0]])) @ Q_.T   # dense, five unit-root Jordan blocks
        Fd = Q_ @ np.tile([1.0, 0.0], 5)
        mod = Dlm(lambda t: Fd.reshape(-1, 1), lambda dt: Gd)
        expect = "mfma16"
    else:
        mod = Dlm.polynomial(2)
        for _ in range(9):
            mod = mod * Dlm.polynomial(2)              # d = 20, p = 10
        expect = "wave-mfma"
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A = rng.standard_normal((d, d))
    p = DlmParameters(np.eye(q) * 1.3, A @ A.T / d + 0.1 * np.eye(d), np.zeros(d), np.eye(d))
    y = rng.standard_normal((2, T, q)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.05] = np.nan
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == eng.expect(expect) and np.all(out["status"] == 0)
    for n in range(2):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); sm, S = split(out["smooth"][n], d)
        scale = max(1.0, np.abs(f["C"]).max())
        np.testing.assert_allclose(m, f["m"], rtol=1e-7, atol=1e-8 * scale)
        np.testing.assert_allclose(C, f["C"], rtol=1e-7, atol=1e-8 * scale)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-6, atol=1e-7 * scale)
        np.testing.assert_allclose(S, s["S"], rtol=1e-6, atol=1e-7 * scale)


# ------------------------------------------------------------------------------------------
# Gibbs samplers end to end on the engine (GibbsSampling.sample / GibbsWishart.sample)
# ------------------------------------------------------------------------------------------
def test_gibbs_dinvgamma_recovers_variances(eng):
    """Pooled d-Inverse-Gamma Gibbs on 200 simulated local-linear-trend series: posterior means of V and
    W land near the truth (a statistical end-to-end check of FFBS + statistics + conjugate draws)."""
    from bayesian_dlms_amd.gibbs import GibbsSampling, InverseGamma
    mod = Dlm.polynomial(2)
    times = np.arange(1, 151, dtype=np.float64)
    mat = materialise(mod, times)
    truth = DlmParameters([[2.0]], np.diag([0.5, 0.05]), [0.0, 0.0], np.eye(2))
    y = simulate(mat, truth, 200, seed=101)
    init = DlmParameters([[1.0]], np.diag([1.0, 1.0]), [0.0, 0.0], np.eye(2) * 10)
    for sim in (False, True):
        chain = list(GibbsSampling.sample(mod, InverseGamma(3.0, 3.0), InverseGamma(3.0, 1.0), init, times, y, eng,
                                          n_iter=60, seed=5, pooled=True, simulation_smoother=sim))
        v = np.mean([s.p.v[0, 0] for s in chain[20:]]); w = np.mean([np.diag(s.p.w) for s in chain[20:]], axis=0)
        assert abs(v - 2.0) < 0.15 and abs(w[0] - 0.5) < 0.08 and abs(w[1] - 0.05) < 0.02, (sim, v, w)


def test_gibbs_per_series_engine_matches_oracle_ffbs(eng):
    """Per-series Gibbs through the engine equals the same chain driven by the oracle's FFBS."""
    from bayesian_dlms_amd.gibbs import GibbsSampling, InverseGamma
    from test_host_logic import oracle_ffbs
    mod = Dlm.polynomial(2)
    times = np.arange(1, 41, dtype=np.float64)
    mat = materialise(mod, times)
    p0 = DlmParameters([[2.0]], np.diag([0.5, 0.2]), [0.0, 0.0], np.eye(2) * 10)
    y = simulate(mat, p0, 5, seed=3, missing=0.1)
    a = list(GibbsSampling.sample(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p0, times, y, eng, n_iter=3, seed=9,
                                  simulation_smoother=False))   # the literal backward sampler, draw for draw
    b = list(GibbsSampling.sample(mod, InverseGamma(5.0, 4.0), InverseGamma(17.0, 4.0), p0, times, y, None, n_iter=3, seed=9,
                                  ffbs=oracle_ffbs))
    for sa, sb in zip(a, b):
        for k in range(5):
            np.testing.assert_allclose(sa.p[k].v, sb.p[k].v, rtol=1e-6)
            np.testing.assert_allclose(sa.p[k].w, sb.p[k].w, rtol=1e-6)


def test_gibbs_wishart_engine(eng):
    from bayesian_dlms_amd.gibbs import GibbsWishart, InverseGamma, InverseWishart
    mod = Dlm.polynomial(1) * Dlm.polynomial(1)
    times = np.arange(1, 101, dtype=np.float64)
    mat = materialise(mod, times)
    Wt = np.array([[0.5, 0.3], [0.3, 0.4]])
    truth = DlmParameters(np.eye(2), Wt, [0.0, 0.0], np.eye(2))
    y = simulate(mat, truth, 100, seed=7)
    init = DlmParameters(np.eye(2), np.eye(2), [0.0, 0.0], np.eye(2) * 10)
    chain = list(GibbsWishart.sample(mod, InverseGamma(3.0, 3.0), InverseWishart(4.0, np.eye(2)), init, times, y, eng,
                                     n_iter=50, seed=2, pooled=True))
    W = np.mean([s.p.w for s in chain[15:]], axis=0)
    np.testing.assert_allclose(W, Wt, atol=0.08)


def test_rccl_allreduce_single_rank_and_stats_pool(eng):
    """The C-ABI RCCL wrapper on a 1-rank communicator (the multi-rank path is the same call)."""
    import torch
    from bayesian_dlms_amd.engine import Engine
    e2 = Engine(0)
    e2.comm_init_rank(1, 0, e2.comm_unique_id())
    t = torch.arange(16, dtype=torch.float64, device="cuda:0")
    e2.allreduce_stats(t)
    np.testing.assert_array_equal(t.cpu().numpy(), np.arange(16.0))
    e2.close()


# ------------------------------------------------------------------------------------------
# edge cases
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flags", VARIANTS)
def test_edge_shapes(eng, flags):
    """T = 1, N = 1, d = 1; every observation missing; a series that starts with missing data."""
    p = DlmParameters([[2.0]], [[3.0]], [1.0], [[10.0]])
    mat = materialise(Dlm.polynomial(1), [5.0])
    out = eng.filter_smooth(mat, p, np.array([[[4.0]]]), flags=flags)
    f, s = oracle_filter_smooth(mat, p, np.array([[4.0]]))
    np.testing.assert_allclose(out["filt"][0], np.concatenate([f["m"], f["C"]], axis=1), rtol=1e-13)
    np.testing.assert_allclose(out["smooth"][0], np.concatenate([s["s"], s["S"]], axis=1), rtol=1e-12)
    mat = materialise(Dlm.polynomial(1), np.arange(1, 9, dtype=np.float64))
    y = np.full((2, 8, 1), np.nan); y[1, 3:, 0] = [1.0, 2.0, np.nan, 2.5, 3.0]
    out = eng.filter_smooth(mat, p, y, flags=flags)
    assert np.all(out["status"] == 0)
    for n in range(2):
        f, s = oracle_filter_smooth(mat, p, y[n])
        np.testing.assert_allclose(out["filt"][n], np.concatenate([f["m"], f["C"]], axis=1), rtol=1e-12)
        np.testing.assert_allclose(out["smooth"][n], np.concatenate([s["s"], s["S"]], axis=1), rtol=1e-11)
    np.testing.assert_allclose(out["filt"][0][:, 1], 10.0 + 3.0 * np.arange(9))   # nothing observed: C_t = C0 + t W


def test_status_flags_nonfinite_and_not_pd(eng):
    mod, mat, p = seasonal_model(T=20)
    y = simulate(mat, p, 3, seed=1)
    y[1, 5, 0] = np.inf                               # a non-finite observation poisons only its own series
    out = eng.filter_smooth(mat, p, y)
    st = out["status"]
    assert st[0] == 0 and st[2] == 0 and (st[1] & _lib.ST_NONFINITE)
    bad = DlmParameters([[0.0]], p.w, p.m0, p.c0)     # V = 0: the information-form backward pass needs V > 0
    out = eng.filter_smooth(mat, bad, y[:1])
    assert out["status"][0] & _lib.ST_NOT_PD
    ok = eng.filter_smooth(mat, bad, y[:1], flags=_lib.OPT_FORCE_GENERIC)   # the generic RTS path handles V = 0
    assert ok["status"][0] == 0 and np.all(np.isfinite(ok["smooth"]))


def dk_reference_draw_mv(mat, p, y, z):
    """Multivariate version of dk_reference_draw: z [T+1][d+p] (state noise, then observation noise)."""
    d, q, T = mat.d, mat.p, mat.T
    G = oracle.from_cm(mat.G[: d * d], d, d); F = oracle.from_cm(mat.F[: d * q], d, q)
    Lc, Lv = np.linalg.cholesky(p.c0), np.linalg.cholesky(p.v)
    Lws = [np.linalg.cholesky(w) for w in p.w] if p.w.ndim == 3 else [np.linalg.cholesky(p.w)] * T     # (a W_t stream: W_t drives the transition into record t)
    x = p.m0 + Lc @ z[0, :d]
    xs, yp = [x], np.empty((T, q))
    for t in range(1, T + 1):
        x = G @ x + Lws[t - 1] @ z[t, :d]
        xs.append(x)
        yp[t - 1] = F.T @ x + Lv @ z[t, d:]
    om = omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, np.zeros(d), p.c0, y - yp)
    s = oracle.smoother(om, f, compat_q1=False)
    return s["s"] + np.array(xs)


def test_ffbs_simulation_smoother_multivariate(eng):
    """DLM_OPT_FFBS_SIMSMOOTH on the tiled path (d = 20, p = 10, dense W, correlated V, missing data)."""
    rng = np.random.default_rng(77)
    mod = Dlm.polynomial(2)
    for _ in range(9):
        mod = mod * Dlm.polynomial(2)
    T, N, d, q = 60, 3, 20, 10
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    A = rng.standard_normal((d, d)); B = rng.standard_normal((q, q))
    p = DlmParameters(B @ B.T / q + 0.5 * np.eye(q), A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2.0)
    y = rng.standard_normal((N, T, q)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    z = rng.standard_normal((N, T + 1, d + q))
    om = omodel(mat)
    for flags in (_lib.OPT_FFBS_SIMSMOOTH, _lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_STATS_OUTER):
        out = eng.ffbs(mat, p, y, z=z, flags=flags)
        assert eng.last_variant == eng.expect("wave-simsmooth") and np.all(out["status"] == 0)   # structured G: dlm_wave48.hip
        old = eng.ffbs(mat, p, y, z=z, flags=flags | _lib.OPT_NO_WAVE)
        assert eng.last_variant == "tiled-simsmooth"
        np.testing.assert_allclose(out["theta"], old["theta"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(out["stats"], old["stats"], rtol=1e-7, atol=1e-8)
        for n in range(N):
            ref = dk_reference_draw_mv(mat, p, y[n], z[n])
            np.testing.assert_allclose(out["theta"][n], ref, rtol=1e-6, atol=1e-7)
            st = oracle.gibbs_stats(om, y[n], ref, want_outer=bool(flags & _lib.OPT_STATS_OUTER))
            np.testing.assert_allclose(out["stats"][n, :q], st["ssy"], rtol=1e-6)
            np.testing.assert_array_equal(out["stats"][n, q:2 * q], st["n"])
            if flags & _lib.OPT_STATS_OUTER:
                np.testing.assert_allclose(out["stats"][n, 2 * q:2 * q + d * d], st["outer"], rtol=1e-5, atol=1e-6)
            else:
                np.testing.assert_allclose(out["stats"][n, 2 * q:2 * q + d], st["ss"], rtol=1e-6)
    out = eng.ffbs(mat, p, y, seed=11, series_offset=4, flags=_lib.OPT_FFBS_SIMSMOOTH)
    zz = oracle.normals(11, 4 + 2, T + 1, d + q)
    np.testing.assert_allclose(out["theta"][2], dk_reference_draw_mv(mat, p, y[2], zz), rtol=1e-6, atol=1e-7)
    # irregular grid with a repeated time (several G tables, W dt, the dt = 0 identity advance): both implementations
    mat2 = materialise(mod, np.cumsum(np.array([1, 1, 2, 0, 3, 1] * 8, dtype=np.float64)) + 1.0)
    y2 = rng.standard_normal((N, mat2.T, q)).cumsum(axis=1)
    y2[rng.random(y2.shape) < 0.1] = np.nan
    new2 = eng.ffbs(mat2, p, y2, seed=5, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_STATS_OUTER)
    assert eng.last_variant == eng.expect("wave-simsmooth")
    old2 = eng.ffbs(mat2, p, y2, seed=5, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_STATS_OUTER | _lib.OPT_NO_WAVE)
    np.testing.assert_allclose(new2["theta"], old2["theta"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(new2["stats"], old2["stats"], rtol=1e-7, atol=1e-8)
    # a dense F next to a structured G (MFMA products with F on the per-wave path)
    Fd = rng.standard_normal((20, 6)); Gd = 0.85 * np.eye(20) + 0.1 * np.eye(20, k=1)
    mat3 = materialise(Dlm(lambda t: Fd, lambda dt: Gd), np.arange(1, 41, dtype=np.float64))
    p3 = DlmParameters(np.eye(6) * 0.8, A @ A.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d))
    y3 = rng.standard_normal((N, 40, 6)).cumsum(axis=1)
    y3[rng.random(y3.shape) < 0.1] = np.nan
    new3 = eng.ffbs(mat3, p3, y3, seed=8, flags=_lib.OPT_FFBS_SIMSMOOTH)
    assert eng.last_variant == eng.expect("wave-simsmooth")
    old3 = eng.ffbs(mat3, p3, y3, seed=8, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_NO_WAVE)
    np.testing.assert_allclose(new3["theta"], old3["theta"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(new3["stats"], old3["stats"], rtol=1e-7, atol=1e-8)


# ------------------------------------------------------------------------------------------
# irregular time grids on the structured fast path (several G(dt) tables, W dt, dt == 0)
# ------------------------------------------------------------------------------------------
def test_irregular_grid_structured_fast_path(eng):
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    gaps = np.array([1, 1, 2, 1, 5, 1, 1, 3, 1, 1, 24, 1, 2, 1, 1] * 8, dtype=np.float64)
    times = np.cumsum(gaps)
    mat = materialise(mod, times)
    assert mat.n_g > 2 and mat.dt is not None and mat.g_index is not None
    w = np.diag([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
    p = DlmParameters([[1.0]], w, np.zeros(13), np.eye(13))
    rng = np.random.default_rng(31)
    y = rng.standard_normal((4, mat.T, 1)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == "sparse16" and np.all(out["status"] == 0)
    for n in range(4):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], 13); sm, S = split(out["smooth"][n], 13)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
    # the simulation smoother on the same grid
    z = rng.standard_normal((4, mat.T + 1, 14))
    o2 = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH)
    assert eng.last_variant == "sparse16-simsmooth"
    om = omodel(mat)
    G_by_step = [oracle.from_cm(mat.G[gi * 169:(gi + 1) * 169], 13, 13) for gi in mat.g_index]
    for n in range(2):
        x = p.m0 + np.linalg.cholesky(p.c0) @ z[n, 0, :13]
        xs, yp = [x], np.empty((mat.T, 1))
        for t in range(1, mat.T + 1):
            x = G_by_step[t - 1] @ x + np.sqrt(np.diag(w) * mat.dt[t - 1]) * z[n, t, :13]
            xs.append(x); yp[t - 1, 0] = mat.F[:13] @ x + z[n, t, 13]
        f = oracle.kf_filter(om, p.v, p.w, np.zeros(13), p.c0, y[n] - yp)
        ref = oracle.smoother(om, f)["s"] + np.array(xs)
        np.testing.assert_allclose(o2["theta"][n], ref, rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], ref)
        np.testing.assert_allclose(o2["stats"][n, 2:15], st["ss"], rtol=1e-7)


def test_zero_time_increment(eng):
    """Two observations at the same time: dt == 0 means no advance (KalmanFilter.scala:279-280)."""
    for mod, d in ((Dlm.polynomial(2), 2), (Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
                                             * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2), 16)):
        times = np.array([1.0, 2.0, 2.0, 3.0, 5.0, 5.0, 6.0])
        mat = materialise(mod, times)
        q = mat.p
        rng = np.random.default_rng(d)
        p = DlmParameters(np.eye(q) * 0.7, np.eye(d) * 0.3, np.zeros(d), np.eye(d) * 2)
        y = rng.standard_normal((2, 7, q))
        for flags in (0, _lib.OPT_FORCE_GENERIC):
            out = eng.filter_smooth(mat, p, y, flags=flags)
            for n in range(2):
                f, s = oracle_filter_smooth(mat, p, y[n])
                np.testing.assert_allclose(split(out["filt"][n], d)[1], f["C"], rtol=1e-10, atol=1e-11)
                np.testing.assert_allclose(split(out["smooth"][n], d)[0], s["s"], rtol=1e-9, atol=1e-10)
                np.testing.assert_allclose(split(out["smooth"][n], d)[1], s["S"], rtol=1e-9, atol=1e-10)


def test_prior_records_on_fast_paths(eng):
    """(a_t, R_t) -- KfState.at / rt -- from the structured and the tiled forward kernels."""
    mod, mat, p = seasonal_model(T=50)
    y = simulate(mat, p, 2, seed=6, missing=0.1)
    out = eng.filter(mat, p, y, want_prior=True, want_fq=True)
    assert eng.last_variant == "sparse16"
    f = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[1])
    a, R = split(out["prior"][1], 13)
    np.testing.assert_allclose(a, f["a"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(R, f["R"], rtol=1e-9, atol=1e-10)
    mod2 = Dlm.polynomial(2)
    for _ in range(7):
        mod2 = mod2 * Dlm.polynomial(2)
    mat2 = materialise(mod2, np.arange(1, 21, dtype=np.float64))
    p2 = DlmParameters(np.eye(8), np.eye(16) * 0.2, np.zeros(16), np.eye(16))
    y2 = np.random.default_rng(1).standard_normal((2, 20, 8))
    out2 = eng.filter(mat2, p2, y2, want_prior=True)
    assert eng.last_variant == eng.expect("wave-mfma")
    f2 = oracle.kf_filter(omodel(mat2), p2.v, p2.w, p2.m0, p2.c0, y2[0])
    a2, R2 = split(out2["prior"][0], 16)
    np.testing.assert_allclose(a2, f2["a"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(R2, f2["R"], rtol=1e-9, atol=1e-10)


def test_time_varying_f_regression_on_fast_path(eng):
    """Dlm.regression (Dlm.scala:159-169): F_t = [1, x_t], composed with a seasonal block; the structured
    d <= 15 kernels reload F every step."""
    T = 60
    rng = np.random.default_rng(77)
    x = [np.array([v]) for v in rng.standard_normal(T)]
    mod = Dlm.regression(x) + Dlm.seasonal(12, 2)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d = mat.d
    assert d == 6 and mat.f_stride == d
    p = DlmParameters([[0.7]], np.diag(rng.uniform(0.05, 0.3, d)), np.zeros(d), np.eye(d) * 2.0)
    y = rng.standard_normal((3, T, 1)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == "sparse16" and np.all(out["status"] == 0)
    for n in range(3):
        f, sm = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); s_, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(s_, sm["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, sm["S"], rtol=1e-8, atol=1e-9)
    # simulation smoother with the same time-varying F
    z = rng.standard_normal((3, T + 1, d + 1))
    o2 = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH)
    assert eng.last_variant == "sparse16-simsmooth"
    om = omodel(mat)
    G = oracle.from_cm(mat.G[:d * d], d, d)
    Ft = np.asarray(mat.F).reshape(T, d)
    for n in range(2):
        xx = p.m0 + np.linalg.cholesky(p.c0) @ z[n, 0, :d]
        xs, yp = [xx], np.empty((T, 1))
        for t in range(1, T + 1):
            xx = G @ xx + np.sqrt(np.diag(p.w)) * z[n, t, :d]
            xs.append(xx); yp[t - 1, 0] = Ft[t - 1] @ xx + np.sqrt(0.7) * z[n, t, d]
        f = oracle.kf_filter(om, p.v, p.w, np.zeros(d), p.c0, y[n] - yp)
        ref = oracle.smoother(om, f)["s"] + np.array(xs)
        np.testing.assert_allclose(o2["theta"][n], ref, rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], ref)
        np.testing.assert_allclose(o2["stats"][n, 0], st["ssy"][0], rtol=1e-7)
        np.testing.assert_allclose(o2["stats"][n, 2:2 + d], st["ss"], rtol=1e-7)


@pytest.mark.parametrize("case", ["sparse16_d13", "mfma16_dense_d6", "generic_d8_p4", "tiled_d17_p3", "per_series_params",
                                  "wave_d20_p10", "wave_d24_p18_dense_f"])
def test_log_likelihood_prediction_error_decomposition(eng, case):
    """dlm_loglik_batch = sum_t KalmanFilter.conditionalLikelihood(f_t, Q_t, y_t) (KalmanFilter.scala:138-153) on every
    forward variant, with missing observations; nothing but N numbers leaves the GPU."""
    rng = np.random.default_rng(abs(hash(case)) % 1000 if False else {"sparse16_d13": 1, "mfma16_dense_d6": 2, "generic_d8_p4": 3,
                                                                      "tiled_d17_p3": 4, "per_series_params": 5, "wave_d20_p10": 6,
                                                                      "wave_d24_p18_dense_f": 7}[case])
    N = 3
    if case in ("sparse16_d13", "per_series_params"):
        mod, mat, p = seasonal_model(T=80)
        expect = "sparse16"
    elif case == "mfma16_dense_d6":
        A = rng.standard_normal((6, 6)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((6, 1))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 61, dtype=np.float64))
        p = DlmParameters([[0.8]], np.eye(6) * 0.3, np.zeros(6), np.eye(6))
        expect = "mfma16"
    elif case == "generic_d8_p4":
        mod = Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
        mat = materialise(mod, np.cumsum(np.array([1, 2, 1, 1, 3] * 8, dtype=np.float64)))
        B = rng.standard_normal((4, 4))
        p = DlmParameters(B @ B.T / 4 + 0.5 * np.eye(4), np.eye(8) * 0.2, np.zeros(8), np.eye(8))
        expect = "wave-mfma"   # d <= 15 with p > 1 and a structured G: the per-wave kernels, one tile per dimension
    elif case == "wave_d20_p10":   # structured G and F on an irregular grid: the per-wave kernel, determinant by a one-wave Cholesky
        mod = Dlm.polynomial(2)
        for _ in range(9):
            mod = mod * Dlm.polynomial(2)
        mat = materialise(mod, np.cumsum(np.array([1, 2, 1, 0, 3] * 8, dtype=np.float64)) + 1.0)
        B = rng.standard_normal((10, 10)); A2 = rng.standard_normal((20, 20))
        p = DlmParameters(B @ B.T / 10 + 0.5 * np.eye(10), A2 @ A2.T / 20 + 0.1 * np.eye(20), rng.standard_normal(20), np.eye(20))
        expect = "wave-mfma"
    elif case == "wave_d24_p18_dense_f":   # two-tile Q (p > 16) with a dense F
        d, q = 24, 18
        G1 = 0.8 * np.eye(d) + 0.1 * np.eye(d, k=1)
        F = rng.standard_normal((d, q))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 31, dtype=np.float64))
        B = rng.standard_normal((q, q))
        p = DlmParameters(B @ B.T / q + 0.5 * np.eye(q), np.eye(d) * 0.3, rng.standard_normal(d), np.eye(d))
        expect = "wave-mfma"
    else:
        d, q = 17, 3
        A = rng.standard_normal((d, d)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((d, q))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 41, dtype=np.float64))
        B = rng.standard_normal((q, q)); A2 = rng.standard_normal((d, d))
        p = DlmParameters(B @ B.T / q + 0.5 * np.eye(q), A2 @ A2.T / d + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d))
        expect = "tiled-mfma"
    y = rng.standard_normal((N, mat.T, mat.p)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.15] = np.nan
    y[:, 7, :] = np.nan
    params = p
    if case == "per_series_params":   # a bank of parameter sets in one launch
        params = [DlmParameters(p.v * s, p.w * s, p.m0, p.c0) for s in (0.5, 1.0, 2.0)]
    out = eng.loglik(mat, params, y)
    assert eng.last_variant == eng.expect(expect) and np.all(out["status"] == 0)
    for n in range(N):
        pn = params[n] if isinstance(params, list) else params
        f = oracle.kf_filter(omodel(mat), pn.v, pn.w, pn.m0, pn.c0, y[n])
        np.testing.assert_allclose(out["loglik"][n], oracle.loglik(omodel(mat), f, y[n]), rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("case", ["sparse16_d13", "dense_d6_to_generic", "generic_d8_p4", "tiled_d17_p3_dense", "tiled_d20_p10_structured"])
def test_time_varying_variance_streams(eng, case):
    """V_t / W_t streams (StudentT.filter, StudentTGibbs.scala:100-136; DlmFsvSystem.ffbs,
    DlmFsvSystem.scala:137-208): filter, smoother, log-likelihood and the literal backward sampler against the oracle."""
    seed = {"sparse16_d13": 1, "dense_d6_to_generic": 2, "generic_d8_p4": 3, "tiled_d17_p3_dense": 4, "tiled_d20_p10_structured": 5}[case]
    rng = np.random.default_rng(100 + seed)
    if case == "sparse16_d13":
        mod, mat, p0 = seasonal_model(T=60)
        expect = "sparse16"
    elif case == "dense_d6_to_generic":
        A = rng.standard_normal((6, 6)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((6, 1))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 41, dtype=np.float64))
        p0 = DlmParameters([[0.8]], np.eye(6) * 0.3, np.zeros(6), np.eye(6))
        expect = "generic"          # the dense-G MFMA kernels take time-invariant variances only
    elif case == "generic_d8_p4":
        mod = Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
        mat = materialise(mod, np.cumsum(np.array([1, 2, 1, 1, 3] * 6, dtype=np.float64)))
        p0 = DlmParameters(np.eye(4), np.eye(8) * 0.2, np.zeros(8), np.eye(8))
        expect = "wave-mfma"        # structured G, p > 1: the per-wave kernels reload V_t / W_t every step
    elif case == "tiled_d17_p3_dense":
        d, q = 17, 3
        A = rng.standard_normal((d, d)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((d, q))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 31, dtype=np.float64))
        p0 = DlmParameters(np.eye(q), np.eye(d) * 0.3, rng.standard_normal(d), np.eye(d))
        expect = "tiled-mfma"
    else:
        mod = Dlm.polynomial(2)
        for _ in range(9):
            mod = mod * Dlm.polynomial(2)
        mat = materialise(mod, np.arange(1, 31, dtype=np.float64))
        p0 = DlmParameters(np.eye(10), np.eye(20) * 0.2, np.zeros(20), np.eye(20))
        expect = "wave-mfma"
    d, q, T = mat.d, mat.p, mat.T
    # SPD streams: V_t = s_t (B B^T / q + I/2), W_t = diag scale + a dense SPD part
    B = rng.standard_normal((q, q)); Vb = B @ B.T / q + 0.5 * np.eye(q)
    A2 = rng.standard_normal((d, d)); Wb = A2 @ A2.T / (10 * d)
    Vs = np.stack([Vb * rng.uniform(0.3, 3.0) for _ in range(T)])
    Ws = np.stack([np.diag(np.diag(p0.w) * rng.uniform(0.2, 4.0, d)) + Wb * rng.uniform(0.0, 1.0) for _ in range(T)])
    if case == "sparse16_d13":
        Ws = np.stack([np.diag(np.diag(p0.w) * rng.uniform(0.2, 4.0, d)) for _ in range(T)])
    p = DlmParameters(Vs, Ws, p0.m0, p0.c0)
    N = 2
    y = rng.standard_normal((N, T, q)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == eng.expect(expect) and np.all(out["status"] == 0)
    ll = eng.loglik(mat, p, y)["loglik"]
    z = rng.standard_normal((N, T + 1, d))
    draws = eng.ffbs(mat, p, y, z=z)
    for n in range(N):
        f = oracle.kf_filter(omodel(mat), Vs, Ws, p.m0, p.c0, y[n])
        sm = oracle.smoother(omodel(mat), f)
        m, C = split(out["filt"][n], d); s_, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(C, f["C"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(s_, sm["s"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(S, sm["S"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(ll[n], oracle.loglik(omodel(mat), f, y[n]), rtol=1e-9, atol=1e-8)
        ref = oracle.backward_sample(omodel(mat), Ws, f, z[n], factor="chol")
        np.testing.assert_allclose(draws["theta"][n], ref["theta"], rtol=1e-6, atol=1e-7)
    if d <= 15 and q == 1 and expect == "sparse16":   # the simulation smoother takes the streams on the structured d <= 15, p = 1 path (round 4) ...
        assert np.all(eng.ffbs(mat, p, y, flags=_lib.OPT_FFBS_SIMSMOOTH)["status"] == 0) and eng.last_variant == "sparse16-simsmooth"
    else:                                             # ... and refuses them elsewhere (the reference-form sampler above serves those callers)
        with pytest.raises(EngineError):
            eng.ffbs(mat, p, y, flags=_lib.OPT_FFBS_SIMSMOOTH)
    if d <= 16 and q <= 16:   # the SVD entry points take the streams too (round 2): equal to the standard filter
        sv = eng.svd_filter(mat, p, y)
        assert np.all(sv["status"] == 0)
        if q == 1:   # (with several components the SVD update masks sqrt(V)^-1 by selection -- valid for diagonal V only)
            msv, Csv = _svd_cov(sv["svd"][0], d)
            f0 = oracle.kf_filter(omodel(mat), Vs, Ws, p.m0, p.c0, y[0])
            np.testing.assert_allclose(msv, f0["m"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(Csv.transpose(0, 2, 1).reshape(mat.T + 1, d * d), f0["C"], rtol=1e-6, atol=1e-7)


def test_ar1_lane_per_series_golden_and_draws(eng, golden_dir):
    """dlm_ar1_ffbs_batch (FilterAr.scala:15-82): the reference's ar_dlm_filtered.csv, per-series
    parameters and variance streams with missing values, draws under injected and Philox normals."""
    from bayesian_dlms_amd.api import FilterAr, SvParameters
    obs = np.loadtxt(os.path.join(golden_dir, "ar_dlm.csv"), delimiter=",", skiprows=1)
    want = np.loadtxt(os.path.join(golden_dir, "ar_dlm_filtered.csv"), delimiter=",", skiprows=1)
    m, c = FilterAr.filter_univariate(obs[:, 1], 0.5, SvParameters(0.8, 1.0, 0.3), eng)
    assert eng.last_variant == "ar1-lane" and m.shape == (1, 5001)
    np.testing.assert_allclose(m[0], want[:, 1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(c[0], want[:, 2], rtol=0, atol=1e-13)
    # a batch: per-series (phi, mu, sigma), per-series variance streams, missing observations, more series than a block
    rng = np.random.default_rng(8)
    N, T = 300, 97
    sv = np.stack([rng.uniform(-0.95, 0.95, N), rng.standard_normal(N), rng.uniform(0.1, 1.0, N)], axis=1)
    v = rng.uniform(0.2, 2.0, (N, T))
    y = rng.standard_normal((N, T)).cumsum(axis=1) * 0.3
    y[rng.random(y.shape) < 0.1] = np.nan
    z = rng.standard_normal((N, T + 1))
    out = eng.ar1_ffbs(y, v, sv, z=z)
    assert np.all(out["status"] == 0)
    for n in (0, 63, 64, 255, 256, 299):
        f = oracle.ar1_filter(y[n], v[n], *sv[n])
        np.testing.assert_allclose(out["filt"][n, :, 0], f["m"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(out["filt"][n, :, 1], f["c"], rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(out["theta"][n], oracle.ar1_backward_sample(f, sv[n, 0], z[n]), rtol=1e-11, atol=1e-12)
    # Philox stream (seed, series_offset + n, t, 0), identical for a shard starting at series 100
    o1 = eng.ar1_ffbs(y, v, sv, seed=5, want_filt=False)
    o2 = eng.ar1_ffbs(y[100:], v[100:], sv[100:], seed=5, series_offset=100, want_filt=False)
    np.testing.assert_array_equal(o1["theta"][100:], o2["theta"])
    zz = oracle.normals(5, 7, T + 1, 1).reshape(-1)
    f7 = oracle.ar1_filter(y[7], v[7], *sv[7])
    np.testing.assert_allclose(o1["theta"][7], oracle.ar1_backward_sample(f7, sv[7, 0], zz), rtol=1e-11, atol=1e-12)
    # a non-stationary phi is flagged, not fatal
    bad = eng.ar1_ffbs(y[:2], 0.5, np.array([[1.2, 0.0, 0.3], [0.5, 0.0, 0.3]]))
    assert bad["status"][0] & _lib.ST_NOT_PD and bad["status"][1] == 0


def test_ou_lane_per_series_irregular_grid(eng):
    """dlm_ou_ffbs_batch (FilterOu.scala:7-79): irregular grid, per-series parameters, missing values, draws."""
    rng = np.random.default_rng(9)
    N, T = 130, 61
    times = np.cumsum(rng.choice([0.5, 1.0, 2.5, 4.0], T))
    sv = np.stack([rng.uniform(0.05, 1.5, N), rng.standard_normal(N), rng.uniform(0.1, 1.0, N)], axis=1)
    v = rng.uniform(0.2, 2.0, T)                      # one shared variance stream
    y = rng.standard_normal((N, T)).cumsum(axis=1) * 0.3
    y[rng.random(y.shape) < 0.1] = np.nan
    z = rng.standard_normal((N, T + 1))
    out = eng.ar1_ffbs(y, v, sv, z=z, times=times)
    assert eng.last_variant == "ou-lane" and np.all(out["status"] == 0)
    for n in (0, 64, 129):
        f = oracle.ou_filter(times, y[n], v, *sv[n])
        np.testing.assert_allclose(out["filt"][n, :, 0], f["m"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(out["filt"][n, :, 1], f["c"], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(out["theta"][n], oracle.ou_backward_sample(times, f, sv[n, 0], z[n]), rtol=1e-10, atol=1e-11)
    assert out["filt"][5, 0, 1] == pytest.approx(sv[5, 2] ** 2)   # the literal c0 = sigma * sigma / phi * phi


def test_device_dinvgamma_step_stream_distribution_and_gibbs(eng):
    """dlm_dinvgamma_step_batch (Gibbs.scala:23-78): (i) equals the oracle's restatement of the same Philox /
    Marsaglia-Tsang stream, (ii) the draws follow InverseGamma(shape, scale) (KS test against scipy), (iii) the
    device-resident Gibbs loop recovers the variances of a local-level model."""
    import torch
    from scipy import stats as sps
    from bayesian_dlms_amd.gibbs import InverseGamma, gibbs_dinvgamma_device
    rng = np.random.default_rng(12)
    d, p_, N = 3, 2, 4000
    st = np.empty((N, 2 * p_ + d + 1))
    st[:, :p_] = rng.uniform(5, 50, (N, p_)); st[:, p_:2 * p_] = rng.integers(10, 60, (N, p_))
    st[:, 2 * p_:2 * p_ + d] = rng.uniform(1, 30, (N, d)); st[:, -1] = 55
    V, W = eng.dinvgamma_step(d, p_, st, (2.0, 3.0), (0.4, 0.5), iteration=7, seed=11, series_offset=100)
    assert eng.last_variant == "dinvgamma-step"
    V = np.asarray(V).reshape(N, p_, p_); W = np.asarray(W).reshape(N, d, d)
    assert np.all(V[:, 0, 1] == 0) and np.all(W[:, 2, 0] == 0)
    for n in (0, 17, 3999):
        v, w = oracle.dinvgamma_step(d, p_, st[n], 2.0, 3.0, 0.4, 0.5, 11, 100 + n, 7)
        np.testing.assert_allclose(np.diag(V[n]), v, rtol=1e-12)
        np.testing.assert_allclose(np.diag(W[n]), w, rtol=1e-12)
    # distribution: identical statistics for every series -> N iid draws per component
    st2 = np.tile(st[0], (N, 1))
    V2, W2 = eng.dinvgamma_step(d, p_, st2, (2.0, 3.0), (0.4, 0.5), iteration=1, seed=3)
    V2 = np.asarray(V2).reshape(N, p_, p_); W2 = np.asarray(W2).reshape(N, d, d)
    ks_v = sps.kstest(V2[:, 1, 1], sps.invgamma(2.0 + 0.5 * st[0, p_ + 1], scale=3.0 + 0.5 * st[0, 1]).cdf)
    ks_w = sps.kstest(W2[:, 0, 0], sps.invgamma(0.4 + 0.5 * 55, scale=0.5 + 0.5 * st[0, 2 * p_]).cdf)
    assert ks_v.pvalue > 1e-3 and ks_w.pvalue > 1e-3
    small = eng.dinvgamma_step(1, 1, np.array([[0.2, 0.0, 0.3, 0.0]] * N), (0.3, 1.0), (0.3, 1.0), iteration=0, seed=5)   # shape < 1 boost
    assert sps.kstest(np.asarray(small[0]).reshape(-1), sps.invgamma(0.3, scale=1.1).cdf).pvalue > 1e-3
    # end to end on the device: local level, V = 2, W = 0.5
    T, Ns = 300, 64
    mod = Dlm.polynomial(1)
    x = np.cumsum(rng.standard_normal((Ns, T)) * np.sqrt(0.5), axis=1)
    yy = torch.as_tensor((x + rng.standard_normal((Ns, T)) * np.sqrt(2.0))[:, :, None], device="cuda")
    acc = []
    gibbs_dinvgamma_device(mod, InverseGamma(3.0, 4.0), InverseGamma(3.0, 1.0), DlmParameters([[1.0]], [[1.0]], [0.0], [[10.0]]),
                           np.arange(1, T + 1, dtype=np.float64), yy, eng, n_iter=150, seed=2,
                           on_iteration=lambda it, V_, W_, s_: acc.append((V_.mean().item(), W_.mean().item())) if it >= 50 else None)
    vm, wm = np.mean(acc, axis=0)
    assert 1.6 < vm < 2.5 and 0.3 < wm < 0.8


def test_forecast_is_the_filter_without_observations(eng):
    """api.forecast = Dlm.forecast / stepForecast (Dlm.scala:296-338): checked against the plain recursion
    a = G m, R = G C G^T + W, f = F^T a, Q = F^T R F + V for a batch of filtering distributions."""
    from bayesian_dlms_amd.api import forecast
    mod, mat, p = seasonal_model(T=30)
    rng = np.random.default_rng(4)
    y = simulate(mat, p, 3, seed=9)
    filt = eng.filter(mat, p, y)["filt"]
    mt, ct = filt[:, -1, :13], np.transpose(filt[:, -1, 13:].reshape(3, 13, 13), (0, 2, 1))
    times, f, Q = forecast(mod, mt, ct, 30.0, p, eng, steps=12)
    G = oracle.from_cm(mat.G[:169], 13, 13); F = np.asarray(mat.F[:13]).reshape(13, 1)
    assert times[0] == 30.0 and times[-1] == 41.0
    for n in range(3):
        m, C = mt[n].copy(), ct[n].copy()
        for k in range(12):
            if k > 0:
                m, C = G @ m, G @ C @ G.T + p.w
            np.testing.assert_allclose(f[n, k], F.T @ m, rtol=1e-10, atol=1e-11)
            np.testing.assert_allclose(Q[n, k], F.T @ C @ F + p.v, rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("case", ["sparse16_d13", "sparse16_d14_irregular", "tiled_d17", "generic_d8_p4"])
def test_fused_call_without_filtered_output(eng, case):
    """dlm_filter_smooth_batch with filt = NULL: identical smoothed moments, the filtered records stay in an engine
    workspace (packed on the structured d <= 15 path; d = 14 exercises the 16-byte padding of a packed record)."""
    rng = np.random.default_rng(21)
    if case == "sparse16_d13":
        mod, mat, p = seasonal_model(T=70); expect = "sparse16"
    elif case == "sparse16_d14_irregular":
        mod = Dlm.polynomial(2) + Dlm.seasonal(12, 6)
        mat = materialise(mod, np.cumsum(np.array([1, 1, 3, 1, 2] * 10, dtype=np.float64)))
        p = DlmParameters([[0.9]], np.diag(rng.uniform(0.05, 0.4, 14)), np.zeros(14), np.eye(14)); expect = "sparse16"
    elif case == "tiled_d17":
        A = rng.standard_normal((17, 17)); G1 = 0.9 * A / np.abs(np.linalg.eigvals(A)).max()
        F = rng.standard_normal((17, 3))
        mat = materialise(Dlm(lambda t: F, lambda dt: G1), np.arange(1, 31, dtype=np.float64))
        p = DlmParameters(np.eye(3), np.eye(17) * 0.3, np.zeros(17), np.eye(17)); expect = "tiled-mfma"
    else:
        mod = Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
        mat = materialise(mod, np.arange(1, 31, dtype=np.float64))
        p = DlmParameters(np.eye(4), np.eye(8) * 0.2, np.zeros(8), np.eye(8)); expect = "wave-mfma"   # d <= 15 with p > 1: the per-wave kernels, one tile per dimension
    y = rng.standard_normal((5, mat.T, mat.p)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    full = eng.filter_smooth(mat, p, y)
    only = eng.filter_smooth(mat, p, y, want_filt=False)
    assert eng.last_variant == eng.expect(expect) and only["filt"] is None and np.all(only["status"] == 0)
    # not bit-identical on the packed path: it reads C_t as an exactly symmetric matrix from its lower triangle, the
    # dense records carry the two triangles as the forward pass rounded them
    np.testing.assert_allclose(only["smooth"], full["smooth"], rtol=1e-9, atol=1e-11)


def test_simulate_on_device_matches_oracle_and_model(eng):
    """dlm_simulate_batch (Dlm.simulateRegular, Dlm.scala:245-292): equals the oracle's restatement on the same Philox
    stream (irregular grid, dense W and V, p = 2) and has the model's moments."""
    rng = np.random.default_rng(6)
    mod = Dlm.polynomial(2) * Dlm.polynomial(1)
    mat = materialise(mod, np.cumsum(np.array([1, 2, 1, 0.5, 3] * 8, dtype=np.float64)))
    B = rng.standard_normal((2, 2)); A = rng.standard_normal((3, 3))
    p = DlmParameters(B @ B.T + np.eye(2), A @ A.T / 3 + 0.1 * np.eye(3), rng.standard_normal(3), np.eye(3) * 2.0)
    out = eng.simulate(mat, p, 4, seed=21, series_offset=10)
    assert eng.last_variant == "generic-simulate" and np.all(out["status"] == 0)
    for n in range(4):
        x, y = oracle.simulate(omodel(mat), p.v, p.w, p.m0, p.c0, 21, 10 + n)
        np.testing.assert_allclose(out["x"][n], x, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(out["y"][n], y, rtol=1e-12, atol=1e-12)
    # moments on a local level: Var(y_1) = C0 + W + V, Var(x_0) = C0
    ll = materialise(Dlm.polynomial(1), np.arange(1, 4, dtype=np.float64))
    big = eng.simulate(ll, DlmParameters([[2.0]], [[0.5]], [1.0], [[3.0]]), 20000, seed=3)
    assert abs(big["x"][:, 0, 0].mean() - 1.0) < 0.05 and abs(big["x"][:, 0, 0].var() - 3.0) < 0.1
    assert abs(big["y"][:, 0, 0].var() - 5.5) < 0.2


@pytest.mark.parametrize("shape", ["d16_p1", "d48_p24", "d33_p17"])
def test_per_wave_kernels_edge_shapes_and_status(eng, shape):
    """dlm_wave48.hip at the corners of its range: one observation component, the largest structured shape, tile counts that
    leave one row / column in the last tile; T = 1; a series without a single observation; status flags."""
    rng = np.random.default_rng({"d16_p1": 161, "d48_p24": 4824, "d33_p17": 3317}[shape])
    if shape == "d16_p1":
        mod = Dlm.polynomial(2) + Dlm.seasonal(12, 7)     # one observation component, eight nonzeros in F's column (dense-F products)
    elif shape == "d48_p24":
        mod = Dlm.polynomial(2)
        for _ in range(23):
            mod = mod * Dlm.polynomial(2)
    else:
        mod = Dlm.polynomial(1)
        for _ in range(16):
            mod = mod * Dlm.polynomial(2)
    times = np.arange(1, 13, dtype=np.float64)
    mat = materialise(mod, times)
    d, q = mat.d, mat.p
    assert (d, q) == {"d16_p1": (16, 1), "d48_p24": (48, 24), "d33_p17": (33, 17)}[shape]
    A = rng.standard_normal((d, d)); B = rng.standard_normal((q, q))
    p = DlmParameters(B @ B.T / q + 0.5 * np.eye(q), A @ A.T / d * 0.05 + 0.02 * np.eye(d), rng.standard_normal(d) * 0.1, np.eye(d) * 0.5)
    y = rng.standard_normal((3, mat.T, q)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.15] = np.nan
    y[2] = np.nan                                        # nothing observed at all
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == eng.expect("wave-mfma") and np.all(out["status"] == 0)
    ll = eng.loglik(mat, p, y)
    assert eng.last_variant == eng.expect("wave-mfma")
    for n in range(3):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); sm, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(C, f["C"], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(S, s["S"], rtol=1e-7, atol=1e-7)
        np.testing.assert_allclose(ll["loglik"][n], oracle.loglik(omodel(mat), oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n]), y[n]),
                                   rtol=1e-9, atol=1e-8)
    assert ll["loglik"][2] == 0.0
    # T = 1
    mat1 = materialise(mod, [3.0])
    o1 = eng.filter_smooth(mat1, p, y[:2, :1])
    assert eng.last_variant == eng.expect("wave-mfma")
    f, s = oracle_filter_smooth(mat1, p, y[0, :1])
    m, C = split(o1["filt"][0], d); sm, S = split(o1["smooth"][0], d)
    np.testing.assert_allclose(C, f["C"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(S, s["S"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(sm, s["s"], rtol=1e-7, atol=1e-7)
    # simulation-smoother draw: construction against the oracle with injected normals
    z = rng.standard_normal((2, mat.T + 1, d + q))
    o2 = eng.ffbs(mat, p, y[:2], z=z, flags=_lib.OPT_FFBS_SIMSMOOTH)
    assert eng.last_variant == eng.expect("wave-simsmooth") and np.all(o2["status"] == 0)
    for n in range(2):
        np.testing.assert_allclose(o2["theta"][n], dk_reference_draw_mv(mat, p, y[n], z[n]), rtol=1e-6, atol=1e-6)
    # status: a non-finite observation poisons only its own series; an indefinite V is reported
    yb = y.copy(); yb[1, 4, 0] = np.inf
    st = eng.filter_smooth(mat, p, yb)["status"]
    assert st[0] == 0 and st[2] == 0 and (st[1] & _lib.ST_NONFINITE)
    bad = DlmParameters(-np.eye(q), p.w, p.m0, p.c0)
    assert eng.filter_smooth(mat, bad, y[:1])["status"][0] & (_lib.ST_NOT_PD | _lib.ST_NONFINITE)


@pytest.mark.parametrize("kind", ["local_level", "linear_growth_irregular", "harmonic_plus_level_d3", "regression_d2_timevarying_f",
                                  "growth_plus_harmonic_d4", "level_plus_two_harmonics_d5"])
def test_lane_per_series_small_models(eng, kind):
    """d <= 5, p = 1 over thousands of series: one lane per series (dlm_lane.hip) -- filter with prior / forecast records,
    log-likelihood, the textbook RTS smoother (fused and standalone) against the oracle and against the wavefront-per-series path."""
    rng = np.random.default_rng({"local_level": 11, "linear_growth_irregular": 12, "harmonic_plus_level_d3": 13,
                                 "regression_d2_timevarying_f": 14, "growth_plus_harmonic_d4": 15, "level_plus_two_harmonics_d5": 16}[kind])
    N = 8192
    if kind == "local_level":
        mod = Dlm.polynomial(1); times = np.arange(1, 38, dtype=np.float64)
    elif kind == "linear_growth_irregular":
        mod = Dlm.polynomial(2); times = np.cumsum(np.array([1, 2, 1, 0, 3, 1, 1] * 5, dtype=np.float64)) + 1.0
    elif kind == "harmonic_plus_level_d3":
        mod = Dlm.polynomial(1) + Dlm.seasonal(12, 1); times = np.arange(1, 30, dtype=np.float64)
    elif kind == "growth_plus_harmonic_d4":
        mod = Dlm.polynomial(2) + Dlm.seasonal(12, 1); times = np.cumsum(np.array([1, 1, 2, 1, 0, 1] * 5, dtype=np.float64)) + 1.0
    elif kind == "level_plus_two_harmonics_d5":
        mod = Dlm.polynomial(1) + Dlm.seasonal(12, 2); times = np.arange(1, 28, dtype=np.float64)
    else:
        xs = rng.standard_normal(33)
        mod = Dlm(lambda t: np.array([[1.0], [xs[int(t) - 1]]]), lambda dt: np.eye(2)); times = np.arange(1, 34, dtype=np.float64)
    mat = materialise(mod, times)
    d, T = mat.d, mat.T
    A = rng.standard_normal((d, d))
    p = DlmParameters([[0.7]], A @ A.T / d * 0.3 + 0.1 * np.eye(d), rng.standard_normal(d), np.eye(d) * 2.0)
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    y[5] = np.nan
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == "lane" and np.all(out["status"] == 0)
    pf = eng.filter(mat, p, y, want_prior=True, want_fq=True)
    assert eng.last_variant == "lane"
    ll = eng.loglik(mat, p, y)
    assert eng.last_variant == "lane"
    sm2 = eng.smooth(mat, p, out["filt"])                 # standalone smoother on the same kernel
    assert eng.last_variant == "lane"
    np.testing.assert_array_equal(sm2["smooth"], out["smooth"])
    for n in (0, 5, 63, 64, 4097, N - 1):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); sm, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-9, atol=1e-9)
        a_, R_ = split(pf["prior"][n], d)
        np.testing.assert_allclose(a_[1:], f["a"][1:], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(R_[1:], f["R"][1:], rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(pf["fq"][n][1:, 0], np.asarray(f["f"])[1:].reshape(T), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(pf["fq"][n][1:, 1], np.asarray(f["Q"])[1:].reshape(T), rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(ll["loglik"][n], oracle.loglik(omodel(mat), f, y[n]), rtol=1e-10, atol=1e-9)
    # simulation-smoother FFBS + statistics on the lane kernels: injected normals against the reference construction
    # (regular grids) and, Philox stream, against the wavefront-per-series implementation of the same draw
    zs = rng.standard_normal((N, T + 1, d + 1))
    dz = eng.ffbs(mat, p, y, z=zs, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_STATS_OUTER)
    assert eng.last_variant == "lane-simsmooth" and np.all(dz["status"] == 0)
    if mat.dt is None and mat.f_stride == 0:
        for n in (0, 5, 64, N - 1):
            refd = dk_reference_draw_mv(mat, p, y[n], zs[n])
            np.testing.assert_allclose(dz["theta"][n], refd, rtol=1e-8, atol=1e-8)
            stt = oracle.gibbs_stats(omodel(mat), y[n], refd, want_outer=True)
            np.testing.assert_allclose(dz["stats"][n, 0], stt["ssy"][0], rtol=1e-8)
            np.testing.assert_array_equal(dz["stats"][n, 1], stt["n"][0])
            np.testing.assert_allclose(dz["stats"][n, 2:2 + d * d], stt["outer"], rtol=1e-7, atol=1e-8)
    # the literal backward sampler (Smoothing.sampleDlm) on lanes: draws, conditional moments and statistics equal the generic kernel's
    zl = rng.standard_normal((N, T + 1, d))
    lit = eng.ffbs(mat, p, y[:256], z=zl[:256], want_cond=True, flags=_lib.OPT_STATS_OUTER)
    assert eng.last_variant == "lane-sampler"
    gen = eng.ffbs(mat, p, y[:256], z=zl[:256], want_cond=True, flags=_lib.OPT_STATS_OUTER | _lib.OPT_FORCE_GENERIC)
    assert eng.last_variant == "generic"
    np.testing.assert_array_equal(lit["status"], gen["status"])
    if mat.dt is None or not np.any(np.asarray(mat.dt) == 0.0):
        # (a zero time increment makes the reference's conditional covariance H exactly singular when g(0) is not the identity:
        # both kernels flag DLM_ST_NOT_PD there and the draw along the null directions is undefined)
        assert np.all(lit["status"] == 0)
        np.testing.assert_allclose(lit["theta"], gen["theta"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(lit["cond"], gen["cond"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(lit["stats"], gen["stats"], rtol=1e-8, atol=1e-9)
        lp = eng.ffbs(mat, p, y[:256], seed=4, series_offset=7)
        gp = eng.ffbs(mat, p, y[:256], seed=4, series_offset=7, flags=_lib.OPT_FORCE_GENERIC)
        np.testing.assert_allclose(lp["theta"], gp["theta"], rtol=1e-9, atol=1e-9)
        for n in (0, 5, 255):   # and the oracle's literal sampler with the canonical (Cholesky) factor
            fo = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[n])
            np.testing.assert_allclose(lit["theta"][n], oracle.backward_sample(omodel(mat), p.w, fo, zl[n], factor="chol")["theta"], rtol=1e-8, atol=1e-8)
    dp = eng.ffbs(mat, p, y, seed=9, series_offset=3, flags=_lib.OPT_FFBS_SIMSMOOTH)
    ref = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_LANE)
    assert eng.last_variant != "lane"
    rp = eng.ffbs(mat, p, y, seed=9, series_offset=3, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_NO_LANE)
    assert eng.last_variant == "sparse16-simsmooth"
    np.testing.assert_allclose(out["filt"], ref["filt"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["smooth"], ref["smooth"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(dp["theta"], rp["theta"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(dp["stats"], rp["stats"], rtol=1e-7, atol=1e-8)


def test_simulation_smoother_with_scalar_variance_stream(eng):
    """The Student-t DLM's FFBS (StudentTGibbs.scala:100-136: V_t = V / lambda_t, p = 1) on the structured fast path with
    DLM_OPT_FFBS_SIMSMOOTH: the draw is the Durbin-Koopman construction with the per-step variances, statistics included;
    V_t on other paths is refused (W_t streams: test_ffbs_simulation_smoother_with_variance_streams)."""
    mod, mat, p = seasonal_model(T=60)
    d, T, N = 13, mat.T, 3
    rng = np.random.default_rng(808)
    Vt = rng.uniform(0.3, 3.0, (T, 1, 1))
    pt = DlmParameters(Vt, p.w, p.m0, p.c0)
    y = simulate(mat, p, N, seed=5, missing=0.1)
    z = rng.standard_normal((N, T + 1, d + 1))
    out = eng.ffbs(mat, pt, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH)
    assert eng.last_variant == "sparse16-simsmooth" and np.all(out["status"] == 0)
    om = omodel(mat)
    G = oracle.from_cm(mat.G[:d * d], d, d); F = np.asarray(mat.F[:d])
    Lc, Lw = np.linalg.cholesky(p.c0), np.linalg.cholesky(p.w)
    for n in range(N):
        x = p.m0 + Lc @ z[n, 0, :d]
        xs, yp = [x], np.empty((T, 1))
        for t in range(1, T + 1):
            x = G @ x + Lw @ z[n, t, :d]
            xs.append(x); yp[t - 1, 0] = F @ x + np.sqrt(Vt[t - 1, 0, 0]) * z[n, t, d]
        f = oracle.kf_filter(om, Vt, p.w, np.zeros(d), p.c0, y[n] - yp)
        ref = oracle.smoother(om, f)["s"] + np.array(xs)
        np.testing.assert_allclose(out["theta"][n], ref, rtol=1e-7, atol=1e-8)
        st = oracle.gibbs_stats(om, y[n], ref)
        np.testing.assert_allclose(out["stats"][n, 0], st["ssy"][0], rtol=1e-7)
        np.testing.assert_allclose(out["stats"][n, 2:2 + d], st["ss"], rtol=1e-7)
    # a W_t stream goes through too since round 4 (test_ffbs_simulation_smoother_with_variance_streams): the same W at every step is the
    # time-invariant call (the factor recomputed per step: the same numbers up to rounding)
    Wt = np.tile(p.w, (T, 1, 1))
    same_w = eng.ffbs(mat, DlmParameters(Vt, Wt, p.m0, p.c0), y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH)
    np.testing.assert_allclose(same_w["theta"], out["theta"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("shape", ["bivariate_local_level", "d6_p3_irregular", "d12_p4_harmonics"])
def test_small_multivariate_models_on_per_wave_kernels(eng, shape):
    """d <= 15 with several observation components (|*| of small models, Dlm.scala:197-208): the per-wave kernels with one tile per
    dimension -- filter + smoother, log-likelihood and the simulation-smoother draw against the oracle, and against the generic path."""
    rng = np.random.default_rng({"bivariate_local_level": 22, "d6_p3_irregular": 63, "d12_p4_harmonics": 124}[shape])
    if shape == "bivariate_local_level":
        mod = Dlm.polynomial(1) * Dlm.polynomial(1); times = np.arange(1, 41, dtype=np.float64)
    elif shape == "d6_p3_irregular":
        mod = Dlm.polynomial(2) * Dlm.polynomial(2) * Dlm.polynomial(2)
        times = np.cumsum(np.array([1, 2, 1, 0, 3, 1] * 6, dtype=np.float64)) + 1.0
    else:
        one = Dlm.polynomial(1) + Dlm.seasonal(12, 1)
        mod = one * one * one * one; times = np.arange(1, 31, dtype=np.float64)
    mat = materialise(mod, times)
    d, q, T = mat.d, mat.p, mat.T
    A = rng.standard_normal((d, d)); B = rng.standard_normal((q, q))
    p = DlmParameters(B @ B.T / q + 0.5 * np.eye(q), A @ A.T / d * 0.2 + 0.05 * np.eye(d), rng.standard_normal(d), np.eye(d) * 1.5)
    N = 5
    y = rng.standard_normal((N, T, q)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.15] = np.nan
    y[:, 3, :] = np.nan
    out = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == eng.expect("wave-mfma") and np.all(out["status"] == 0)
    ll = eng.loglik(mat, p, y)
    assert eng.last_variant == eng.expect("wave-mfma")
    gen = eng.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_GENERIC)
    assert eng.last_variant == "generic"
    np.testing.assert_allclose(out["filt"], gen["filt"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out["smooth"], gen["smooth"], rtol=1e-8, atol=1e-9)
    z = rng.standard_normal((N, T + 1, d + q))
    dr = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_STATS_OUTER)
    assert eng.last_variant == eng.expect("wave-simsmooth") and np.all(dr["status"] == 0)
    # the reference-form sampler on register tiles (dlm_sampler16.hip with the multivariate tables) against the generic kernel
    zl = rng.standard_normal((N, T + 1, d))
    lit = eng.ffbs(mat, p, y, z=zl, want_cond=True, flags=_lib.OPT_STATS_OUTER)
    assert eng.last_variant == "sparse16-sampler"
    gl = eng.ffbs(mat, p, y, z=zl, want_cond=True, flags=_lib.OPT_STATS_OUTER | _lib.OPT_FORCE_GENERIC)
    assert eng.last_variant == "generic"
    np.testing.assert_array_equal(lit["status"], gl["status"])
    if mat.dt is None:
        np.testing.assert_allclose(lit["theta"], gl["theta"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(lit["cond"], gl["cond"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(lit["stats"], gl["stats"], rtol=1e-8, atol=1e-9)
    om = omodel(mat)
    for n in range(N):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, C = split(out["filt"][n], d); sm, S = split(out["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(C, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(ll["loglik"][n], oracle.loglik(om, f, y[n]), rtol=1e-10, atol=1e-9)
        if mat.dt is None:   # (the reference construction below is written for a regular grid)
            ref = dk_reference_draw_mv(mat, p, y[n], z[n])
            np.testing.assert_allclose(dr["theta"][n], ref, rtol=1e-7, atol=1e-8)
            st = oracle.gibbs_stats(om, y[n], ref, want_outer=True)
            np.testing.assert_allclose(dr["stats"][n, :q], st["ssy"], rtol=1e-7)
            np.testing.assert_allclose(dr["stats"][n, 2 * q:2 * q + d * d], st["outer"], rtol=1e-6, atol=1e-7)


def test_reference_form_sampler_steady_state_reuse(eng):
    """A long regular series without missing values: once the filtered covariance has stopped moving the register-tile sampler
    reuses J, H and its factor (dlm_sampler16.hip); draws and conditional moments still equal the generic kernel's."""
    mod, mat, p = seasonal_model(T=900)
    y = simulate(mat, p, 3, seed=12)
    y[1, 300:310, 0] = np.nan                      # a gap in one series: its covariance moves again, then settles
    z = np.random.default_rng(5).standard_normal((3, 901, 13))
    out = eng.ffbs(mat, p, y, z=z, want_cond=True)
    assert eng.last_variant == "sparse16-sampler" and np.all(out["status"] == 0)
    gen = eng.ffbs(mat, p, y, z=z, want_cond=True, flags=_lib.OPT_NO_SAMPLER16)
    assert eng.last_variant == "generic"
    np.testing.assert_allclose(out["theta"], gen["theta"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["cond"], gen["cond"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(out["stats"], gen["stats"], rtol=1e-9)
    # the covariance of the undisturbed series is stationary over the second half: the conditional covariances repeat
    H = out["cond"][0][:, 13:]
    assert np.abs(H[700] - H[500]).max() <= 1e-12 * np.abs(H[500]).max()


@pytest.mark.parametrize("shape", ["seasonal_d13", "linear_growth_d2", "multivariate_d20_p10"])
def test_backward_sampling_from_existing_filter_records(eng, shape):
    """dlm_backward_sample_batch (Smoothing.sampleDlm on a stored filter, Smoothing.scala:151-180): the same draws, conditional
    moments and statistics as dlm_ffbs_batch, whichever sampler kernel the shape selects (register tiles, lanes, generic)."""
    rng = np.random.default_rng({"seasonal_d13": 1, "linear_growth_d2": 2, "multivariate_d20_p10": 3}[shape])
    if shape == "seasonal_d13":
        mod, mat, p = seasonal_model(T=60); expect = "sparse16-sampler"
    elif shape == "linear_growth_d2":
        mat = materialise(Dlm.polynomial(2), np.cumsum(np.array([1, 2, 1, 1, 3] * 10, dtype=np.float64)))
        p = DlmParameters([[1.3]], np.array([[0.5, 0.1], [0.1, 0.2]]), [0.0, 0.0], np.eye(2) * 4.0); expect = "lane-sampler"
    else:
        mod = Dlm.polynomial(2)
        for _ in range(9):
            mod = mod * Dlm.polynomial(2)
        mat = materialise(mod, np.arange(1, 31, dtype=np.float64))
        A = rng.standard_normal((20, 20))
        p = DlmParameters(np.eye(10) * 1.2, A @ A.T / 20 + 0.1 * np.eye(20), np.zeros(20), np.eye(20)); expect = "wave-sampler"
    d, q, T = mat.d, mat.p, mat.T
    y = rng.standard_normal((4, T, q)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    z = rng.standard_normal((4, T + 1, d))
    full = eng.ffbs(mat, p, y, z=z, want_cond=True, flags=_lib.OPT_STATS_OUTER)
    assert eng.last_variant == eng.expect(expect)
    filt = eng.filter(mat, p, y)["filt"]
    again = eng.ffbs(mat, p, y, z=z, want_cond=True, flags=_lib.OPT_STATS_OUTER, filt=filt)
    assert eng.last_variant == eng.expect(expect)
    np.testing.assert_allclose(again["theta"], full["theta"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(again["cond"], full["cond"], rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(again["stats"], full["stats"], rtol=1e-9, atol=1e-10)


# ------------------------------------------------------------------------------------------
# round 2: packed symmetric records, engine-owned buffers, stream ordering, device-resident Gibbs
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["seasonal_d13", "seasonal_d13_missing_irregular", "poly2_d2", "poly3_plus_harmonic_d5"])
def test_packed_symmetric_records_equal_dense(eng, case):
    """DLM_OPT_PACKED_SYM: [mean | lower triangle by rows] records carry exactly the dense call's numbers."""
    rng = np.random.default_rng(3)
    if case.startswith("seasonal"):
        mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
        times = np.cumsum(np.array([1, 1, 2, 1, 3] * 12, dtype=np.float64)) if "irregular" in case else np.arange(1, 61, dtype=np.float64)
        p = DlmParameters([[1.0]], np.diag([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4]), np.zeros(13), np.eye(13))
    elif case == "poly2_d2":
        mod, times = Dlm.polynomial(2), np.arange(1, 41, dtype=np.float64)
        p = DlmParameters([[2.0]], np.array([[0.5, 0.1], [0.1, 0.3]]), np.zeros(2), np.eye(2) * 5.0)
    else:
        mod, times = Dlm.polynomial(3) + Dlm.seasonal(12, 1), np.arange(1, 41, dtype=np.float64)
        p = DlmParameters([[1.5]], np.eye(5) * 0.2, np.zeros(5), np.eye(5))
    mat = materialise(mod, times)
    d = mat.d
    y = rng.standard_normal((5, mat.T, 1)).cumsum(axis=1)
    if "missing" in case:
        y[rng.random(y.shape) < 0.15] = np.nan
    dense = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_LANE)
    packed = eng.filter_smooth(mat, p, y, flags=_lib.OPT_PACKED_SYM)
    assert eng.last_variant == "sparse16" and np.all(packed["status"] == 0)
    prec = d + d * (d + 1) // 2
    assert packed["filt"].shape[-1] == prec + (prec & 1) == _lib.load().dlm_packed_record_doubles(d)
    for name in ("filt", "smooth"):
        un = eng.unpack_records(d, packed[name])
        m, C = split(dense[name], d)
        Cs = 0.5 * (C.reshape(C.shape[:-1] + (d, d)) + np.swapaxes(C.reshape(C.shape[:-1] + (d, d)), -1, -2))
        # the dense record holds both triangles (equal to rounding); the packed one the lower triangle, mirrored on unpack --
        # and the backward pass then reads a mirrored C_t where the dense call reads both triangles: rounding-level differences
        np.testing.assert_allclose(un[..., :d], m, rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(un[..., d:].reshape(Cs.shape), Cs, rtol=1e-11, atol=1e-12)
    pf = eng.filter(mat, p, y, flags=_lib.OPT_PACKED_SYM)
    np.testing.assert_array_equal(pf["filt"][..., :prec], packed["filt"][..., :prec])
    with pytest.raises(EngineError):        # shapes outside the structured d <= 15, p = 1 path say so
        eng.filter_smooth(mat, p, y, flags=_lib.OPT_PACKED_SYM | _lib.OPT_FORCE_GENERIC)


def test_engine_owned_buffers_round_trip_and_device_mode(eng):
    """dlm_buffer_alloc / upload / download: a caller without torch keeps a fused call device-resident."""
    import ctypes
    mod, mat, p = seasonal_model(T=40)
    N, d = 6, 13
    y = simulate(mat, p, N, seed=5, missing=0.05)
    host = eng.filter_smooth(mat, p, y)
    free0, total = eng.mem_info()
    assert total > 100e9 and free0 <= total
    rec = d + d * d
    V, vs, W, ws, m0, m0s, C0, c0s, _, _ = __import__("bayesian_dlms_amd.engine", fromlist=["pack_params"]).pack_params(p, N)
    arrays = {"y": y.reshape(-1), "F": mat.F, "G": mat.G, "V": V, "W": W, "m0": m0, "C0": C0}
    ptrs = {}
    for k, a in arrays.items():
        a = np.ascontiguousarray(a, dtype=np.float64)
        ptrs[k] = eng.buffer_alloc(a.nbytes)
        eng.buffer_upload(ptrs[k], a)
    nrec = N * (mat.T + 1) * rec
    for k in ("filt", "smooth"):
        ptrs[k] = eng.buffer_alloc(nrec * 8)
    ptrs["status"] = eng.buffer_alloc(N * 4)
    md = _lib.ModelDesc(mat.d, mat.p, mat.T, N, ptrs["F"], 0, ptrs["G"], 1, None, None)
    pd = _lib.ParamsDesc(ptrs["V"], 0, ptrs["W"], 0, ptrs["m0"], 0, ptrs["C0"], 0, 0, 0)
    op = _lib.Options(0, _lib.DLM_MEM_DEVICE, 0, 0)
    lib = _lib.load()
    V_ = ctypes.c_void_p
    rc = lib.dlm_filter_smooth_batch(eng.h, md, pd, V_(ptrs["y"]), op, V_(ptrs["filt"]), V_(ptrs["smooth"]), V_(ptrs["status"]))
    assert rc == 0
    for k in ("filt", "smooth"):
        back = eng.buffer_download(ptrs[k], np.empty(nrec))
        np.testing.assert_array_equal(back.reshape(N, mat.T + 1, rec), host[k])
    one = eng.buffer_download(ptrs["smooth"], np.empty(rec), offset=(3 * (mat.T + 1) + 17) * rec * 8)    # a single record
    np.testing.assert_array_equal(one, host["smooth"][3, 17])
    for ptr in ptrs.values():
        eng.buffer_free(ptr)
    with pytest.raises(EngineError):
        eng.buffer_free(12345)          # not a buffer of this engine


def test_inputs_produced_on_the_torch_stream_are_ordered_before_the_engine(eng):
    """The engine launches on its own stream: a tensor still being written by a long-running torch kernel must be
    complete before the filter reads it (dlm_engine_wait_stream)."""
    import torch
    mod, mat, p = seasonal_model(T=64)
    N = 2048
    dev = torch.device("cuda", 0)
    base = torch.as_tensor(simulate(mat, p, 4, seed=9), device=dev).repeat(N // 4, 1, 1).contiguous()
    want = eng.filter_smooth(mat, p, base)["smooth"].clone()
    big = torch.randn(6144, 6144, device=dev)
    for _ in range(3):
        torch.cuda.synchronize()
        acc = big
        for _ in range(12):
            acc = (acc @ big).tanh()                 # a few tens of ms of queued work on torch's stream
        y = base + 0.0 * acc[0, 0].double()          # the observations exist only when that chain has finished
        got = eng.filter_smooth(mat, p, y)["smooth"]
        assert torch.equal(got, want)


def test_gibbs_sample_svd_and_device_resident_pooled_chain(eng):
    """GibbsSampling.sampleSvd (Gibbs.scala:182-217) on the SVD sampler's statistics, and the pooled chain with the
    observations as a device tensor (statistics pooled and reduced in HBM: the bench's C3 control flow, one rank)."""
    import torch
    from bayesian_dlms_amd.gibbs import GibbsSampling, InverseGamma
    rng = np.random.default_rng(8)
    mod = Dlm.polynomial(1)
    T, N = 300, 12
    times = np.arange(1, T + 1, dtype=np.float64)
    x = np.cumsum(rng.standard_normal((N, T)) * np.sqrt(3.0), axis=1)
    y = (x + rng.standard_normal((N, T)) * np.sqrt(2.0))[..., None]
    init = DlmParameters([[1.0]], [[1.0]], [0.0], [[10.0]])
    chain = list(GibbsSampling.sample_svd(mod, InverseGamma(4.0, 6.0), InverseGamma(4.0, 9.0), init, times, y, eng, n_iter=80, seed=2, pooled=True))
    v = np.mean([s.p.v[0, 0] for s in chain[20:]]); w = np.mean([s.p.w[0, 0] for s in chain[20:]])
    assert 1.4 < v < 2.8 and 2.0 < w < 4.2, (v, w)
    uid = eng.comm_unique_id()
    eng.comm_init_rank(1, 0, uid)
    yd = torch.as_tensor(y, device="cuda:0")
    host = list(GibbsSampling.sample(mod, InverseGamma(4.0, 6.0), InverseGamma(4.0, 9.0), init, times, y, eng, n_iter=4, seed=2, pooled=True))
    devc = list(GibbsSampling.sample(mod, InverseGamma(4.0, 6.0), InverseGamma(4.0, 9.0), init, times, yd, eng, n_iter=4, seed=2, pooled=True,
                                     allreduce=eng.allreduce_stats))
    for a, b in zip(host, devc):   # same Philox draws, same pooled statistics, same numpy conjugate draws
        np.testing.assert_allclose(a.p.v, b.p.v, rtol=1e-10)
        np.testing.assert_allclose(a.p.w, b.p.w, rtol=1e-10)
    assert hasattr(devc[-1].stats, "data_ptr")          # the per-series statistics stayed on the device


@pytest.mark.parametrize("stream", ["w", "v", "w_literal"])
def test_svd_filter_and_sampler_with_variance_streams(eng, stream):
    """V_t / W_t streams on the SVD path (second half): DlmFsvSystem.ffbsSvd feeds
    transformParams(p.copy(w = W_t)) to SvdFilter.step / SvdSampler.step (DlmFsvSystem.scala:176-208), DlmFsv.ffbsSvd
    the same with V_t (DlmFsv.scala:208-228).  Against the oracle with per-step parameters, and -- the consistent form --
    against the standard Kalman filter with the same streams."""
    rng = np.random.default_rng(12)
    mod = Dlm.polynomial(1) + Dlm.seasonal(12, 2)
    T, d, N = 30, 5, 2
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    base_w = np.diag([0.05, 0.3, 0.2, 0.4, 0.1])
    if stream.startswith("w"):
        A = rng.standard_normal((T, d, d)) * 0.15
        W = base_w[None] * (1.0 + 0.5 * rng.random((T, 1, 1))) + A @ A.transpose(0, 2, 1)      # dense SPD, different every step
        V = np.array([[1.2]])
    else:
        W = base_w
        V = (0.5 + rng.random((T, 1, 1)) * 2.0)
    literal = stream == "w_literal"
    p = DlmParameters(V, W, np.zeros(d), np.diag(np.linspace(0.5, 2.0, d)))
    y = rng.standard_normal((N, T, 1)).cumsum(axis=1)
    y[rng.random(y.shape) < 0.1] = np.nan
    flags = (_lib.OPT_SVD_RAW_W_Q2 | _lib.OPT_SVD_SAMPLER_Q9) if literal else 0
    out = eng.svd_filter(mat, p, y, flags=flags)
    assert eng.last_variant == "svd-jacobi" and np.all(out["status"] == 0)
    z = rng.standard_normal((N, T + 1, d))
    draw = eng.svd_ffbs(mat, p, y, z=z, flags=flags)
    om = omodel(mat)
    for n in range(N):
        o = oracle.svd_filter(om, V, W, p.m0, p.c0, y[n], raw_w_q2=literal)
        m, C = _svd_cov(out["svd"][n], d)
        om_, oC = _svd_cov(np.concatenate([o["m"], o["dc"], o["uc"]], axis=1), d)
        np.testing.assert_allclose(m, om_, rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(C, oC, rtol=1e-7, atol=1e-8)
        ob = oracle.svd_backward_sample(om, W, o, z[n], literal_q9=literal)
        np.testing.assert_allclose(draw["theta"][n], ob["theta"], rtol=1e-6, atol=1e-7)
        if not literal:
            kf = oracle.kf_filter(om, V, W, p.m0, p.c0, y[n])
            np.testing.assert_allclose(m, kf["m"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(C.transpose(0, 2, 1).reshape(T + 1, d * d), kf["C"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("shape", ["c4_d40_p20", "d20_p6_irregular_missing", "d33_p17_wstream"])
def test_reference_form_sampler_on_register_tiles_16_to_48(eng, shape):
    """Smoothing.sampleDlm (Smoothing.scala:74-122) for 16 <= d <= 48 on register tiles (dlm_wave48.hip: k_sampler_w48):
    conditional moments, draws and statistics equal the generic kernel's and the oracle's literal sampler with the
    canonical (Cholesky) factor -- including the steady-state reuse of J, H and the factor on a long regular stretch."""
    rng = np.random.default_rng(14)
    wst = False
    if shape == "c4_d40_p20":
        mod = Dlm.polynomial(2)
        for _ in range(19):
            mod = mod * Dlm.polynomial(2)
        T, q, d = 160, 20, 40
        times = np.arange(1, T + 1, dtype=np.float64)
        miss = 0.0
    elif shape == "d20_p6_irregular_missing":
        mod = Dlm.polynomial(2) + Dlm.seasonal(12, 3)
        mod = mod * (Dlm.polynomial(1) + Dlm.seasonal(24, 2)) * Dlm.polynomial(3) * Dlm.polynomial(1) * Dlm.polynomial(2) * Dlm.polynomial(1)
        times = np.cumsum(np.array([1, 1, 2, 1, 2, 3, 1] * 6, dtype=np.float64))
        T, q, d = len(times), 6, 20
        miss = 0.15
    else:
        mod = Dlm.polynomial(2)
        for _ in range(15):
            mod = mod * Dlm.polynomial(2)
        mod = mod * Dlm.polynomial(1)
        T, q, d = 30, 17, 33
        times = np.arange(1, T + 1, dtype=np.float64)
        miss, wst = 0.05, True
    mat = materialise(mod, times)
    assert (mat.d, mat.p) == (d, q)
    A = rng.standard_normal((d, d))
    W = A @ A.T / d + 0.1 * np.eye(d)
    if wst:
        B = rng.standard_normal((T, d, d)) * 0.1
        W = W[None] + B @ B.transpose(0, 2, 1)
    p = DlmParameters(np.eye(q) * 0.8, W, rng.standard_normal(d), np.eye(d) * 2.0)
    N = 3
    y = rng.standard_normal((N, T, q)).cumsum(axis=1)
    if miss:
        y[rng.random(y.shape) < miss] = np.nan
    z = rng.standard_normal((N, T + 1, d))
    for flags in (0, _lib.OPT_STATS_OUTER):
        out = eng.ffbs(mat, p, y, z=z, flags=flags, want_cond=True)
        assert eng.last_variant == eng.expect("wave-sampler") and np.all(out["status"] == 0)
        gen = eng.ffbs(mat, p, y, z=z, flags=flags | _lib.OPT_FORCE_GENERIC, want_cond=True)
        assert eng.last_variant == "generic"
        np.testing.assert_allclose(out["theta"], gen["theta"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(out["cond"], gen["cond"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(out["stats"], gen["stats"], rtol=1e-7, atol=1e-8)
    om = omodel(mat)
    fo = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[1])
    ob = oracle.backward_sample(om, p.w if not wst else W, fo, z[1], factor="chol")
    np.testing.assert_allclose(out["theta"][1], ob["theta"], rtol=1e-6, atol=1e-7)
    ph = eng.ffbs(mat, p, y, seed=3, series_offset=11)      # Philox stream, no conditional records
    pg = eng.ffbs(mat, p, y, seed=3, series_offset=11, flags=_lib.OPT_FORCE_GENERIC)
    np.testing.assert_allclose(ph["theta"], pg["theta"], rtol=1e-7, atol=1e-8)



def test_steady_state_steps_of_the_structured_kernels(eng):
    """Once the covariance recursion has settled (regular grid, no missing observation) the d <= 15 kernels advance only the
    mean (forward: a = G m, m = a + K e; backward: q alone) and mark it in the side record; DLM_OPT_NO_STEADY recomputes
    everything every step.  Same numbers to 1e-11, against the oracle to the usual tolerance, and a gap in the data takes
    the full path again."""
    rng = np.random.default_rng(31)
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    T = 700
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    p = DlmParameters([[1.0]], np.diag([0.3, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4]), np.zeros(13), np.eye(13))   # settles within ~80 steps
    y = rng.standard_normal((6, T, 1)).cumsum(axis=1)
    y[1, 300:305, 0] = np.nan            # a gap: leaves the steady state, re-enters it later
    y[2, 10::97, 0] = np.nan             # scattered missing values
    fast = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == "sparse16" and np.all(fast["status"] == 0)
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    np.testing.assert_allclose(fast["filt"], full["filt"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(fast["smooth"], full["smooth"], rtol=1e-10, atol=1e-11)
    # the covariance really is frozen on the settled stretch of an undisturbed series ...
    C = fast["filt"][0][:, 13:]
    assert np.array_equal(C[400], C[399]) and np.array_equal(C[600], C[400])
    # ... and moves again after the gap
    C1 = fast["filt"][1][:, 13:]
    assert not np.array_equal(C1[305], C1[299])
    for n in (0, 1, 2):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, Cc = split(fast["filt"][n], 13); sm, S = split(fast["smooth"][n], 13)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(Cc, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
    ll = eng.loglik(mat, p, y)["loglik"]
    ll0 = eng.loglik(mat, p, y, flags=_lib.OPT_NO_STEADY)["loglik"]
    np.testing.assert_allclose(ll, ll0, rtol=1e-11)
    np.testing.assert_allclose(ll[0], oracle.loglik(omodel(mat), oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0]), y[0]), rtol=1e-9)
    fq = eng.filter(mat, p, y, want_fq=True)["fq"]
    f0 = oracle.kf_filter(omodel(mat), p.v, p.w, p.m0, p.c0, y[0])
    np.testing.assert_allclose(fq[0, 1:, 0], f0["f"][1:, 0], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(fq[0, 1:, 1], f0["Q"][1:, 0], rtol=1e-9, atol=1e-10)
    # the backward pass fetches only the mean of a record whose covariance it already holds: the same with packed records on
    # either side (DLM_OPT_PACKED_SYM: packed in and out; want_filt=False: packed engine-internal records in, dense out)
    pk = eng.filter_smooth(mat, p, y, flags=_lib.OPT_PACKED_SYM)
    np.testing.assert_allclose(eng.unpack_records(13, pk["smooth"]), fast["smooth"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(eng.unpack_records(13, pk["filt"]), fast["filt"], rtol=1e-12, atol=1e-13)
    so = eng.filter_smooth(mat, p, y, want_filt=False)
    np.testing.assert_allclose(so["smooth"], fast["smooth"], rtol=1e-12, atol=1e-13)
    # a series that ends inside its steady stretch, one that never reaches it (T = 40), and T = 1
    for Ts in (40, 1):
        ms = materialise(mod, np.arange(1, Ts + 1, dtype=np.float64))
        a1 = eng.filter_smooth(ms, p, y[:, :Ts])
        b1 = eng.filter_smooth(ms, p, y[:, :Ts], flags=_lib.OPT_NO_STEADY)
        np.testing.assert_allclose(a1["smooth"], b1["smooth"], rtol=1e-11, atol=1e-12)
    # the simulation smoother's forward pass (y* has the covariance recursion of y) takes the same steady steps, its backward
    # pass the mean-only records: same draws and statistics to 1e-10
    z = rng.standard_normal((6, T + 1, 14))
    a_ = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH, want_stats=True)
    b_ = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_NO_STEADY, want_stats=True)
    np.testing.assert_allclose(a_["theta"], b_["theta"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(a_["stats"], b_["stats"], rtol=1e-9)
    ph = eng.ffbs(mat, p, y, seed=5, flags=_lib.OPT_FFBS_SIMSMOOTH)          # the Philox stream
    pg = eng.ffbs(mat, p, y, seed=5, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_NO_STEADY)
    np.testing.assert_allclose(ph["theta"], pg["theta"], rtol=1e-10, atol=1e-11)


def test_steady_state_steps_of_the_per_wave_kernels(eng):
    """16 <= d <= 48 (k_filter_w48 / k_smoother_w48): once the covariance recursion has settled on a regular, fully observed
    stretch the forward step is a = G m, e = y - F^T a, m = a + K e with the gain parked in LDS, and the record is marked for the
    backward pass.  Same numbers as with DLM_OPT_NO_STEADY to 1e-10, the usual tolerance against the oracle, frozen covariances
    on the settled stretch, the full path again after a (partially) missing observation."""
    rng = np.random.default_rng(53)
    mod = Dlm.polynomial(2)
    for _ in range(9):
        mod = mod * Dlm.polynomial(2)              # d = 20, p = 10
    T = 160
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A = rng.standard_normal((d, d))
    p = DlmParameters(np.eye(q), A @ A.T / d + 0.1 * np.eye(d), np.zeros(d), np.eye(d))
    y = rng.standard_normal((5, T, q)).cumsum(axis=1)
    y[1, 80, :] = np.nan                 # a wholly missing observation
    y[2, 90, 3] = np.nan                 # a partially missing one
    y[3, rng.random((T, q)) < 0.1] = np.nan
    fast = eng.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE)
    assert eng.last_variant == "wave-mfma" and np.all(fast["status"] == 0)
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE | _lib.OPT_NO_STEADY)
    np.testing.assert_allclose(fast["filt"], full["filt"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(fast["smooth"], full["smooth"], rtol=1e-9, atol=1e-10)
    C = fast["filt"][0][:, d:]
    assert np.array_equal(C[120], C[119]) and np.array_equal(C[150], C[100])          # frozen on the settled stretch
    Cn = full["filt"][0][:, d:]
    assert not np.array_equal(Cn[120], Cn[119])                                        # (the full path keeps rounding about)
    C1 = fast["filt"][1][:, d:]
    assert not np.array_equal(C1[81], C1[79]) and np.array_equal(C1[150], C1[149])     # leaves at the gap, settles again
    for n in range(4):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, Cc = split(fast["filt"][n], d); sm, S = split(fast["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(Cc, f["C"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(S, s["S"], rtol=1e-7, atol=1e-8)
    # filter alone (no innovation buffer, no marks) and the simulation smoother's forward pass (filters y* in place)
    fo = eng.filter(mat, p, y, flags=_lib.OPT_FORCE_WAVE)
    np.testing.assert_allclose(fo["filt"], full["filt"], rtol=1e-10, atol=1e-11)
    z = rng.standard_normal((5, T + 1, d + q))
    a_ = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_FORCE_WAVE)
    b_ = eng.ffbs(mat, p, y, z=z, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_FORCE_WAVE | _lib.OPT_NO_STEADY)
    np.testing.assert_allclose(a_["theta"], b_["theta"], rtol=1e-9, atol=1e-10)


def test_svd_filter_steady_state_reuse(eng):
    """Once the posterior factors have stopped moving (largest change <= 1e-11 of the largest entry) the SVD filter reuses both
    decompositions and only the mean moves; DLM_OPT_FORCE_GENERIC recomputes them every step.  Same covariances and means to 1e-8
    (the path is held to 1e-7), against the Kalman filter of the oracle, and the full path again after a missing observation."""
    rng = np.random.default_rng(61)
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    T = 400
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    p = DlmParameters([[1.0]], np.diag([0.3, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4]), np.zeros(13), np.eye(13))   # settles within ~80 steps
    y = rng.standard_normal((3, T, 1)).cumsum(axis=1)
    y[1, 250, 0] = np.nan
    fast = eng.svd_filter(mat, p, y)
    assert eng.last_variant == "svd-jacobi" and np.all(fast["status"] == 0)
    full = eng.svd_filter(mat, p, y, flags=_lib.OPT_FORCE_GENERIC)
    om = omodel(mat)
    for n in range(3):
        m, C = _svd_cov(fast["svd"][n], 13)
        m2, C2 = _svd_cov(full["svd"][n], 13)
        np.testing.assert_allclose(m, m2, rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(C, C2, rtol=1e-8, atol=1e-9)
        kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y[n])
        np.testing.assert_allclose(m, kf["m"], rtol=1e-7, atol=1e-8)
        np.testing.assert_allclose(C.transpose(0, 2, 1).reshape(T + 1, 169), kf["C"], rtol=1e-7, atol=1e-8)
    # frozen factors on the settled stretch of the undisturbed series; moving again after the gap of series 1
    r0 = fast["svd"][0]
    assert np.array_equal(r0[300, 13:], r0[299, 13:])
    r1 = fast["svd"][1]
    assert not np.array_equal(r1[252, 13:], r1[249, 13:])


def test_structure_analysis_cache_follows_the_model(eng):
    """The engine keeps the structure analysis of the last call (DLM_OPT_MODEL_UNCHANGED from the Python layer when the staged F, G
    and time grid are bit for bit the same; host-memory calls compare the tables themselves).  Same model, other parameters:
    reused; another model of the same shape, or an entry point in between that analyses nothing: analysed afresh."""
    from bayesian_dlms_amd.engine import Engine
    import torch
    rng = np.random.default_rng(67)
    T = 50
    times = np.arange(1, T + 1, dtype=np.float64)
    modA = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    modB = Dlm.polynomial(1) + Dlm.seasonal(12, 6)          # same d = 13, another G
    matA, matB = materialise(modA, times), materialise(modB, times)
    p1 = DlmParameters([[1.0]], np.diag(rng.uniform(0.1, 0.5, 13)), np.zeros(13), np.eye(13))
    p2 = DlmParameters([[2.0]], np.diag(rng.uniform(0.1, 0.5, 13)), np.ones(13), 2.0 * np.eye(13))
    y = torch.as_tensor(rng.standard_normal((4, T, 1)).cumsum(axis=1), device="cuda")
    fresh = lambda mat, p: {k: v.cpu().numpy() for k, v in Engine(0).filter_smooth(mat, p, y).items() if k in ("filt", "smooth")}
    seq = [(matA, p1), (matA, p1), (matA, p2), (matB, p2), (matB, p1), (matA, p1)]
    for i, (mat, p) in enumerate(seq):
        out = eng.filter_smooth(mat, p, y)
        ref = fresh(mat, p)
        np.testing.assert_array_equal(out["filt"].cpu().numpy(), ref["filt"], err_msg=f"call {i}")
        np.testing.assert_array_equal(out["smooth"].cpu().numpy(), ref["smooth"], err_msg=f"call {i}")
        if i == 3:   # an entry point that analyses nothing, on yet another model of the same shape, in between
            eng.svd_filter(matA, p1, y)
    # host-memory calls: the tables are compared by the engine itself
    yh = y.cpu().numpy()
    for mat, p in seq[2:5]:
        out = eng.filter_smooth(mat, p, yh)
        ref = fresh(mat, p)
        np.testing.assert_array_equal(out["filt"], ref["filt"])


def test_dispatch_rule_of_the_multivariate_kernels():
    """16 <= d <= 48 with a structured G: more than 256 series, or a time-invariant model on a regular grid at any batch size (its
    steady-state steps), take the wave-per-series kernels; a handful of series on an irregular grid the workgroup-per-series ones."""
    from bayesian_dlms_amd.engine import Engine
    e = Engine(0)
    rng = np.random.default_rng(71)
    mod = Dlm.polynomial(2)
    for _ in range(9):
        mod = mod * Dlm.polynomial(2)              # d = 20, p = 10
    A = rng.standard_normal((20, 20))
    p = DlmParameters(np.eye(10), A @ A.T / 20 + 0.1 * np.eye(20), np.zeros(20), np.eye(20))
    T = 40
    reg = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    irr = materialise(mod, np.cumsum(rng.integers(1, 3, T)).astype(np.float64))
    y = rng.standard_normal((4, T, 10)).cumsum(axis=1)
    e.filter_smooth(reg, p, y)
    assert e.last_variant == "wave-mfma"
    e.filter_smooth(reg, p, y, flags=_lib.OPT_NO_STEADY)
    assert e.last_variant == "tiled-mfma"
    e.filter_smooth(irr, p, y)
    assert e.last_variant == "tiled-mfma"
    e.filter_smooth(irr, p, rng.standard_normal((300, T, 10)).cumsum(axis=1))
    assert e.last_variant == "wave-mfma"
    e.close()


@pytest.mark.parametrize("knz", [1, 3, 4])
def test_steady_state_steps_for_every_sparsity_instantiation(eng, knz):
    """The structured d <= 15 kernels are instantiated for 1..4 nonzeros per row and column of G (the C2 model has 2): a stable
    circulant G with `knz` nonzeros per row, d = 8, long enough to settle.  Fast path against DLM_OPT_NO_STEADY and the oracle,
    with a gap in one series, packed records, and the log-likelihood."""
    rng = np.random.default_rng(80 + knz)
    d, T = 8, 500
    Gm = np.zeros((d, d))
    coef = {1: [0.9], 3: [0.6, 0.25, -0.2], 4: [0.5, 0.3, -0.2, 0.15]}[knz]
    for i in range(d):
        for s_, cf in enumerate(coef):
            Gm[i, (i + s_) % d] = cf
    Fv = np.array([1.0, 0.0, 1.0, 0.5, 0.0, 1.0, 0.0, -0.5]).reshape(-1, 1)
    mod = Dlm(lambda t: Fv, lambda dt: Gm)
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    p = DlmParameters([[0.7]], np.diag(rng.uniform(0.1, 0.5, d)), np.zeros(d), np.eye(d))
    y = rng.standard_normal((4, T, 1))
    y[1, 200:203, 0] = np.nan
    fast = eng.filter_smooth(mat, p, y)
    assert eng.last_variant == "sparse16" and np.all(fast["status"] == 0)
    full = eng.filter_smooth(mat, p, y, flags=_lib.OPT_NO_STEADY)
    np.testing.assert_allclose(fast["filt"], full["filt"], rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(fast["smooth"], full["smooth"], rtol=1e-10, atol=1e-11)
    C = fast["filt"][0][:, d:]
    assert np.array_equal(C[450], C[300])                    # it did settle
    for n in (0, 1):
        f, s = oracle_filter_smooth(mat, p, y[n])
        m, Cc = split(fast["filt"][n], d); sm, S = split(fast["smooth"][n], d)
        np.testing.assert_allclose(m, f["m"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(Cc, f["C"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(sm, s["s"], rtol=1e-8, atol=1e-9)
        np.testing.assert_allclose(S, s["S"], rtol=1e-8, atol=1e-9)
    pk = eng.filter_smooth(mat, p, y, flags=_lib.OPT_PACKED_SYM)
    np.testing.assert_allclose(eng.unpack_records(d, pk["smooth"]), fast["smooth"], rtol=1e-12, atol=1e-13)
    ll = eng.loglik(mat, p, y)["loglik"]
    ll0 = eng.loglik(mat, p, y, flags=_lib.OPT_NO_STEADY)["loglik"]
    np.testing.assert_allclose(ll, ll0, rtol=1e-11)
    a_ = eng.ffbs(mat, p, y, seed=3, flags=_lib.OPT_FFBS_SIMSMOOTH)
    b_ = eng.ffbs(mat, p, y, seed=3, flags=_lib.OPT_FFBS_SIMSMOOTH | _lib.OPT_NO_STEADY)
    np.testing.assert_allclose(a_["theta"], b_["theta"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("k", [4, 8, 12, 17, 24])
def test_steady_state_steps_across_the_per_wave_instantiations(k):
    """|*| of k linear-growth models: d = 2k, p = k -- one tile per dimension (k = 4, 8), two tiles for d (12), three for d and two
    for p (17, and 24: d = 48), with Vm^-1 in registers or in LDS as the instantiation has it.  Steady-state steps against
    DLM_OPT_NO_STEADY and the oracle, on the per-wave kernels, with a partially missing observation in one series."""
    from bayesian_dlms_amd.engine import Engine
    e = Engine(0)
    rng = np.random.default_rng(90 + k)
    mod = Dlm.polynomial(2)
    for _ in range(k - 1):
        mod = mod * Dlm.polynomial(2)
    T = 140
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    d, q = mat.d, mat.p
    A = rng.standard_normal((d, d))
    p = DlmParameters(np.eye(q) * 0.8, A @ A.T / d + 0.2 * np.eye(d), np.zeros(d), np.eye(d))
    y = rng.standard_normal((3, T, q)).cumsum(axis=1)
    y[1, 70, q // 2] = np.nan
    fast = e.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE)
    assert e.last_variant == "wave-mfma" and np.all(fast["status"] == 0)
    full = e.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE | _lib.OPT_NO_STEADY)
    np.testing.assert_allclose(fast["filt"], full["filt"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(fast["smooth"], full["smooth"], rtol=1e-9, atol=1e-10)
    C = fast["filt"][0][:, d:]
    assert np.array_equal(C[130], C[110])                    # settled (and frozen) well before the end
    f, s = oracle_filter_smooth(mat, p, y[1])
    m, Cc = split(fast["filt"][1], d); sm, S = split(fast["smooth"][1], d)
    np.testing.assert_allclose(m, f["m"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(Cc, f["C"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(sm, s["s"], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(S, s["S"], rtol=1e-7, atol=1e-8)
    # F treated as dense (DLM_OPT_NO_SPARSE_F): the forward pass keeps its steady steps (products with F on the matrix pipe), the
    # backward pass has none
    dn = e.filter_smooth(mat, p, y, flags=_lib.OPT_FORCE_WAVE | _lib.OPT_NO_SPARSE_F)
    np.testing.assert_allclose(dn["filt"], full["filt"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(dn["smooth"], full["smooth"], rtol=1e-8, atol=1e-9)
    e.close()
