"""Pins the CPU oracle against the reference's own known-answer tests and golden CSVs.

CPU-only (no GPU marker).  Sources of truth:
  * core/src/test/scala/KalmanFilter.scala:78-189  -> tests/golden/kalman_filter_test.json
  * core/src/test/scala/Smoothing.scala:10-76      -> closed-form scalar RTS identities
  * core/src/test/scala/SvdFilter.scala:16-34,102-158 -> sqrt identities, SVD filter == KF
  * examples/data/first_order_dlm*.csv             -> tests/golden/*.csv (config C1)
Everything d>1 in the smoother / FFBS / Gibbs is unpinned by the reference and is checked
here through algebraic identities instead.
"""
import csv
import json
import os

import numpy as np
import pytest

import oracle
from bayesian_dlms_amd.dlm import Dlm, DlmParameters, materialise


def _read_csv(path):
    with open(path) as fh:
        rows = list(csv.reader(fh))
    return rows[0], rows[1:]


def _omodel(mat):
    return oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)


# ----------------------------------------------------------------------------------------
# KalmanFilterTest known-answer table (bivariate, missing data, dt = 2)
# ----------------------------------------------------------------------------------------
def _kf_fixture(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "kalman_filter_test.json")))
    mod = Dlm.polynomial(1) * Dlm.polynomial(1)
    p = DlmParameters(np.diag(g["v"]), np.diag(g["w"]), np.array(g["m0"]), np.diag(g["c0"]))
    y = np.array([[np.nan if v is None else v for v in row] for row in g["obs"]], dtype=np.float64)
    mat = materialise(mod, g["times"])
    return g, mod, p, y, mat


def test_kalman_filter_known_answers(golden_dir):
    g, mod, p, y, mat = _kf_fixture(golden_dir)
    assert mat.d == 2 and mat.p == 2 and mat.n_g == 1  # polynomial G does not depend on dt
    assert list(mat.dt) == [1, 1, 1, 1, 1, 2]
    out = oracle.kf_filter(_omodel(mat), p.v, p.w, p.m0, p.c0, y)
    assert out["rc"] == 0
    tol = g["tol"]
    for k, exp in g["steps"].items():
        t = int(k)
        for name in ("a", "f", "m"):
            if name in exp:
                np.testing.assert_allclose(out[name][t], exp[name], atol=tol, rtol=0)
        for name, dim in (("R", 2), ("Q", 2), ("C", 2)):
            if name in exp:
                got = oracle.from_cm(out[name][t], dim, dim)
                np.testing.assert_allclose(got, np.diag(exp[name]), atol=tol, rtol=0)
    # the values the reference author left commented out (KalmanFilter.scala:187-188)
    s6 = g["steps"]["6"]
    assert abs(out["m"][6][0] - s6["m_first_commented"]) < 1e-4
    assert abs(out["C"][6][0] - s6["C_first_commented"]) < 1e-4
    # record 0 is the initial state at t0 - 1, f/Q absent (None)
    np.testing.assert_array_equal(out["m"][0], p.m0)
    assert np.all(np.isnan(out["f"][0])) and np.all(np.isnan(out["Q"][0]))


def test_gain_identity(golden_dir):
    """KfSpec: (Q^T \\ R^T)^T == R inv(Q) (KalmanFilter.scala:21-30) via the first update."""
    g, mod, p, y, mat = _kf_fixture(golden_dir)
    out = oracle.kf_filter(_omodel(mat), p.v, p.w, p.m0, p.c0, y)
    R1 = oracle.from_cm(out["R"][1], 2, 2); Q1 = oracle.from_cm(out["Q"][1], 2, 2)
    K = R1 @ np.linalg.inv(Q1)
    m1 = out["a"][1] + K @ (y[0] - out["f"][1])
    np.testing.assert_allclose(out["m"][1], m1, atol=1e-12)
    np.testing.assert_allclose(oracle.from_cm(out["C"][1], 2, 2), R1 - K @ R1, atol=1e-12)


# ----------------------------------------------------------------------------------------
# first_order_dlm golden CSVs (config C1): exact Double.toString output of the reference
# ----------------------------------------------------------------------------------------
def _c1(golden_dir):
    _, rows = _read_csv(os.path.join(golden_dir, "first_order_dlm.csv"))
    times = np.array([float(r[0]) for r in rows]); y = np.array([float(r[1]) for r in rows])
    mat = materialise(Dlm.polynomial(1), times)
    p = DlmParameters([[2.0]], [[3.0]], [0.0], [[10.0]])
    return mat, p, y.reshape(-1, 1)


def test_first_order_filtered_golden(golden_dir):
    mat, p, y = _c1(golden_dir)
    assert mat.T == 1000 and mat.dt is None and mat.g_index is None
    out = oracle.kf_filter(_omodel(mat), p.v, p.w, p.m0, p.c0, y)
    _, rows = _read_csv(os.path.join(golden_dir, "first_order_dlm_filtered.csv"))
    assert len(rows) == 1001
    exp_m = np.array([float(r[1]) for r in rows]); exp_c = np.array([float(r[2]) for r in rows])
    exp_f = np.array([float(r[3]) for r in rows[1:]]); exp_q = np.array([float(r[4]) for r in rows[1:]])
    np.testing.assert_allclose(out["m"][:, 0], exp_m, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out["C"][:, 0], exp_c, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out["f"][1:, 0], exp_f, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(out["Q"][1:, 0], exp_q, rtol=1e-13, atol=1e-13)
    assert float(rows[0][0]) == 0.0  # initial state at t0 - 1


@pytest.mark.parametrize("compat", [True, False])
def test_first_order_smoothed_golden(golden_dir, compat):
    mat, p, y = _c1(golden_dir)
    om = _omodel(mat)
    filt = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    sm = oracle.smoother(om, filt, compat_q1=compat)  # d = 1: Q1 form == textbook form
    _, rows = _read_csv(os.path.join(golden_dir, "first_order_dlm_smoothed.csv"))
    exp_s = np.array([float(r[1]) for r in rows]); exp_S = np.array([float(r[2]) for r in rows])
    np.testing.assert_allclose(sm["s"][:, 0], exp_s, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sm["S"][:, 0], exp_S, rtol=1e-12, atol=1e-12)


# ----------------------------------------------------------------------------------------
# SmoothingTest (Smoothing.scala:10-76): scalar RTS closed forms
# ----------------------------------------------------------------------------------------
def test_smoothing_identities():
    times = [1.0, 2.0, 3.0, 4.0, 5.0, 7.0]
    y = np.array([4.5, 3.0, 6.3, np.nan, 10.1, 15.2]).reshape(-1, 1)
    mat = materialise(Dlm.polynomial(1), times)
    om = _omodel(mat)
    p = DlmParameters([[3.0]], [[1.0]], [0.0], [[1.0]])
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    s = oracle.smoother(om, f, compat_q1=True)
    assert s["s"].shape[0] == len(times) + 1
    tol = 1e-4
    s7, S7 = s["s"][-1, 0], s["S"][-1, 0]
    assert abs(f["m"][-1, 0] - s7) < tol and abs(f["C"][-1, 0] - S7) < tol
    m5, c5, r7, a7 = f["m"][5, 0], f["C"][5, 0], f["R"][6, 0], f["a"][6, 0]
    s5 = m5 + c5 / r7 * (s7 - a7); S5 = c5 - c5 * c5 / (r7 * r7) * (r7 - S7)
    assert abs(s["s"][5, 0] - s5) < tol and abs(s["S"][5, 0] - S5) < tol
    m4, c4, r5, a5 = f["m"][4, 0], f["C"][4, 0], f["R"][5, 0], f["a"][5, 0]
    s4 = m4 + c4 / r5 * (s5 - a5); S4 = c4 - c4 * c4 / (r5 * r5) * (r5 - S5)
    assert abs(s["s"][4, 0] - s4) < tol and abs(s["S"][4, 0] - S4) < tol


# ----------------------------------------------------------------------------------------
# SvdKfSpec / SvdFilterTest (SvdFilter.scala test :16-34, :102-158)
# ----------------------------------------------------------------------------------------
def _spd(rng, n, cond=10.0):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    return q @ np.diag(np.linspace(1.0, cond, n)) @ q.T


def test_sqrt_svd_identities():
    rng = np.random.default_rng(7)
    for n in (1, 2, 5, 13):
        m = _spd(rng, n)
        r = oracle.sqrt_svd(m)
        np.testing.assert_allclose(r.T @ r, m, atol=1e-10)
        ri = oracle.sqrt_svd(m, inverse=True)
        np.testing.assert_allclose(np.linalg.inv(ri.T @ ri), m, atol=1e-9)


def test_svd_filter_matches_kalman_filter(golden_dir):
    g, mod, p, y, mat = _kf_fixture(golden_dir)
    om = _omodel(mat)
    kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    for raw in (True, False):  # W = I here, so Q2 is invisible exactly as in the reference test
        sf = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y, raw_w_q2=raw)
        for t in range(1, 7):
            np.testing.assert_allclose(sf["m"][t], kf["m"][t], atol=1e-9)
            uc = oracle.from_cm(sf["uc"][t], 2, 2)
            cov = uc @ np.diag(sf["dc"][t] ** 2) @ uc.T
            np.testing.assert_allclose(cov, oracle.from_cm(kf["C"][t], 2, 2), atol=1e-9)


# ----------------------------------------------------------------------------------------
# unpinned territory: d = 13 seasonal (config C2 shape), algebraic cross-checks
# ----------------------------------------------------------------------------------------
def seasonal_setup(T=60, seed=3, missing=0.0):
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
    times = np.arange(1, T + 1, dtype=np.float64)
    mat = materialise(mod, times)
    w = np.diag([0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4])
    p = DlmParameters([[1.0]], w, np.zeros(13), np.eye(13))
    rng = np.random.default_rng(seed)
    G = oracle.from_cm(mat.G, 13, 13); F = mat.F.reshape(13)
    x = rng.standard_normal(13); y = np.empty((T, 1))
    for t in range(T):
        x = G @ x + np.sqrt(np.diag(w)) * rng.standard_normal(13)
        y[t, 0] = F @ x + rng.standard_normal()
    if missing > 0:
        y[rng.random(T) < missing, 0] = np.nan
    return mod, mat, p, y


def test_seasonal_model_shape():
    mod, mat, p, y = seasonal_setup(T=5)
    assert mat.d == 13 and mat.p == 1
    np.testing.assert_array_equal(mat.F, [1, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0])
    G = oracle.from_cm(mat.G, 13, 13)
    assert G[0, 0] == 1.0
    np.testing.assert_allclose(G[1:3, 1:3], [[np.cos(2 * np.pi / 24), -np.sin(2 * np.pi / 24)],
                                             [np.sin(2 * np.pi / 24), np.cos(2 * np.pi / 24)]])
    np.testing.assert_allclose(G @ G.T, np.eye(13), atol=1e-14)


def test_seasonal_filter_vs_numpy_and_joseph_identity():
    mod, mat, p, y = seasonal_setup(T=80, missing=0.1)
    om = _omodel(mat)
    out = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    G = oracle.from_cm(mat.G, 13, 13); F = mat.F.reshape(13, 1)
    m, C = p.m0.copy(), p.c0.copy()
    for t in range(80):
        a = G @ m; R = G @ C @ G.T + p.w
        if np.isnan(y[t, 0]):
            m, C = a, R
        else:
            Q = F.T @ R @ F + p.v; K = R @ F / Q
            m = a + (K * (y[t, 0] - F.T @ a)).reshape(-1)
            C = R - K @ Q @ K.T  # textbook form == Joseph form in exact arithmetic
        np.testing.assert_allclose(out["m"][t + 1], m, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(oracle.from_cm(out["C"][t + 1], 13, 13), C, rtol=1e-9, atol=1e-11)


def test_seasonal_smoother_q1_and_textbook():
    """Q1: the literal reference form J X J differs from J X J^T for d > 1; means agree."""
    mod, mat, p, y = seasonal_setup(T=60)
    om = _omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    lit = oracle.smoother(om, f, compat_q1=True)
    txt = oracle.smoother(om, f, compat_q1=False)
    np.testing.assert_allclose(lit["s"], txt["s"], rtol=1e-12, atol=1e-12)
    S = oracle.from_cm(txt["S"][10], 13, 13)
    np.testing.assert_allclose(S, S.T, atol=1e-10)
    assert np.linalg.eigvalsh((S + S.T) / 2).min() > 0
    assert np.abs(lit["S"] - txt["S"]).max() > 1e-3  # the quirk is real
    # textbook RTS against an independent numpy recursion
    G = oracle.from_cm(mat.G, 13, 13)
    s_next = f["m"][-1]; S_next = oracle.from_cm(f["C"][-1], 13, 13)
    for t in range(59, 50, -1):
        C = oracle.from_cm(f["C"][t], 13, 13); R1 = oracle.from_cm(f["R"][t + 1], 13, 13)
        J = C @ G.T @ np.linalg.inv(R1)
        s_next = f["m"][t] + J @ (s_next - f["a"][t + 1])
        S_next = C - J @ (R1 - S_next) @ J.T
        np.testing.assert_allclose(txt["s"][t], s_next, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(oracle.from_cm(txt["S"][t], 13, 13), S_next, rtol=1e-8, atol=1e-10)


def test_block_diagonal_consistency():
    """Filtering two series as one |*| model equals two independent filters (SURVEY 8c ii)."""
    times = np.arange(1, 31, dtype=np.float64)
    rng = np.random.default_rng(11)
    y = rng.standard_normal((30, 2)).cumsum(axis=0)
    one = materialise(Dlm.polynomial(2), times)
    two = materialise(Dlm.polynomial(2) * Dlm.polynomial(2), times)
    p1 = DlmParameters([[2.0]], np.diag([0.5, 0.1]), [0.0, 0.0], np.eye(2))
    p2 = p1 * p1
    big = oracle.kf_filter(_omodel(two), p2.v, p2.w, p2.m0, p2.c0, y)
    sm_big = oracle.smoother(_omodel(two), big, compat_q1=False)
    for k in range(2):
        small = oracle.kf_filter(_omodel(one), p1.v, p1.w, p1.m0, p1.c0, y[:, k:k + 1])
        sm = oracle.smoother(_omodel(one), small, compat_q1=False)
        np.testing.assert_allclose(big["m"][:, 2 * k:2 * k + 2], small["m"], atol=1e-12)
        np.testing.assert_allclose(sm_big["s"][:, 2 * k:2 * k + 2], sm["s"], atol=1e-11)
        Cb = big["C"].reshape(31, 4, 4).transpose(0, 2, 1)[:, 2 * k:2 * k + 2, 2 * k:2 * k + 2]
        Cs = small["C"].reshape(31, 2, 2).transpose(0, 2, 1)
        np.testing.assert_allclose(Cb, Cs, atol=1e-12)


def test_backward_sample_moments_and_factors():
    mod, mat, p, y = seasonal_setup(T=40)
    om = _omodel(mat)
    f = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    z = oracle.normals(1234, 0, 41, 13)
    assert abs(z.mean()) < 0.2 and 0.8 < z.std() < 1.2
    e = oracle.backward_sample(om, p.w, f, z, factor="eig")
    c = oracle.backward_sample(om, p.w, f, z, factor="chol")
    assert e["rc"] == 0 and c["rc"] == 0
    G = oracle.from_cm(mat.G, 13, 13)
    for t in (39, 20, 0):
        # H == C - C G^T R^-1 G C (the algebraically equal form the engine uses)
        C = oracle.from_cm(f["C"][t], 13, 13); R1 = oracle.from_cm(f["R"][t + 1], 13, 13)
        H = C - C @ G.T @ np.linalg.solve(R1, G @ C)
        np.testing.assert_allclose(oracle.from_cm(e["H"][t], 13, 13), H, rtol=1e-8, atol=1e-10)
    # with z = 0 both factors return the conditional-mean path; it is NOT the smoothed mean
    z0 = np.zeros_like(z)
    e0 = oracle.backward_sample(om, p.w, f, z0, factor="eig")
    c0 = oracle.backward_sample(om, p.w, f, z0, factor="chol")
    np.testing.assert_allclose(e0["theta"], c0["theta"], atol=1e-12)
    sm = oracle.smoother(om, f, compat_q1=False)
    np.testing.assert_allclose(e0["theta"], sm["s"], rtol=1e-8, atol=1e-9)  # zero-noise path == RTS mean


def test_gibbs_stats_match_numpy():
    mod, mat, p, y = seasonal_setup(T=50, missing=0.2)
    om = _omodel(mat)
    rng = np.random.default_rng(5)
    theta = rng.standard_normal((51, 13))
    st = oracle.gibbs_stats(om, y, theta, want_outer=True)
    G = oracle.from_cm(mat.G, 13, 13); F = mat.F.reshape(13)
    obs = ~np.isnan(y[:, 0])
    res = y[obs, 0] - theta[1:][obs] @ F
    np.testing.assert_allclose(st["ssy"][0], (res ** 2).sum(), rtol=1e-12)
    assert st["n"][0] == obs.sum()
    diff = theta[1:] - theta[:-1] @ G.T
    np.testing.assert_allclose(st["ss"], (diff ** 2).sum(axis=0), rtol=1e-12)
    np.testing.assert_allclose(oracle.from_cm(st["outer"], 13, 13), diff.T @ diff, rtol=1e-11, atol=1e-11)


def test_svd_filter_d13_and_q2():
    mod, mat, p, y = seasonal_setup(T=40, missing=0.1)
    om = _omodel(mat)
    kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    good = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y, raw_w_q2=False)
    np.testing.assert_allclose(good["m"], kf["m"], rtol=1e-7, atol=1e-8)
    for t in (1, 17, 40):
        uc = oracle.from_cm(good["uc"][t], 13, 13)
        np.testing.assert_allclose(uc @ np.diag(good["dc"][t] ** 2) @ uc.T,
                                   oracle.from_cm(kf["C"][t], 13, 13), rtol=1e-7, atol=1e-8)
    lit = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y, raw_w_q2=True)
    assert np.abs(lit["m"] - kf["m"]).max() > 1e-3  # Q2: raw W is treated as sqrt(W) when W != I


def test_svd_sampler_zero_noise_matches_rts():
    mod, mat, p, y = seasonal_setup(T=30)
    om = _omodel(mat)
    sf = oracle.svd_filter(om, p.v, p.w, p.m0, p.c0, y, raw_w_q2=False)
    kf = oracle.kf_filter(om, p.v, p.w, p.m0, p.c0, y)
    sm = oracle.smoother(om, kf, compat_q1=False)
    out = oracle.svd_backward_sample(om, p.w, sf, np.zeros((31, 13)), literal_q9=False)
    np.testing.assert_allclose(out["theta"], sm["s"], rtol=1e-6, atol=1e-7)
    # Q9: the literal reference form (sqrt(W) where sqrt(W)^-1 is needed) differs when W != I ...
    lit = oracle.svd_backward_sample(om, p.w, sf, np.zeros((31, 13)), literal_q9=True)
    assert np.abs(lit["theta"] - sm["s"]).max() > 1e-2
    # ... and coincides when W = I, which is all the reference's own test exercises
    p1 = DlmParameters(p.v, np.eye(13), p.m0, p.c0)
    sf1 = oracle.svd_filter(om, p1.v, p1.w, p1.m0, p1.c0, y, raw_w_q2=True)
    kf1 = oracle.kf_filter(om, p1.v, p1.w, p1.m0, p1.c0, y)
    sm1 = oracle.smoother(om, kf1, compat_q1=False)
    lit1 = oracle.svd_backward_sample(om, p1.w, sf1, np.zeros((31, 13)), literal_q9=True)
    np.testing.assert_allclose(lit1["theta"], sm1["s"], rtol=1e-6, atol=1e-7)


def test_loglik_matches_scipy_and_scalar_closed_form():
    """oracle_loglik = sum of KalmanFilter.conditionalLikelihood (KalmanFilter.scala:138-153): checked against
    scipy.stats on the forecasts, observed components only; all-missing steps contribute nothing."""
    from scipy.stats import multivariate_normal, norm
    rng = np.random.default_rng(11)
    mod = Dlm.polynomial(2) * Dlm.polynomial(1)
    mat = materialise(mod, np.arange(1, 31, dtype=np.float64))
    B = rng.standard_normal((2, 2))
    p = DlmParameters(B @ B.T + np.eye(2), np.eye(3) * 0.4, np.zeros(3), np.eye(3) * 2)
    y = rng.standard_normal((mat.T, 2)).cumsum(axis=0)
    y[rng.random(y.shape) < 0.25] = np.nan
    y[4] = np.nan
    M = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
    f = oracle.kf_filter(M, p.v, p.w, p.m0, p.c0, y)
    want = 0.0
    for t in range(mat.T):
        ob = ~np.isnan(y[t])
        if not ob.any():
            continue
        Q = f["Q"][t + 1].reshape(2, 2).T[np.ix_(ob, ob)]
        mu = f["f"][t + 1][ob]
        want += norm(mu[0], np.sqrt(Q[0, 0])).logpdf(y[t][ob][0]) if ob.sum() == 1 else multivariate_normal(mu, Q).logpdf(y[t][ob])
    assert oracle.loglik(M, f, y) == pytest.approx(want, rel=1e-12, abs=1e-10)


def test_likelihood_q7_on_the_reference_s_filtered_means(golden_dir):
    """KalmanFilter.likelihood (KalmanFilter.scala:299-306) = KalmanFilter.logLikelihood (:175-183) over the filtered means:
    sum_t log N(m_t; g(dt) m_{t-1}, W dt).  Pinned by the reference's own output: the filtered means of
    first_order_dlm_filtered.csv (config C1: w = 3, dt = 1, g = 1) put through scipy's normal density, and -- dense W, d = 3,
    irregular grid, missing data -- through scipy's multivariate normal."""
    from scipy.stats import multivariate_normal, norm
    mat, p, y = _c1(golden_dir)
    M = _omodel(mat)
    f = oracle.kf_filter(M, p.v, p.w, p.m0, p.c0, y)
    _, rows = _read_csv(os.path.join(golden_dir, "first_order_dlm_filtered.csv"))
    m_ref = np.array([float(r[1]) for r in rows])                      # the reference's m_0 .. m_T
    want = norm(m_ref[:-1], np.sqrt(3.0)).logpdf(m_ref[1:]).sum()
    assert oracle.likelihood_q7(M, f, p.w) == pytest.approx(want, rel=1e-12)
    rng = np.random.default_rng(12)
    mod = Dlm.polynomial(1) + Dlm.seasonal(24, 1)        # G depends on dt: several tables
    times = np.cumsum(np.array([1, 2, 1, 0.5, 3] * 6, dtype=np.float64))
    mat = materialise(mod, times)
    A = rng.standard_normal((3, 3))
    p = DlmParameters([[0.7]], A @ A.T + 0.3 * np.eye(3), rng.standard_normal(3), np.eye(3))
    y = rng.standard_normal((mat.T, 1)).cumsum(axis=0)
    y[rng.random(y.shape) < 0.2] = np.nan
    M = _omodel(mat)
    f = oracle.kf_filter(M, p.v, p.w, p.m0, p.c0, y)
    assert mat.n_g > 1
    want = 0.0
    for t in range(mat.T):
        G = oracle.from_cm(np.asarray(mat.G)[mat.g_index[t] * 9:(mat.g_index[t] + 1) * 9], 3, 3)
        want += multivariate_normal(G @ f["m"][t], p.w * mat.dt[t]).logpdf(f["m"][t + 1])
    assert oracle.likelihood_q7(M, f, p.w) == pytest.approx(want, rel=1e-11)
    # it is NOT the prediction-error likelihood
    assert abs(oracle.likelihood_q7(M, f, p.w) - oracle.loglik(M, f, y)) > 1.0


def test_ar1_filter_reproduces_reference_csv(golden_dir):
    """FilterAr.filterUnivariate on examples/data/ar_dlm.csv with SvParameters(0.8, 1.0, 0.3), v = 0.5
    (examples/src/main/scala/dlm/ar.scala:47-61) reproduces ar_dlm_filtered.csv (5001 rows)."""
    obs = np.loadtxt(os.path.join(golden_dir, "ar_dlm.csv"), delimiter=",", skiprows=1)
    want = np.loadtxt(os.path.join(golden_dir, "ar_dlm_filtered.csv"), delimiter=",", skiprows=1)
    f = oracle.ar1_filter(obs[:, 1], 0.5, 0.8, 1.0, 0.3)
    assert want.shape == (5001, 3)
    np.testing.assert_allclose(f["m"], want[:, 1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(f["c"], want[:, 2], rtol=0, atol=1e-13)
    # backward sampler: with z = 0 it is the conditional-mean recursion; check one step by hand
    th = oracle.ar1_backward_sample(f, 0.8, np.zeros(5002 - 1))
    t = 4000
    assert th[t] == pytest.approx(f["m"][t] + f["c"][t] * 0.8 / f["r"][t + 1] * (th[t + 1] - f["a"][t + 1]), rel=1e-14)


def test_ou_filter_closed_forms():
    """oracle_ou_filter follows FilterOu.stepUni literally: dt-dependent AR coefficient and innovation variance, the
    literal c0 = sigma^2 and first dt = 0; on a unit grid it equals the AR(1) filter with phi' = exp(-phi),
    sigma'^2 = sigma^2 (1 - exp(-2 phi)) / (2 phi) from the second record on."""
    rng = np.random.default_rng(2)
    T = 40
    y = rng.standard_normal(T).cumsum() * 0.2
    y[[3, 17]] = np.nan
    phi, mu, sig = 0.3, 0.5, 0.4
    f = oracle.ou_filter(np.arange(1.0, T + 1), y, 0.7, phi, mu, sig)
    assert f["c"][0] == pytest.approx(sig * sig) and f["r"][1] == pytest.approx(sig * sig)   # first dt = 0
    ph, s2 = np.exp(-phi), sig * sig * (1 - np.exp(-2 * phi)) / (2 * phi)
    m, c = f["m"][1], f["c"][1]
    for t in range(1, T):
        a, r = mu + ph * (m - mu), ph * ph * c + s2
        if np.isnan(y[t]):
            m, c = a, r
        else:
            k = r / (r + 0.7); m, c = a + k * (y[t] - a), k * 0.7
        assert f["m"][t + 1] == pytest.approx(m, rel=1e-13) and f["c"][t + 1] == pytest.approx(c, rel=1e-13)


def test_simulate_moments_and_consistency():
    """oracle.simulate (Dlm.simulateRegular, Dlm.scala:245-292): the recursion x_t = G x_{t-1} + L_w sqrt(dt) z,
    y_t = F^T x_t + L_v z holds record by record against the Philox normals, and the moments are the model's."""
    mod = Dlm.polynomial(2)
    mat = materialise(mod, np.cumsum([1.0, 2.0, 0.5, 1.0, 3.0]))
    W = np.array([[0.5, 0.1], [0.1, 0.3]]); V = np.array([[0.7]]); C0 = np.array([[2.0, 0.3], [0.3, 1.0]])
    x, y = oracle.simulate(_omodel(mat), V, W, np.array([1.0, -1.0]), C0, 9, 4)
    z = oracle.normals(9, 4, mat.T + 1, 3)       # [T+1][d+p]
    np.testing.assert_allclose(x[0], np.array([1.0, -1.0]) + np.linalg.cholesky(C0) @ z[0, :2], atol=1e-13)
    Lw, Lv = np.linalg.cholesky(W), np.linalg.cholesky(V)
    for t in range(mat.T):
        G = np.asarray(mat.G).reshape(-1, 2, 2)[mat.g_index[t] if mat.g_index is not None else 0].T   # column-major
        xn = G @ x[t] + np.sqrt(mat.dt[t]) * (Lw @ z[t + 1, :2])
        np.testing.assert_allclose(x[t + 1], xn, atol=1e-12)
        Ft = mat.F.reshape(-1)[t * mat.f_stride:t * mat.f_stride + 2]
        np.testing.assert_allclose(y[t], Ft @ xn + Lv @ z[t + 1, 2:3], atol=1e-12)
    ll = materialise(Dlm.polynomial(1), np.arange(1, 3, dtype=np.float64))
    ys = np.array([oracle.simulate(_omodel(ll), [[2.0]], [[0.5]], [1.0], [[3.0]], 1, n)[1][0, 0] for n in range(4000)])
    assert abs(ys.mean() - 1.0) < 0.15 and abs(ys.var() - 5.5) < 0.5


def test_svd_filter_with_variance_streams_is_the_kalman_filter():
    """V_t / W_t streams on the oracle's SVD path (DlmFsvSystem.ffbsSvd, DlmFsv.ffbsSvd): with the square roots taken per
    step the SVD filter's U D^2 U^T must equal the standard filter's C under the same streams (the identity the
    reference's SvdFilterTest checks for constant parameters, core/src/test/scala/SvdFilter.scala:145-157)."""
    from bayesian_dlms_amd.dlm import Dlm, materialise
    rng = np.random.default_rng(5)
    mod = Dlm.polynomial(2) + Dlm.seasonal(12, 1)
    T, d = 25, 4
    mat = materialise(mod, np.arange(1, T + 1, dtype=np.float64))
    M = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
    A = rng.standard_normal((T, d, d)) * 0.2
    W = np.eye(d)[None] * 0.3 + A @ A.transpose(0, 2, 1)
    V = 0.5 + rng.random((T, 1, 1))
    y = rng.standard_normal((T, 1)).cumsum(axis=0)
    y[[4, 11]] = np.nan
    sf = oracle.svd_filter(M, V, W, np.zeros(d), np.eye(d) * 2.0, y)
    kf = oracle.kf_filter(M, V, W, np.zeros(d), np.eye(d) * 2.0, y)
    np.testing.assert_allclose(sf["m"], kf["m"], rtol=1e-8, atol=1e-9)
    for t in range(T + 1):
        U = oracle.from_cm(sf["uc"][t], d, d)
        np.testing.assert_allclose(U @ np.diag(sf["dc"][t] ** 2) @ U.T, oracle.from_cm(kf["C"][t], d, d), rtol=1e-8, atol=1e-9)
    # the sampler with zero noise and per-step W reproduces the RTS mean of the same model
    z = np.zeros((T + 1, d))
    th = oracle.svd_backward_sample(M, W, sf, z, literal_q9=False)["theta"]
    np.testing.assert_allclose(th[T], kf["m"][T], rtol=1e-8, atol=1e-9)
    assert np.all(np.isfinite(th))


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """`make -C oracle asan` (ASan + UBSan build of the C restatement), then the filter / smoother / sampler / SVD /
    statistics entry points on the golden file and on a d = 13 model in a child process that loads that build: any
    out-of-bounds access or undefined behaviour aborts it."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    script = tmp_path / "run.py"
    script.write_text('''
import csv, os, sys
import numpy as np
sys.path.insert(0, os.environ["DLM_ROOT"])
import oracle
from bayesian_dlms_amd.dlm import Dlm, materialise
rows = list(csv.reader(open(os.path.join(os.environ["DLM_ROOT"], "tests", "golden", "first_order_dlm.csv"))))[1:]
y = np.array([float(r[1]) for r in rows])[:, None]
mat = materialise(Dlm.polynomial(1), np.array([float(r[0]) for r in rows]))
M = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
f = oracle.kf_filter(M, [[2.0]], [[3.0]], [0.0], [[10.0]], y)
s = oracle.smoother(M, f, compat_q1=True)
assert np.isfinite(s["S"]).all()
mod = Dlm.polynomial(1) + Dlm.seasonal(24, 6)
mat = materialise(mod, np.cumsum(np.array([1, 1, 2, 0, 3] * 8, dtype=float)) + 1)
M = oracle.Model(mat.d, mat.p, mat.T, mat.F, mat.G, mat.g_index, mat.dt, mat.f_stride)
rng = np.random.default_rng(0)
yy = rng.standard_normal((mat.T, 1)); yy[[3, 9]] = np.nan
W = np.diag(np.linspace(0.1, 0.5, 13))
f = oracle.kf_filter(M, [[1.0]], W, np.zeros(13), np.eye(13), yy)
oracle.smoother(M, f)
z = oracle.normals(3, 1, mat.T + 1, 13)
th = oracle.backward_sample(M, W, f, z, factor="chol")["theta"]
th2 = oracle.backward_sample(M, W, f, z, factor="eig")["theta"]
st = oracle.gibbs_stats(M, yy, th, want_outer=True)
sf = oracle.svd_filter(M, [[1.0]], W, np.zeros(13), np.diag(np.linspace(0.5, 2, 13)), yy)
oracle.svd_backward_sample(M, W, sf, z, literal_q9=False)
oracle.simulate(M, [[1.0]], W, np.zeros(13), np.eye(13), 5, 2)
oracle.filter_smooth_batch(2, M, np.array([[1.0]]), W, np.zeros(13), np.eye(13), np.stack([yy, yy]), want_out=True)
print("ASAN RUN OK")
''')
    env = dict(os.environ, DLM_ROOT=root, DLM_ORACLE_LIB=os.path.join(root, "oracle", "libdlm_oracle_asan.so"), LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ASAN RUN OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
