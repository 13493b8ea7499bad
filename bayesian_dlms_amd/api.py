"""Host-side mirror of the reference's operator interface for the hot path.

Same names, argument order and per-series semantics as the Scala objects, widened to a batch
(a `Vector[Vector[Data]]`, i.e. N series on one time grid) and backed by the HIP engine:

  KalmanFilter.filter_dlm(mod, ys, p)        KalmanFilter.scala:291-294 (+ `.filter`, Filter.scala:41-45)
  Smoothing.backwards_smoother(mod, kf)      Smoothing.scala:57-64
  Smoothing.ffbs_dlm(mod, ys, p)             Smoothing.scala:173-180
  Smoothing.sample_dlm(mod, kf, w)           Smoothing.scala:164-165
  SvdFilter.filter_dlm(mod, ys, p)           SvdFilter.scala:158-161
  SvdSampler.ffbs_dlm(mod, ys, p)            SvdSampler.scala:79-82

`ys` is either one series (`Sequence[Data]`) or a batch (`Sequence[Sequence[Data]]`); results
come back as lists of light state objects whose fields are views into the engine's flat
output (the JVM-object explosion the survey warns about is avoided the same way in the
Scala shim: lazily built `KfState`s over a DoubleBuffer).

Every function needs an `Engine`; there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _lib
from .dlm import Data, Dlm, DlmParameters, MaterialisedModel, materialise
from .engine import Engine


@dataclass
class KfState:
    """KfState (KalmanFilter.scala:22-30)."""
    time: float
    mt: np.ndarray
    ct: np.ndarray
    at: Optional[np.ndarray]
    rt: Optional[np.ndarray]
    ft: Optional[np.ndarray]
    qt: Optional[np.ndarray]


@dataclass
class SmoothingState:
    """Smoothing.SmoothingState (Smoothing.scala:18-22)."""
    time: float
    mean: np.ndarray
    covariance: np.ndarray


@dataclass
class SamplingState:
    """SamplingState (Smoothing.scala:10-15): time, sample (+ the conditional mean/cov)."""
    time: float
    sample: np.ndarray
    mean: Optional[np.ndarray] = None
    cov: Optional[np.ndarray] = None


@dataclass
class SvdState:
    """SvdState (SvdFilter.scala:7-14): C = uc diag(dc^2) uc^T."""
    time: float
    mt: np.ndarray
    dc: np.ndarray
    uc: np.ndarray


def _is_batch(ys) -> bool:
    return len(ys) > 0 and not isinstance(ys[0], Data)


def pack_observations(ys):
    """Vector[Data] or Vector[Vector[Data]] -> (times [T], y [N][T][p], batched?).  All series
    of a batch must share their observation times (irregular grids are fine)."""
    batch = ys if _is_batch(ys) else [ys]
    if len(batch) == 0 or len(batch[0]) == 0:
        raise ValueError("empty observation vector (the reference throws on t0.get, KalmanFilter.scala:116-117)")
    times = np.array([d.time for d in batch[0]], dtype=np.float64)
    y = np.stack([np.stack([np.asarray(d.observation, dtype=np.float64) for d in s]) for s in batch])
    for s in batch[1:]:
        if not np.array_equal(np.array([d.time for d in s]), times):
            raise ValueError("all series of a batch must share one time grid")
    return times, y, _is_batch(ys)


def _mat(a, d, c=None):
    """column-major flat -> (d x c) ndarray"""
    c = d if c is None else c
    return np.asarray(a).reshape(c, d).T


def _times_with_init(times):
    return np.concatenate([[times.min() - 1.0], times])


class KalmanFilter:
    @staticmethod
    def filter_dlm(mod: Dlm, ys, p, engine: Engine, *, keep_init: bool = False, flags: int = 0):
        """KalmanFilter.filterDlm: one KfState per observation (`filterTraverse` drops the initial
        state); `keep_init=True` is `.filter` (T+1 states), the input backwards_smoother wants."""
        times, y, batched = pack_observations(ys)
        mat = materialise(mod, times)
        out = engine.filter(mat, p, y, want_prior=True, want_fq=True, flags=flags)
        tt = _times_with_init(times)
        d, q = mat.d, mat.p
        res = []
        for n in range(y.shape[0]):
            states = []
            for t in range(0 if keep_init else 1, mat.T + 1):
                f, pr, fq = out["filt"][n, t], out["prior"][n, t], out["fq"][n, t]
                states.append(KfState(float(tt[t]), f[:d], _mat(f[d:], d), pr[:d], _mat(pr[d:], d),
                                      None if t == 0 else fq[:q], None if t == 0 else _mat(fq[q:], q)))
            res.append(states)
        return res if batched else res[0]

    @staticmethod
    def filter(mod, ys, p, engine, **kw):
        return KalmanFilter.filter_dlm(mod, ys, p, engine, keep_init=True, **kw)

    @staticmethod
    def log_likelihood(mod: Dlm, ys, p, engine: Engine, *, flags: int = 0):
        """Sum over the series of KalmanFilter.conditionalLikelihood(f_t, Q_t, y_t) (KalmanFilter.scala:138-153): the
        prediction-error log-likelihood log p(y_{1:T} | V, W).  `p` may be one DlmParameters or one per series (a bank
        of parameter sets evaluated in one launch, as MetropolisHastings.dlm / RaoBlackwellFilter.kfStep need)."""
        times, y, batched = pack_observations(ys)
        mat = materialise(mod, times)
        ll = np.asarray(engine.loglik(mat, p, y, flags=flags)["loglik"])
        return ll if batched else float(ll[0])

    @staticmethod
    def likelihood(mod: Dlm, ys, p, engine: Engine, *, flags: int = 0):
        """KalmanFilter.likelihood(mod, ys)(p) as the reference writes it (KalmanFilter.scala:299-306) -- the function
        MetropolisHastings.dlm evaluates (MetropolisHastings.scala:134, :205): filter, then the TRANSITION density of the filtered
        means, sum_t log N(m_t; g(dt_t) m_{t-1}, W dt_t) (KalmanFilter.logLikelihood, :175-183).  For the
        prediction-error log-likelihood log p(y | V, W) use log_likelihood."""
        from . import _lib
        return KalmanFilter.log_likelihood(mod, ys, p, engine, flags=flags | _lib.OPT_LOGLIK_LITERAL_Q7)


def _records_from_states(kf_states: Sequence[Sequence[KfState]], d: int) -> np.ndarray:
    N, T1 = len(kf_states), len(kf_states[0])
    rec = np.empty((N, T1, d + d * d))
    for n, s in enumerate(kf_states):
        for t, k in enumerate(s):
            rec[n, t, :d] = k.mt
            rec[n, t, d:] = np.asarray(k.ct).T.reshape(-1)
    return rec


class Smoothing:
    @staticmethod
    def backwards_smoother(mod: Dlm, kf_states, p, engine: Engine, *, compat_q1: bool = False):
        """Smoothing.backwardsSmoother(mod)(kfStates); kfStates are the T+1 states of `.filter`.
        `p` supplies W (the reference reads R_{t+1} from the states; the engine recomputes it).
        compat_q1=True reproduces the literal Smoothing.scala:44 covariance (no transpose)."""
        batch = kf_states if isinstance(kf_states[0], (list, tuple)) else [kf_states]
        d = batch[0][0].mt.shape[0]
        times = np.array([k.time for k in batch[0][1:]], dtype=np.float64)
        mat = materialise(mod, times)
        rec = _records_from_states(batch, d)
        out = engine.smooth(mat, p, rec, flags=_lib.OPT_SMOOTHER_COMPAT_Q1 if compat_q1 else 0)
        tt = _times_with_init(times)
        res = [[SmoothingState(float(tt[t]), out["smooth"][n, t, :d], _mat(out["smooth"][n, t, d:], d))
                for t in range(mat.T + 1)] for n in range(len(batch))]
        return res if isinstance(kf_states[0], (list, tuple)) else res[0]

    @staticmethod
    def filter_smooth(mod: Dlm, ys, p, engine: Engine):
        """Fused KalmanFilter.filter + backwardsSmoother (the engine's headline path)."""
        times, y, batched = pack_observations(ys)
        mat = materialise(mod, times)
        out = engine.filter_smooth(mat, p, y)
        tt = _times_with_init(times); d = mat.d
        sm = [[SmoothingState(float(tt[t]), out["smooth"][n, t, :d], _mat(out["smooth"][n, t, d:], d))
               for t in range(mat.T + 1)] for n in range(y.shape[0])]
        return sm if batched else sm[0]

    @staticmethod
    def ffbs_dlm(mod: Dlm, ys, p, engine: Engine, *, seed: int = 0, series_offset: int = 0):
        """Smoothing.ffbsDlm: one draw of the state path (T+1 SamplingStates) per series.
        Unlike the reference (Rand.always of one eager draw, unseedable) the draw
        is a pure function of (seed, series index)."""
        times, y, batched = pack_observations(ys)
        mat = materialise(mod, times)
        out = engine.ffbs(mat, p, y, seed=seed, series_offset=series_offset, want_cond=True, want_stats=False)
        tt = _times_with_init(times); d = mat.d
        res = [[SamplingState(float(tt[t]), out["theta"][n, t], out["cond"][n, t, :d], _mat(out["cond"][n, t, d:], d))
                for t in range(mat.T + 1)] for n in range(y.shape[0])]
        return res if batched else res[0]

    @staticmethod
    def sample_dlm(mod: Dlm, kf_states, p, engine: Engine, *, seed: int = 0):
        """Smoothing.sampleDlm(mod, filtered, w): backward sampling from existing filter states."""
        batch = kf_states if isinstance(kf_states[0], (list, tuple)) else [kf_states]
        d = batch[0][0].mt.shape[0]
        times = np.array([k.time for k in batch[0][1:]], dtype=np.float64)
        mat = materialise(mod, times)
        rec = _records_from_states(batch, d)
        y = np.full((len(batch), mat.T, mat.p), np.nan)
        out = engine.ffbs(mat, p, y, seed=seed, filt=rec, want_stats=False)
        tt = _times_with_init(times)
        res = [[SamplingState(float(tt[t]), out["theta"][n, t]) for t in range(mat.T + 1)] for n in range(len(batch))]
        return res if isinstance(kf_states[0], (list, tuple)) else res[0]


class SvdFilter:
    @staticmethod
    def filter_dlm(mod: Dlm, ys, p, engine: Engine, *, literal_q2: bool = False):
        """SvdFilter.filterDlm.  literal_q2=True passes the raw W where sqrt(W) is expected, as
        the reference's filterDlm does (SvdFilter.scala:158-161)."""
        times, y, batched = pack_observations(ys)
        mat = materialise(mod, times)
        out = engine.svd_filter(mat, p, y, flags=_lib.OPT_SVD_RAW_W_Q2 if literal_q2 else 0)
        d = mat.d
        res = [[SvdState(float(times[t - 1]), out["svd"][n, t, :d], out["svd"][n, t, d:2 * d], _mat(out["svd"][n, t, 2 * d:], d))
                for t in range(1, mat.T + 1)] for n in range(y.shape[0])]
        return res if batched else res[0]


class SvdSampler:
    @staticmethod
    def ffbs_dlm(mod: Dlm, ys, p, engine: Engine, *, seed: int = 0, literal: bool = False):
        """SvdSampler.ffbsDlm.  literal=True reproduces Q2 and Q9 (raw W in the time update,
        sqrt(W) where sqrt(W)^-1 is needed in the backward step)."""
        times, y, batched = pack_observations(ys)
        mat = materialise(mod, times)
        flags = (_lib.OPT_SVD_RAW_W_Q2 | _lib.OPT_SVD_SAMPLER_Q9) if literal else 0
        out = engine.svd_ffbs(mat, p, y, seed=seed, flags=flags, want_stats=False)
        tt = _times_with_init(times)
        res = [[SamplingState(float(tt[t]), out["theta"][n, t]) for t in range(mat.T + 1)] for n in range(y.shape[0])]
        return res if batched else res[0]


@dataclass
class SvParameters:
    """SvParameters(phi, mu, sigmaEta) of the AR(1) log-volatility (StochasticVolatility.scala)."""
    phi: float
    mu: float
    sigma_eta: float


class FilterAr:
    """FilterAr (FilterAr.scala:15-82) for a batch of scalar series: ys [N][T] with NaN for None, vs the per-step
    observation variances ([N][T], [T] or a scalar), p one SvParameters or one per series."""

    @staticmethod
    def _sv(p):
        if isinstance(p, SvParameters):
            return np.array([p.phi, p.mu, p.sigma_eta])
        return np.array([[q.phi, q.mu, q.sigma_eta] for q in p])

    @staticmethod
    def filter_univariate(ys, vs, p, engine: Engine):
        """filterUnivariate: (m_t, c_t) for t = 0..T, record 0 = (mu, sigma_eta^2 / (1 - phi^2))."""
        out = engine.ar1_ffbs(np.atleast_2d(np.asarray(ys, dtype=np.float64)), vs, FilterAr._sv(p), want_theta=False)
        f = np.asarray(out["filt"])
        return f[..., 0], f[..., 1]

    @staticmethod
    def ffbs(p, ys, vs, engine: Engine, *, seed: int = 0, series_offset: int = 0, z=None):
        """ffbs: one draw of the state path per series (T+1 values, the first at t0 - 1)."""
        out = engine.ar1_ffbs(np.atleast_2d(np.asarray(ys, dtype=np.float64)), vs, FilterAr._sv(p), z=z, seed=seed,
                              series_offset=series_offset, want_filt=False)
        return np.asarray(out["theta"])


class FilterOu:
    """FilterOu (FilterOu.scala:7-79): the Ornstein-Uhlenbeck state on an irregular grid `times` [T] shared by the batch;
    p.phi > 0 is the mean-reversion rate.  Literal reference behaviour is kept: c0 = sigma^2 and a first dt of 0."""

    @staticmethod
    def filter_univariate(times, ys, vs, p, engine: Engine):
        out = engine.ar1_ffbs(np.atleast_2d(np.asarray(ys, dtype=np.float64)), vs, FilterAr._sv(p), want_theta=False, times=times)
        f = np.asarray(out["filt"])
        return f[..., 0], f[..., 1]

    @staticmethod
    def ffbs(p, times, ys, vs, engine: Engine, *, seed: int = 0, series_offset: int = 0, z=None):
        out = engine.ar1_ffbs(np.atleast_2d(np.asarray(ys, dtype=np.float64)), vs, FilterAr._sv(p), z=z, seed=seed,
                              series_offset=series_offset, want_filt=False, times=times)
        return np.asarray(out["theta"])


def forecast(mod: Dlm, mt, ct, time: float, p: DlmParameters, engine: Engine, steps: int):
    """Dlm.forecast (Dlm.scala:320-338) for a batch of filtering distributions: mt [N][d], ct [N][d][d] (the posterior at
    `time`) -> (times [steps], f [N][steps][p], Q [N][steps][p][p]).  Element 0 is the one-step prediction at `time`
    itself (no advance, as the reference's Stream starts), element k advances k unit steps: stepForecast (:296-307) is
    the Kalman filter without an observation, so elements 1.. come from dlm_filter_batch on all-missing data with the
    per-series (mt, ct) as initial state."""
    mt = np.atleast_2d(np.asarray(mt, dtype=np.float64))
    ct = np.asarray(ct, dtype=np.float64).reshape(mt.shape[0], mt.shape[1], mt.shape[1])
    N, d = mt.shape
    times = time + np.arange(steps, dtype=np.float64)
    F0 = np.asarray(mod.f(time), dtype=np.float64)                      # d x p, used as F^T
    q = F0.shape[1]
    f = np.empty((N, steps, q)); Q = np.empty((N, steps, q, q))
    v0 = p.v if p.v.ndim == 2 else p.v[0]
    f[:, 0] = mt @ F0
    Q[:, 0] = np.einsum("dp,ndD,Dq->npq", F0, ct, F0) + v0
    if steps > 1:
        mat = materialise(mod, times[1:])                               # first increment = 1: time -> time + 1
        params = [DlmParameters(p.v, p.w, mt[n], ct[n]) for n in range(N)]
        out = engine.filter(mat, params, np.full((N, steps - 1, q), np.nan), want_fq=True)
        fq = np.asarray(out["fq"])[:, 1:]
        f[:, 1:] = fq[..., :q]
        Q[:, 1:] = np.transpose(fq[..., q:].reshape(N, steps - 1, q, q), (0, 1, 3, 2))
    return times, f, Q


def simulate(mod: Dlm, times, p: DlmParameters, engine: Engine, n_series: int = 1, *, seed: int = 0, series_offset: int = 0):
    """Dlm.simulateRegular / Dlm.simulate (Dlm.scala:245-292) for a batch of independent realisations on the device
    (dlm_simulate_batch): `times` are the observation times, the first increment is times[1] - times[0] like the
    filter's.  Returns (x [N][T+1][d] with record 0 the initial state, y [N][T][p])."""
    out = engine.simulate(materialise(mod, times), p, n_series, seed=seed, series_offset=series_offset)
    return np.asarray(out["x"]), np.asarray(out["y"])
