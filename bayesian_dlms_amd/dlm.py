"""Host-side mirror of the reference's model layer (`Dlm`, `DlmParameters`, `Data`).

Reference: core/src/main/scala/dlm/model/Dlm.scala
  Dlm(f, g)                 :14-31     composeModels (|+|) :107-111
  DlmParameters(v,w,m0,c0)  :36-89     outerSumModel (|*|) :117-122
  Data(time, observation)   :94        polynomial :139-153, regression :159-169,
  rotationMatrix :190-192, blockDiagonal :197-208, seasonalG :213-221,
  angle :226-228, seasonal :236-243.

The reference model is two closures, `f: time => d x p` and `g: dt => d x d`.  The
engine cannot call closures per timestep across a C ABI, so `materialise` evaluates
them once on the host into flat column-major tables (the layout Breeze's
`DenseMatrix.data` has) that the C ABI (`include/dlm_engine.h`) takes.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Optional, Sequence

import numpy as np


def block_diagonal(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Dlm.blockDiagonal (Dlm.scala:197-208)."""
    a = np.atleast_2d(a); b = np.atleast_2d(b)
    out = np.zeros((a.shape[0] + b.shape[0], a.shape[1] + b.shape[1]))
    out[: a.shape[0], : a.shape[1]] = a
    out[a.shape[0]:, a.shape[1]:] = b
    return out


def rotation_matrix(theta: float) -> np.ndarray:
    """Dlm.rotationMatrix (Dlm.scala:190-192)."""
    return np.array([[math.cos(theta), -math.sin(theta)], [math.sin(theta), math.cos(theta)]])


def angle(period: int, dt: float) -> float:
    """Dlm.angle (Dlm.scala:226-228); Scala `%` on doubles is fmod."""
    return 2.0 * math.pi * math.fmod(dt, period) / period


def seasonal_g(period: int, harmonics: int, dt: float) -> np.ndarray:
    """Dlm.seasonalG (Dlm.scala:213-221)."""
    g = rotation_matrix(1 * angle(period, dt))
    for h in range(2, harmonics + 1):
        g = block_diagonal(g, rotation_matrix(h * angle(period, dt)))
    return g


@dataclass(frozen=True)
class Dlm:
    """A DLM: f(time) -> d x p observation matrix (used as F^T), g(dt) -> d x d."""

    f: Callable[[float], np.ndarray]
    g: Callable[[float], np.ndarray]

    def compose(self, y: "Dlm") -> "Dlm":
        """`|+|` (Dlm.composeModels, Dlm.scala:107-111): richer univariate model."""
        x = self
        return Dlm(lambda t: np.vstack([x.f(t), y.f(t)]), lambda dt: block_diagonal(x.g(dt), y.g(dt)))

    def outer(self, y: "Dlm") -> "Dlm":
        """`|*|` (Dlm.outerSumModel, Dlm.scala:117-122): multivariate model."""
        x = self
        return Dlm(lambda t: block_diagonal(x.f(t), y.f(t)), lambda dt: block_diagonal(x.g(dt), y.g(dt)))

    __add__ = compose
    __mul__ = outer

    @staticmethod
    def polynomial(order: int) -> "Dlm":
        """Dlm.polynomial (Dlm.scala:139-153)."""
        def f(t):
            e = np.zeros((order, 1)); e[0, 0] = 1.0
            return e

        def g(dt):
            return np.eye(order) + np.eye(order, k=1)
        return Dlm(f, g)

    @staticmethod
    def seasonal(period: int, harmonics: int) -> "Dlm":
        """Dlm.seasonal (Dlm.scala:236-243)."""
        def f(t):
            e = np.zeros((2 * harmonics, 1)); e[0::2, 0] = 1.0
            return e
        return Dlm(f, lambda dt: seasonal_g(period, harmonics, dt))

    @staticmethod
    def regression(x: Sequence[np.ndarray]) -> "Dlm":
        """Dlm.regression (Dlm.scala:159-169): F_t = [1, x_t]; G = I_2 as in the reference."""
        def f(t):
            xi = np.asarray(x[int(t) - 1], dtype=np.float64).reshape(-1)
            return np.concatenate([[1.0], xi]).reshape(-1, 1)
        return Dlm(f, lambda dt: np.eye(2))

    @staticmethod
    def autoregressive(*phi: float) -> "Dlm":
        """Dlm.autoregressive (Dlm.scala:176-185).  The reference builds g as a
        (len(phi) x 1) column, which is only a valid system matrix for AR(1); we
        materialise diag(phi), identical for the AR(1) case the reference uses."""
        k = len(phi)

        def f(t):
            e = np.zeros((k, 1)); e[0, 0] = 1.0
            return e
        return Dlm(f, lambda dt: np.diag(np.asarray(phi, dtype=np.float64)))


@dataclass
class DlmParameters:
    """DlmParameters(v, w, m0, c0) (Dlm.scala:36-39).  `v` / `w` may also be [T] streams of matrices (shape
    (T, p, p) / (T, d, d)): V_t of observation t and W_t of the transition into it -- the per-step variances of
    StudentT.filter (StudentTGibbs.scala:100-136) and DlmFsvSystem.ffbs (DlmFsvSystem.scala:137-208)."""

    v: np.ndarray
    w: np.ndarray
    m0: np.ndarray
    c0: np.ndarray

    def __post_init__(self):
        self.v = np.asarray(self.v, dtype=np.float64)
        self.w = np.asarray(self.w, dtype=np.float64)
        if self.v.ndim != 3:
            self.v = np.atleast_2d(self.v)
        if self.w.ndim != 3:
            self.w = np.atleast_2d(self.w)
        self.m0 = np.atleast_1d(np.asarray(self.m0, dtype=np.float64))
        self.c0 = np.atleast_2d(np.asarray(self.c0, dtype=np.float64))

    def outer(self, y: "DlmParameters") -> "DlmParameters":
        """`|*|` on parameters (Dlm.outerSumParameters, Dlm.scala:127-134)."""
        return DlmParameters(block_diagonal(self.v, y.v), block_diagonal(self.w, y.w),
                             np.concatenate([self.m0, y.m0]), block_diagonal(self.c0, y.c0))

    __mul__ = outer

    def to_list(self):
        """diag(v), diag(w), m0, diag(c0) (DlmParameters.toList, Dlm.scala:82-83)."""
        return list(np.concatenate([np.diag(self.v), np.diag(self.w), self.m0, np.diag(self.c0)]))

    @staticmethod
    def from_list(v_dim: int, w_dim: int, l: Sequence[float]) -> "DlmParameters":
        """DlmParameters.fromList (Dlm.scala:74-80)."""
        l = list(l)
        return DlmParameters(np.diag(l[:v_dim]), np.diag(l[v_dim:v_dim + w_dim]),
                             np.array(l[v_dim + w_dim:v_dim + 2 * w_dim]),
                             np.diag(l[v_dim + 2 * w_dim:v_dim + 3 * w_dim]))

    def map(self, fn) -> "DlmParameters":
        g = np.vectorize(fn)
        return DlmParameters(g(self.v), g(self.w), g(self.m0), g(self.c0))


@dataclass
class Data:
    """Data(time, observation) (Dlm.scala:94); `None`/NaN entries are missing."""

    time: float
    observation: np.ndarray

    def __post_init__(self):
        obs = [np.nan if o is None else float(o) for o in np.atleast_1d(np.asarray(self.observation, dtype=object))]
        self.observation = np.asarray(obs, dtype=np.float64)


@dataclass
class MaterialisedModel:
    """Flat tables the C ABI takes (all fp64, matrices column-major)."""

    d: int
    p: int
    T: int
    F: np.ndarray          # [nF * d * p]
    f_stride: int          # 0 => time-invariant F
    G: np.ndarray          # [nG * d * d]
    n_g: int
    g_index: Optional[np.ndarray]   # int32 [T] or None (single G)
    dt: Optional[np.ndarray]        # fp64 [T] or None (all 1.0)
    times: np.ndarray      # fp64 [T]


def _cm(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.float64)
    return np.ascontiguousarray(a.T).reshape(-1)


def materialise(mod: Dlm, times: Sequence[float]) -> MaterialisedModel:
    """Evaluate the model closures on a shared time grid.

    The initial state sits at t0 - 1 (KalmanFilter.initialiseState,
    KalmanFilter.scala:112-118) so the first increment is times[0] - (min(times) - 1).
    """
    times = np.asarray(times, dtype=np.float64).reshape(-1)
    if times.size == 0:
        raise ValueError("empty observation vector (the reference throws on t0.get)")
    T = times.size
    prev = np.concatenate([[times.min() - 1.0], times[:-1]])
    dts = times - prev
    f0 = np.atleast_2d(np.asarray(mod.f(float(times[0])), dtype=np.float64))
    d, p = f0.shape
    Fs = [np.atleast_2d(np.asarray(mod.f(float(t)), dtype=np.float64)) for t in times]
    if all(np.array_equal(f0, f) for f in Fs):
        F = _cm(f0); f_stride = 0
    else:
        F = np.concatenate([_cm(f) for f in Fs]); f_stride = d * p
    uniq, inv = np.unique(dts, return_inverse=True)
    # one table entry per distinct G matrix (g may ignore dt, e.g. Dlm.polynomial)
    table, remap = [], []
    for u in uniq:
        gm = _cm(np.atleast_2d(mod.g(float(u))))
        for k, have in enumerate(table):
            if np.array_equal(have, gm):
                remap.append(k)
                break
        else:
            remap.append(len(table)); table.append(gm)
    g_index = np.asarray(remap, dtype=np.int32)[inv]
    single = len(table) == 1
    return MaterialisedModel(
        d=d, p=p, T=T, F=F, f_stride=f_stride, G=np.concatenate(table), n_g=len(table),
        g_index=None if single else g_index,
        dt=None if bool(np.all(dts == 1.0)) else dts.copy(), times=times)
