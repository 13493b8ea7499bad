"""Gibbs samplers for the DLM variances on top of the batched FFBS engine.

Mirrors GibbsSampling.sample / dinvGammaStep (Gibbs.scala:134-180) and GibbsWishart.sample /
wishartStep (GibbsWishart.scala:40-80).  Per iteration the engine runs FFBS for every series
and accumulates the sufficient statistics on the device; only those cross to the host, where
the conjugate draws (tiny, once per iteration) are made exactly as the reference makes them:

  V_jj ~ InverseGamma(alpha + n_j / 2, beta + ssy_j / 2)          Gibbs.scala:41-48
  W_ii ~ InverseGamma(alpha + T / 2,   beta + ss_i / 2)           Gibbs.scala:72-77 (shape uses T)
  W    ~ InverseWishart(nu + T, Psi + sum diff diff^T / dt)       GibbsWishart.scala:31-34

Two modes:
  per-series (reference semantics for a block-diagonal `|*|` model): every series owns its V, W;
      series shard across GPUs with no communication at all;
  pooled: one (V, W) shared by all series; statistics are summed over the shard on the device
      (dlm_stats_pool) and all-reduced across ranks (RCCL through dlm_gibbs_suffstats_allreduce,
      or any callable for tests) -- the only collective on the path -- and every rank makes the
      same draw from the same seed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Iterator, Optional, Sequence

import numpy as np

from . import _lib
from .dlm import Dlm, DlmParameters, materialise


def shard_bounds(n_series: int, world: int, rank: int):
    """Contiguous block of series owned by `rank`: [g * ceil(N/G), (g+1) * ceil(N/G))."""
    per = -(-n_series // world)
    lo = min(n_series, rank * per)
    return lo, min(n_series, lo + per)


@dataclass
class InverseGamma:
    """InverseGamma(shape, scale): draw = 1 / Gamma(shape, 1/scale).draw (InverseGamma.scala:14)."""
    shape: float
    scale: float

    def draw(self, rng: np.random.Generator, size=None):
        return 1.0 / rng.gamma(self.shape, 1.0 / np.asarray(self.scale), size=size)


@dataclass
class InverseWishart:
    """InverseWishart(nu, psi): draw through the Bartlett factor (InverseWishart.scala:17-25,
    Wishart.scala:34-43): A lower with sqrt(chi2(nu - i)) on the diagonal, N(0,1) below;
    L = chol(inv(psi)); draw = inv(L)^T inv(A)^T inv(A) inv(L)."""
    nu: float
    psi: np.ndarray

    def draw(self, rng: np.random.Generator):
        psi = np.asarray(self.psi, dtype=np.float64)
        d = psi.shape[0]
        A = np.zeros((d, d))
        for i in range(d):
            A[i, i] = np.sqrt(rng.chisquare(self.nu - i))
            A[i, :i] = rng.standard_normal(i)
        L = np.linalg.cholesky(np.linalg.inv(psi))
        invl, inva = np.linalg.inv(L), np.linalg.inv(A)
        return invl.T @ inva.T @ inva @ invl


def split_stats(stats: np.ndarray, d: int, p: int, outer: bool):
    """[.., L] statistics -> (ssy [.., p], n [.., p], ss [.., d] or outer [.., d, d], T [..])."""
    ssy, n = stats[..., :p], stats[..., p:2 * p]
    if outer:
        body = stats[..., 2 * p:2 * p + d * d].reshape(stats.shape[:-1] + (d, d))
    else:
        body = stats[..., 2 * p:2 * p + d]
    return ssy, n, body, stats[..., -1]


def draw_v(prior: InverseGamma, ssy, n, rng) -> np.ndarray:
    """sampleObservationMatrix (Gibbs.scala:23-50): diagonal V from the residual statistics."""
    return 1.0 / rng.gamma(prior.shape + 0.5 * np.asarray(n), 1.0 / (prior.scale + 0.5 * np.asarray(ssy)))


def draw_w_diag(prior: InverseGamma, ss, t_count, rng) -> np.ndarray:
    """sampleSystemMatrix (Gibbs.scala:56-78): diagonal W; shape uses the number of transitions."""
    ss = np.asarray(ss)
    shape = prior.shape + 0.5 * np.asarray(t_count)[..., None] if np.ndim(t_count) else prior.shape + 0.5 * t_count
    return 1.0 / rng.gamma(shape, 1.0 / (prior.scale + 0.5 * ss))


@dataclass
class GibbsState:
    """GibbsSampling.State (Gibbs.scala:8-11), batched: parameters (shared or per series) and,
    optionally, the last state draw theta [N][T+1][d]."""
    p: object
    theta: Optional[np.ndarray]
    stats: np.ndarray
    status: Optional[object] = None     # per-series status flags of this iteration's FFBS call (DLM_ST_*), as the engine returned them


def _params_list(p, N):
    return [p] * N if isinstance(p, DlmParameters) else list(p)


class GibbsSampling:
    @staticmethod
    def sample(mod: Dlm, prior_v: InverseGamma, prior_w: InverseGamma, init_params, times, y, engine,
               *, n_iter: int, seed: int = 0, pooled: bool = False, series_offset: int = 0,
               allreduce: Optional[Callable[[np.ndarray], np.ndarray]] = None,
               keep_theta: bool = False, ffbs: Optional[Callable] = None,
               simulation_smoother: bool = False) -> Iterator[GibbsState]:
        """d-Inverse-Gamma Gibbs (GibbsSampling.sample, Gibbs.scala:165-180).  `y` is this rank's shard [N][T][p] (numpy:
        host mode; a torch device tensor: everything stays in HBM); `series_offset` its first global series index (keeps
        the Philox streams identical to a single-GPU run).  Yields one GibbsState per iteration.
        Default: the reference's operation sequence -- forward filter, then the backward sampler of Smoothing.step
        (register-tile kernels for structured d <= 15 and lanes for d <= 5, the generic kernel elsewhere).
        simulation_smoother=True draws the states with the Durbin-Koopman simulation smoother instead: the same
        conditional distribution without a factorisation per step (4x faster at C3), not the reference's construction."""
        return _gibbs(mod, prior_v, prior_w, init_params, times, y, engine, n_iter, seed, pooled,
                      series_offset, allreduce, keep_theta, ffbs, wishart=False, simsmooth=simulation_smoother)

    @staticmethod
    def sample_svd(mod: Dlm, prior_v: InverseGamma, prior_w: InverseGamma, init_params, times, y, engine,
                   *, n_iter: int, seed: int = 0, pooled: bool = False, series_offset: int = 0,
                   allreduce: Optional[Callable] = None, keep_theta: bool = False, literal: bool = False) -> Iterator[GibbsState]:
        """GibbsSampling.sampleSvd / stepSvd (Gibbs.scala:182-217): the same conjugate updates with the state draw from
        the SVD filter and SvdSampler (dlm_svd_ffbs_batch, statistics accumulated on the device).  literal=True runs
        the reference's arithmetic: raw W in the time update and sqrt(W) in the sampler (Q9)."""
        quirks = (_lib.OPT_SVD_RAW_W_Q2 | _lib.OPT_SVD_SAMPLER_Q9) if literal else 0

        def run(mat, params, yy, seed, series_offset, flags, want_theta, want_stats, want_filt=False):
            return engine.svd_ffbs(mat, params, yy, seed=seed, series_offset=series_offset, flags=flags | quirks, want_stats=want_stats)
        return _gibbs(mod, prior_v, prior_w, init_params, times, y, engine, n_iter, seed, pooled,
                      series_offset, allreduce, keep_theta, run, wishart=False, simsmooth=False)


def gibbs_dinvgamma_device(mod: Dlm, prior_v: InverseGamma, prior_w: InverseGamma, init_params: DlmParameters, times, y,
                           engine, *, n_iter: int, seed: int = 0, series_offset: int = 0, simulation_smoother: bool = False,
                           on_iteration: Optional[Callable] = None):
    """Per-series d-Inverse-Gamma Gibbs that never leaves the GPU: FFBS + statistics (dlm_ffbs_batch) and the conjugate
    draws (dlm_dinvgamma_step_batch) alternate on device-resident parameter arrays; nothing crosses PCIe per iteration.
    `y` is a device tensor [N][T][p]; every series starts from `init_params`.  Returns (V [N][p*p], W [N][d*d]) of the last
    iteration (device tensors); `on_iteration(it, V, W, stats)` may copy out whatever chain summaries are wanted.
    The draws use the Philox stream (seed, global series index, iteration, component), so a sharded run reproduces the
    single-GPU run; they are the same distributions as GibbsSampling.sample draws on the host, not the same numbers."""
    import torch
    N = int(y.shape[0])
    mat = materialise(mod, times)
    d, p = mat.d, mat.p
    dev = y.device
    flags = _lib.OPT_FFBS_SIMSMOOTH if simulation_smoother else 0
    cmf = lambda a: torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64).T).reshape(-1), device=dev)
    V = cmf(init_params.v).repeat(N, 1).contiguous(); W = cmf(init_params.w).repeat(N, 1).contiguous()
    m0 = torch.as_tensor(np.asarray(init_params.m0, dtype=np.float64), device=dev)
    C0 = cmf(init_params.c0)
    for it in range(n_iter):
        packed = (V.reshape(-1), p * p, W.reshape(-1), d * d, m0, 0, C0, 0)
        out = engine.ffbs(mat, packed, y, seed=seed * 1000003 + it, series_offset=series_offset, flags=flags,
                          want_theta=False, want_stats=True, want_filt=False)
        stats = out["stats"]
        del out     # (the filter workspace returns to the allocator before the next call asks for one)
        V, W = engine.dinvgamma_step(d, p, stats, prior_v, prior_w, iteration=it, seed=seed, series_offset=series_offset)
        if on_iteration is not None:
            on_iteration(it, V, W, stats)
    return V, W


class GibbsWishart:
    @staticmethod
    def sample(mod: Dlm, prior_v: InverseGamma, prior_w: InverseWishart, init_params, times, y, engine,
               *, n_iter: int, seed: int = 0, pooled: bool = False, series_offset: int = 0,
               allreduce=None, keep_theta: bool = False, ffbs=None,
               simulation_smoother: bool = False) -> Iterator[GibbsState]:
        """Inverse-Wishart Gibbs for W (GibbsWishart.sample; order theta, W, V as wishartStep)."""
        return _gibbs(mod, prior_v, prior_w, init_params, times, y, engine, n_iter, seed, pooled,
                      series_offset, allreduce, keep_theta, ffbs, wishart=True, simsmooth=simulation_smoother)


def _gibbs(mod, prior_v, prior_w, init_params, times, y, engine, n_iter, seed, pooled, series_offset,
           allreduce, keep_theta, ffbs, wishart, simsmooth=False):
    on_device = hasattr(y, "data_ptr")          # a torch device tensor: statistics are pooled and reduced in HBM
    if not on_device:
        y = np.asarray(y, dtype=np.float64)
    N = y.shape[0]
    mat = materialise(mod, times)
    d, p = mat.d, mat.p
    flags = (_lib.OPT_STATS_OUTER if wishart else 0) | (_lib.OPT_FFBS_SIMSMOOTH if simsmooth else 0)
    run = ffbs if ffbs is not None else engine.ffbs
    params = init_params
    rng = np.random.default_rng(seed)            # identical on every rank (pooled draws agree)
    for it in range(n_iter):
        out = run(mat, params, y, seed=seed * 1000003 + it, series_offset=series_offset, flags=flags,
                  want_theta=keep_theta, want_stats=True, want_filt=False)     # (a Gibbs iteration never looks at the filter's records)
        stats = out["stats"] if on_device else np.asarray(out["stats"])
        theta = ((out["theta"] if on_device else np.asarray(out["theta"])) if keep_theta else None)
        status = out.get("status")
        del out     # the filter workspace (N (T + 1) (d + d^2) doubles: 26 GB at C4) goes back to the allocator BEFORE the next call asks for one
        if pooled:
            # sum over the series of this shard (dlm_stats_pool), then over the ranks: the only collective on the path
            tot = stats.sum(axis=0) if engine is None else engine.stats_pool(stats)
            if allreduce is not None:
                tot = allreduce(tot)
            tot = tot.cpu().numpy() if hasattr(tot, "cpu") else np.asarray(tot)
            ssy, n, body, tcount = split_stats(tot, d, p, wishart)
            base = _params_list(params, 1)[0]
            if wishart:
                w = InverseWishart(prior_w.nu + tcount, np.asarray(prior_w.psi) + body).draw(rng)
            else:
                w = np.diag(draw_w_diag(prior_w, body, tcount, rng))
            v = np.diag(draw_v(prior_v, ssy, n, rng))
            params = DlmParameters(v, w, base.m0, base.c0)
        else:
            # independent V, W per series; the draws of series n use a generator keyed by its
            # GLOBAL index so that sharding does not change them
            ssy, n, body, tcount = split_stats(stats.cpu().numpy() if on_device else stats, d, p, wishart)
            old = _params_list(params, N)
            new = []
            for k in range(N):
                r = np.random.default_rng([seed, it, series_offset + k])
                if wishart:
                    w = InverseWishart(prior_w.nu + tcount[k], np.asarray(prior_w.psi) + body[k]).draw(r)
                    v = np.diag(draw_v(prior_v, ssy[k], n[k], r))
                else:
                    v = np.diag(draw_v(prior_v, ssy[k], n[k], r))
                    w = np.diag(draw_w_diag(prior_w, body[k], tcount[k], r))
                new.append(DlmParameters(v, w, old[k].m0, old[k].c0))
            params = new
        yield GibbsState(params, theta, stats, status)
