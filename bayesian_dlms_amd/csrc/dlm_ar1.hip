// Scalar AR(1) state-space FFBS, one LANE per series.
//
// Restates FilterAr.filterUnivariate / stepUni / univariateSample / backStepUni / ffbs (FilterAr.scala:15-82), the
// filter the stochastic-volatility samplers run once per series and per MCMC iteration with per-step observation
// variances v_t (StochasticVolatility.scala:142-162, FactorSv.scala:415-512):
//   alpha_t = mu + phi (alpha_{t-1} - mu) + eta_t,  eta_t ~ N(0, sigma_eta^2);   y_t = alpha_t + eps_t,  eps_t ~ N(0, v_t)
// With d = 1 a wavefront per series would idle 63 lanes, so a lane owns a series and walks its T steps; a lane's
// stream of y, v, (m, c) and theta is contiguous, so every cache line it touches is used in full over 4-8 steps.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

// OU = true: the Ornstein-Uhlenbeck variant on an irregular grid (FilterOu.scala:7-79) -- the same recursion with
//   phi -> exp(-phi dt),  sigma_eta^2 -> sigma_eta^2 (1 - exp(-2 phi dt)) / (2 phi),  dt_t = times[t] - times[t-1],
// and, LITERALLY as filterUnivariate (:34-46) has it, c0 = sigma * sigma / phi * phi (= sigma^2) and an initial state
// that carries the time of the first observation (first dt = 0).
//
// A lane's recursion is a dependent chain with one load pair per step; issued step by step it runs at memory latency.
// The passes therefore walk TILES of AR_TILE steps: the (y, v) or (m, c) of the NEXT tile are requested (independent
// 16-byte loads, 64-128 contiguous bytes per lane) before the current tile is computed, the tile's results leave in one
// burst of stores, and the backward pass draws the tile's normals ahead of the chain.
constexpr int AR_TILE = 8;    // backward: 8 (m, c) pairs = one 128-byte line per lane and tile (16 costs a resident wave per SIMD)
constexpr int FW_TILE = 16;   // forward: 16 y and 16 v = one 128-byte line per lane, stream and tile
typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));

template <bool OU>
__global__ __launch_bounds__(64) void k_ar1_ffbs(int N, int T, const double* __restrict__ times,
                                                const double* __restrict__ y, const double* __restrict__ v,
                                                long long v_stride, const double* __restrict__ sv, long long sv_stride,
                                                const double* __restrict__ z, unsigned long long seed,
                                                unsigned long long series_offset, double* __restrict__ filt,
                                                double* __restrict__ theta, int* __restrict__ status) {
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= N) return;
  const double* p = sv + (size_t)n * sv_stride;
  const double phi = p[0], mu = p[1], sig = p[2], s2 = sig * sig;
  int st = ((OU ? phi > 0.0 : fabs(phi) < 1.0) && sig > 0.0) ? 0 : DLM_ST_NOT_PD;   // a stationary state
  const double* yn = y + (size_t)n * T;
  const double* vn = v + (size_t)n * v_stride;
  double* f = filt + (size_t)n * (T + 1) * 2;
  // filterUnivariate (:34-49): m0 = mu, c0 = sigma_eta^2 / (1 - phi^2)
  double m = mu, c = OU ? sig * sig / phi * phi : s2 / (1.0 - phi * phi);
  f[0] = m; f[1] = c;
  const int Tl = T - 1;
  dbl2 yb[FW_TILE / 2], vb[FW_TILE / 2];
  auto request = [&](int t0) {                // (y, v) of steps t0 .. t0 + FW_TILE - 1 (clamped at the last step)
    if (t0 + FW_TILE <= T) {
#pragma unroll
      for (int j = 0; j < FW_TILE / 2; ++j) { yb[j] = *(const dbl2*)(yn + t0 + 2 * j); vb[j] = *(const dbl2*)(vn + t0 + 2 * j); }
    } else {
#pragma unroll
      for (int j = 0; j < FW_TILE / 2; ++j) {
        const int ta = t0 + 2 * j < Tl ? t0 + 2 * j : Tl, tb = t0 + 2 * j + 1 < Tl ? t0 + 2 * j + 1 : Tl;
        yb[j] = dbl2{yn[ta], yn[tb]}; vb[j] = dbl2{vn[ta], vn[tb]};
      }
    }
  };
  if (T > 0) request(0);
  for (int t0 = 0; t0 < T; t0 += FW_TILE) {   // stepUni (FilterAr.scala:17-32 / FilterOu.scala:7-27)
    double yc[FW_TILE], vc[FW_TILE];
#pragma unroll
    for (int j = 0; j < FW_TILE / 2; ++j) { yc[2 * j] = yb[j].x; yc[2 * j + 1] = yb[j].y; vc[2 * j] = vb[j].x; vc[2 * j + 1] = vb[j].y; }
    if (t0 + FW_TILE < T) request(t0 + FW_TILE);
    dbl2 out[FW_TILE];
#pragma unroll
    for (int j = 0; j < FW_TILE; ++j) {
      const int t = t0 + j;
      double at, rt;
      if (OU) {
        const int tc = t < Tl ? t : Tl;
        const double dt = times[tc] - (tc == 0 ? times[0] : times[tc - 1]);
        const double variance = (s2 * (1 - exp(-2 * phi * dt))) / (2 * phi);
        at = mu + exp(-phi * dt) * (m - mu);
        rt = exp(-2 * phi * dt) * c + variance;
      } else {
        at = mu + phi * (m - mu);
        rt = phi * phi * c + s2;
      }
      const double yt = yc[j], vt = vc[j];
      double mn, cn;
      if (yt == yt) {
        const double kt = rt / (rt + vt);
        mn = at + kt * (yt - at);
        cn = kt * vt;
        if (t < T && !(vt > 0.0)) st |= DLM_ST_NOT_PD;
      } else { mn = at; cn = rt; }
      if (t < T) { m = mn; c = cn; }
      out[j] = dbl2{m, c};
    }
    dbl2* fo = (dbl2*)(f + 2 * (t0 + 1));
    if (t0 + FW_TILE <= T) {
#pragma unroll
      for (int j = 0; j < FW_TILE; ++j) fo[j] = out[j];
    } else {
#pragma unroll
      for (int j = 0; j < FW_TILE; ++j) if (t0 + j < T) fo[j] = out[j];
    }
  }
  if (theta) {   // univariateSample / backStepUni (:58-76)
    const unsigned long long series = series_offset + (unsigned long long)n;
    const double* zn = z ? z + (size_t)n * (T + 1) : nullptr;
    double* th = theta + (size_t)n * (T + 1);
    double x = m + sqrt(c) * (zn ? zn[T] : philox_normal(seed, series, (unsigned)T, 0u));
    th[T] = x;
    // tiles of records t0 .. t0 + AR_TILE - 1, walked downwards; the top tile may be ragged
    const dbl2* fr = (const dbl2*)f;
    dbl2 fb[AR_TILE];
    int t0 = ((T - 1) / AR_TILE) * AR_TILE;
    if (T > 0) {
#pragma unroll
      for (int j = 0; j < AR_TILE; ++j) { const int t = t0 + j < T ? t0 + j : T - 1; fb[j] = fr[t]; }
    }
    for (; T > 0 && t0 >= 0; t0 -= AR_TILE) {
      dbl2 fc[AR_TILE];
      double zz[AR_TILE], xo[AR_TILE];
#pragma unroll
      for (int j = 0; j < AR_TILE; ++j) fc[j] = fb[j];
      if (t0 > 0) {
#pragma unroll
        for (int j = 0; j < AR_TILE; ++j) fb[j] = fr[t0 - AR_TILE + j];
      }
#pragma unroll
      for (int j = 0; j < AR_TILE; ++j) {
        const int t = t0 + j < T ? t0 + j : T - 1;
        zz[j] = zn ? zn[t] : philox_normal(seed, series, (unsigned)t, 0u);
      }
#pragma unroll
      for (int j = AR_TILE - 1; j >= 0; --j) {
        const int t = t0 + j;
        const double mt = fc[j].x, ct = fc[j].y;
        double ph = phi, a1, r1;                           // at, rt of the state after t (FilterState.at / rt)
        if (OU) {
          const int tc = t < T ? t : T - 1;
          const double dt = times[tc] - (tc == 0 ? times[0] : times[tc - 1]);
          ph = exp(-phi * dt);
          a1 = mu + ph * (mt - mu);
          r1 = exp(-2 * phi * dt) * ct + (s2 * (1 - exp(-2 * phi * dt))) / (2 * phi);
        } else {
          a1 = mu + phi * (mt - mu);
          r1 = phi * phi * ct + s2;
        }
        const double mean = mt + (ct * ph / r1) * (x - a1);
        double cov = ct - (ct * ct) * (ph * ph) / r1;
        // dt = 0 (always the first OU step, FilterOu.scala:37-38) makes this exactly c - c^2 / c: a rounding-level negative
        // value would draw NaN in the reference; it is clamped here and in the oracle, larger ones are flagged
        if (t < T && !(cov >= -1e-12 * ct)) st |= DLM_ST_NOT_PD;
        cov = cov > 0.0 ? cov : 0.0;
        const double xn = mean + sqrt(cov) * zz[j];
        if (t < T) x = xn;
        xo[j] = x;
      }
      if (t0 + AR_TILE <= T) {
#pragma unroll
        for (int j = 0; j < AR_TILE / 2; ++j) *(dbl2*)(th + t0 + 2 * j) = dbl2{xo[2 * j], xo[2 * j + 1]};
      } else {
#pragma unroll
        for (int j = 0; j < AR_TILE; ++j) if (t0 + j < T) th[t0 + j] = xo[j];
      }
    }
    if (!isfinite(x)) st |= DLM_ST_NONFINITE;
  }
  if (!isfinite(m) || !isfinite(c)) st |= DLM_ST_NONFINITE;
  if (status && st) atomicOr(&status[n], st);
}

hipError_t launch_ar1_ffbs(int N, int T, const double* times, const double* y, const double* v, long long v_stride,
                           const double* sv, long long sv_stride, const double* z, unsigned long long seed,
                           unsigned long long series_offset, double* filt, double* theta, int* status, hipStream_t s) {
  if (times) hipLaunchKernelGGL(k_ar1_ffbs<true>, dim3((N + 63) / 64), dim3(64), 0, s, N, T, times, y, v, v_stride, sv,
                                sv_stride, z, seed, series_offset, filt, theta, status);
  else hipLaunchKernelGGL(k_ar1_ffbs<false>, dim3((N + 63) / 64), dim3(64), 0, s, N, T, times, y, v, v_stride, sv,
                          sv_stride, z, seed, series_offset, filt, theta, status);
  return hipGetLastError();
}

}  // namespace dlm
