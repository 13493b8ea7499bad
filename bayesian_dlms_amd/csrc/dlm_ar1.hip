// Scalar AR(1) state-space FFBS, one LANE per series.
//
// Restates FilterAr.filterUnivariate / stepUni / univariateSample / backStepUni / ffbs (FilterAr.scala:15-82), the
// filter the stochastic-volatility samplers run once per series and per MCMC iteration with per-step observation
// variances v_t (StochasticVolatility.scala:142-162, FactorSv.scala:415-512):
//   alpha_t = mu + phi (alpha_{t-1} - mu) + eta_t,  eta_t ~ N(0, sigma_eta^2);   y_t = alpha_t + eps_t,  eps_t ~ N(0, v_t)
// With d = 1 a wavefront per series would idle 63 lanes, so a lane owns a series and walks its T steps; a lane's
// stream of y, v, (m, c) and theta is contiguous, so every cache line it touches is used in full over 4-8 steps
// (the pass is a few hundred microseconds at N = 10^4, T = 10^3 either way).
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

// OU = true: the Ornstein-Uhlenbeck variant on an irregular grid (FilterOu.scala:7-79) -- the same recursion with
//   phi -> exp(-phi dt),  sigma_eta^2 -> sigma_eta^2 (1 - exp(-2 phi dt)) / (2 phi),  dt_t = times[t] - times[t-1],
// and, LITERALLY as filterUnivariate (:34-46) has it, c0 = sigma * sigma / phi * phi (= sigma^2) and an initial state
// that carries the time of the first observation (first dt = 0).
template <bool OU>
__global__ __launch_bounds__(256) void k_ar1_ffbs(int N, int T, const double* __restrict__ times,
                                                 const double* __restrict__ y, const double* __restrict__ v,
                                                 long long v_stride, const double* __restrict__ sv, long long sv_stride,
                                                 const double* __restrict__ z, unsigned long long seed,
                                                 unsigned long long series_offset, double* __restrict__ filt,
                                                 double* __restrict__ theta, int* __restrict__ status) {
  const int n = blockIdx.x * 256 + threadIdx.x;
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
  for (int t = 0; t < T; ++t) {   // stepUni (FilterAr.scala:17-32 / FilterOu.scala:7-27)
    double at, rt;
    if (OU) {
      const double dt = times[t] - (t == 0 ? times[0] : times[t - 1]);
      const double variance = (s2 * (1 - exp(-2 * phi * dt))) / (2 * phi);
      at = mu + exp(-phi * dt) * (m - mu);
      rt = exp(-2 * phi * dt) * c + variance;
    } else {
      at = mu + phi * (m - mu);
      rt = phi * phi * c + s2;
    }
    const double yt = yn[t], vt = vn[t];
    if (yt == yt) {
      const double kt = rt / (rt + vt);
      m = at + kt * (yt - at);
      c = kt * vt;
      if (!(vt > 0.0)) st |= DLM_ST_NOT_PD;
    } else { m = at; c = rt; }
    f[2 * (t + 1)] = m; f[2 * (t + 1) + 1] = c;
  }
  if (theta) {   // univariateSample / backStepUni (:58-76)
    const unsigned long long series = series_offset + (unsigned long long)n;
    const double* zn = z ? z + (size_t)n * (T + 1) : nullptr;
    double* th = theta + (size_t)n * (T + 1);
    double x = m + sqrt(c) * (zn ? zn[T] : philox_normal(seed, series, (unsigned)T, 0u));
    th[T] = x;
    for (int t = T - 1; t >= 0; --t) {
      const double mt = f[2 * t], ct = f[2 * t + 1];
      double ph = phi, a1, r1;                           // at, rt of the state after t (FilterState.at / rt)
      if (OU) {
        const double dt = times[t] - (t == 0 ? times[0] : times[t - 1]);
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
      if (!(cov >= -1e-12 * ct)) st |= DLM_ST_NOT_PD;
      cov = cov > 0.0 ? cov : 0.0;
      x = mean + sqrt(cov) * (zn ? zn[t] : philox_normal(seed, series, (unsigned)t, 0u));
      th[t] = x;
    }
    if (!isfinite(x)) st |= DLM_ST_NONFINITE;
  }
  if (!isfinite(m) || !isfinite(c)) st |= DLM_ST_NONFINITE;
  if (status && st) atomicOr(&status[n], st);
}

hipError_t launch_ar1_ffbs(int N, int T, const double* times, const double* y, const double* v, long long v_stride,
                           const double* sv, long long sv_stride, const double* z, unsigned long long seed,
                           unsigned long long series_offset, double* filt, double* theta, int* status, hipStream_t s) {
  if (times) hipLaunchKernelGGL(k_ar1_ffbs<true>, dim3((N + 255) / 256), dim3(256), 0, s, N, T, times, y, v, v_stride, sv,
                                sv_stride, z, seed, series_offset, filt, theta, status);
  else hipLaunchKernelGGL(k_ar1_ffbs<false>, dim3((N + 255) / 256), dim3(256), 0, s, N, T, times, y, v, v_stride, sv,
                          sv_stride, z, seed, series_offset, filt, theta, status);
  return hipGetLastError();
}

}  // namespace dlm
