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

__global__ __launch_bounds__(256) void k_ar1_ffbs(int N, int T, const double* __restrict__ y, const double* __restrict__ v,
                                                 long long v_stride, const double* __restrict__ sv, long long sv_stride,
                                                 const double* __restrict__ z, unsigned long long seed,
                                                 unsigned long long series_offset, double* __restrict__ filt,
                                                 double* __restrict__ theta, int* __restrict__ status) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double* p = sv + (size_t)n * sv_stride;
  const double phi = p[0], mu = p[1], sig = p[2], s2 = sig * sig;
  int st = (fabs(phi) < 1.0 && sig > 0.0) ? 0 : DLM_ST_NOT_PD;   // c0 = sigma^2 / (1 - phi^2) needs a stationary state
  const double* yn = y + (size_t)n * T;
  const double* vn = v + (size_t)n * v_stride;
  double* f = filt + (size_t)n * (T + 1) * 2;
  // filterUnivariate (:34-49): m0 = mu, c0 = sigma_eta^2 / (1 - phi^2)
  double m = mu, c = s2 / (1.0 - phi * phi);
  f[0] = m; f[1] = c;
  for (int t = 0; t < T; ++t) {   // stepUni (:17-32)
    const double at = mu + phi * (m - mu);
    const double rt = phi * phi * c + s2;
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
      const double a1 = mu + phi * (mt - mu);            // at, rt of the state after t (FilterState.at / rt)
      const double r1 = phi * phi * ct + s2;
      const double mean = mt + (ct * phi / r1) * (x - a1);
      const double cov = ct - (ct * ct) * (phi * phi) / r1;
      if (!(cov >= 0.0)) st |= DLM_ST_NOT_PD;
      x = mean + sqrt(cov) * (zn ? zn[t] : philox_normal(seed, series, (unsigned)t, 0u));
      th[t] = x;
    }
    if (!isfinite(x)) st |= DLM_ST_NONFINITE;
  }
  if (!isfinite(m) || !isfinite(c)) st |= DLM_ST_NONFINITE;
  if (status && st) atomicOr(&status[n], st);
}

hipError_t launch_ar1_ffbs(int N, int T, const double* y, const double* v, long long v_stride, const double* sv,
                           long long sv_stride, const double* z, unsigned long long seed, unsigned long long series_offset,
                           double* filt, double* theta, int* status, hipStream_t s) {
  hipLaunchKernelGGL(k_ar1_ffbs, dim3((N + 255) / 256), dim3(256), 0, s, N, T, y, v, v_stride, sv, sv_stride, z, seed,
                     series_offset, filt, theta, status);
  return hipGetLastError();
}

}  // namespace dlm
