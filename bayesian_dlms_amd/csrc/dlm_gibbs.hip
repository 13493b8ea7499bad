// d-Inverse-Gamma conjugate step on the device (GibbsSampling.dinvGammaStep: Gibbs.scala:23-78, :134-151).
//
// The FFBS kernels leave [ssy (p) | n (p) | ss (d) | T] per series on the device; with per-series parameters (the
// reference's semantics for a block-diagonal `|*|` model) the draws
//   V_jj ~ InverseGamma(alpha_v + n_j / 2, beta_v + ssy_j / 2)            Gibbs.scala:41-48
//   W_ii ~ InverseGamma(alpha_w + T / 2,   beta_w + ss_i / 2)             Gibbs.scala:72-77 (shape uses T)
// are N (p + d) independent scalars: one thread each, so that a Gibbs iteration never leaves the GPU (the host loop
// over 10^4 series costs more than the FFBS pass it follows).  InverseGamma(shape, scale).draw = 1 / Gamma(shape,
// 1 / scale).draw (InverseGamma.scala:14) = scale / Gamma(shape, 1).draw; the unit-scale Gamma is Marsaglia-Tsang (2000)
// on Philox normals and uniforms keyed by (seed, global series, iteration, component, attempt): reproducible and
// independent of the sharding.  The reference's generator cannot be seeded: only the distribution is
// comparable with it; oracle/dlm_oracle.c restates this very construction.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

__device__ __forceinline__ void gibbs_rand(unsigned long long seed, unsigned long long series, unsigned long long iteration,
                                           unsigned comp, unsigned attempt, unsigned which, double& u1, double& u2) {
  unsigned c[4] = {(unsigned)series, (unsigned)(series >> 32), (unsigned)iteration, comp * 2048u + attempt * 2u + which};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32) ^ 0x47494242u);
  u1 = ((double)c[0] * 4294967296.0 + (double)c[1] + 1.0) * (1.0 / 18446744073709551616.0);   // (0, 1]
  u2 = ((double)c[2] * 4294967296.0 + (double)c[3]) * (1.0 / 18446744073709551616.0);         // [0, 1)
}

// Gamma(a, 1): Marsaglia & Tsang, "A simple method for generating gamma variables" (2000); a < 1 by the u^(1/a) boost
__device__ double gamma_unit(double a, unsigned long long seed, unsigned long long series, unsigned long long iteration, unsigned comp) {
  double boost = 1.0;
  if (a < 1.0) {
    double u1, u2;
    gibbs_rand(seed, series, iteration, comp, 1023u, 0u, u1, u2);
    boost = pow(u1, 1.0 / a);
    a += 1.0;
  }
  const double dd = a - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
  for (unsigned k = 0; k < 1023u; ++k) {
    double u1, u2, w1, w2;
    gibbs_rand(seed, series, iteration, comp, k, 0u, u1, u2);
    const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    double v = 1.0 + cc * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    gibbs_rand(seed, series, iteration, comp, k, 1u, w1, w2);
    if (log(w1) < 0.5 * x * x + dd - dd * v + dd * log(v)) return dd * v * boost;
  }
  return dd * boost;   // unreachable in practice (acceptance > 95 % per attempt)
}

__global__ __launch_bounds__(256) void k_dinvgamma_step(int d, int p, int N, const double* __restrict__ stats, double av, double bv,
                                                        double aw, double bw, unsigned long long seed,
                                                        unsigned long long series_offset, unsigned long long iteration,
                                                        double* __restrict__ Vout, double* __restrict__ Wout) {
  const int per = p + d;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= (long long)N * per) return;
  const int n = (int)(gid / per), comp = (int)(gid % per);
  const int L = 2 * p + d + 1;
  const double* s = stats + (size_t)n * L;
  const unsigned long long series = series_offset + (unsigned long long)n;
  double shape, rate;
  if (comp < p) { shape = av + 0.5 * s[p + comp]; rate = bv + 0.5 * s[comp]; }
  else { shape = aw + 0.5 * s[L - 1]; rate = bw + 0.5 * s[2 * p + (comp - p)]; }
  const double val = rate / gamma_unit(shape, seed, series, iteration, (unsigned)comp);
  if (comp < p) {   // column comp of the dense diagonal V
    double* col = Vout + (size_t)n * p * p + (size_t)comp * p;
    for (int i = 0; i < p; ++i) col[i] = (i == comp) ? val : 0.0;
  } else {
    const int i0 = comp - p;
    double* col = Wout + (size_t)n * d * d + (size_t)i0 * d;
    for (int i = 0; i < d; ++i) col[i] = (i == i0) ? val : 0.0;
  }
}

hipError_t launch_dinvgamma_step(int d, int p, int N, const double* stats, double av, double bv, double aw, double bw,
                                 unsigned long long seed, unsigned long long series_offset, unsigned long long iteration,
                                 double* Vout, double* Wout, hipStream_t s) {
  const long long total = (long long)N * (p + d);
  hipLaunchKernelGGL(k_dinvgamma_step, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, p, N, stats, av, bv, aw, bw,
                     seed, series_offset, iteration, Vout, Wout);
  return hipGetLastError();
}

}  // namespace dlm
