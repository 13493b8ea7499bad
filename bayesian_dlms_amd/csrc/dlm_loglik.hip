// KalmanFilter.likelihood literally (DLM_OPT_LOGLIK_LITERAL_Q7 on dlm_loglik_batch).
//
// The reference's `likelihood` (KalmanFilter.scala:299-306) -- the function MetropolisHastings.dlm evaluates
// (MetropolisHastings.scala:134, :205) -- is NOT the prediction-error decomposition: it filters and then sums the
// TRANSITION density of the filtered means (KalmanFilter.logLikelihood, :175-183),
//     sum_{t = 1..T}  MultivariateGaussian(mod.g(dt_t) * m_{t-1}, w * dt_t).logPdf(m_t),
// with m_0 the initial state at t0 - 1.  Breeze's logPdf is -1/2 |L^-1 (x - mu)|^2 - (n/2 log 2 pi + sum log diag L),
// L = cholesky(w dt).  Every consecutive pair counts, also across a missing observation (m_t = a_t there: the quadratic
// form is 0); w is time-invariant (the reference passes p.w).
//
// Two kernels behind whichever forward kernel filtered the batch into an engine workspace:
//   k_spd_inverse_logdet  one wave per parameter set: W^-1 and log det W by a Cholesky factorisation in LDS (once per call)
//   k_loglik_q7           one workgroup per series: c_t = m_t - G_t m_{t-1}, c_t^T W^-1 c_t / dt_t, summed in a fixed order
//                         (no atomics: the result does not depend on the launch)
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

// W [nset][d*d] (stride w_stride doubles, column-major, symmetric) -> Winv [nset][d*d], logdet [nset]; bad[n] |= NOT_PD
__global__ __launch_bounds__(64) void k_spd_inverse_logdet(int d, int nset, const double* __restrict__ W, long long w_stride,
                                                           double* __restrict__ Winv, double* __restrict__ logdet,
                                                           int* __restrict__ bad) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x, ld = d + 1;
  if (n >= nset) return;
  double* A = sm;                 // d x ld, row-major: lower triangle becomes L
  double* X = sm + d * ld;        // L^-1, row-major
  const double* Wn = W + (size_t)n * w_stride;
  for (int k = lane; k < d * d; k += 64) A[(k % d) * ld + k / d] = Wn[k];
  __syncthreads();
  bool notpd = false;
  for (int k = 0; k < d; ++k) {
    const double akk = A[k * ld + k];
    if (!(akk > 0.0)) notpd = true;
    const double lkk = sqrt(akk > 0.0 ? akk : 1.0), inv = 1.0 / lkk;
    __syncthreads();
    for (int i = k + lane; i < d; i += 64) A[i * ld + k] = (i == k) ? lkk : A[i * ld + k] * inv;
    __syncthreads();
    for (int i = k + 1 + lane; i < d; i += 64) {
      const double lik = A[i * ld + k];
      for (int j = k + 1; j <= i; ++j) A[i * ld + j] = fma(-lik, A[j * ld + k], A[i * ld + j]);
    }
    __syncthreads();
  }
  // X = L^-1 by forward substitution, one column per lane; then Winv = X^T X
  for (int j = lane; j < d; j += 64) {
    for (int i = 0; i < d; ++i) {
      double v = (i == j) ? 1.0 : 0.0;
      for (int l = j; l < i; ++l) v = fma(-A[i * ld + l], X[l * ld + j], v);
      X[i * ld + j] = (i < j) ? 0.0 : v / A[i * ld + i];
    }
  }
  __syncthreads();
  for (int k = lane; k < d * d; k += 64) {
    const int i = k % d, j = k / d;
    double s = 0.0;
    for (int l = (i > j ? i : j); l < d; ++l) s = fma(X[l * ld + i], X[l * ld + j], s);
    Winv[(size_t)n * d * d + k] = s;
  }
  if (lane == 0) {
    double s = 0.0;
    for (int i = 0; i < d; ++i) s += log(A[i * ld + i]);
    logdet[n] = notpd ? __builtin_nan("") : 2.0 * s;
    if (notpd && bad) { if (w_stride) atomicOr(&bad[n], DLM_ST_NOT_PD); else bad[0] = DLM_ST_NOT_PD; }
  }
}

// records [N][T+1][rec] (mean first) -> loglik [N].  One workgroup of four waves per series; lane i of a wave owns component i
// of the step the wave is on (d <= 64), the waves take the steps round robin, and each wave sums its own steps in order.
__global__ __launch_bounds__(256) void k_loglik_q7(KArgs a, const double* __restrict__ records, const double* __restrict__ Winv,
                                                   const double* __restrict__ logdet, const int* __restrict__ wbad, int g_in_lds) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = a.d, T = a.T, rec = d + d * d, dd = d * d;
  double* Wi = sm;                         // d x d, column-major (symmetric)
  double* Gl = sm + dd;                    // n_g tables when they fit
  double* vec = Gl + (g_in_lds ? a.n_g * dd : 0);   // per wave: m_{t-1} (64) | c (64)
  double* part = vec + 4 * 128;            // 4 partial sums
  const size_t wset = a.w_stride ? (size_t)n : 0;
  for (int k = tid; k < dd; k += 256) Wi[k] = Winv[wset * dd + k];
  if (g_in_lds) for (int k = tid; k < a.n_g * dd; k += 256) Gl[k] = a.G[k];
  __syncthreads();
  const double* r = records + (size_t)n * (T + 1) * rec;
  double* m0 = vec + wave * 128;
  double* cv = m0 + 64;
  const double ldw = logdet[wset];
  double acc = 0.0;
  bool bad = false;
  for (int t = wave; t < T; t += 4) {      // pair (record t, record t + 1); transition table of observation t
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    const double* Gt = g_in_lds ? Gl + (size_t)gi * dd : a.G + (size_t)gi * dd;
    const double x1 = lane < d ? r[(size_t)(t + 1) * rec + lane] : 0.0;
    if (lane < d) m0[lane] = r[(size_t)t * rec + lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double c = 0.0;
    if (lane < d) {
      double v = 0.0;
      for (int k = 0; k < d; ++k) v = fma(Gt[lane + k * d], m0[k], v);
      c = x1 - v;
      cv[lane] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double q = 0.0;
    if (lane < d) {
      double u = 0.0;
      for (int k = 0; k < d; ++k) u = fma(Wi[lane + k * d], cv[k], u);
      q = c * u;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (!(dt > 0.0)) bad = true;           // w * 0 is the zero matrix: Breeze's cholesky throws (a NaN here, flagged)
    acc += -0.5 * q / dt - 0.5 * ((double)d * (1.8378770664093453 + log(dt)) + ldw);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (lane == 0) part[wave] = bad ? __builtin_nan("") : acc;
  __syncthreads();
  if (tid == 0) {
    const double s = ((part[0] + part[1]) + part[2]) + part[3];
    a.loglik[n] = s;
    int st = 0;
    if (!(s == s) || wbad[wset]) st |= DLM_ST_NOT_PD;
    if (!isfinite(s) && !st) st |= DLM_ST_NONFINITE;
    if (a.status && st) atomicOr(&a.status[n], st);
  }
}

size_t loglik_q7_ws_bytes(const KArgs& a) {
  const size_t nset = a.w_stride ? (size_t)a.N : 1;
  return sizeof(double) * nset * ((size_t)a.d * a.d + 1) + sizeof(int) * nset + 64;
}

// ws: loglik_q7_ws_bytes(a) bytes of device scratch.  a.loglik [N], a.status [N] (nullable), a.W / a.w_stride, the model.
hipError_t launch_loglik_q7(const KArgs& a, const double* records, void* ws, hipStream_t s) {
  const int d = a.d;
  const size_t nset = a.w_stride ? (size_t)a.N : 1;
  double* Winv = (double*)ws;
  double* logdet = Winv + nset * (size_t)d * d;
  int* wbad = (int*)(logdet + nset);
  hipError_t err = hipMemsetAsync(wbad, 0, sizeof(int) * nset, s);
  if (err != hipSuccess) return err;
  const size_t lds1 = sizeof(double) * 2 * (size_t)d * (d + 1);
  if (lds1 > 64 * 1024) {   // d = 64: 66 560 bytes, above what a kernel gets without asking
    static const hipError_t big = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spd_inverse_logdet), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (big != hipSuccess) return big;
  }
  hipLaunchKernelGGL(k_spd_inverse_logdet, dim3((unsigned)nset), dim3(64), lds1, s, d, (int)nset, a.W, (long long)a.w_stride, Winv, logdet, wbad);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  const int g_in_lds = ((size_t)a.n_g * d * d * 8 <= 32768) ? 1 : 0;
  const size_t lds2 = sizeof(double) * ((size_t)d * d + (g_in_lds ? (size_t)a.n_g * d * d : 0) + 4 * 128 + 4);
  hipLaunchKernelGGL(k_loglik_q7, dim3(a.N), dim3(256), lds2, s, a, records, (const double*)Winv, (const double*)logdet, (const int*)wbad, g_in_lds);
  return hipGetLastError();
}

}  // namespace dlm
