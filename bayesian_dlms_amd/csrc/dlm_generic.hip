// Generic wave-per-series kernels: any d <= 64, p <= 64, missing data, irregular dt,
// time-varying F, per-series parameters, Q1 compatibility switch.
//
// One 64-lane wavefront owns one series for all T steps; the state (m, C), the prior
// (a, R) and every scratch matrix live in that wave's private LDS slice, the shared
// model tables (F, G) and parameters are read through L2.  These kernels follow the
// reference's operation sequence (Joseph-form update, solve for the gain and for
// J = C G^T R^-1) and are the correctness path for every shape; the d <= 16, p == 1
// fast path lives in dlm_mfma16.hip.
//
// Reference (under /root/reference/core/src/main/scala/dlm/model/):
//   KalmanFilter.scala:64-118, :273-286, :311-321   filter step
//   Smoothing.scala:31-64                           RTS smoother
//   Smoothing.scala:74-122                          backward sampling
//   Gibbs.scala:23-78, GibbsWishart.scala:16-35     sufficient statistics
// The linear solves use a Cholesky factorisation (Q, R and H are symmetric positive
// definite) where Breeze's `\` uses LU; results agree to fp64 rounding.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

#define CM(i, j, ld) ((i) + (j) * (ld))

__device__ __forceinline__ void wsync() { __syncthreads(); }  // block == one wavefront

// C(m x n) = op(A)(m x k) * op(B)(k x n).  Column-major; generic (LDS or global) pointers.
template <bool TA, bool TB>
__device__ __forceinline__ void gemm(int lane, int m, int n, int k, const double* A, int lda,
                                     const double* B, int ldb, double* C, int ldc) {
  for (int idx = lane; idx < m * n; idx += 64) {
    const int i = idx % m, j = idx / m;
    double s = 0.0;
    for (int l = 0; l < k; ++l) {
      const double av = TA ? A[CM(l, i, lda)] : A[CM(i, l, lda)];
      const double bv = TB ? B[CM(j, l, ldb)] : B[CM(l, j, ldb)];
      s = fma(av, bv, s);
    }
    C[CM(i, j, ldc)] = s;
  }
}

// In-place lower Cholesky of the n x n LDS matrix A (column-major).  Returns true when a
// non-positive pivot was met (the factor is then completed with a tiny pivot).
// psd: the factor of a covariance to DRAW with -- a non-positive pivot gives a zero column (the direction has no variance:
// what the oracle's chol_lower does) instead of the tiny pivot that keeps a later triangular solve finite.
__device__ bool chol_lds(int lane, int n, double* A, bool psd = false) {
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    wsync();
    double akk = A[CM(k, k, n)];
    const bool np = !(akk > 0.0);
    if (np) { bad = true; akk = 1e-300; }
    const double lkk = (np && psd) ? 0.0 : sqrt(akk), inv = (np && psd) ? 0.0 : 1.0 / lkk;
    wsync();
    for (int i = k + lane; i < n; i += 64) A[CM(i, k, n)] = (i == k) ? lkk : A[CM(i, k, n)] * inv;
    wsync();
    const int r = n - k - 1;
    for (int idx = lane; idx < r * r; idx += 64) {
      const int i = k + 1 + idx % r, j = k + 1 + idx / r;
      if (i >= j) A[CM(i, j, n)] = fma(-A[CM(i, k, n)], A[CM(j, k, n)], A[CM(i, j, n)]);
    }
  }
  wsync();
  return bad;
}

// Symmetric eigendecomposition by cyclic Jacobi on one wave -- the factor the REFERENCE draws with (MultivariateGaussianSvd.scala:13-22:
// eigSym(cov), draw = mu + E sqrt(Lambda) z), in the oracle's restatement of it (oracle/dlm_oracle.c: eig_sym -- the same pair order,
// rotation formulas and stopping rule; Breeze's eigSym is LAPACK dsyevd, whose eigenvector signs are implementation-defined: both
// sides make the largest-|.| entry of each column positive).  S (n x n, column-major, symmetric) is destroyed and receives the
// eigenvectors, columns in ascending order of the eigenvalues `lam`; E: n x n scratch; tmp: n doubles of scratch.
__device__ void eig_lds(int lane, int n, double* S, double* E, double* lam, double* tmp) {
  for (int i = lane; i < n * n; i += 64) E[i] = (i % n == i / n) ? 1.0 : 0.0;
  wsync();
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int i = lane; i < n * n; i += 64) { const double v = S[i] * S[i]; if (i % n == i / n) dia += v; else off += v; }
    for (int o_ = 32; o_ > 0; o_ >>= 1) { off += __shfl_xor(off, o_); dia += __shfl_xor(dia, o_); }
    if (off <= 1e-60 + 1e-32 * dia) break;                       // (wave-uniform: every lane holds the same sums)
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = S[CM(p, q, n)];
        if (apq == 0.0) continue;
        const double app = S[CM(p, p, n)], aqq = S[CM(q, q, n)];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        wsync();                                                   // every lane has read the pivot entries
        for (int k = lane; k < n; k += 64) {                       // columns p, q
          const double skp = S[CM(k, p, n)], skq = S[CM(k, q, n)];
          S[CM(k, p, n)] = c * skp - s * skq;
          S[CM(k, q, n)] = s * skp + c * skq;
          const double ekp = E[CM(k, p, n)], ekq = E[CM(k, q, n)];
          E[CM(k, p, n)] = c * ekp - s * ekq;
          E[CM(k, q, n)] = s * ekp + c * ekq;
        }
        wsync();
        for (int k = lane; k < n; k += 64) {                       // rows p, q
          const double spk = S[CM(p, k, n)], sqk = S[CM(q, k, n)];
          S[CM(p, k, n)] = c * spk - s * sqk;
          S[CM(q, k, n)] = s * spk + c * sqk;
        }
        wsync();
      }
  }
  wsync();
  for (int j = lane; j < n; j += 64) tmp[j] = S[CM(j, j, n)];
  wsync();
  for (int j = lane; j < n; j += 64) {                             // ascending order (ties by index), canonical sign, into S
    const double lj = tmp[j];
    int rank = 0;
    for (int k = 0; k < n; ++k) rank += (tmp[k] < lj) || (tmp[k] == lj && k < j);
    int arg = 0; double best = -1.0;
    for (int i = 0; i < n; ++i) { const double v = fabs(E[CM(i, j, n)]); if (v > best * (1.0 + 1e-12)) { best = v; arg = i; } }
    const double sg = E[CM(arg, j, n)] < 0.0 ? -1.0 : 1.0;
    lam[rank] = lj;
    for (int i = 0; i < n; ++i) S[CM(i, rank, n)] = sg * E[CM(i, j, n)];
  }
  wsync();
}

// Solve (L L^T) X = B in place for the n x nrhs LDS matrix B; one lane per right-hand side.
__device__ void chol_solve_lds(int lane, int n, int nrhs, const double* L, double* B) {
  for (int c = lane; c < nrhs; c += 64) {
    double* x = B + (size_t)c * n;
    for (int i = 0; i < n; ++i) {
      double s = x[i];
      for (int l = 0; l < i; ++l) s = fma(-L[CM(i, l, n)], x[l], s);
      x[i] = s / L[CM(i, i, n)];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i];
      for (int l = i + 1; l < n; ++l) s = fma(-L[CM(l, i, n)], x[l], s);
      x[i] = s / L[CM(i, i, n)];
    }
  }
  wsync();
}

__device__ __forceinline__ bool any_nonfinite(int lane, int n, const double* v) {
  bool bad = false;
  for (int i = lane; i < n; i += 64) bad |= !isfinite(v[i]);
  return __ballot(bad) != 0ull;
}

// ---------------------------------------------------------------------------------------
// Kalman filter
// ---------------------------------------------------------------------------------------
size_t generic_filter_lds_bytes(int d, int p) {
  return sizeof(double) * (size_t)(3 * d + 4 * d * d + 2 * d * p + 2 * p * p + 3 * p) + sizeof(int) * p + 16;
}

__global__ __launch_bounds__(64) void k_filter_generic(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, frec = p + p * p;
  double* m = sm;            double* av = m + d;       double* tv = av + d;
  double* C = tv + d;        double* R = C + dd;       double* T1 = R + dd;     double* T2 = T1 + dd;
  double* RF = T2 + dd;      double* Kt = RF + d * p;  double* Q = Kt + d * p;  double* Qm = Q + p * p;
  double* fv = Qm + p * p;   double* ev = fv + p;      double* yv = ev + p;
  int* idx = (int*)(yv + p);

  const double* V = a.V + (size_t)n * a.v_stride;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const double* y = a.y + (size_t)n * T * p;
  double* out = a.filt ? a.filt + (size_t)n * (T + 1) * rec : nullptr;   // null: likelihood only, nothing stored
  double ll = 0.0;   // sum_t log N(y_t^obs; f_t^obs, Q_t^obs) (KalmanFilter.scala:138-153)
  double* pri = a.prior ? a.prior + (size_t)n * (T + 1) * rec : nullptr;
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * frec : nullptr;
  int st = 0;

  // initialiseState (KalmanFilter.scala:112-118): m = a = m0, C = R = C0, f/Q absent
  for (int i = lane; i < d; i += 64) { m[i] = m0[i]; if (out) out[i] = m0[i]; if (pri) pri[i] = m0[i]; }
  for (int i = lane; i < dd; i += 64) { C[i] = C0[i]; if (out) out[d + i] = C0[i]; if (pri) pri[d + i] = C0[i]; }
  if (fq) for (int i = lane; i < frec; i += 64) fq[i] = __builtin_nan("");

  const double* V0 = V; const double* W0 = W;
  for (int t = 0; t < T; ++t) {
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
    const double* Ft = a.F + (size_t)t * a.f_stride;
    const double dt = a.dt ? a.dt[t] : 1.0;
    V = V0 + (size_t)t * a.v_tstride;   // time-varying variances (StudentTGibbs.scala:100-136, DlmFsvSystem.scala:137-208)
    W = W0 + (size_t)t * a.w_tstride;
    wsync();
    // advState (KalmanFilter.scala:273-286)
    if (dt == 0.0) {
      for (int i = lane; i < d; i += 64) av[i] = m[i];
      for (int i = lane; i < dd; i += 64) R[i] = C[i];
    } else {
      gemm<false, false>(lane, d, 1, d, Gt, d, m, d, av, d);
      gemm<false, false>(lane, d, d, d, Gt, d, C, d, T1, d);
      wsync();
      gemm<false, true>(lane, d, d, d, T1, d, Gt, d, R, d);
      wsync();
      for (int i = lane; i < dd; i += 64) R[i] = fma(W[i], dt, R[i]);
    }
    wsync();
    // oneStepPrediction, unmasked (KalmanFilter.scala:311-321)
    gemm<true, false>(lane, p, 1, d, Ft, d, av, d, fv, p);
    gemm<false, false>(lane, d, p, d, R, d, Ft, d, RF, d);
    wsync();
    gemm<true, false>(lane, p, p, d, Ft, d, RF, d, Q, p);
    wsync();
    for (int i = lane; i < p * p; i += 64) Q[i] += V[i];
    wsync();
    if (pri) {
      double* pr = pri + (size_t)(t + 1) * rec;
      for (int i = lane; i < d; i += 64) pr[i] = av[i];
      for (int i = lane; i < dd; i += 64) pr[d + i] = R[i];
    }
    if (fq) {
      double* fr = fq + (size_t)(t + 1) * frec;
      for (int i = lane; i < p; i += 64) fr[i] = fv[i];
      for (int i = lane; i < p * p; i += 64) fr[p + i] = Q[i];
    }
    // observed components (flattenObs / indexNonMissing, KalmanFilter.scala:130-132,190-197)
    const double yl = (lane < p) ? y[(size_t)t * p + lane] : __builtin_nan("");
    const unsigned long long mask = __ballot(yl == yl);
    const int pm = __popcll(mask);
    if (yl == yl) {
      const int pos = __popcll(mask & ((1ull << lane) - 1ull));
      idx[pos] = lane; yv[pos] = yl;
    }
    wsync();
    if (pm == 0) {  // updateState :74-75
      for (int i = lane; i < d; i += 64) m[i] = av[i];
      for (int i = lane; i < dd; i += 64) C[i] = R[i];
    } else {
      // masked predcov / gain (missingF, missingV :202-218; oneStepMissing :44-53)
      for (int k = lane; k < pm * pm; k += 64) Qm[k] = Q[CM(idx[k % pm], idx[k / pm], p)];
      for (int k = lane; k < pm * d; k += 64) { const int j = k % pm, i = k / pm; Kt[CM(j, i, pm)] = RF[CM(i, idx[j], d)]; }
      for (int j = lane; j < pm; j += 64) ev[j] = yv[j] - fv[idx[j]];
      wsync();
      if (chol_lds(lane, pm, Qm)) st |= DLM_ST_NOT_PD;
      chol_solve_lds(lane, pm, d, Qm, Kt);  // Kt = Qm^-1 (R Fm)^T  => K = Kt^T
      if (a.loglik) {   // -1/2 (pm log 2pi + log det Qm + e^T Qm^-1 e) from the factor just computed
        for (int j = lane; j < pm; j += 64) yv[j] = ev[j];    // yv (the packed observations) is free now
        wsync();
        chol_solve_lds(lane, pm, 1, Qm, yv);
        double part = 0.0;
        for (int j = lane; j < pm; j += 64) part += 2.0 * log(Qm[CM(j, j, pm)]) + ev[j] * yv[j];
        for (int o_ = 32; o_ > 0; o_ >>= 1) part += __shfl_xor(part, o_);
        ll -= 0.5 * (1.8378770664093453 * pm + part);
      }
      // m = a + K e
      for (int i = lane; i < d; i += 64) {
        double s = av[i];
        for (int j = 0; j < pm; ++j) s = fma(Kt[CM(j, i, pm)], ev[j], s);
        m[i] = s;
      }
      // Joseph form (KalmanFilter.scala:87-90): D = I - K Fm^T ; C = D R D^T + K Vm K^T
      for (int k = lane; k < dd; k += 64) {
        const int i = k % d, j = k / d;
        double s = (i == j) ? 1.0 : 0.0;
        for (int l = 0; l < pm; ++l) s = fma(-Kt[CM(l, i, pm)], Ft[CM(j, idx[l], d)], s);
        T1[k] = s;
      }
      for (int k = lane; k < d * pm; k += 64) {  // KV = K Vm  (d x pm) -> RF buffer
        const int i = k % d, l = k / d;
        double s = 0.0;
        for (int r = 0; r < pm; ++r) s = fma(Kt[CM(r, i, pm)], V[CM(idx[r], idx[l], p)], s);
        RF[CM(i, l, d)] = s;
      }
      wsync();
      gemm<false, false>(lane, d, d, d, T1, d, R, d, T2, d);
      wsync();
      for (int k = lane; k < dd; k += 64) {
        const int i = k % d, j = k / d;
        double s = 0.0;
        for (int l = 0; l < d; ++l) s = fma(T2[CM(i, l, d)], T1[CM(j, l, d)], s);
        for (int l = 0; l < pm; ++l) s = fma(RF[CM(i, l, d)], Kt[CM(l, j, pm)], s);
        C[k] = s;
      }
    }
    wsync();
    if (out) {
      double* o = out + (size_t)(t + 1) * rec;
      for (int i = lane; i < d; i += 64) o[i] = m[i];
      for (int i = lane; i < dd; i += 64) o[d + i] = C[i];
    }
    if (any_nonfinite(lane, dd, C) || any_nonfinite(lane, d, m)) st |= DLM_ST_NONFINITE;
  }
  if (a.loglik && lane == 0) a.loglik[n] = ll;
  if (a.status && lane == 0) a.status[n] |= st;
}

// ---------------------------------------------------------------------------------------
// shared backward-pass prologue: from (m_t, C_t) recompute a_{t+1}, R_{t+1}, G C and
// J^T = R_{t+1}^-1 G C (cgrinv, Smoothing.scala:41 / :85).  On return T1 holds J^T,
// R holds chol(R_{t+1}), Rfull (optional) holds R_{t+1}.
// ---------------------------------------------------------------------------------------
__device__ bool backward_prologue(int lane, int d, const double* Gt, const double* W, double dt,
                                  const double* m, const double* C, double* a1, double* R,
                                  double* T1, double* Rfull) {
  const int dd = d * d;
  gemm<false, false>(lane, d, d, d, Gt, d, C, d, T1, d);  // G C
  if (dt == 0.0) {
    for (int i = lane; i < d; i += 64) a1[i] = m[i];
    for (int i = lane; i < dd; i += 64) R[i] = C[i];
    wsync();
  } else {
    gemm<false, false>(lane, d, 1, d, Gt, d, m, d, a1, d);
    wsync();
    gemm<false, true>(lane, d, d, d, T1, d, Gt, d, R, d);
    wsync();
    for (int i = lane; i < dd; i += 64) R[i] = fma(W[i], dt, R[i]);
    wsync();
  }
  if (Rfull) { for (int i = lane; i < dd; i += 64) Rfull[i] = R[i]; }
  wsync();
  const bool bad = chol_lds(lane, d, R);
  chol_solve_lds(lane, d, d, R, T1);  // T1 = R^-1 G C = J^T
  return bad;
}

// ---------------------------------------------------------------------------------------
// RTS smoother
// ---------------------------------------------------------------------------------------
size_t generic_smoother_lds_bytes(int d, int p) { return sizeof(double) * (size_t)(4 * d + 6 * d * d) + 16; }

__global__ __launch_bounds__(64) void k_smoother_generic(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, T = a.T, dd = d * d, rec = d + dd;
  double* m = sm;  double* a1 = m + d;  double* s = a1 + d;  double* u = s + d;
  double* C = u + d;  double* R = C + dd;  double* T1 = R + dd;  double* S = T1 + dd;
  double* X = S + dd;  double* T2 = X + dd;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* fin = a.filt_in + (size_t)n * (T + 1) * rec;
  double* out = a.smooth + (size_t)n * (T + 1) * rec;
  const bool compat = (a.flags & DLM_OPT_SMOOTHER_COMPAT_Q1) != 0;
  int st = 0;
  {  // init = last filter state (Smoothing.scala:59-61)
    const double* r = fin + (size_t)T * rec; double* o = out + (size_t)T * rec;
    for (int i = lane; i < d; i += 64) { s[i] = r[i]; o[i] = r[i]; }
    for (int i = lane; i < dd; i += 64) { S[i] = r[d + i]; o[d + i] = r[d + i]; }
  }
  for (int t = T - 1; t >= 0; --t) {
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
    const double dt = a.dt ? a.dt[t] : 1.0;
    const double* r = fin + (size_t)t * rec;
    wsync();
    for (int i = lane; i < d; i += 64) m[i] = r[i];
    for (int i = lane; i < dd; i += 64) C[i] = r[d + i];
    wsync();
    const double* Wt = W + (size_t)t * a.w_tstride;   // the transition into record t+1
    if (backward_prologue(lane, d, Gt, Wt, dt, m, C, a1, R, T1, X)) st |= DLM_ST_NOT_PD;
    // X = R_{t+1} - S_{t+1}; u = s_{t+1} - a_{t+1}
    for (int i = lane; i < dd; i += 64) X[i] -= S[i];
    for (int i = lane; i < d; i += 64) u[i] = s[i] - a1[i];
    wsync();
    // mean = m + J u ; T2 = J X  (J = T1^T)
    for (int i = lane; i < d; i += 64) {
      double acc = m[i];
      for (int k = 0; k < d; ++k) acc = fma(T1[CM(k, i, d)], u[k], acc);
      s[i] = acc;
    }
    gemm<true, false>(lane, d, d, d, T1, d, X, d, T2, d);
    wsync();
    // covariance = C - (J X) J^T  (textbook)   or   C - (J X) J  (Smoothing.scala:44, Q1)
    if (compat) gemm<false, true>(lane, d, d, d, T2, d, T1, d, X, d);
    else        gemm<false, false>(lane, d, d, d, T2, d, T1, d, X, d);
    wsync();
    for (int i = lane; i < dd; i += 64) S[i] = C[i] - X[i];
    wsync();
    double* o = out + (size_t)t * rec;
    for (int i = lane; i < d; i += 64) o[i] = s[i];
    for (int i = lane; i < dd; i += 64) o[d + i] = S[i];
    if (any_nonfinite(lane, dd, S) || any_nonfinite(lane, d, s)) st |= DLM_ST_NONFINITE;
  }
  if (a.status && lane == 0) a.status[n] |= st;
}

// ---------------------------------------------------------------------------------------
// Backward sampling + Gibbs sufficient statistics
// ---------------------------------------------------------------------------------------
size_t generic_sampler_lds_bytes(int d, int p) {
  return sizeof(double) * (size_t)(7 * d + 7 * d * d + 2 * p) + 16;
}

__global__ __launch_bounds__(64) void k_sampler_generic(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd;
  double* m = sm;   double* a1 = m + d;   double* th = a1 + d;  double* u = th + d;
  double* h = u + d;  double* zv = h + d;  double* ssv = zv + d;
  double* C = ssv + d;  double* R = C + dd;  double* T1 = R + dd;  double* D = T1 + dd;
  double* T2 = D + dd;  double* H = T2 + dd;  double* OUT = H + dd;
  double* ssy = OUT + dd;  double* nob = ssy + p;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* fin = a.filt_in + (size_t)n * (T + 1) * rec;
  const double* y = a.y ? a.y + (size_t)n * T * p : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * d : nullptr;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  double* cond = a.cond ? a.cond + (size_t)n * (T + 1) * rec : nullptr;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;
  const bool eig = (a.flags & DLM_OPT_DRAW_EIG) != 0;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  int st = 0;

  for (int i = lane; i < d; i += 64) ssv[i] = 0.0;
  for (int i = lane; i < dd; i += 64) OUT[i] = 0.0;
  for (int i = lane; i < p; i += 64) { ssy[i] = 0.0; nob[i] = 0.0; }
  // initialise (Smoothing.scala:105-109): theta_T ~ N(m_T, C_T)
  {
    const double* r = fin + (size_t)T * rec;
    for (int i = lane; i < d; i += 64) {
      h[i] = r[i];
      zv[i] = zin ? zin[(size_t)T * d + i] : philox_normal(a.seed, series, (unsigned)T, (unsigned)i);
    }
    for (int i = lane; i < dd; i += 64) H[i] = r[d + i];
    wsync();
    if (cond) {
      double* c = cond + (size_t)T * rec;
      for (int i = lane; i < d; i += 64) c[i] = h[i];
      for (int i = lane; i < dd; i += 64) c[d + i] = H[i];
    }
    wsync();
    if (eig) {   // the reference's factor: theta = m + E sqrt(Lambda) z (a rounding-level negative eigenvalue counts as zero and is flagged: the reference would draw NaN)
      eig_lds(lane, d, H, D, a1, u);
      for (int i = lane; i < d; i += 64) {
        double acc = h[i];
        for (int k = 0; k < d; ++k) { const double lk = a1[k]; acc = fma(H[CM(i, k, d)] * sqrt(lk > 0.0 ? lk : 0.0), zv[k], acc); }
        th[i] = acc;
      }
      if (a1[0] < 0.0) st |= DLM_ST_NOT_PD;
    } else {
    if (chol_lds(lane, d, H, true)) st |= DLM_ST_NOT_PD;
    for (int i = lane; i < d; i += 64) {
      double acc = h[i];
      for (int k = 0; k <= i; ++k) acc = fma(H[CM(i, k, d)], zv[k], acc);
      th[i] = acc;
    }
    }
    wsync();
    if (thout) for (int i = lane; i < d; i += 64) thout[(size_t)T * d + i] = th[i];
  }
  for (int t = T - 1; t >= 0; --t) {
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
    const double* Ft = a.F + (size_t)t * a.f_stride;
    const double dt = a.dt ? a.dt[t] : 1.0;
    const double* r = fin + (size_t)t * rec;
    wsync();
    // observation residual statistic of theta_{t+1} (Gibbs.scala:29-39)
    if (a.stats && y) {
      for (int j = lane; j < p; j += 64) {
        const double yj = y[(size_t)t * p + j];
        if (yj == yj) {
          double f = 0.0;
          for (int k = 0; k < d; ++k) f = fma(Ft[CM(k, j, d)], th[k], f);
          ssy[j] += (yj - f) * (yj - f);
          nob[j] += 1.0;
        }
      }
    }
    for (int i = lane; i < d; i += 64) {
      m[i] = r[i];
      zv[i] = zin ? zin[(size_t)t * d + i] : philox_normal(a.seed, series, (unsigned)t, (unsigned)i);
    }
    for (int i = lane; i < dd; i += 64) C[i] = r[d + i];
    wsync();
    const double* Wt = W + (size_t)t * a.w_tstride;   // the transition into record t+1
    if (backward_prologue(lane, d, Gt, Wt, dt, m, C, a1, R, T1, nullptr)) st |= DLM_ST_NOT_PD;
    for (int i = lane; i < d; i += 64) u[i] = th[i] - a1[i];
    wsync();
    // h = m + J u ; D = I - J G   (Smoothing.scala:88-93)
    for (int i = lane; i < d; i += 64) {
      double acc = m[i];
      for (int k = 0; k < d; ++k) acc = fma(T1[CM(k, i, d)], u[k], acc);
      h[i] = acc;
    }
    for (int k = lane; k < dd; k += 64) {
      const int i = k % d, j = k / d;
      double acc = (i == j) ? 1.0 : 0.0;
      for (int l = 0; l < d; ++l) acc = fma(-T1[CM(l, i, d)], Gt[CM(l, j, d)], acc);
      D[k] = acc;
    }
    wsync();
    gemm<false, false>(lane, d, d, d, D, d, C, d, T2, d);     // D C
    wsync();
    gemm<false, true>(lane, d, d, d, T2, d, D, d, H, d);      // D C D^T
    gemm<true, false>(lane, d, d, d, T1, d, Wt, d, R, d);      // J W -> R (its Cholesky factor is spent)
    wsync();
    for (int k = lane; k < dd; k += 64) {                      // H += dt * (J W) J^T
      const int i = k % d, j = k / d;
      double acc = 0.0;
      for (int l = 0; l < d; ++l) acc = fma(R[CM(i, l, d)], T1[CM(l, j, d)], acc);
      H[k] = fma(acc, dt, H[k]);
    }
    wsync();
    for (int k = lane; k < dd; k += 64) {                      // (H + H^T) / 2  (:95)
      const int i = k % d, j = k / d;
      if (i > j) { const double v = (H[CM(i, j, d)] + H[CM(j, i, d)]) / 2.0; T2[CM(i, j, d)] = v; T2[CM(j, i, d)] = v; }
      else if (i == j) T2[k] = H[k];
    }
    wsync();
    if (cond) {
      double* c = cond + (size_t)t * rec;
      for (int i = lane; i < d; i += 64) c[i] = h[i];
      for (int i = lane; i < dd; i += 64) c[d + i] = T2[i];
    }
    if (eig) {
      wsync();
      eig_lds(lane, d, T2, D, a1, u);                            // (D, a1 and u are free here)
      if (a1[0] < 0.0) st |= DLM_ST_NOT_PD;
      for (int i = lane; i < d; i += 64) {
        double acc = h[i];
        for (int k = 0; k < d; ++k) { const double lk = a1[k]; acc = fma(T2[CM(i, k, d)] * sqrt(lk > 0.0 ? lk : 0.0), zv[k], acc); }
        R[i] = acc;                                              // (R's d x d block is spent: theta_t parks there while u is scratch)
      }
      wsync();
      for (int i = lane; i < d; i += 64) u[i] = R[i];
    } else {
    if (chol_lds(lane, d, T2, true)) st |= DLM_ST_NOT_PD;
    // theta_t = h + L z ; system-innovation statistic of (theta_t, theta_{t+1})
    for (int i = lane; i < d; i += 64) {
      double acc = h[i];
      for (int k = 0; k <= i; ++k) acc = fma(T2[CM(i, k, d)], zv[k], acc);
      u[i] = acc;  // theta_t
    }
    }
    wsync();
    if (a.stats) {
      for (int i = lane; i < d; i += 64) {   // diff = theta_{t+1} - G theta_t  -> a1
        double acc = th[i];
        for (int k = 0; k < d; ++k) acc = fma(-Gt[CM(i, k, d)], u[k], acc);
        a1[i] = acc;
      }
      wsync();
      const double dts = (dt == 0.0) ? 1.0 : dt;
      if (outer) for (int k = lane; k < dd; k += 64) OUT[k] += a1[k % d] * a1[k / d] / dts;
      for (int i = lane; i < d; i += 64) ssv[i] += a1[i] * a1[i] / dts;
    }
    wsync();
    for (int i = lane; i < d; i += 64) th[i] = u[i];
    if (thout) for (int i = lane; i < d; i += 64) thout[(size_t)t * d + i] = u[i];
    if (any_nonfinite(lane, d, u)) st |= DLM_ST_NONFINITE;
  }
  wsync();
  if (a.stats) {
    const int L = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * L;
    for (int j = lane; j < p; j += 64) { so[j] = ssy[j]; so[p + j] = nob[j]; }
    if (outer) for (int k = lane; k < dd; k += 64) so[2 * p + k] = OUT[k];
    else for (int i = lane; i < d; i += 64) so[2 * p + i] = ssv[i];
    if (lane == 0) so[L - 1] = (double)T;
  }
  if (a.status && lane == 0) a.status[n] |= st;
}

// column sums of stats[N][L] -> pooled[L]; deterministic (fixed summation order per column)
__global__ void k_stats_pool(const double* stats, int N, int L, double* pooled) {
  const int c = blockIdx.x;
  __shared__ double part[256];
  double s = 0.0;
  for (int n = threadIdx.x; n < N; n += 256) s += stats[(size_t)n * L + c];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) pooled[c] = part[0];
}

// ---------------------------------------------------------------------------------------
// Simulation from the model (Dlm.simulateRegular / simStep / stepState / observation, Dlm.scala:245-292):
//   x_0 ~ N(m0, C0);  x_t = G_t x_{t-1} + w_t, w_t ~ N(0, W dt_t);  y_t = F_t^T x_t + v_t, v_t ~ N(0, V)
// over the model's time grid (the reference's regular grid is the special case dt = const).  Draws: lower
// Cholesky factors (the reference draws through an eigen-factor, MultivariateGaussianSvd.scala:13-22 -- the same
// distribution; its generator cannot be seeded) on the Philox stream (seed, series, record t, i):
// i < d the state noise (record 0: the initial state), d <= i < d + p the observation noise -- the convention of the
// simulation smoother.  a.theta = x [N][T+1][d] (nullable), a.smooth = y [N][T][p].
// ---------------------------------------------------------------------------------------
size_t generic_simulate_lds_bytes(int d, int p) { return sizeof(double) * (size_t)(2 * d * d + p * p + 3 * d + p + d + p) + 16; }

__global__ __launch_bounds__(64) void k_simulate_generic(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d;
  double* Lw = sm;           double* Lc = Lw + dd;     double* Lv = Lc + dd;
  double* x = Lv + p * p;    double* xn = x + d;       double* gx = xn + d;      double* fy = gx + d;
  double* z = fy + p;        // d + p normals of the current record
  const double* V0 = a.V + (size_t)n * a.v_stride;
  const double* W0 = a.W + (size_t)n * a.w_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  double* xo = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  double* yo = a.smooth + (size_t)n * T * p;
  int st = 0;
  for (int i = lane; i < dd; i += 64) { Lw[i] = W0[i]; Lc[i] = C0[i]; }
  for (int i = lane; i < p * p; i += 64) Lv[i] = V0[i];
  wsync();
  if (chol_lds(lane, d, Lw)) st |= DLM_ST_NOT_PD;
  if (chol_lds(lane, d, Lc)) st |= DLM_ST_NOT_PD;
  if (chol_lds(lane, p, Lv)) st |= DLM_ST_NOT_PD;
  for (int i = lane; i < d; i += 64) z[i] = philox_normal(a.seed, series, 0u, (unsigned)i);
  wsync();
  for (int i = lane; i < d; i += 64) {   // initialiseState (:268-273): x0 = m0 + chol(C0) z
    double acc = m0[i];
    for (int k = 0; k <= i; ++k) acc = fma(Lc[CM(i, k, d)], z[k], acc);
    x[i] = acc;
    if (xo) xo[i] = acc;
  }
  for (int t = 0; t < T; ++t) {
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
    const double* Ft = a.F + (size_t)t * a.f_stride;
    const double dt = a.dt ? a.dt[t] : 1.0, sdt = sqrt(dt);
    if (a.w_tstride || a.v_tstride) {   // time-varying variances: factor this step's matrices
      wsync();
      if (a.w_tstride) for (int i = lane; i < dd; i += 64) Lw[i] = (W0 + (size_t)t * a.w_tstride)[i];
      if (a.v_tstride) for (int i = lane; i < p * p; i += 64) Lv[i] = (V0 + (size_t)t * a.v_tstride)[i];
      wsync();
      if (a.w_tstride && chol_lds(lane, d, Lw)) st |= DLM_ST_NOT_PD;
      if (a.v_tstride && chol_lds(lane, p, Lv)) st |= DLM_ST_NOT_PD;
    }
    wsync();
    for (int i = lane; i < d + p; i += 64) z[i] = philox_normal(a.seed, series, (unsigned)(t + 1), (unsigned)i);
    gemm<false, false>(lane, d, 1, d, Gt, d, x, d, gx, d);      // G x
    wsync();
    for (int i = lane; i < d; i += 64) {                          // stepState (:245-255)
      double acc = gx[i];
      for (int k = 0; k <= i; ++k) acc = fma(Lw[CM(i, k, d)] * sdt, z[k], acc);
      xn[i] = acc;
    }
    wsync();
    gemm<true, false>(lane, p, 1, d, Ft, d, xn, d, fy, p);        // F^T x
    wsync();
    for (int j = lane; j < p; j += 64) {                          // observation (:257-266)
      double acc = fy[j];
      for (int k = 0; k <= j; ++k) acc = fma(Lv[CM(j, k, p)], z[d + k], acc);
      yo[(size_t)t * p + j] = acc;
    }
    for (int i = lane; i < d; i += 64) { x[i] = xn[i]; if (xo) xo[(size_t)(t + 1) * d + i] = xn[i]; }
  }
  wsync();
  if (any_nonfinite(lane, d, x)) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0) a.status[n] |= st;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static hipError_t check_lds(const void* fn, size_t bytes) {
  if (bytes > 160 * 1024) return hipErrorInvalidValue;
  if (bytes > 64 * 1024) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

hipError_t launch_generic_filter(const KArgs& a, hipStream_t s) {
  const size_t lds = generic_filter_lds_bytes(a.d, a.p);
  hipError_t e = check_lds((const void*)k_filter_generic, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_filter_generic, dim3(a.N), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_generic_smoother(const KArgs& a, hipStream_t s) {
  const size_t lds = generic_smoother_lds_bytes(a.d, a.p);
  hipError_t e = check_lds((const void*)k_smoother_generic, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_smoother_generic, dim3(a.N), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_generic_sampler(const KArgs& a, hipStream_t s) {
  const size_t lds = generic_sampler_lds_bytes(a.d, a.p);
  hipError_t e = check_lds((const void*)k_sampler_generic, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_sampler_generic, dim3(a.N), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_generic_simulate(const KArgs& a, hipStream_t s) {
  const size_t lds = generic_simulate_lds_bytes(a.d, a.p);
  hipError_t e = check_lds((const void*)k_simulate_generic, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_simulate_generic, dim3(a.N), dim3(64), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_stats_pool(const double* stats, int N, int L, double* pooled, hipStream_t s) {
  hipLaunchKernelGGL(k_stats_pool, dim3(L), dim3(256), 0, s, stats, N, L, pooled);
  return hipGetLastError();
}

}  // namespace dlm
