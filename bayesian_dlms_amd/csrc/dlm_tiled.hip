// Multivariate path: 16 <= d <= 48, p <= 32 (config C4: d = 40, p = 20).
//
// One 256-thread workgroup (4 wavefronts) per series.  Every matrix of the recursion lives in the
// workgroup's LDS, zero-padded to multiples of 16 with a leading dimension of 49 (d-wide) or 33
// (p-wide) doubles -- odd, so that the strided A-operand reads are bank-conflict free -- and every
// O(d^3) / O(d^2 p) product is an MFMA GEMM: the 16 x 16 output tiles are dealt round-robin to the
// four waves, each tile accumulating ceil(k/4) v_mfma_f64_16x16x4_f64 whose A / B operands are one
// ds_read_b64 per lane each (500x less LDS traffic than a scalar LDS GEMM).  Missing observation
// components, irregular dt / several G, time-varying F and per-series parameters are supported.
//
// Forward (KalmanFilter.scala:64-118,273-286,311-321):
//   R = G C G^T + W dt,  Q = F^T R F + V,  K = R F Qm^-1,  m = a + K e,  C = R - K (R F)^T
//   (the last is the Joseph form of the reference with K Qm = R Fm substituted; equal in exact
//   arithmetic).  Missing components are decoupled by replacing their rows/columns of Q with the
//   identity before the inverse and zeroing them afterwards, which yields exactly inv(Q[obs, obs]).
// Backward (Smoothing.scala:31-64): the RTS smoothing distribution in information form,
//   s_t = m_t + C_t q_t,  S_t = C_t - C_t P_t C_t,
//   K_t = C_t F Vm^-1,  Qm^-1 = Vm^-1 - Vm^-1 F^T K,  e_t = y_t - F^T G m_{t-1},  u = Qm^-1 e,
//   q_{t-1} = G^T [ q + F (u - K^T q) ],
//   P_{t-1} = G^T [ P + F (Qm^-1 + K^T P K) F^T - F (P K)^T - (P K) F^T ] G,
//   which needs no d x d solve and no R_{t+1}; only a p x p SPD inverse per distinct mask.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int DL = 49;          // leading dimension of d-wide matrices (up to 48 columns)
constexpr int PL = 33;          // leading dimension of p-wide matrices (up to 32 columns)
constexpr int BIG = 48 * DL;    // doubles in a 48 x d-wide matrix
constexpr int MID = 48 * PL;    // doubles in a 48 x p-wide matrix
constexpr int SML = 32 * PL;    // doubles in a 32 x p-wide matrix

// C (mt x nt tiles, mt * nt <= 12) = D -/+ op(A) op(B), all row-major in LDS.  MODE 0: C = acc, 1: C = D + acc,
// 2: C = D - acc (D may alias C).  kb = number of 4-deep k-blocks.
template <bool TA, bool TB, int MODE>
__device__ __forceinline__ void gemm_t(int tid, int mt, int nt, int kb, const double* A, int lda, const double* B,
                                       int ldb, double* C, int ldc, const double* D = nullptr) {
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int ntiles = mt * nt;   // <= 9 on this path: every wave owns tiles wave, wave + 4, wave + 8
  // The (up to) three tiles of a wave accumulate in three independent MFMA chains, so the dependent-
  // accumulator latency of one chain (~3x the issue interval) is covered by the other two.
  int ao[3], bo[3];
  bool on[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int tile = wave + 4 * q;
    on[q] = tile < ntiles;
    const int i0 = on[q] ? (tile / nt) * 16 : 0, j0 = on[q] ? (tile % nt) * 16 : 0;
    ao[q] = TA ? g * lda + i0 + c : (i0 + c) * lda + g;      // op(A)[i0 + c][g]     (+ 4 kk along k)
    bo[q] = TB ? (j0 + c) * ldb + g : g * ldb + j0 + c;      // op(B)[g][j0 + c]
  }
  const int as = TA ? 4 * lda : 4, bs = TB ? 4 : 4 * ldb;    // stride of one k-block
  d4 acc[3] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  for (int kk = 0; kk < kb; ++kk) {
    double av[3], bv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { av[q] = A[ao[q] + kk * as]; bv[q] = B[bo[q] + kk * bs]; }
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc[q], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (!on[q]) continue;
    const int tile = wave + 4 * q;
    const int i0 = (tile / nt) * 16, j0 = (tile % nt) * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (i0 + 4 * r + g) * ldc + j0 + c;
      if (MODE == 0) C[o] = acc[q][r];
      else if (MODE == 1) C[o] = D[o] + acc[q][r];
      else C[o] = D[o] - acc[q][r];
    }
  }
}

// In-place inverse of the SPD n x n (n <= 32) LDS matrix A (ld PL) via Cholesky; Li is scratch.
// Returns true if a non-positive pivot was met.  All 256 threads must call.
__device__ bool spd_inverse(int tid, int n, double* A, double* Li) {
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    __syncthreads();
    double akk = A[k * PL + k];
    if (!(akk > 0.0)) { bad = true; akk = 1e-300; }
    const double lkk = sqrt(akk), inv = 1.0 / lkk;
    __syncthreads();
    if (tid >= k && tid < n) A[tid * PL + k] = (tid == k) ? lkk : A[tid * PL + k] * inv;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += 256) {
      const int i = idx / n, j = idx % n;
      if (j > k && i >= j) A[i * PL + j] = fma(-A[i * PL + k], A[j * PL + k], A[i * PL + j]);
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 32 * PL; idx += 256) Li[idx] = 0.0;
  __syncthreads();
  if (tid < n) {   // column tid of L^-1 by forward substitution
    const int j = tid;
    Li[j * PL + j] = 1.0 / A[j * PL + j];
    for (int i = j + 1; i < n; ++i) {
      double s = 0.0;
      for (int l = j; l < i; ++l) s = fma(A[i * PL + l], Li[l * PL + j], s);
      Li[i * PL + j] = -s / A[i * PL + i];
    }
  }
  __syncthreads();
  const int nt = (n + 15) / 16, kb = (n + 3) / 4;
  gemm_t<true, false, 0>(tid, nt, nt, kb, Li, PL, Li, PL, A, PL);   // A^-1 = L^-T L^-1
  __syncthreads();
  return bad;
}

__device__ __forceinline__ void zero_lds(int tid, double* p, int n) { for (int i = tid; i < n; i += 256) p[i] = 0.0; }

// column-major global d x d (or d x p) -> zero-padded row-major LDS
__device__ __forceinline__ void load_cm(int tid, const double* src, int rows, int cols, double* dst, int ld) {
  for (int idx = tid; idx < rows * cols; idx += 256) { const int i = idx % rows, j = idx / rows; dst[i * ld + j] = src[idx]; }
}

bool tiled_supported(const KArgs& a) { return a.d >= 16 && a.d <= 48 && a.p <= 32; }

constexpr int FILT_DOUBLES = 4 * BIG + 3 * MID + 3 * SML + 8 * 48;    // 141.7 KB
constexpr int SMTH_DOUBLES = 4 * BIG + 3 * MID + 4 * SML + 10 * 48;   // 150.9 KB
size_t tiled_filter_lds_bytes() { return sizeof(double) * FILT_DOUBLES + 16; }
size_t tiled_smoother_lds_bytes() { return sizeof(double) * SMTH_DOUBLES + 16; }

// ---------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_filter_tiled(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, frec = p + p * p;
  const int dt16 = (d + 15) / 16, pt16 = (p + 15) / 16, kd = (d + 3) / 4, kp = (p + 3) / 4;
  double* C = sm;            double* R = C + BIG;     double* Tm = R + BIG;    double* Gm = Tm + BIG;
  double* Fm = Gm + BIG;     double* RF = Fm + MID;   double* Kg = RF + MID;
  double* Q = Kg + MID;      double* Qi = Q + SML;    double* Li = Qi + SML;
  double* mv = Li + SML;     double* av = mv + 48;    double* ev = av + 48;    double* fv = ev + 48;
  double* ob = fv + 48;      // observed flags (1.0 / 0.0)
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* y = a.y + (size_t)n * T * p;
  double* out = a.filt + (size_t)n * (T + 1) * rec;
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * frec : nullptr;
  int st = 0;

  zero_lds(tid, sm, FILT_DOUBLES);
  __syncthreads();
  load_cm(tid, a.C0 + (size_t)n * a.c0_stride, d, d, C, DL);
  load_cm(tid, a.F, d, p, Fm, PL);
  int gcur = a.g_index ? a.g_index[0] : 0;
  load_cm(tid, a.G + (size_t)gcur * dd, d, d, Gm, DL);
  if (tid < d) mv[tid] = (a.m0 + (size_t)n * a.m0_stride)[tid];
  __syncthreads();
  for (int idx = tid; idx < dd; idx += 256) out[d + idx] = C[(idx % d) * DL + idx / d];
  if (tid < d) out[tid] = mv[tid];
  if (fq) for (int i = tid; i < frec; i += 256) fq[i] = __builtin_nan("");

  for (int t = 0; t < T; ++t) {
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) { __syncthreads(); load_cm(tid, a.G + (size_t)gi * dd, d, d, Gm, DL); gcur = gi; }
    if (a.f_stride) { __syncthreads(); load_cm(tid, a.F + (size_t)t * a.f_stride, d, p, Fm, PL); }
    __syncthreads();
    // advState: a = G m, R = G C G^T + W dt   (dt == 0: a = m, R = C)
    if (dt == 0.0) {
      for (int idx = tid; idx < 48 * DL; idx += 256) R[idx] = C[idx];
      if (tid < d) av[tid] = mv[tid];
    } else {
      gemm_t<false, false, 0>(tid, dt16, dt16, kd, Gm, DL, C, DL, Tm, DL);
      if (tid < d) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Gm[tid * DL + k], mv[k], s); av[tid] = s; }
      __syncthreads();
      gemm_t<false, true, 0>(tid, dt16, dt16, kd, Tm, DL, Gm, DL, R, DL);
      __syncthreads();
      for (int idx = tid; idx < dd; idx += 256) { const int i = idx % d, j = idx / d; R[i * DL + j] = fma(W[idx], dt, R[i * DL + j]); }
    }
    __syncthreads();
    // forecast: f = F^T a, RF = R F, Q = F^T R F + V
    gemm_t<false, false, 0>(tid, dt16, pt16, kd, R, DL, Fm, PL, RF, PL);
    if (tid < p) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Fm[k * PL + tid], av[k], s); fv[tid] = s; }
    __syncthreads();
    gemm_t<true, false, 0>(tid, pt16, pt16, kd, Fm, PL, RF, PL, Q, PL);
    __syncthreads();
    for (int idx = tid; idx < p * p; idx += 256) { const int i = idx % p, j = idx / p; Q[i * PL + j] += V[idx]; }
    if (tid < p) {
      const double yv = y[(size_t)t * p + tid];
      ob[tid] = (yv == yv) ? 1.0 : 0.0;
      ev[tid] = (yv == yv) ? yv - fv[tid] : 0.0;
    }
    __syncthreads();
    if (fq) {
      double* fr = fq + (size_t)(t + 1) * frec;
      if (tid < p) fr[tid] = fv[tid];
      for (int idx = tid; idx < p * p; idx += 256) fr[p + idx] = Q[(idx % p) * PL + idx / p];
    }
    bool any = false;
    for (int j = 0; j < p; ++j) any |= ob[j] != 0.0;
    if (!any) {   // updateState :74-75
      __syncthreads();
      for (int idx = tid; idx < 48 * DL; idx += 256) C[idx] = R[idx];
      if (tid < d) mv[tid] = av[tid];
    } else {
      // Qm: missing rows/columns -> identity; inverse; back to zero
      __syncthreads();
      for (int idx = tid; idx < p * p; idx += 256) {
        const int i = idx / p, j = idx % p;
        Qi[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? Q[i * PL + j] : (i == j ? 1.0 : 0.0);
      }
      if (spd_inverse(tid, p, Qi, Li)) st |= DLM_ST_NOT_PD;
      for (int idx = tid; idx < p * p; idx += 256) {
        const int i = idx / p, j = idx % p;
        if (!(ob[i] != 0.0 && ob[j] != 0.0)) Qi[i * PL + j] = 0.0;
      }
      __syncthreads();
      gemm_t<false, false, 0>(tid, dt16, pt16, kp, RF, PL, Qi, PL, Kg, PL);          // K = R F Qm^-1
      __syncthreads();
      if (tid < d) { double s = av[tid]; for (int j = 0; j < p; ++j) s = fma(Kg[tid * PL + j], ev[j], s); mv[tid] = s; }
      gemm_t<false, true, 2>(tid, dt16, dt16, kp, Kg, PL, RF, PL, C, DL, R);          // C = R - K (R F)^T
    }
    __syncthreads();
    double* o = out + (size_t)(t + 1) * rec;
    if (tid < d) o[tid] = mv[tid];
    for (int idx = tid; idx < dd; idx += 256) o[d + idx] = C[(idx % d) * DL + idx / d];
  }
  __syncthreads();
  bool bad = false;
  for (int idx = tid; idx < dd; idx += 256) bad |= !isfinite(C[(idx % d) * DL + idx / d]);
  if (tid < d) bad |= !isfinite(mv[tid]);
  if (__syncthreads_or(bad)) st |= DLM_ST_NONFINITE;
  if (a.status && tid == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// backward pass (information form, general p)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_smoother_tiled(KArgs a) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd;
  const int dt16 = (d + 15) / 16, pt16 = (p + 15) / 16, kd = (d + 3) / 4, kp = (p + 3) / 4;
  double* C = sm;            double* P = C + BIG;     double* T1 = P + BIG;    double* T2 = T1 + BIG;
  double* Fm = T2 + BIG;     double* Kg = Fm + MID;   double* PK = Kg + MID;
  double* Vi = PK + MID;     double* Qi = Vi + SML;   double* X = Qi + SML;    double* Li = X + SML;
  double* mv = Li + SML;     double* mp = mv + 48;    double* qv = mp + 48;    double* rv = qv + 48;
  double* ev = rv + 48;      double* uv = ev + 48;    double* ob = uv + 48;    double* obp = ob + 48;
  double* tv = obp + 48;     double* cq = tv + 48;
  double* CF = T2;           // d x p scratch aliases (T2 is free while K is built); uses ld PL
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* y = a.y + (size_t)n * T * p;
  const double* fin = a.filt_in + (size_t)n * (T + 1) * rec;
  double* out = a.smooth + (size_t)n * (T + 1) * rec;
  int st = 0;

  zero_lds(tid, sm, SMTH_DOUBLES);
  __syncthreads();
  load_cm(tid, a.F, d, p, Fm, PL);
  if (tid < p) obp[tid] = -1.0;   // mask of the cached Vm^-1 (none yet)
  __syncthreads();

  for (int t = T; t >= 0; --t) {
    const double* r = fin + (size_t)t * rec;
    __syncthreads();
    load_cm(tid, r + d, d, d, C, DL);
    if (tid < d) { mv[tid] = r[tid]; mp[tid] = (t > 0) ? (r - rec)[tid] : 0.0; }
    if (a.f_stride && t > 0) load_cm(tid, a.F + (size_t)(t - 1) * a.f_stride, d, p, Fm, PL);
    if (tid < p) { const double yv = (t > 0) ? y[(size_t)(t - 1) * p + tid] : __builtin_nan(""); ob[tid] = (yv == yv) ? 1.0 : 0.0; tv[tid] = yv; }
    __syncthreads();
    bool any = false, same = true;
    for (int j = 0; j < p; ++j) { any |= ob[j] != 0.0; same &= ob[j] == obp[j]; }
    const double* Gt = a.G + (size_t)((a.g_index && t > 0) ? a.g_index[t - 1] : 0) * dd;   // G of the step INTO record t
    const double dtt = (a.dt && t > 0) ? a.dt[t - 1] : 1.0;

    if (any) {
      if (!same) {   // Vm^-1 for this missingness pattern (cached while the pattern repeats)
        __syncthreads();
        for (int idx = tid; idx < p * p; idx += 256) {
          const int i = idx / p, j = idx % p;
          Vi[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? V[i + j * p] : (i == j ? 1.0 : 0.0);
        }
        if (spd_inverse(tid, p, Vi, Li)) st |= DLM_ST_NOT_PD;
        for (int idx = tid; idx < p * p; idx += 256) {
          const int i = idx / p, j = idx % p;
          if (!(ob[i] != 0.0 && ob[j] != 0.0)) Vi[i * PL + j] = 0.0;
        }
        if (tid < p) obp[tid] = ob[tid];
        __syncthreads();
      }
      gemm_t<false, false, 0>(tid, dt16, pt16, kd, C, DL, Fm, PL, CF, PL);            // C F
      // e = y - F^T G m_{t-1}  (a_t = G m_{t-1}; identity advance when dt == 0)
      if (tid < d) {
        double s = 0.0;
        if (dtt == 0.0) s = mp[tid];
        else for (int k = 0; k < d; ++k) s = fma(Gt[tid + k * d], mp[k], s);
        rv[tid] = s;   // a_t (rv is free here)
      }
      __syncthreads();
      gemm_t<false, false, 0>(tid, dt16, pt16, kp, CF, PL, Vi, PL, Kg, PL);           // K = C F Vm^-1
      if (tid < p) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Fm[k * PL + tid], rv[k], s); ev[tid] = (ob[tid] != 0.0) ? tv[tid] - s : 0.0; }
      __syncthreads();
      gemm_t<true, false, 0>(tid, pt16, pt16, kd, Fm, PL, Kg, PL, X, PL);             // F^T K
      __syncthreads();
      gemm_t<false, false, 2>(tid, pt16, pt16, kp, Vi, PL, X, PL, Qi, PL, Vi);        // Qm^-1 = Vm^-1 - Vm^-1 F^T K
      __syncthreads();
      if (tid < p) { double s = 0.0; for (int j = 0; j < p; ++j) s = fma(Qi[tid * PL + j], ev[j], s); uv[tid] = s; }
    }
    __syncthreads();
    // x1 = P C -> T1 ; P K ; x2 = C (P C) -> T2 ; C q
    gemm_t<false, false, 0>(tid, dt16, dt16, kd, P, DL, C, DL, T1, DL);
    if (any) gemm_t<false, false, 0>(tid, dt16, pt16, kd, P, DL, Kg, PL, PK, PL);
    if (tid < d) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(C[tid * DL + k], qv[k], s); cq[tid] = s; }
    __syncthreads();
    gemm_t<false, false, 0>(tid, dt16, dt16, kd, C, DL, T1, DL, T2, DL);
    __syncthreads();
    double* o = out + (size_t)t * rec;
    if (tid < d) o[tid] = mv[tid] + cq[tid];                                          // s_t = m_t + C_t q_t
    for (int idx = tid; idx < dd; idx += 256) { const int i = idx % d, j = idx / d; o[d + idx] = C[i * DL + j] - T2[i * DL + j]; }  // S_t
    if (t == 0) break;

    // (q_{t-1}, P_{t-1})
    __syncthreads();
    if (any) {
      gemm_t<true, false, 1>(tid, pt16, pt16, kd, Kg, PL, PK, PL, X, PL, Qi);         // X = Qm^-1 + K^T P K
      if (tid < p) { double s = uv[tid]; for (int k = 0; k < d; ++k) s = fma(-Kg[k * PL + tid], qv[k], s); tv[tid] = s; }   // u - K^T q
      __syncthreads();
      gemm_t<false, false, 0>(tid, dt16, pt16, kp, Fm, PL, X, PL, CF, PL);            // F X   (T2 is free again)
      if (tid < d) { double s = qv[tid]; for (int j = 0; j < p; ++j) s = fma(Fm[tid * PL + j], tv[j], s); rv[tid] = s; }     // r
      __syncthreads();
      gemm_t<false, true, 1>(tid, dt16, dt16, kp, CF, PL, Fm, PL, P, DL, P);          // P += F X F^T
      __syncthreads();
      gemm_t<false, true, 2>(tid, dt16, dt16, kp, Fm, PL, PK, PL, P, DL, P);          // P -= F (P K)^T
      __syncthreads();
      gemm_t<false, true, 2>(tid, dt16, dt16, kp, PK, PL, Fm, PL, P, DL, P);          // P -= (P K) F^T   => M
      __syncthreads();
      // The expanded update treats P as exactly symmetric (it uses (P K)^T for K^T P).  Without this
      // symmetrisation the antisymmetric rounding component is NOT contracted by (I - F K^T) and grows
      // exponentially for unit-root models (polynomial trends with dense W): see DESIGN.md 4.3.
      for (int idx = tid; idx < dd; idx += 256) {
        const int i = idx / d, j = idx % d;
        if (i < j) { const double v = 0.5 * (P[i * DL + j] + P[j * DL + i]); P[i * DL + j] = v; P[j * DL + i] = v; }
      }
    } else if (tid < d) rv[tid] = qv[tid];
    __syncthreads();
    if (dtt == 0.0) {
      if (tid < d) qv[tid] = rv[tid];                                                  // identity advance: P = M, q = r
    } else {
      zero_lds(tid, T2, BIG);                                                          // keep the zero padding exact
      __syncthreads();
      load_cm(tid, Gt, d, d, T2, DL);                                                  // G -> T2
      __syncthreads();
      gemm_t<false, false, 0>(tid, dt16, dt16, kd, P, DL, T2, DL, T1, DL);             // M G
      if (tid < d) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(T2[k * DL + tid], rv[k], s); qv[tid] = s; }        // q = G^T r
      __syncthreads();
      gemm_t<true, false, 0>(tid, dt16, dt16, kd, T2, DL, T1, DL, P, DL);              // P = G^T M G
    }
  }
  __syncthreads();
  bool bad = false;
  for (int idx = tid; idx < dd; idx += 256) bad |= !isfinite(T2[(idx % d) * DL + idx / d]) || !isfinite(C[(idx % d) * DL + idx / d]);
  if (__syncthreads_or(bad)) st |= DLM_ST_NONFINITE;
  if (a.status && tid == 0 && st) atomicOr(&a.status[n], st);
}

static hipError_t set_lds(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_tiled_filter(const KArgs& a, hipStream_t s) {
  const size_t lds = tiled_filter_lds_bytes();
  hipError_t e = set_lds((const void*)k_filter_tiled, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_filter_tiled, dim3(a.N), dim3(256), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_tiled_smoother(const KArgs& a, hipStream_t s) {
  const size_t lds = tiled_smoother_lds_bytes();
  hipError_t e = set_lds((const void*)k_smoother_tiled, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_smoother_tiled, dim3(a.N), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace dlm
