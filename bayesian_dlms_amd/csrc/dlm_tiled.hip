// Multivariate path: 16 <= d <= 48, p <= 32 (config C4: d = 40, p = 20).
//
// One 512-thread workgroup (8 wavefronts, two per SIMD) per series.  Every matrix of the recursion lives in the
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

#include <type_traits>

namespace dlm {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512;         // threads per workgroup: 8 waves, two per SIMD (latency hiding, 1 workgroup per CU)
constexpr int NW = NT / 64;
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
  const int ntiles = mt * nt;   // <= 9 on this path: wave w owns tiles w and w + 8
  if (wave >= ntiles) return;
  // two independent MFMA accumulation chains per wave (and two waves per SIMD) cover the
  // dependent-accumulator latency of a chain
  int ao[2], bo[2];
  bool on[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave + NW * q;
    on[q] = tile < ntiles;
    const int i0 = on[q] ? (tile / nt) * 16 : 0, j0 = on[q] ? (tile % nt) * 16 : 0;
    ao[q] = TA ? g * lda + i0 + c : (i0 + c) * lda + g;      // op(A)[i0 + c][g]     (+ 4 kk along k)
    bo[q] = TB ? (j0 + c) * ldb + g : g * ldb + j0 + c;      // op(B)[g][j0 + c]
  }
  const int as = TA ? 4 * lda : 4, bs = TB ? 4 : 4 * ldb;    // stride of one k-block
  d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  if (on[1]) {
    for (int kk = 0; kk < kb; ++kk) {
      const double a0 = A[ao[0] + kk * as], b0 = B[bo[0] + kk * bs], a1 = A[ao[1] + kk * as], b1 = B[bo[1] + kk * bs];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1], 0, 0, 0);
    }
  } else {
    for (int kk = 0; kk < kb; ++kk)
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(A[ao[0] + kk * as], B[bo[0] + kk * bs], acc[0], 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (!on[q]) continue;
    const int tile = wave + NW * q;
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

__device__ __forceinline__ void wsync() {   // LDS hand-off inside ONE wavefront (in-order LDS queue)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double bcast_lane(double v, int src) {   // lane `src` -> SGPR pair (uniform)
  const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// In-place inverse of the SPD n x n (n <= 32) LDS matrix A (ld PL); Li is scratch.
// Register-resident on wave 0: lane i holds row i (32 doubles); the Cholesky pivot quantities and the
// entries of L needed by every lane are broadcast with v_readlane into SGPRs, so the n sequential
// pivots cost no LDS round trip and no barrier (the LDS version spent 64k of 89k cycles per step
// here).  Then X = L^-1 by forward substitution (lane = column of X) and A^-1 = X^T X by MFMA.
// Every thread must call; ends with one workgroup barrier.  Returns whether a pivot was non-positive.
// compile-time loop: the register arrays below must only ever be indexed by constants
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ bool spd_inverse(int tid, int n_, double* A, double* Li, int* bad_flag) {
  const int n = __builtin_amdgcn_readfirstlane(n_);   // uniform: the guards below must be scalar branches
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    double a[32], x[32];
    static_for<0, 32>([&](auto J) { constexpr int j = J; a[j] = (lane < n && j < n) ? A[lane * PL + j] : ((lane == j) ? 1.0 : 0.0); });
    bool bad = false;
    static_for<0, 32>([&](auto Kc) {
      constexpr int k = Kc;
      if (k < n) {
        double akk = bcast_lane(a[k], k);
        if (!(akk > 0.0)) { bad = true; akk = 1e-300; }
        const double lkk = sqrt(akk), inv = 1.0 / lkk;
        a[k] = (lane == k) ? lkk : a[k] * inv;               // column k of L (rows >= k are meaningful)
        static_for<k + 1, 32>([&](auto J) {
          constexpr int j = J;
          if (j < n) a[j] = fma(-a[k], bcast_lane(a[k], j), a[j]);   // A[i][j] -= L[i][k] L[j][k]
        });
      }
    });
    // X = L^-1, lane c = column c:  x[i] = ((i == c) - sum_{l<i} L[i][l] x[l]) / L[i][i]
    double dinv = 1.0;
    static_for<0, 32>([&](auto Kc) { constexpr int k = Kc; dinv = (lane == k) ? 1.0 / a[k] : dinv; });   // 1 / L[lane][lane]
    static_for<0, 32>([&](auto Ic) {
      constexpr int i = Ic;
      x[i] = 0.0;
      if (i < n) {
        double acc = (lane == i) ? 1.0 : 0.0;
        static_for<0, i>([&](auto Lc) { constexpr int l = Lc; acc = fma(-bcast_lane(a[l], i), x[l], acc); });
        x[i] = acc * bcast_lane(dinv, i);
      }
    });
    for (int idx = lane; idx < 32 * PL; idx += 64) Li[idx] = 0.0;
    wsync();
    if (lane < n) static_for<0, 32>([&](auto Ic) { constexpr int i = Ic; if (i < n) Li[i * PL + lane] = x[i]; });
    wsync();
    // A^-1 = X^T X : up to 2 x 2 tiles, all on this wave
    const int nt = (n + 15) / 16, kb = (n + 3) / 4, g = lane >> 4, c = lane & 15;
    for (int tile = 0; tile < nt * nt; ++tile) {
      const int i0 = (tile / nt) * 16, j0 = (tile % nt) * 16;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int kk = 0; kk < kb; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Li[(4 * kk + g) * PL + i0 + c], Li[(4 * kk + g) * PL + j0 + c], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) A[(i0 + 4 * r + g) * PL + j0 + c] = acc[r];
    }
    if (lane == 0) *bad_flag = bad ? 1 : 0;
  }
  __syncthreads();
  return *bad_flag != 0;
}

__device__ __forceinline__ void zero_lds(int tid, double* p, int n) { for (int i = tid; i < n; i += NT) p[i] = 0.0; }

// element loop over a rows x cols column-major block without integer division: wave -> column, lane -> row
#define FOR_CM(rows, cols, i, j) for (int j = tid >> 6; j < (cols); j += NW) for (int i = tid & 63; i < (rows); i += 64)

// column-major global d x d (or d x p) -> zero-padded row-major LDS
__device__ __forceinline__ void load_cm(int tid, const double* src, int rows, int cols, double* dst, int ld) {
  FOR_CM(rows, cols, i, j) dst[i * ld + j] = src[i + j * rows];
}

// In-place lower Cholesky of the n x n LDS matrix A (row-major, leading dimension ld); one-off per series
// (simulation-smoother set-up), so a plain right-looking factorisation with workgroup barriers.
__device__ bool chol_block(int tid, int n, double* A, int ld) {
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    __syncthreads();
    double akk = A[k * ld + k];
    if (!(akk > 0.0)) { bad = true; akk = 1e-300; }
    const double lkk = sqrt(akk), inv = 1.0 / lkk;
    __syncthreads();
    if (tid >= k && tid < n) A[tid * ld + k] = (tid == k) ? lkk : A[tid * ld + k] * inv;
    __syncthreads();
    for (int j = (tid >> 6) + k + 1; j < n; j += NW)
      for (int i = (tid & 63); i < n; i += 64)
        if (i >= j) A[i * ld + j] = fma(-A[i * ld + k], A[j * ld + k], A[i * ld + j]);
  }
  __syncthreads();
  for (int j = (tid >> 6); j < n; j += NW) for (int i = (tid & 63); i < n; i += 64) if (i < j) A[i * ld + j] = 0.0;
  __syncthreads();
  return bad;
}

// Inverse of the SPD n x n matrix Q (LDS, ld PL) by Newton-Schulz refinement of a warm start:
//   E = I - Q X ,  X <- X + X E        (||E|| squares every iteration)
// X holds the inverse of the previous time step's Q on entry (Q_t changes slowly: one or two iterations
// at steady state) and the verified inverse on exit.  If the
// warm start is too far off (first step, missingness pattern changed: n max|E| >= 0.5) or it has not
// converged (max|E| <= 2e-10) in 6 iterations, the direct register Cholesky takes over.  E and Tn are p x p scratch.
// Every thread must call.  Returns whether the direct path met a non-positive pivot.
__device__ __forceinline__ bool spd_inverse_warm(int tid, int n, const double* Q, double* X, double* E, double* Tn,
                                                 double* Li, int* flag, bool have_warm, int* dbg = nullptr) {
  const int nt = (n + 15) / 16, kb = (n + 3) / 4;
  const double tol = 2e-10;   // the fp64 floor of max|I - Q X| is ~n cond(Q) eps (1e-11 here, no better for the direct inverse)
  bool done = false;
  if (have_warm) {
    for (int it = 0; it < 6 && !done; ++it) {
      __syncthreads();
      gemm_t<false, false, 0>(tid, nt, nt, kb, Q, PL, X, PL, E, PL);            // Q X
      __syncthreads();
      bool big = false, far = false;
      FOR_CM(n, n, i, j) {
        const double e = ((i == j) ? 1.0 : 0.0) - E[i * PL + j];
        E[i * PL + j] = e;
        big |= !(fabs(e) <= tol);
        far |= !(fabs(e) * n < 0.5);
      }
      const int any_far = __syncthreads_or(far);
      const int any_big = __syncthreads_or(big);
      if (dbg && tid == 0) { dbg[0] += 1; if (any_far) dbg[2] += 1; }
      if (any_far && any_big) break;                                             // not contractive enough: go direct
      if (!any_big) done = true;   // ||E|| <= 2e-10: the update below squares it, i.e. lands on the fp64 floor
      gemm_t<false, false, 1>(tid, nt, nt, kb, X, PL, E, PL, Tn, PL, X);         // X + X E
      __syncthreads();
      FOR_CM(n, n, i, j) X[i * PL + j] = 0.5 * (Tn[i * PL + j] + Tn[j * PL + i]);  // keep it symmetric
    }
    __syncthreads();
  }
  if (done) return false;
  if (dbg && tid == 0) dbg[1] += 1;
  FOR_CM(n, n, i, j) X[i * PL + j] = Q[i * PL + j];
  return spd_inverse(tid, n, X, Li, flag);
}

#ifdef DLM_STAMP
#define TSTAMP(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); seg[k] += _t - tlast; tlast = _t; }
#else
#define TSTAMP(k)
#endif

bool tiled_supported(const KArgs& a) { return a.d >= 16 && a.d <= 48 && a.p <= 32; }

constexpr int FILT_DOUBLES = 4 * BIG + 3 * MID + 2 * SML + 8 * 48;    // 133 KB (inverse scratch aliases Tm / Kg)
constexpr int FILT_SIM_DOUBLES = FILT_DOUBLES + BIG + SML + 3 * 48;   // + chol(W), chol(V), x+ (48), normals (96): 158 KB
constexpr int SIMS_DOUBLES = 2 * BIG + 3 * MID + 4 * SML + 16 * 48;   // mean-only backward pass: 115 KB
constexpr int SMTH_DOUBLES = 4 * BIG + 3 * MID + 4 * SML + 10 * 48;   // 150.9 KB
size_t tiled_filter_lds_bytes() { return sizeof(double) * FILT_DOUBLES + 16; }
size_t tiled_smoother_lds_bytes() { return sizeof(double) * SMTH_DOUBLES + 16; }

// ---------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------
// SIM: first half of the Durbin-Koopman simulation smoother (see dlm_sparse16.hip): simulate (x+, y+),
// filter y* = y - y+ from a zero prior mean, write x+ [N][T+1][d] and y* [N][T][p].  Normals of record t:
// components 0..d-1 state noise (record 0: the initial state), d..d+p-1 observation noise.
template <bool SIM>
__global__ __launch_bounds__(NT) void k_filter_tiled(KArgs a, double* __restrict__ xplus, double* __restrict__ ystar) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, frec = p + p * p;
  const int dt16 = (d + 15) / 16, pt16 = (p + 15) / 16, kd = (d + 3) / 4, kp = (p + 3) / 4;
  double* C = sm;            double* R = C + BIG;     double* Tm = R + BIG;    double* Gm = Tm + BIG;
  double* Fm = Gm + BIG;     double* RF = Fm + MID;   double* Kg = RF + MID;
  double* Q = Kg + MID;      double* Qi = Q + SML;    // Qi persists: warm start of the next step's inverse
  double* mv = Qi + SML;     double* av = mv + 48;    double* ev = av + 48;    double* fv = ev + 48;
  double* ob = fv + 48;      // observed flags (1.0 / 0.0)
  // scratch of the inverse, aliasing buffers that are idle between the forecast and the gain
  double* Qm = Tm;           double* Es = Tm + SML;   double* Tn = Kg;         double* Li = Es;
  bool warm = false;
  double* Lw = sm + FILT_DOUBLES;  double* Lv = Lw + BIG;   double* xv = Lv + SML;   double* zv = xv + 48;   // SIM only
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* y = a.y + (size_t)n * T * p;
  double* out = a.filt + (size_t)n * (T + 1) * rec;
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * frec : nullptr;
  double* pri = a.prior ? a.prior + (size_t)n * (T + 1) * rec : nullptr;   // optional (a_t, R_t) records
  int st = 0;

  zero_lds(tid, sm, SIM ? FILT_SIM_DOUBLES : FILT_DOUBLES);
  __syncthreads();
  load_cm(tid, a.C0 + (size_t)n * a.c0_stride, d, d, C, DL);
  load_cm(tid, a.F, d, p, Fm, PL);
  int gcur = a.g_index ? a.g_index[0] : 0;
  load_cm(tid, a.G + (size_t)gcur * dd, d, d, Gm, DL);
  if (tid < d) mv[tid] = (a.m0 + (size_t)n * a.m0_stride)[tid];
  __syncthreads();
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  const double* zin = (SIM && a.z) ? a.z + (size_t)n * (T + 1) * (d + p) : nullptr;
  double* xp = SIM ? xplus + (size_t)n * (T + 1) * d : nullptr;
  double* ys = SIM ? ystar + (size_t)n * T * p : nullptr;
  if (SIM) {
    // factors chol(W), chol(V), and x+_0 = m0 + chol(C0) z_0 (chol(C0) in R, which is idle here)
    load_cm(tid, W, d, d, Lw, DL);
    load_cm(tid, V, p, p, Lv, PL);
    for (int idx = tid; idx < 48 * DL; idx += NT) R[idx] = C[idx];
    if (tid < d) zv[tid] = zin ? zin[tid] : philox_normal(a.seed, series, 0u, (unsigned)tid);
    const bool b1 = chol_block(tid, d, Lw, DL), b2 = chol_block(tid, p, Lv, PL), b3 = chol_block(tid, d, R, DL);
    if (b1 || b2 || b3) st |= DLM_ST_NOT_PD;
    if (tid < d) { double sx = mv[tid]; for (int k = 0; k <= tid; ++k) sx = fma(R[tid * DL + k], zv[k], sx); xv[tid] = sx; }
    __syncthreads();
    if (tid < d) { xp[tid] = xv[tid]; mv[tid] = 0.0; }     // y* is filtered from a zero prior mean
    __syncthreads();
  }
  FOR_CM(d, d, i, j) out[d + i + j * d] = C[i * DL + j];
  if (tid < d) out[tid] = mv[tid];
  if (pri) { FOR_CM(d, d, i, j) pri[d + i + j * d] = C[i * DL + j]; if (tid < d) pri[tid] = mv[tid]; }
  if (fq) for (int i = tid; i < frec; i += NT) fq[i] = __builtin_nan("");

#ifdef DLM_STAMP
  int dbgc[3] = {0, 0, 0};
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
  for (int t = 0; t < T; ++t) {
    TSTAMP(7)
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) { __syncthreads(); load_cm(tid, a.G + (size_t)gi * dd, d, d, Gm, DL); gcur = gi; }
    if (a.f_stride) { __syncthreads(); load_cm(tid, a.F + (size_t)t * a.f_stride, d, p, Fm, PL); }
    __syncthreads();
    // advState: a = G m, R = G C G^T + W dt   (dt == 0: a = m, R = C)
    if (dt == 0.0) {
      for (int idx = tid; idx < 48 * DL; idx += NT) R[idx] = C[idx];
      if (tid < d) av[tid] = mv[tid];
    } else {
      gemm_t<false, false, 0>(tid, dt16, dt16, kd, Gm, DL, C, DL, Tm, DL);
      if (tid < d) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Gm[tid * DL + k], mv[k], s); av[tid] = s; }
      __syncthreads();
      gemm_t<false, true, 0>(tid, dt16, dt16, kd, Tm, DL, Gm, DL, R, DL);
      __syncthreads();
      FOR_CM(d, d, i, j) R[i * DL + j] = fma(W[i + j * d], dt, R[i * DL + j]);
    }
    __syncthreads();
    TSTAMP(0)
    if (pri) {
      double* pr = pri + (size_t)(t + 1) * rec;
      if (tid < d) pr[tid] = av[tid];
      FOR_CM(d, d, i, j) pr[d + i + j * d] = R[i * DL + j];
    }
    // forecast: f = F^T a, RF = R F, Q = F^T R F + V
    gemm_t<false, false, 0>(tid, dt16, pt16, kd, R, DL, Fm, PL, RF, PL);
    if (tid < p) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Fm[k * PL + tid], av[k], s); fv[tid] = s; }
    __syncthreads();
    gemm_t<true, false, 0>(tid, pt16, pt16, kd, Fm, PL, RF, PL, Q, PL);
    __syncthreads();
    FOR_CM(p, p, i, j) Q[i * PL + j] += V[i + j * p];
    if (SIM) {
      // x+_t = G x+_{t-1} + L_W z_x ;  y+_t = F^T x+_t + L_V z_y   (tv = new x+, then copied back)
      if (tid < d + p) zv[tid] = zin ? zin[(size_t)(t + 1) * (d + p) + tid] : philox_normal(a.seed, series, (unsigned)(t + 1), (unsigned)tid);
      __syncthreads();
      double xn = 0.0;
      if (tid < d) {
        if (dt == 0.0) xn = xv[tid];
        else {
          const double sdt = sqrt(dt);
          for (int k = 0; k < d; ++k) xn = fma(Gm[tid * DL + k], xv[k], xn);
          for (int k = 0; k <= tid; ++k) xn = fma(Lw[tid * DL + k] * sdt, zv[k], xn);
        }
      }
      __syncthreads();
      if (tid < d) { xv[tid] = xn; xp[(size_t)(t + 1) * d + tid] = xn; }
      __syncthreads();
    }
    if (tid < p) {
      double yv = y[(size_t)t * p + tid];
      if (SIM) {
        double yp = 0.0;
        for (int k = 0; k < d; ++k) yp = fma(Fm[k * PL + tid], xv[k], yp);
        for (int k = 0; k <= tid; ++k) yp = fma(Lv[tid * PL + k], zv[d + k], yp);
        yv = yv - yp;                                  // NaN (missing) stays NaN
        ys[(size_t)t * p + tid] = yv;
      }
      ob[tid] = (yv == yv) ? 1.0 : 0.0;
      ev[tid] = (yv == yv) ? yv - fv[tid] : 0.0;
    }
    __syncthreads();
    if (fq) {
      double* fr = fq + (size_t)(t + 1) * frec;
      if (tid < p) fr[tid] = fv[tid];
      FOR_CM(p, p, i, j) fr[p + i + j * p] = Q[i * PL + j];
    }
    bool any = false;
    for (int j = 0; j < p; ++j) any |= ob[j] != 0.0;
    TSTAMP(1)
    if (!any) {   // updateState :74-75
      __syncthreads();
      for (int idx = tid; idx < 48 * DL; idx += NT) C[idx] = R[idx];
      if (tid < d) mv[tid] = av[tid];
    } else {
      // Qm: missing rows/columns -> identity; inverse; back to zero
      __syncthreads();
      zero_lds(tid, Tm, 2 * SML);
      zero_lds(tid, Kg, SML);
      __syncthreads();
      FOR_CM(p, p, i, j) Qm[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? Q[i * PL + j] : (i == j ? 1.0 : 0.0);
      if (tid < p && ob[tid] == 0.0) Qi[tid * PL + tid] = 1.0;   // warm start: identity on the missing block
#ifdef DLM_STAMP
      if (spd_inverse_warm(tid, p, Qm, Qi, Es, Tn, Li, (int*)(ob + 40), warm, (n == 0) ? dbgc : nullptr)) st |= DLM_ST_NOT_PD;
#else
      if (spd_inverse_warm(tid, p, Qm, Qi, Es, Tn, Li, (int*)(ob + 40), warm)) st |= DLM_ST_NOT_PD;
#endif
      warm = true;
      FOR_CM(p, p, i, j) if (!(ob[i] != 0.0 && ob[j] != 0.0)) Qi[i * PL + j] = 0.0;
      __syncthreads();
      TSTAMP(2)
      gemm_t<false, false, 0>(tid, dt16, pt16, kp, RF, PL, Qi, PL, Kg, PL);          // K = R F Qm^-1
      __syncthreads();
      if (tid < d) { double s = av[tid]; for (int j = 0; j < p; ++j) s = fma(Kg[tid * PL + j], ev[j], s); mv[tid] = s; }
      gemm_t<false, true, 2>(tid, dt16, dt16, kp, Kg, PL, RF, PL, C, DL, R);          // C = R - K (R F)^T
    }
    __syncthreads();
    TSTAMP(3)
    double* o = out + (size_t)(t + 1) * rec;
    if (tid < d) o[tid] = mv[tid];
    FOR_CM(d, d, i, j) o[d + i + j * d] = C[i * DL + j];
    TSTAMP(4)
  }
#ifdef DLM_STAMP
  if (n == 0 && tid == 0 && a.status) { for (int k = 0; k < 8; ++k) a.status[1 + k] = (int)(seg[k] / (unsigned long long)T); a.status[6] = dbgc[0]; a.status[7] = dbgc[1]; a.status[9] = dbgc[2]; }
#endif
  __syncthreads();
  bool bad = false;
  FOR_CM(d, d, i, j) bad |= !isfinite(C[i * DL + j]);
  if (tid < d) bad |= !isfinite(mv[tid]);
  if (__syncthreads_or(bad)) st |= DLM_ST_NONFINITE;
  if (a.status && tid == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// backward pass (information form, general p)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_smoother_tiled(KArgs a) {
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

  // register prefetch of the record stream: C_t one step ahead, the means two steps ahead (the
  // innovation of step t needs m_{t-1})
  double pre[6], mcur = 0.0, mnext = 0.0;
  {
    const double* r = fin + (size_t)T * rec;
    int q = 0;
    for (int j = tid >> 6; j < d; j += NW, ++q) pre[q] = ((tid & 63) < d) ? r[d + (tid & 63) + j * d] : 0.0;
    if (tid < d) { mcur = r[tid]; mnext = (T > 0) ? (r - rec)[tid] : 0.0; }
  }
  for (int t = T; t >= 0; --t) {
    __syncthreads();
    {
      int q = 0;
      for (int j = tid >> 6; j < d; j += NW, ++q) if ((tid & 63) < d) C[(tid & 63) * DL + j] = pre[q];
      if (tid < d) { mv[tid] = mcur; mp[tid] = mnext; mcur = mnext; }
      if (t > 0) {
        const double* r = fin + (size_t)(t - 1) * rec;
        q = 0;
        for (int j = tid >> 6; j < d; j += NW, ++q) pre[q] = ((tid & 63) < d) ? r[d + (tid & 63) + j * d] : 0.0;
        if (tid < d) mnext = (t > 1) ? (r - rec)[tid] : 0.0;
      }
    }
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
        FOR_CM(p, p, i, j) Vi[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? V[i + j * p] : (i == j ? 1.0 : 0.0);
        if (spd_inverse(tid, p, Vi, Li, (int*)(cq + 40))) st |= DLM_ST_NOT_PD;
        FOR_CM(p, p, i, j) if (!(ob[i] != 0.0 && ob[j] != 0.0)) Vi[i * PL + j] = 0.0;
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
    FOR_CM(d, d, i, j) o[d + i + j * d] = C[i * DL + j] - T2[i * DL + j];                       // S_t
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
      FOR_CM(d, d, i, j) if (i < j) { const double v = 0.5 * (P[i * DL + j] + P[j * DL + i]); P[i * DL + j] = v; P[j * DL + i] = v; }
    } else if (tid < d) rv[tid] = qv[tid];
    __syncthreads();
    {   // like Smoothing.smoothStep (Smoothing.scala:41): always the table entry g(dt), also for dt == 0
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
  FOR_CM(d, d, i, j) bad |= !isfinite(T2[i * DL + j]) || !isfinite(C[i * DL + j]);
  if (__syncthreads_or(bad)) st |= DLM_ST_NONFINITE;
  if (a.status && tid == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// second half of the simulation smoother for general p: mean-only backward pass on y*, theta = s* + x+,
// Gibbs statistics (Gibbs.scala:23-78, GibbsWishart.scala:16-35) on the fly.  No covariance recursion:
//   q_{t-1} = G^T [ q + F (Qm^-1 e - K^T q) ],  K = C F Vm^-1,  Qm^-1 = Vm^-1 - Vm^-1 F^T K
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_simsmooth_tiled(KArgs a, const double* __restrict__ xplus,
                                                        const double* __restrict__ ystar) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd;
  const int dt16 = (d + 15) / 16, pt16 = (p + 15) / 16, kd = (d + 3) / 4, kp = (p + 3) / 4;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;
  double* C = sm;            double* OUT = C + BIG;
  double* Fm = OUT + BIG;    double* CF = Fm + MID;   double* Kg = CF + MID;
  double* Vi = Kg + MID;     double* Qi = Vi + SML;   double* X = Qi + SML;    double* Li = X + SML;
  double* mv = Li + SML;     double* mp = mv + 48;    double* qv = mp + 48;    double* rv = qv + 48;
  double* ev = rv + 48;      double* uv = ev + 48;    double* ob = uv + 48;    double* obp = ob + 48;
  double* tv = obp + 48;     double* thn = tv + 48;   double* thc = thn + 48;  double* dfv = thc + 48;
  double* ssy = dfv + 48;    double* nob = ssy + 48;  double* ssd = nob + 48;  double* flagv = ssd + 48;
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* y = a.y ? a.y + (size_t)n * T * p : nullptr;
  const double* ys = ystar + (size_t)n * T * p;
  const double* xp = xplus + (size_t)n * (T + 1) * d;
  const double* fin = a.filt_in + (size_t)n * (T + 1) * rec;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  int st = 0;

  zero_lds(tid, sm, SIMS_DOUBLES);
  __syncthreads();
  load_cm(tid, a.F, d, p, Fm, PL);
  if (tid < p) obp[tid] = -1.0;
  double pre[6], mcur = 0.0, mnext = 0.0;
  {
    const double* r = fin + (size_t)T * rec;
    int q = 0;
    for (int j = tid >> 6; j < d; j += NW, ++q) pre[q] = ((tid & 63) < d) ? r[d + (tid & 63) + j * d] : 0.0;
    if (tid < d) { mcur = r[tid]; mnext = (T > 0) ? (r - rec)[tid] : 0.0; }
  }
  for (int t = T; t >= 0; --t) {
    __syncthreads();
    {
      int q = 0;
      for (int j = tid >> 6; j < d; j += NW, ++q) if ((tid & 63) < d) C[(tid & 63) * DL + j] = pre[q];
      if (tid < d) { mv[tid] = mcur; mp[tid] = mnext; mcur = mnext; }
      if (t > 0) {
        const double* r = fin + (size_t)(t - 1) * rec;
        q = 0;
        for (int j = tid >> 6; j < d; j += NW, ++q) pre[q] = ((tid & 63) < d) ? r[d + (tid & 63) + j * d] : 0.0;
        if (tid < d) mnext = (t > 1) ? (r - rec)[tid] : 0.0;
      }
    }
    if (a.f_stride && t > 0) load_cm(tid, a.F + (size_t)(t - 1) * a.f_stride, d, p, Fm, PL);
    if (tid < p) { const double yv = (t > 0) ? ys[(size_t)(t - 1) * p + tid] : __builtin_nan(""); ob[tid] = (yv == yv) ? 1.0 : 0.0; tv[tid] = yv; }
    const double xcur = (tid < d) ? xp[(size_t)t * d + tid] : 0.0;
    __syncthreads();
    bool any = false, same = true;
    for (int j = 0; j < p; ++j) { any |= ob[j] != 0.0; same &= ob[j] == obp[j]; }
    const double* Gt = a.G + (size_t)((a.g_index && t > 0) ? a.g_index[t - 1] : 0) * dd;   // G of the step INTO record t
    const double dtt = (a.dt && t > 0) ? a.dt[t - 1] : 1.0;

    // theta_t = m*_t + C_t q_t + x+_t
    if (tid < d) {
      double s = mv[tid] + xcur;
      for (int k = 0; k < d; ++k) s = fma(C[tid * DL + k], qv[k], s);
      thc[tid] = s;
      if (thout) thout[(size_t)t * d + tid] = s;
    }
    __syncthreads();
    if (a.stats) {
      if (t < T) {   // system innovation theta_{t+1} - G_{t+1} theta_t
        const double* Gn = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
        const double dtn = a.dt ? a.dt[t] : 1.0;
        if (tid < d) {
          double s = thn[tid];
          if (dtn == 0.0) s -= thc[tid];
          else for (int k = 0; k < d; ++k) s = fma(-Gn[tid + k * d], thc[k], s);
          const double dts = (dtn == 0.0) ? 1.0 : dtn;
          dfv[tid] = s / sqrt(dts);
          ssd[tid] = fma(s, s / dts, ssd[tid]);
        }
        __syncthreads();
        if (outer) FOR_CM(d, d, i, j) OUT[i * DL + j] = fma(dfv[i], dfv[j], OUT[i * DL + j]);
      }
      if (t > 0 && y && tid < p) {   // observation residual of theta_t against the ORIGINAL y_t
        const double yv = y[(size_t)(t - 1) * p + tid];
        if (yv == yv) {
          double f = 0.0;
          for (int k = 0; k < d; ++k) f = fma(Fm[k * PL + tid], thc[k], f);
          ssy[tid] = fma(yv - f, yv - f, ssy[tid]); nob[tid] += 1.0;
        }
      }
    }
    __syncthreads();
    if (tid < d) thn[tid] = thc[tid];
    if (t == 0) break;

    if (any) {
      if (!same) {
        __syncthreads();
        FOR_CM(p, p, i, j) Vi[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? V[i + j * p] : (i == j ? 1.0 : 0.0);
        if (spd_inverse(tid, p, Vi, Li, (int*)flagv)) st |= DLM_ST_NOT_PD;
        FOR_CM(p, p, i, j) if (!(ob[i] != 0.0 && ob[j] != 0.0)) Vi[i * PL + j] = 0.0;
        if (tid < p) obp[tid] = ob[tid];
        __syncthreads();
      }
      gemm_t<false, false, 0>(tid, dt16, pt16, kd, C, DL, Fm, PL, CF, PL);            // C F
      if (tid < d) {
        double s = 0.0;
        if (dtt == 0.0) s = mp[tid];
        else for (int k = 0; k < d; ++k) s = fma(Gt[tid + k * d], mp[k], s);
        rv[tid] = s;   // a*_t
      }
      __syncthreads();
      gemm_t<false, false, 0>(tid, dt16, pt16, kp, CF, PL, Vi, PL, Kg, PL);           // K = C F Vm^-1
      if (tid < p) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Fm[k * PL + tid], rv[k], s); ev[tid] = (ob[tid] != 0.0) ? tv[tid] - s : 0.0; }
      __syncthreads();
      gemm_t<true, false, 0>(tid, pt16, pt16, kd, Fm, PL, Kg, PL, X, PL);             // F^T K
      __syncthreads();
      gemm_t<false, false, 2>(tid, pt16, pt16, kp, Vi, PL, X, PL, Qi, PL, Vi);        // Qm^-1
      __syncthreads();
      if (tid < p) {
        double s = 0.0;
        for (int j = 0; j < p; ++j) s = fma(Qi[tid * PL + j], ev[j], s);              // u = Qm^-1 e
        for (int k = 0; k < d; ++k) s = fma(-Kg[k * PL + tid], qv[k], s);             // - K^T q
        uv[tid] = s;
      }
      __syncthreads();
      if (tid < d) { double s = qv[tid]; for (int j = 0; j < p; ++j) s = fma(Fm[tid * PL + j], uv[j], s); rv[tid] = s; }
    } else if (tid < d) rv[tid] = qv[tid];
    __syncthreads();
    if (tid < d) {
      double s = 0.0;
      if (dtt == 0.0) s = rv[tid];
      else for (int k = 0; k < d; ++k) s = fma(Gt[k + tid * d], rv[k], s);             // q = G^T r
      qv[tid] = s;
    }
  }
  __syncthreads();
  bool bad = (tid < d) && !isfinite(thn[tid]);
  if (__syncthreads_or(bad)) st |= DLM_ST_NONFINITE;
  if (a.stats) {
    const int L = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * L;
    if (tid < p) { so[tid] = ssy[tid]; so[p + tid] = nob[tid]; }
    if (outer) FOR_CM(d, d, i, j) so[2 * p + i + j * d] = OUT[i * DL + j];
    else if (tid < d) so[2 * p + tid] = ssd[tid];
    if (tid == 0) so[L - 1] = (double)T;
  }
  if (a.status && tid == 0 && st) atomicOr(&a.status[n], st);
}

static hipError_t set_lds(const void* fn, size_t bytes) {
  return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

hipError_t launch_tiled_filter(const KArgs& a, hipStream_t s) {
  const size_t lds = tiled_filter_lds_bytes();
  hipError_t e = set_lds((const void*)k_filter_tiled<false>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_filter_tiled<false>, dim3(a.N), dim3(NT), lds, s, a, (double*)nullptr, (double*)nullptr);
  return hipGetLastError();
}

hipError_t launch_tiled_simsmooth(const KArgs& a, double* xplus, double* ystar, hipStream_t s) {
  size_t lds = sizeof(double) * FILT_SIM_DOUBLES + 16;
  hipError_t e = set_lds((const void*)k_filter_tiled<true>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_filter_tiled<true>, dim3(a.N), dim3(NT), lds, s, a, xplus, ystar);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  KArgs b = a;
  b.filt_in = a.filt;
  lds = sizeof(double) * SIMS_DOUBLES + 16;
  e = set_lds((const void*)k_simsmooth_tiled, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_simsmooth_tiled, dim3(a.N), dim3(NT), lds, s, b, (const double*)xplus, (const double*)ystar);
  return hipGetLastError();
}

hipError_t launch_tiled_smoother(const KArgs& a, hipStream_t s) {
  const size_t lds = tiled_smoother_lds_bytes();
  hipError_t e = set_lds((const void*)k_smoother_tiled, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_smoother_tiled, dim3(a.N), dim3(NT), lds, s, a);
  return hipGetLastError();
}

}  // namespace dlm
