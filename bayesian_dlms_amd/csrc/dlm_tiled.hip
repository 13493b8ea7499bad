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
#include <cstdlib>
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

#include <type_traits>

namespace dlm {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NT = 512;         // threads per workgroup: 8 waves, two per SIMD (latency hiding, 1 workgroup per CU)
constexpr int NW = NT / 64;
constexpr int VW = NW - 1;      // the wave that does a step's vector work (it owns at most one tile of any phase)
constexpr int DL = 49;          // leading dimension of d-wide matrices (up to 48 columns)
constexpr int PL = 33;          // leading dimension of p-wide matrices (up to 32 columns)
constexpr int BIG = 48 * DL;    // doubles in a 48 x d-wide matrix
constexpr int MID = 48 * PL;    // doubles in a 48 x p-wide matrix
constexpr int SML = 32 * PL;    // doubles in a 32 x p-wide matrix

// Workgroup barrier for LDS hand-offs only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence
// makes the compiler wait for ALL outstanding memory operations (vmcnt(0)): every barrier that follows a step's
// record stores or the next record's prefetch loads would then wait for HBM.  Nothing in these kernels exchanges data
// through global memory inside a launch, so only the LDS queue has to drain.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void wsync() {   // LDS hand-off inside ONE wavefront (in-order LDS queue)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- record I/O with a FIXED number of instructions per thread -----------------------------------------
// Raw buffer loads/stores whose padded lanes carry an out-of-range offset (loads give 0, stores are dropped), in
// loops with compile-time trip counts: the compiler then knows exactly how many vector-memory operations sit
// between a prefetch and its use and waits with a counted vmcnt.  With data-dependent store loops it falls back to
// vmcnt(0) at the first use of a prefetched value -- which, vector-memory operations retiring in order, also waits
// for the record stores of the step before (8k of 51k cycles per backward step at C4).
typedef unsigned u2v __attribute__((ext_vector_type(2)));
constexpr int OOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double bld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u2v v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __hiloint2double((int)v[1], (int)v[0]);
}
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) {
  const u2v v = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// Byte offsets of this thread's share of a record [vector (n) | matrix (n x n, column-major)]: row i = lane,
// columns j = wave + 8 q, q < 6; the vector element tid (threads 0..n-1).
struct RecOff { int m[6]; int v; };
__device__ __forceinline__ RecOff rec_offsets(int tid, int n) {
  RecOff o;
  const int i = tid & 63;
#pragma unroll
  for (int q = 0; q < 6; ++q) { const int j = (tid >> 6) + NW * q; o.m[q] = (i < n && j < n) ? (n + i + j * n) * 8 : OOB; }
  o.v = tid < n ? tid * 8 : OOB;
  return o;
}

// C (mt x nt tiles) = op(A) op(B) combined with D, all row-major in LDS.  kb = number of 4-deep k-blocks.
//   MODE 0: C = acc      1: C = D + acc      2: C = D - acc      (D in LDS, may alias C)
//   MODE 3: C = acc + dscale * Dg, Dg a column-major drows x dcols matrix in GLOBAL memory (W dt, V): the
//           addend is fetched before the MFMA chain, so its latency hides behind the chain and the separate
//           element-wise pass with its barrier disappears.
// SYM (mt == nt, symmetric result): only the tiles on and above the diagonal are computed, each written to
// both places.  The fp64 MFMA issues about once per 110-140 cycles per SIMD whatever the number of independent
// accumulators (profiles/r01_mfma_f64_microbench.txt), so a phase lasts as long as the MFMAs of its busiest
// SIMD: 6 tiles instead of 9 is 20 against 30 on it.  Waves w and w + 4 share a SIMD; tile q goes to wave q % 8.
// `shift` rotates the tile -> wave assignment (tile q on wave (q + shift) % 8) so that two products issued in
// the same phase load the SIMDs evenly.
// AVG (with SYM): a diagonal tile holds both triangles, rounded differently; average them so that the result is
// EXACTLY symmetric (the wave that computed the tile owns all of it: a wave-level hand-off through LDS suffices).
template <bool TA, bool TB, int MODE, bool SYM = false, bool AVG = false>
__device__ __forceinline__ void gemm_t(int tid, int mt, int nt, int kb, const double* A, int lda, const double* B,
                                       int ldb, double* C, int ldc, const double* D = nullptr, double dscale = 1.0,
                                       int drows = 0, int dcols = 0, int shift = 0) {
  const int wave = ((tid >> 6) - shift) & (NW - 1), lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int ntiles = SYM ? mt * (mt + 1) / 2 : mt * nt;   // <= 16: a wave owns tiles w and w + 8
  if (wave >= ntiles) return;
  int ao[2], bo[2], ti[2], tj[2];
  bool on[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave + NW * q;
    on[q] = tile < ntiles;
    int i = 0, j = 0;
    if (on[q]) {
      if (SYM) { int rem = tile, len = nt; while (rem >= len) { rem -= len; ++i; --len; } j = i + rem; }
      else { i = tile / nt; j = tile % nt; }
    }
    ti[q] = i; tj[q] = j;
    const int i0 = i * 16, j0 = j * 16;
    ao[q] = TA ? g * lda + i0 + c : (i0 + c) * lda + g;      // op(A)[i0 + c][g]     (+ 4 kk along k)
    bo[q] = TB ? (j0 + c) * ldb + g : g * ldb + j0 + c;      // op(B)[g][j0 + c]
  }
  const int as = TA ? 4 * lda : 4, bs = TB ? 4 : 4 * ldb;    // stride of one k-block
  d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  d4 dg[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};   // MODE 3: the global addend, requested now, added after the chain
  if (MODE == 3) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (!on[q]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ti[q] * 16 + 4 * r + g, j = tj[q] * 16 + c;
        dg[q][r] = (i < drows && j < dcols) ? D[i + (size_t)j * drows] : 0.0;
      }
    }
  }
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
    const int i0 = ti[q] * 16, j0 = tj[q] * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = (i0 + 4 * r + g) * ldc + j0 + c;
      double v;
      if (MODE == 0) v = acc[q][r];
      else if (MODE == 3) v = fma(dscale, dg[q][r], acc[q][r]);
      else if (MODE == 1) v = D[o] + acc[q][r];
      else v = D[o] - acc[q][r];
      C[o] = v;
      if (SYM && i0 != j0) C[(j0 + c) * ldc + i0 + 4 * r + g] = v;
      if (AVG) acc[q][r] = v;
    }
    if (SYM && AVG && i0 == j0) {
      wsync();
      d4 w;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = C[(i0 + c) * ldc + i0 + 4 * r + g];
      wsync();
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(i0 + 4 * r + g) * ldc + i0 + c] = 0.5 * (acc[q][r] + w[r]);
    }
  }
}

// y[i] = sum_k M[i * ld + k] x[k] (TR: M[k * ld + i]) for i < rows, by the lanes of ONE wave (which = wave index):
// the O(d^2) vector work of a step runs on a wave that has a light share of the phase's tiles, four
// independent partial sums keep the LDS reads in flight.
template <bool TR>
__device__ __forceinline__ double wave_matvec(int tid, int which, int rows, int cols, const double* M, int ld, const double* x) {
  const int i = tid - which * 64;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i >= 0 && i < rows) {
    int k = 0;
    for (; k + 3 < cols; k += 4) {
      s0 = fma(TR ? M[k * ld + i] : M[i * ld + k], x[k], s0);
      s1 = fma(TR ? M[(k + 1) * ld + i] : M[i * ld + k + 1], x[k + 1], s1);
      s2 = fma(TR ? M[(k + 2) * ld + i] : M[i * ld + k + 2], x[k + 2], s2);
      s3 = fma(TR ? M[(k + 3) * ld + i] : M[i * ld + k + 3], x[k + 3], s3);
    }
    for (; k < cols; ++k) s0 = fma(TR ? M[k * ld + i] : M[i * ld + k], x[k], s0);
  }
  return (s0 + s1) + (s2 + s3);
}


// Products with the d x d transition G of the backward pass.  G is kept in LDS as a compact row-major copy (leading
// dimension gd = d, no padding: both access patterns below run along a row, so any leading dimension is conflict
// free; there is room for d <= 46) or, for d = 47, 48, read straight from global memory (column-major; the same 18 KB
// for every workgroup, so it lives in L1/L2).  Either way all k-blocks of the G operand are fetched before the
// MFMA chain (one latency, not kb) and the padding is made of predicated zeros.
//   GA = false:  C = A * G      (A row-major in LDS)             B[k][j] = G[k][j]
//   GA = true :  C = G^T * B    (B row-major in LDS, SYM result)  A[i][k] = G[k][i]
template <bool GA, bool SYM, bool GLDS>
__device__ __forceinline__ void gemm_g(int tid, int mt, int kb, const double* L, int ldl, const double* Gp, int gd,
                                       double* C, int ldc) {
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int ntiles = SYM ? mt * (mt + 1) / 2 : mt * mt;
  if (wave >= ntiles) return;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int tile = wave + NW * q;
    if (tile >= ntiles) break;
    int i = 0, j = 0;
    if (SYM) { int rem = tile, len = mt; while (rem >= len) { rem -= len; ++i; --len; } j = i + rem; }
    else { i = tile / mt; j = tile % mt; }
    const int i0 = i * 16, j0 = j * 16;
    const int o = (GA ? i0 : j0) + c;                         // the G index that is not summed over
    double gv[12];
#pragma unroll
    for (int kk = 0; kk < 12; ++kk) {
      const int k = g + 4 * kk;
      const bool ok = kk < kb && k < gd && o < gd;
      gv[kk] = ok ? (GLDS ? Gp[k * gd + o] : Gp[k + (size_t)o * gd]) : 0.0;
    }
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    const int lo = GA ? g * ldl + j0 + c : (i0 + c) * ldl + g;   // B[g][j0 + c]  /  A[i0 + c][g]
    const int ls = GA ? 4 * ldl : 4;
#pragma unroll
    for (int kk = 0; kk < 12; ++kk) {
      if (kk < kb) {
        const double lv = L[lo + kk * ls];
        acc = GA ? __builtin_amdgcn_mfma_f64_16x16x4f64(gv[kk], lv, acc, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f64_16x16x4f64(lv, gv[kk], acc, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      C[(i0 + 4 * r + g) * ldc + j0 + c] = acc[r];
      if (SYM && i0 != j0) C[(j0 + c) * ldc + i0 + 4 * r + g] = acc[r];
    }
  }
}

__device__ __forceinline__ double bcast_lane(double v, int src) {   // lane `src` -> SGPR pair (uniform)
  const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Z = T X T^T (+ add) for a structured T (at most K <= 4 nonzeros per row), all d x d row-major in LDS with leading
// dimension DL; Y is scratch, Z may alias X.  Thread (lane i, wave w) owns the elements (i, w + 8 q): pass 1
// Y[i][c] = sum_s X[i][idx_c[s]] val_c[s], barrier, pass 2 Z[r][i] = sum_s val_r[s] Y[idx_r[s]][i] with r = w + 8 q --
// both passes read along conflict-free rows/columns and use the SAME six rows of the table of T per wave.  Those 24
// entries live spread over the lanes of the wave (lane 4 q + s holds entry s of row w + 8 q: tix, tvl, loaded by
// load_wave_table when the table changes) and are broadcast with v_readlane: no memory traffic, no latency.
// 12 K multiply-adds per thread replace two dense MFMA products.
// addw (nullable): this thread's six addends (W dt of the forward pass), for element (r, i).
// A vector rides along as column d of the scratch: Y[.][d] = x on entry of pass 2 gives yv = T x from lane i == d.
__device__ __forceinline__ void load_wave_table(int tid, int d, const SparseBig* __restrict__ tab, int& tix, double& tvl) {
  const int l = tid & 63, r = (tid >> 6) + NW * (l >> 2);
  const bool ok = l < 24 && r < d;
  tix = ok ? tab->idx[r][l & 3] : 0;
  tvl = ok ? tab->val[r][l & 3] : 0.0;
}
// Branch-free: every lane reads in-bounds addresses (rows up to 63 of a 48-row matrix run into the next LDS
// buffer, harmlessly) and lanes without an element store to `trash`, so that the 6 K reads of a pass are all in
// flight together instead of one exec-masked block per element.
template <int K>
__device__ __forceinline__ void sparse_congruence_k(int tid, int d, int tix, double tvl, const double* X, double* Y,
                                                    double* Z, const double* addw, double ascale, const double* x, double* yv,
                                                    double* trash) {
  const int i = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (w == VW && i < d) Y[i * DL + d] = x[i];
  int ix[6][K];
  double vl[6][K];
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int s_ = 0; s_ < K; ++s_) { ix[q][s_] = __builtin_amdgcn_readlane(tix, 4 * q + s_); vl[q][s_] = bcast_lane(tvl, 4 * q + s_); }
  double in[6][K];
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int s_ = 0; s_ < K; ++s_) in[q][s_] = X[i * DL + ix[q][s_]];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int c = w + NW * q;
    double acc = in[q][0] * vl[q][0];
#pragma unroll
    for (int s_ = 1; s_ < K; ++s_) acc = fma(in[q][s_], vl[q][s_], acc);
    double* dst = (c < d && i < d) ? Y + i * DL + c : trash;
    *dst = acc;
  }
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int s_ = 0; s_ < K; ++s_) in[q][s_] = Y[ix[q][s_] * DL + i];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int r = w + NW * q;
    const double y0 = in[q][0] * vl[q][0];
    double acc = (addw && i < d) ? fma(addw[q], ascale, y0) : y0;
#pragma unroll
    for (int s_ = 1; s_ < K; ++s_) acc = fma(in[q][s_], vl[q][s_], acc);
    double* dst = (r < d && i < d) ? Z + r * DL + i : ((r < d && i == d) ? yv + r : trash);
    *dst = acc;
  }
}
__device__ __forceinline__ void sparse_congruence(int tid, int d, int K, int tix, double tvl, const double* X, double* Y,
                                                  double* Z, const double* addw, double ascale, const double* x, double* yv,
                                                  double* trash) {
  switch (K) {   // wave-uniform
    case 1: sparse_congruence_k<1>(tid, d, tix, tvl, X, Y, Z, addw, ascale, x, yv, trash); break;
    case 2: sparse_congruence_k<2>(tid, d, tix, tvl, X, Y, Z, addw, ascale, x, yv, trash); break;
    case 3: sparse_congruence_k<3>(tid, d, tix, tvl, X, Y, Z, addw, ascale, x, yv, trash); break;
    default: sparse_congruence_k<4>(tid, d, tix, tvl, X, Y, Z, addw, ascale, x, yv, trash); break;
  }
}

// M = D + A1 B1^T - A2 B2^T - A3 B3^T for a symmetric result (upper tiles, mirrored): the rank-3p update of the
// backward recursion in ONE phase.  All operands row-major d x p in LDS with leading dimension ld.
__device__ __forceinline__ void gemm_update3(int tid, int mt, int kb, const double* A1, const double* B1, const double* A2,
                                             const double* B2, const double* A3, const double* B3, int ld, double* C,
                                             int ldc, const double* D) {
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  const int ntiles = mt * (mt + 1) / 2;
  if (wave >= ntiles) return;
  int i = 0, rem = wave, len = mt;
  while (rem >= len) { rem -= len; ++i; --len; }
  const int i0 = i * 16, j0 = (i + rem) * 16;
  const int ao = (i0 + c) * ld + g, bo = (j0 + c) * ld + g;
  d4 pos = {0.0, 0.0, 0.0, 0.0}, neg = {0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < kb; ++kk) {
    const double a1 = A1[ao + 4 * kk], b1 = B1[bo + 4 * kk], a2 = A2[ao + 4 * kk], b2 = B2[bo + 4 * kk];
    const double a3 = A3[ao + 4 * kk], b3 = B3[bo + 4 * kk];
    pos = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, pos, 0, 0, 0);
    neg = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, neg, 0, 0, 0);
    neg = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, neg, 0, 0, 0);
  }
  d4 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = (i0 + 4 * r + g) * ldc + j0 + c;
    v[r] = (D[o] + pos[r]) - neg[r];
    C[o] = v[r];
    if (i0 != j0) C[(j0 + c) * ldc + i0 + 4 * r + g] = v[r];
  }
  if (i0 == j0) {   // a diagonal tile holds both triangles, rounded differently: average them (same wave wrote all of it)
    wsync();
    d4 w;
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = C[(i0 + c) * ldc + i0 + 4 * r + g];
    wsync();
#pragma unroll
    for (int r = 0; r < 4; ++r) C[(i0 + 4 * r + g) * ldc + i0 + c] = 0.5 * (v[r] + w[r]);
  }
}

// y[i] = sum_k G[i][k] x[k] (TR: G[k][i]) with G column-major in GLOBAL memory, by the lanes of wave `which`
template <bool TR>
__device__ __forceinline__ double wave_matvec_g(int tid, int which, int n, const double* Gg, const double* x) {
  const int i = tid - which * 64;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (i >= 0 && i < n) {
    int k = 0;
    for (; k + 3 < n; k += 4) {
      s0 = fma(TR ? Gg[k + (size_t)i * n] : Gg[i + (size_t)k * n], x[k], s0);
      s1 = fma(TR ? Gg[k + 1 + (size_t)i * n] : Gg[i + (size_t)(k + 1) * n], x[k + 1], s1);
      s2 = fma(TR ? Gg[k + 2 + (size_t)i * n] : Gg[i + (size_t)(k + 2) * n], x[k + 2], s2);
      s3 = fma(TR ? Gg[k + 3 + (size_t)i * n] : Gg[i + (size_t)(k + 3) * n], x[k + 3], s3);
    }
    for (; k < n; ++k) s0 = fma(TR ? Gg[k + (size_t)i * n] : Gg[i + (size_t)k * n], x[k], s0);
  }
  return (s0 + s1) + (s2 + s3);
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

__device__ __noinline__ bool spd_inverse(int tid, int n_, double* A, double* Li, int* bad_flag) {
  const int n = __builtin_amdgcn_readfirstlane(n_);   // uniform: the guards below must be scalar branches
  lds_barrier();
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
  lds_barrier();
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
    lds_barrier();
    double akk = A[k * ld + k];
    if (!(akk > 0.0)) { bad = true; akk = 1e-300; }
    const double lkk = sqrt(akk), inv = 1.0 / lkk;
    lds_barrier();
    if (tid >= k && tid < n) A[tid * ld + k] = (tid == k) ? lkk : A[tid * ld + k] * inv;
    lds_barrier();
    for (int j = (tid >> 6) + k + 1; j < n; j += NW)
      for (int i = (tid & 63); i < n; i += 64)
        if (i >= j) A[i * ld + j] = fma(-A[i * ld + k], A[j * ld + k], A[i * ld + j]);
  }
  lds_barrier();
  for (int j = (tid >> 6); j < n; j += NW) for (int i = (tid & 63); i < n; i += 64) if (i < j) A[i * ld + j] = 0.0;
  lds_barrier();
  return bad;
}

// Inverse of the SPD n x n matrix Q (LDS, ld PL) by Newton-Schulz refinement of a warm start:
//   E = I - Q X ,  X <- X + X E        (||E|| squares every iteration)
// X holds the inverse of the previous time step's Q on entry (Q_t changes slowly: one or two iterations
// at steady state) and the verified inverse on exit.  If the
// warm start is too far off (first step, missingness pattern changed: n max|E| >= 0.5) or it has not
// converged (max|E| <= 2e-10) in 6 iterations, the direct register Cholesky takes over.  E and Tn are p x p scratch.
// Every thread must call.  Returns whether the direct path met a non-positive pivot.
// E = I - Q X (n x n, up to 2 x 2 tiles on waves 0..3) with the residual tests folded into the epilogue:
// big = some |E_ij| > tol, far = some n |E_ij| >= 0.5 (entries i, j < n only).
__device__ __forceinline__ void gemm_resid(int tid, int nt, int kb, int n, const double* Q, const double* X, double* E,
                                           double tol, bool& big, bool& far) {
  const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, c = lane & 15;
  if (wave >= nt * nt) return;
  const int i0 = (wave / nt) * 16, j0 = (wave % nt) * 16;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int kk = 0; kk < kb; ++kk)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Q[(i0 + c) * PL + g + 4 * kk], X[(g + 4 * kk) * PL + j0 + c], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 4 * r + g, j = j0 + c;
    const double e = ((i == j) ? 1.0 : 0.0) - acc[r];
    E[i * PL + j] = e;
    if (i < n && j < n) { big |= !(fabs(e) <= tol); far |= !(fabs(e) * n < 0.5); }
  }
}
// Workgroup-wide OR of two predicates with ONE barrier: wave ballots, one LDS atomic per wave that has something
// to report.  `slot` must be zero on entry (nobody touches it between the previous barrier and this call); `other`
// (the slot of the next call) is cleared for that call.  Returns bit 0 = any a, bit 1 = any b.
__device__ __forceinline__ int block_or2(int tid, bool a, bool b, int* slot, int* other) {
  const unsigned long long ba = __ballot(a), bb = __ballot(b);
  if ((tid & 63) == 0) { const int v = (ba ? 1 : 0) | (bb ? 2 : 0); if (v) atomicOr(slot, v); }
  lds_barrier();
  const int r = *slot;
  if (tid == 0) *other = 0;
  return r;
}

// In-place inverse of an SPD matrix by Newton-Schulz refinement of a warm start, on the MFMA pipe:
//   E = I - Q X ,  X <- X + X E        (||E|| squares every iteration)
// X holds the inverse of the previous time step's Q on entry (Q_t changes slowly: one or two iterations at steady
// state).  The iterates ping-pong between X and Xalt (any idle n x n buffer); *Xout tells where the verified inverse
// ended up.  An iteration is two small products and two barriers: the residual test rides in the epilogue of the
// first (gemm_resid + block_or2) and X + X E = 2X - X Q X is symmetric, so the second writes its upper tiles to both
// places and averages the two triangles of its diagonal tiles: the antisymmetric part of X is a neutral mode of the
// iteration and, left alone, drifts from step to step until the residual test can no longer be met.  If the warm start is too far off (first step, missingness pattern
// changed: n max|E| >= 0.5) or it has not converged (max|E| <= 2e-10) in 6 iterations, the direct register Cholesky
// takes over (into X).  Padding: Q and X must be zero outside n x n up to 16 nt (X stays so).  Every thread must call.
// Returns whether the direct path met a non-positive pivot.
__device__ __forceinline__ bool spd_inverse_warm(int tid, int n, const double* Q, double* X, double* Xalt, double* E,
                                                 double* Li, int* flag, int* orflags, bool have_warm, double** Xout,
                                                 int* dbg = nullptr) {
  const int nt = (n + 15) / 16, kb = (n + 3) / 4;
  const double tol = 2e-10;   // the fp64 floor of max|I - Q X| is ~n cond(Q) eps (1e-11 here, no better for the direct inverse)
  bool done = false;
  double* cur = X;
  double* alt = Xalt;
  if (have_warm) {
    if (tid == 0) { orflags[0] = 0; orflags[1] = 0; }
    for (int it = 0; it < 6 && !done; ++it) {
      lds_barrier();
      bool big = false, far = false;
      gemm_resid(tid, nt, kb, n, Q, cur, E, tol, big, far);
      const int bits = block_or2(tid, big, far, orflags + (it & 1), orflags + ((it + 1) & 1));
      const bool any_big = (bits & 1) != 0, any_far = (bits & 2) != 0;
      if (dbg && tid == 0) { dbg[0] += 1; if (any_far) dbg[2] += 1; }
      if (any_far && any_big) break;                                             // not contractive enough: go direct
      if (!any_big) done = true;   // ||E|| <= 2e-10: the update below squares it, i.e. lands on the fp64 floor
      gemm_t<false, false, 1, true, true>(tid, nt, nt, kb, cur, PL, E, PL, alt, PL, cur);   // alt = cur + cur E, exactly symmetric
      double* sw = cur; cur = alt; alt = sw;
    }
    lds_barrier();
  }
  if (done) { *Xout = cur; return false; }
  if (dbg && tid == 0) dbg[1] += 1;
  FOR_CM(n, n, i, j) X[i * PL + j] = Q[i * PL + j];
  *Xout = X;
  return spd_inverse(tid, n, X, Li, flag);
}

#ifdef DLM_STAMP
#define TSTAMP(k) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory"); seg[k] += _t - tlast; tlast = _t; }
#else
#define TSTAMP(k)
#endif

int sparse48_analyse(const double* G /* d x d column-major, host */, int d, SparseBig* rows, SparseBig* cols) {
  int kmax = 1;
  for (int pass = 0; pass < 2; ++pass) {
    SparseBig* t = pass ? cols : rows;
    for (int j = 0; j < 48; ++j)
      for (int s = 0; s < 4; ++s) { t->idx[j][s] = 0; t->val[j][s] = 0.0; }
    for (int j = 0; j < d; ++j) {
      int cnt = 0;
      for (int l = 0; l < d; ++l) {
        const double v = pass ? G[l + j * d] /* G[l][j] */ : G[j + l * d] /* G[j][l] */;
        if (v != 0.0) {
          if (cnt == 4) return 99;
          t->idx[j][cnt] = l; t->val[j][cnt] = v; ++cnt;
        }
      }
      if (cnt > kmax) kmax = cnt;
    }
  }
  rows->K = cols->K = kmax;
  rows->pad = cols->pad = 0;
  return kmax;
}

int sparsef_analyse(const double* F /* d x p column-major, host */, int d, int p, SparseF* out) {
  int kmax = 1;
  for (int j = 0; j < 32; ++j) for (int s = 0; s < 4; ++s) { out->cidx[j][s] = 0; out->cval[j][s] = 0.0; }
  for (int i = 0; i < 48; ++i) for (int s = 0; s < 4; ++s) { out->ridx[i][s] = 0; out->rval[i][s] = 0.0; }
  for (int j = 0; j < p; ++j) {
    int cnt = 0;
    for (int i = 0; i < d; ++i) {
      const double v = F[i + j * d];
      if (v != 0.0) { if (cnt == 4) return 99; out->cidx[j][cnt] = i; out->cval[j][cnt] = v; ++cnt; }
    }
    if (cnt > kmax) kmax = cnt;
  }
  for (int i = 0; i < d; ++i) {
    int cnt = 0;
    for (int j = 0; j < p; ++j) {
      const double v = F[i + j * d];
      if (v != 0.0) { if (cnt == 4) return 99; out->ridx[i][cnt] = j; out->rval[i][cnt] = v; ++cnt; }
    }
    if (cnt > kmax) kmax = cnt;
  }
  out->K = kmax; out->pad = 0;
  return kmax;
}

bool tiled_supported(const KArgs& a) {   // records are addressed through 32-bit buffer offsets: < 2 GiB per series
  return a.d >= 16 && a.d <= 48 && a.p <= 32 && ((size_t)a.T + 1) * (size_t)(a.d + a.d * a.d) * 8 < ((size_t)1 << 31);
}

constexpr int FILT_DOUBLES = 4 * BIG + 3 * MID + 2 * SML + 8 * 48;    // 133 KB (inverse scratch aliases Tm / Kg)
constexpr int FILT_SIM_DOUBLES = FILT_DOUBLES + BIG + SML + 3 * 48;   // + chol(W), chol(V), x+ (48), normals (96): 158 KB
constexpr int SIMS_DOUBLES = 2 * BIG + 3 * MID + 4 * SML + 16 * 48;   // mean-only backward pass: 115 KB
constexpr int GL_MAXD = 46;                                             // largest d whose compact G copy fits
constexpr int SMTH_DOUBLES = 4 * BIG + 3 * MID + 3 * SML + 10 * 48 + GL_MAXD * GL_MAXD;   // 159.4 KB (the Cholesky scratch aliases T1)
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
  double* Qm = Tm;           double* Es = Tm + SML;   double* Li = Es;
  bool warm = false;
  double* Lw = sm + FILT_DOUBLES;  double* Lv = Lw + BIG;   double* xv = Lv + SML;   double* zv = xv + 48;   // SIM only
  const double* V = a.V + (size_t)n * a.v_stride;   // V_0 / W_0; advanced every step when time-varying
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* const V0 = V; const double* const W0 = W;
  const double* y = a.y + (size_t)n * T * p;
  double* out = a.filt ? a.filt + (size_t)n * (T + 1) * rec : nullptr;   // null: likelihood only, nothing stored
  double ll = 0.0;   // prediction-error log-likelihood, accumulated by lane 0 of wave VW (KalmanFilter.scala:138-153)
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * frec : nullptr;
  double* pri = a.prior ? a.prior + (size_t)n * (T + 1) * rec : nullptr;   // optional (a_t, R_t) records
  int st = 0;

  zero_lds(tid, sm, SIM ? FILT_SIM_DOUBLES : FILT_DOUBLES);
  lds_barrier();
  load_cm(tid, a.C0 + (size_t)n * a.c0_stride, d, d, C, DL);
  load_cm(tid, a.F, d, p, Fm, PL);
  int gcur = a.g_index ? a.g_index[0] : 0;
  load_cm(tid, a.G + (size_t)gcur * dd, d, d, Gm, DL);
  if (tid < d) mv[tid] = (a.m0 + (size_t)n * a.m0_stride)[tid];
  lds_barrier();
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  const double* zin = (SIM && a.z) ? a.z + (size_t)n * (T + 1) * (d + p) : nullptr;
  double* xp = SIM ? xplus + (size_t)n * (T + 1) * d : nullptr;
  double* ys = ystar ? ystar + (size_t)n * T * p : nullptr;   // SIM: y* = y - y+ ; otherwise (optional) the innovations
  if (SIM) {
    // factors chol(W), chol(V), and x+_0 = m0 + chol(C0) z_0 (chol(C0) in R, which is idle here)
    load_cm(tid, W, d, d, Lw, DL);
    load_cm(tid, V, p, p, Lv, PL);
    for (int idx = tid; idx < 48 * DL; idx += NT) R[idx] = C[idx];
    if (tid < d) zv[tid] = zin ? zin[tid] : philox_normal(a.seed, series, 0u, (unsigned)tid);
    const bool b1 = chol_block(tid, d, Lw, DL), b2 = chol_block(tid, p, Lv, PL), b3 = chol_block(tid, d, R, DL);
    if (b1 || b2 || b3) st |= DLM_ST_NOT_PD;
    if (tid < d) { double sx = mv[tid]; for (int k = 0; k <= tid; ++k) sx = fma(R[tid * DL + k], zv[k], sx); xv[tid] = sx; }
    lds_barrier();
    if (tid < d) { xp[tid] = xv[tid]; mv[tid] = 0.0; }     // y* is filtered from a zero prior mean
    lds_barrier();
  }
  if (out) { FOR_CM(d, d, i, j) out[d + i + j * d] = C[i * DL + j]; if (tid < d) out[tid] = mv[tid]; }
  if (pri) { FOR_CM(d, d, i, j) pri[d + i + j * d] = C[i * DL + j]; if (tid < d) pri[tid] = mv[tid]; }
  if (fq) for (int i = tid; i < frec; i += NT) fq[i] = __builtin_nan("");

  // per-step record I/O with a fixed instruction count per thread (see rec_offsets): y is prefetched one step ahead
  const int recb = rec * 8;
  const __amdgpu_buffer_rsrc_t rfo = mk_rsrc(out, out ? (size_t)(T + 1) * recb : 0);   // zero-sized: every store is dropped
  const __amdgpu_buffer_rsrc_t rpr = mk_rsrc(pri, pri ? (size_t)(T + 1) * recb : 0);
  const __amdgpu_buffer_rsrc_t ry = mk_rsrc(y, (size_t)T * p * 8);
  const __amdgpu_buffer_rsrc_t rys = mk_rsrc(ys, ys ? (size_t)T * p * 8 : 0);
  const RecOff ro = rec_offsets(tid, d);
  const int poff = tid < p ? tid * 8 : OOB;
  double ynext = bld(ry, poff, 0);
  int tabcur = -1, tix = 0;   // the wave's rows of the structured-G table, spread over its lanes (load_wave_table)
  double tvl = 0.0;
  double wreg[6];   // this thread's elements (r = wave + 8 q, i = lane) of W for the structured congruence
#pragma unroll
  for (int q = 0; q < 6; ++q) { const int r = (tid >> 6) + NW * q, i = tid & 63; wreg[q] = (a.spb && r < d && i < d) ? W[r + i * d] : 0.0; }
#ifdef DLM_STAMP
  int dbgc[3] = {0, 0, 0};
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
  for (int t = 0; t < T; ++t) {
    TSTAMP(7)
    int tl = tid;   // opaque copy: no hoisting of the products' operand addresses out of the time loop (see k_smoother_tiled)
    asm volatile("" : "+v"(tl));
    if (a.v_tstride) V = V0 + (size_t)t * a.v_tstride;   // time-varying variances (StudentTGibbs.scala:100-136, DlmFsvSystem.scala:137-208)
    if (a.w_tstride) {
      W = W0 + (size_t)t * a.w_tstride;
#pragma unroll
      for (int q = 0; q < 6; ++q) { const int r = (tid >> 6) + NW * q, i = tid & 63; wreg[q] = (a.spb && r < d && i < d) ? W[r + i * d] : 0.0; }
    }
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) { lds_barrier(); load_cm(tid, a.G + (size_t)gi * dd, d, d, Gm, DL); gcur = gi; }
    if (a.f_stride) { lds_barrier(); load_cm(tid, a.F + (size_t)t * a.f_stride, d, p, Fm, PL); }
    lds_barrier();
    // advState: a = G m, R = G C G^T + W dt   (dt == 0: a = m, R = C)
    if (dt == 0.0) {
      for (int idx = tid; idx < 48 * DL; idx += NT) R[idx] = C[idx];
      if (tid < d) av[tid] = mv[tid];
    } else {
      if (a.spb) {   // structured G: two gather passes (rows of G) instead of two dense products; also a = G m
        const SparseBig* tab = a.spb + 2 * gi;
        if (gi != tabcur) { load_wave_table(tid, d, tab, tix, tvl); tabcur = gi; }
        sparse_congruence(tl, d, tab->K, tix, tvl, C, Tm, R, wreg, dt, mv, av, ob + 46);
      } else {
        gemm_t<false, false, 0>(tl, dt16, dt16, kd, Gm, DL, C, DL, Tm, DL);
        { const double s = wave_matvec<false>(tl, VW, d, d, Gm, DL, mv); if (tid >= VW * 64 && tid - VW * 64 < d) av[tid - VW * 64] = s; }
        lds_barrier();
        gemm_t<false, true, 3, true>(tl, dt16, dt16, kd, Tm, DL, Gm, DL, R, DL, W, dt, d, d);   // (G C) G^T + W dt
      }
    }
    lds_barrier();
    TSTAMP(0)
    if (pri) {
      const int i = tid & 63, so = (t + 1) * recb;
#pragma unroll
      for (int q = 0; q < 6; ++q) bst(rpr, ro.m[q], so, R[i * DL + (tid >> 6) + NW * q]);
      bst(rpr, ro.v, so, av[i]);
    }
    // forecast: f = F^T a, RF = R F, Q = F^T R F + V
    gemm_t<false, false, 0>(tl, dt16, pt16, kd, R, DL, Fm, PL, RF, PL);
    { const double s = wave_matvec<true>(tl, VW, p, d, Fm, PL, av); if (tid >= VW * 64 && tid - VW * 64 < p) fv[tid - VW * 64] = s; }
    lds_barrier();
    gemm_t<true, false, 3, true>(tl, pt16, pt16, kd, Fm, PL, RF, PL, Q, PL, V, 1.0, p, p);   // F^T (R F) + V
    lds_barrier();
    if (SIM) {
      // x+_t = G x+_{t-1} + L_W z_x ;  y+_t = F^T x+_t + L_V z_y   (tv = new x+, then copied back)
      if (tid < d + p) zv[tid] = zin ? zin[(size_t)(t + 1) * (d + p) + tid] : philox_normal(a.seed, series, (unsigned)(t + 1), (unsigned)tid);
      lds_barrier();
      double xn = 0.0;
      if (tid < d) {
        if (dt == 0.0) xn = xv[tid];
        else {
          const double sdt = sqrt(dt);
          for (int k = 0; k < d; ++k) xn = fma(Gm[tid * DL + k], xv[k], xn);
          for (int k = 0; k <= tid; ++k) xn = fma(Lw[tid * DL + k] * sdt, zv[k], xn);
        }
      }
      lds_barrier();
      if (tid < d) { xv[tid] = xn; xp[(size_t)(t + 1) * d + tid] = xn; }
      lds_barrier();
    }
    const double ycur = ynext;
    ynext = bld(ry, t + 1 < T ? poff : OOB, (t + 1 < T ? t + 1 : 0) * p * 8);
    if (tid < p) {
      double yv = ycur;
      if (SIM) {
        double yp = 0.0;
        for (int k = 0; k < d; ++k) yp = fma(Fm[k * PL + tid], xv[k], yp);
        for (int k = 0; k <= tid; ++k) yp = fma(Lv[tid * PL + k], zv[d + k], yp);
        yv = yv - yp;                                  // NaN (missing) stays NaN
        ys[(size_t)t * p + tid] = yv;
      }
      ob[tid] = (yv == yv) ? 1.0 : 0.0;
      ev[tid] = (yv == yv) ? yv - fv[tid] : 0.0;
      const unsigned long long mo = __ballot(yv == yv), mm = __ballot(!(yv == yv));   // the p lanes are all in wave 0
      if (tid == 0) { ob[44] = mo ? 1.0 : 0.0; ob[45] = mm ? 0.0 : 1.0; }
    }
    if (!SIM) bst(rys, poff, t * p * 8, ycur - fv[tid & 31]);   // innovations for the fused backward pass (NaN = missing; dropped when not asked for)
    lds_barrier();
    if (fq) {
      double* fr = fq + (size_t)(t + 1) * frec;
      if (tid < p) fr[tid] = fv[tid];
      FOR_CM(p, p, i, j) fr[p + i + j * p] = Q[i * PL + j];
    }
    const bool any = ob[44] != 0.0, allobs = ob[45] != 0.0;   // wave 0's ballots, published before the last barrier
    TSTAMP(1)
    double* Xf = Qi;   // where this step's Qm^-1 ends up (Qi or the idle C buffer)
    if (!any) {   // updateState :74-75
      lds_barrier();
      for (int idx = tid; idx < 48 * DL; idx += NT) C[idx] = R[idx];
      if (tid < d) mv[tid] = av[tid];
    } else {
      // Qm: Q with the missing rows/columns replaced by the identity, zero padding up to 32 x 32 (it aliases scratch)
      FOR_CM(32, 32, i, j) Qm[i * PL + j] = (i < p && j < p) ? ((ob[i] != 0.0 && ob[j] != 0.0) ? Q[i * PL + j] : (i == j ? 1.0 : 0.0)) : 0.0;
      if (!allobs && tid < p && ob[tid] == 0.0) Qi[tid * PL + tid] = 1.0;   // warm start: identity on the missing block
#ifdef DLM_STAMP
      if (spd_inverse_warm(tid, p, Qm, Qi, C, Es, Li, (int*)(ob + 40), (int*)(ob + 42), warm, &Xf, (n == 0) ? dbgc : nullptr)) st |= DLM_ST_NOT_PD;
#else
      if (spd_inverse_warm(tid, p, Qm, Qi, C, Es, Li, (int*)(ob + 40), (int*)(ob + 42), warm, &Xf)) st |= DLM_ST_NOT_PD;
#endif
      warm = true;
      if (!allobs) {
        FOR_CM(p, p, i, j) if (!(ob[i] != 0.0 && ob[j] != 0.0)) Xf[i * PL + j] = 0.0;
        lds_barrier();
      }
      TSTAMP(2)
      if (a.loglik) {   // -1/2 (n_obs log 2pi + log det Qm + e^T Qm^-1 e): det from a Cholesky factor of a copy of Qm
        for (int idx = tid; idx < 32 * PL; idx += NT) Es[idx] = Qm[idx];
        if (chol_block(tid, p, Es, PL)) st |= DLM_ST_NOT_PD;
        const double sj = wave_matvec<false>(tl, VW, p, p, Xf, PL, ev);
        if (tid >= VW * 64) {
          const int j = tid - VW * 64;
          double part = (j < p) ? 2.0 * log(Es[j * PL + j]) + ev[j] * sj + 1.8378770664093453 * ob[j] : 0.0;
          for (int o_ = 32; o_ > 0; o_ >>= 1) part += __shfl_xor(part, o_);
          ll -= 0.5 * part;
        }
      }
      gemm_t<false, false, 0>(tl, dt16, pt16, kp, RF, PL, Xf, PL, Kg, PL);          // K = R F Qm^-1 (6 tiles: waves 0..5)
      if (Xf != Qi && tid >= 6 * 64)                                                 // keep it as the next warm start
        for (int idx = tid - 6 * 64; idx < 32 * PL; idx += 2 * 64) Qi[idx] = Xf[idx];
      lds_barrier();
      { const double s = wave_matvec<false>(tl, VW, d, p, Kg, PL, ev); if (tid >= VW * 64 && tid - VW * 64 < d) mv[tid - VW * 64] = av[tid - VW * 64] + s; }
      gemm_t<false, true, 2, true>(tl, dt16, dt16, kp, Kg, PL, RF, PL, C, DL, R);    // C = R - K (R F)^T
    }
    lds_barrier();
    TSTAMP(3)
    {
      const int i = tid & 63, so = (t + 1) * recb;
#pragma unroll
      for (int q = 0; q < 6; ++q) bst(rfo, ro.m[q], so, C[i * DL + (tid >> 6) + NW * q]);
      bst(rfo, ro.v, so, mv[i]);
    }
    TSTAMP(4)
  }
#ifdef DLM_STAMP
  if (n == 0 && tid == 0 && a.status) { for (int k = 0; k < 8; ++k) a.status[1 + k] = (int)(seg[k] / (unsigned long long)T); a.status[6] = dbgc[0]; a.status[7] = dbgc[1]; a.status[9] = dbgc[2]; }
#endif
  if (a.loglik && tid == VW * 64) a.loglik[n] = ll;
  lds_barrier();
  bool bad = false;
  FOR_CM(d, d, i, j) bad |= !isfinite(C[i * DL + j]);
  if (tid < d) bad |= !isfinite(mv[tid]);
  if (__syncthreads_or(bad)) st |= DLM_ST_NONFINITE;
  if (a.status && tid == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// backward pass (information form, general p)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_smoother_tiled(KArgs a, const double* __restrict__ innov) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd;
  const int dt16 = (d + 15) / 16, pt16 = (p + 15) / 16, kd = (d + 3) / 4, kp = (p + 3) / 4;
  double* C = sm;            double* P = C + BIG;     double* T1 = P + BIG;    double* T2 = T1 + BIG;
  double* Fm = T2 + BIG;     double* Kg = Fm + MID;   double* PK = Kg + MID;
  double* Vi = PK + MID;     double* Qi = Vi + SML;   double* X = Qi + SML;    double* Li = T1;   // Cholesky scratch: T1 is idle then
  double* mv = X + SML;      double* mp = mv + 48;    double* qv = mp + 48;    double* rv = qv + 48;
  double* ev = rv + 48;      double* uv = ev + 48;    double* ob = uv + 48;    double* obp = ob + 48;
  double* tv = obp + 48;     double* cq = tv + 48;
  double* Glp = cq + 48;           // compact row-major copy of G when it fits (d <= 46); otherwise G is read from global memory
  const bool Gl = d <= GL_MAXD;
  double* CF = T2;           // d x p scratch aliases (T2 is free while K is built); uses ld PL
  const int vt = tid - VW * 64;   // lane index inside the vector-work wave (negative on the others)
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* es = innov + (size_t)n * T * p;   // e_t = y_t - f_t from the forward pass (NaN = missing), record t <-> es[(t-1) p ..]
  const int recb = rec * 8;
  const __amdgpu_buffer_rsrc_t rin = mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const __amdgpu_buffer_rsrc_t rout = mk_rsrc(a.smooth + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const __amdgpu_buffer_rsrc_t rinn = mk_rsrc(es, (size_t)T * p * 8);
  const RecOff ro = rec_offsets(tid, d);
  const int eoff = tid < p ? tid * 8 : OOB;
  int st = 0;

  zero_lds(tid, sm, SMTH_DOUBLES);
  lds_barrier();
  load_cm(tid, a.F, d, p, Fm, PL);
  int gcur = -1, tabcur = -1, tix = 0;   // tix / tvl: the wave's rows of the structured-G table (load_wave_table)
  double tvl = 0.0;
  if (tid < p) obp[tid] = -1.0;   // mask of the cached Vm^-1 (none yet)
  lds_barrier();

  // register prefetch of the record stream and of the innovations, one step ahead
  double pre[6], mcur, ecur;
#pragma unroll
  for (int q = 0; q < 6; ++q) pre[q] = bld(rin, ro.m[q], T * recb);
  mcur = bld(rin, ro.v, T * recb);
  ecur = bld(rinn, T > 0 ? eoff : OOB, (T > 0 ? T - 1 : 0) * p * 8);
#ifdef DLM_STAMP
  unsigned long long seg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tlast)::"memory");
#endif
  for (int t = T; t >= 0; --t) {
    lds_barrier();
    TSTAMP(0)
    // an opaque copy of the thread index: keeps the compiler from hoisting the operand addresses of the dozen
    // products below out of the time loop, where they would occupy (and spill) a few hundred registers
    int tl = tid;
    asm volatile("" : "+v"(tl));
    {   // G of the step INTO record t (the regular grid has one): refresh the LDS copy when it changes
      const int gi = (a.g_index && t > 0) ? a.g_index[t - 1] : 0;
      if (Gl && !a.spb && gi != gcur) {
        const double* Gs = a.G + (size_t)gi * dd;
        for (int idx = tid; idx < dd; idx += NT) { const int k_ = idx % d, o_ = idx / d; Glp[k_ * d + o_] = Gs[idx]; }   // Gs is column-major
        gcur = gi;
      }
    }
    {
#pragma unroll
      for (int q = 0; q < 6; ++q) if (ro.m[q] != OOB) C[(tid & 63) * DL + (tid >> 6) + NW * q] = pre[q];
      if (tid < d) mv[tid] = mcur;
      if (tid < p) {   // the lanes of wave 0: observation mask of this step and whether it repeats the cached one
        const bool obs = (t > 0) && (ecur == ecur);   // (an out-of-range load returns 0, but t > 0 guards it)
        ob[tid] = obs ? 1.0 : 0.0;
        ev[tid] = obs ? ecur : 0.0;
        const unsigned long long mo = __ballot(obs), md = __ballot((obs ? 1.0 : 0.0) != obp[tid]);
        if (tid == 0) { mp[0] = mo ? 1.0 : 0.0; mp[1] = md ? 0.0 : 1.0; }
      }
      {   // next record and innovation (below record 0: record 0 again, unused)
        const int tp = t > 0 ? t - 1 : 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) pre[q] = bld(rin, ro.m[q], tp * recb);
        mcur = bld(rin, ro.v, tp * recb);
        ecur = bld(rinn, t > 1 ? eoff : OOB, (t > 1 ? t - 2 : 0) * p * 8);
      }
    }
    if (a.f_stride && t > 0) load_cm(tid, a.F + (size_t)(t - 1) * a.f_stride, d, p, Fm, PL);
    lds_barrier();
    TSTAMP(1)
    const bool any = mp[0] != 0.0, same = mp[1] != 0.0;
    const double* Gt = a.G + (size_t)((a.g_index && t > 0) ? a.g_index[t - 1] : 0) * dd;   // G of the step INTO record t

    if (any) {
      if (!same || a.v_tstride) {   // Vm^-1 for this missingness pattern (cached while the pattern repeats and V is time-invariant)
        const double* Vt = V + (size_t)(t - 1) * a.v_tstride;   // V of the observation at record t
        lds_barrier();
        FOR_CM(p, p, i, j) Vi[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? Vt[i + j * p] : (i == j ? 1.0 : 0.0);
        if (spd_inverse(tid, p, Vi, Li, (int*)(cq + 40))) st |= DLM_ST_NOT_PD;
        FOR_CM(p, p, i, j) if (!(ob[i] != 0.0 && ob[j] != 0.0)) Vi[i * PL + j] = 0.0;
        if (tid < p) obp[tid] = ob[tid];
        lds_barrier();
      }
      TSTAMP(2)
      // Every phase below is one or two MFMA products on waves 0.. plus the step's vector work on wave VW.
      gemm_t<false, false, 0>(tl, dt16, pt16, kd, C, DL, Fm, PL, CF, PL);            // C F
      lds_barrier();
      TSTAMP(3)
      gemm_t<false, false, 0>(tl, dt16, pt16, kp, CF, PL, Vi, PL, Kg, PL);           // K = C F Vm^-1
      lds_barrier();
      TSTAMP(4)
      gemm_t<true, false, 0>(tl, pt16, pt16, kd, Fm, PL, Kg, PL, X, PL);             // F^T K
      lds_barrier();
      TSTAMP(5)
      gemm_t<false, false, 2, true>(tl, pt16, pt16, kp, Vi, PL, X, PL, Qi, PL, Vi);  // Qm^-1 = Vm^-1 - Vm^-1 F^T K (symmetric)
    }
    lds_barrier();
    TSTAMP(6)
    // x1 = P C -> T1 ; P K ; u = Qm^-1 e ; C q
    gemm_t<false, false, 0>(tl, dt16, dt16, kd, P, DL, C, DL, T1, DL);
    if (any) gemm_t<false, false, 0>(tl, dt16, pt16, kd, P, DL, Kg, PL, PK, PL, nullptr, 1.0, 0, 0, 1);
    {
      const double su = any ? wave_matvec<false>(tl, VW, p, p, Qi, PL, ev) : 0.0;
      const double sc = wave_matvec<false>(tl, VW, d, d, C, DL, qv);
      if (any && vt >= 0 && vt < p) uv[vt] = su;
      if (vt >= 0 && vt < d) cq[vt] = sc;
    }
    lds_barrier();
    TSTAMP(7)
    // x2 = C (P C) -> T2 (symmetric) ; X = Qm^-1 + K^T P K (symmetric) on the waves the first product leaves idle ; u - K^T q
    gemm_t<false, false, 0, true>(tl, dt16, dt16, kd, C, DL, T1, DL, T2, DL);
    if (any && t > 0) {
      gemm_t<true, false, 1, true>(tl, pt16, pt16, kd, Kg, PL, PK, PL, X, PL, Qi, 1.0, 0, 0, 6);
      const double s = wave_matvec<true>(tl, VW, p, d, Kg, PL, qv);
      if (vt >= 0 && vt < p) tv[vt] = uv[vt] - s;
    }
    lds_barrier();
    TSTAMP(8)
    {   // s_t = m_t + C_t q_t ; S_t = C_t - C_t P_t C_t  (7 stores per thread; rows >= 48 of the lane index read
        // in-bounds scratch that is never stored)
      const int i = tid & 63, so = t * recb;
#pragma unroll
      for (int q = 0; q < 6; ++q) { const int j = (tid >> 6) + NW * q; bst(rout, ro.m[q], so, C[i * DL + j] - T2[i * DL + j]); }
      bst(rout, ro.v, so, mv[tid & 63] + cq[tid & 63]);
    }
    if (t == 0) break;

    // (q_{t-1}, P_{t-1})
    lds_barrier();
    TSTAMP(9)
    if (any) {
      gemm_t<false, false, 0>(tl, dt16, pt16, kp, Fm, PL, X, PL, CF, PL);            // F X   (T2 is free again)
      {   // r = q + F (u - K^T q)
        const double s = wave_matvec<false>(tl, VW, d, p, Fm, PL, tv);
        if (vt >= 0 && vt < d) rv[vt] = qv[vt] + s;
      }
      lds_barrier();
      TSTAMP(10)
      // M = P + (F X) F^T - F (P K)^T - (P K) F^T in one symmetric product.  The expanded update treats P as
      // exactly symmetric (it uses (P K)^T for K^T P); an antisymmetric rounding component would NOT be
      // contracted by (I - F K^T) and grows exponentially for unit-root models.  Computing the
      // upper tiles and mirroring them keeps P symmetric by construction.
      gemm_update3(tl, dt16, kp, CF, Fm, Fm, PK, PK, Fm, PL, P, DL, P);
    } else if (vt >= 0 && vt < d) rv[vt] = qv[vt];
    lds_barrier();
    TSTAMP(11)
    // like Smoothing.smoothStep (Smoothing.scala:41): always the table entry g(dt), also for dt == 0
    if (a.spb) {   // structured G: P = G^T M G and q = G^T r as gathers with the columns of G
      const int gi = (a.g_index && t > 0) ? a.g_index[t - 1] : 0;
      const SparseBig* tab = a.spb + 2 * gi + 1;
      if (gi != tabcur) { load_wave_table(tid, d, tab, tix, tvl); tabcur = gi; }
      sparse_congruence(tl, d, tab->K, tix, tvl, P, T1, P, nullptr, 0.0, rv, qv, ob + 46);
    } else {
      if (Gl) gemm_g<false, false, true>(tl, dt16, kd, P, DL, Glp, d, T1, DL);            // M G
      else gemm_g<false, false, false>(tl, dt16, kd, P, DL, Gt, d, T1, DL);
      {   // q = G^T r
        const double s = Gl ? wave_matvec<true>(tl, VW, d, d, Glp, d, rv) : wave_matvec_g<true>(tl, VW, d, Gt, rv);
        if (vt >= 0 && vt < d) qv[vt] = s;
      }
      lds_barrier();
      TSTAMP(12)
      if (Gl) gemm_g<true, true, true>(tl, dt16, kd, T1, DL, Glp, d, P, DL);
      else gemm_g<true, true, false>(tl, dt16, kd, T1, DL, Gt, d, P, DL);
    }                           // P = G^T M G (symmetric)
    TSTAMP(13)
  }
#ifdef DLM_STAMP
  if (n == 0 && tid == 0 && a.status) for (int k = 0; k < 16; ++k) a.status[1 + k] = (int)(seg[k] / (unsigned long long)(T + 1));
#endif
  lds_barrier();
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
  lds_barrier();
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
    lds_barrier();
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
    lds_barrier();
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
    lds_barrier();
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
        lds_barrier();
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
    lds_barrier();
    if (tid < d) thn[tid] = thc[tid];
    if (t == 0) break;

    if (any) {
      if (!same) {
        lds_barrier();
        FOR_CM(p, p, i, j) Vi[i * PL + j] = (ob[i] != 0.0 && ob[j] != 0.0) ? V[i + j * p] : (i == j ? 1.0 : 0.0);
        if (spd_inverse(tid, p, Vi, Li, (int*)flagv)) st |= DLM_ST_NOT_PD;
        FOR_CM(p, p, i, j) if (!(ob[i] != 0.0 && ob[j] != 0.0)) Vi[i * PL + j] = 0.0;
        if (tid < p) obp[tid] = ob[tid];
        lds_barrier();
      }
      gemm_t<false, false, 0>(tid, dt16, pt16, kd, C, DL, Fm, PL, CF, PL);            // C F
      if (tid < d) {
        double s = 0.0;
        if (dtt == 0.0) s = mp[tid];
        else for (int k = 0; k < d; ++k) s = fma(Gt[tid + k * d], mp[k], s);
        rv[tid] = s;   // a*_t
      }
      lds_barrier();
      gemm_t<false, false, 0>(tid, dt16, pt16, kp, CF, PL, Vi, PL, Kg, PL);           // K = C F Vm^-1
      if (tid < p) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Fm[k * PL + tid], rv[k], s); ev[tid] = (ob[tid] != 0.0) ? tv[tid] - s : 0.0; }
      lds_barrier();
      gemm_t<true, false, 0>(tid, pt16, pt16, kd, Fm, PL, Kg, PL, X, PL);             // F^T K
      lds_barrier();
      gemm_t<false, false, 2>(tid, pt16, pt16, kp, Vi, PL, X, PL, Qi, PL, Vi);        // Qm^-1
      lds_barrier();
      if (tid < p) {
        double s = 0.0;
        for (int j = 0; j < p; ++j) s = fma(Qi[tid * PL + j], ev[j], s);              // u = Qm^-1 e
        for (int k = 0; k < d; ++k) s = fma(-Kg[k * PL + tid], qv[k], s);             // - K^T q
        uv[tid] = s;
      }
      lds_barrier();
      if (tid < d) { double s = qv[tid]; for (int j = 0; j < p; ++j) s = fma(Fm[tid * PL + j], uv[j], s); rv[tid] = s; }
    } else if (tid < d) rv[tid] = qv[tid];
    lds_barrier();
    if (tid < d) {
      double s = 0.0;
      if (dtt == 0.0) s = rv[tid];
      else for (int k = 0; k < d; ++k) s = fma(Gt[k + tid * d], rv[k], s);             // q = G^T r
      qv[tid] = s;
    }
  }
  lds_barrier();
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

hipError_t launch_tiled_filter(const KArgs& a, double* innov, hipStream_t s) {
  if (wave48_filter_supported(a)) return launch_wave48_filter(a, a.spb_k, innov, s);
  const size_t lds = tiled_filter_lds_bytes();
  hipError_t e = set_lds((const void*)k_filter_tiled<false>, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_filter_tiled<false>, dim3(a.N), dim3(NT), lds, s, a, (double*)nullptr, innov);
  return hipGetLastError();
}

hipError_t launch_tiled_simsmooth(const KArgs& a, double* xplus, double* ystar, hipStream_t s) {
  if (wave48_simsmooth_supported(a)) return launch_wave48_simsmooth(a, a.spb_k, xplus, ystar, s);
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

hipError_t launch_tiled_smoother(const KArgs& a, const double* innov, hipStream_t s) {
  if (wave48_smoother_supported(a)) return launch_wave48_smoother(a, a.spb_k, innov, s);
  const size_t lds = tiled_smoother_lds_bytes();
  hipError_t e = set_lds((const void*)k_smoother_tiled, lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_smoother_tiled, dim3(a.N), dim3(NT), lds, s, a, innov);
  return hipGetLastError();
}

}  // namespace dlm
