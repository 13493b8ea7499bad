// Fast path for structured system matrices: d <= 15, p == 1, one G, regular grid, and at
// most K <= 4 nonzeros in every row AND every column of G.  Every model the reference can
// build satisfies this (polynomial: bidiagonal, seasonal: 2x2 rotation blocks, regression:
// identity, autoregressive: diagonal; Dlm.scala:139-243), and |+| / |*| keep it
// (block-diagonal composition, Dlm.scala:107-122, :197-208).  Dense G uses dlm_mfma16.hip.
//
// Same register layout as dlm_mfma16.hip (fp64 MFMA accumulator layout, one wavefront per
// series).  What changes: the congruence T X T^T with a sparse T (T = G in the forward
// pass, T = G^T in the backward pass) costs O(K d^2) instead of two dense products.  It runs
// as two gather passes through a wave-private LDS image with an odd leading dimension (17
// doubles), so that both the row-wise and the transposed read are bank-conflict free:
//     pass 1   Y[i][j] = sum_s X[i][idx_s(j)] val_s(j)          (= X T^T)
//     pass 2   Z[i][j] = sum_s Y[idx_s(j)][i] val_s(j)          (= (T Y)^T = T X T^T, symmetric)
// where (idx_s(j), val_s(j)) are the nonzeros of row j of T.  The fp64 MFMA pipe is then
// used only for the two genuinely dense products of the backward pass (P C and C (P C)),
// and runs concurrently with the VALU work of the other resident waves.
//
// Recursions and reference citations: see dlm_mfma16.hip (identical algebra).
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int LD = 17;                        // leading dimension of an LDS image
constexpr int IMG = 16 * LD;                  // doubles per image
constexpr int WAVE_LDS = 2 * IMG + 5 * 16;    // two images + five 16-vectors per wave

#ifndef DLM_SM_WAVES
#define DLM_SM_WAVES 4
#endif
#ifndef DLM_FI_WAVES
#define DLM_FI_WAVES 5
#endif

__device__ __forceinline__ d4 mmT(const d4& x, const d4& y) {  // X^T * Y
#ifdef DLM_MMT_SPLIT
  // two independent accumulation chains halve the dependent-MFMA latency
  const d4 z = {0.0, 0.0, 0.0, 0.0};
  d4 a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], z, 0, 0, 0);
  d4 a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], z, 0, 0, 0);
  a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], a0, 0, 0, 0);
  a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], a1, 0, 0, 0);
  return a0 + a1;
#else
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], acc, 0, 0, 0);
  return acc;
#endif
}

// LDS hand-off between lanes of ONE wavefront: the LDS queue is in order per wave, so only
// the compiler has to be kept from reordering.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N>
__device__ __forceinline__ double row_ror(double v) {  // DPP rotate within a 16-lane row
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a row (over c); every lane of the row gets the sum
__device__ __forceinline__ double row_sum(double v) {
  v += row_ror<8>(v); v += row_ror<4>(v); v += row_ror<2>(v); v += row_ror<1>(v);
  return v;
}
// sum over lanes c, c+16, c+32, c+48 (over g) with the gfx950 permlane swaps; all get the sum
__device__ __forceinline__ double sum_g(double v) {
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}

__device__ __forceinline__ double uniform_from_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Record I/O through raw buffer instructions: padded lanes carry an out-of-range offset, for
// which the hardware returns 0 on loads and drops stores -- no exec-mask branches.
constexpr int OOB = 0x7ffffff0;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
#ifndef DLM_BUF_LOAD
#define DLM_BUF_LOAD 1
#endif
#ifndef DLM_BUF_STORE
#define DLM_BUF_STORE 1
#endif
#ifndef DLM_IMG_XCHG
#define DLM_IMG_XCHG 1
#endif
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, const char* base, int voff, int soff) {
#ifdef DLM_EXP_NOLOAD
  soff = 0;  // timing experiment: always re-read record 0 (cache resident)
#endif
#if DLM_BUF_LOAD
  const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __hiloint2double((int)v[1], (int)v[0]);
#else
  return voff != OOB ? *(const double*)(base + (size_t)soff + voff) : 0.0;
#endif
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, char* base, int voff, int soff, double x) {
#ifdef DLM_EXP_NOSTORE
  asm volatile("" ::"v"(x));  // timing experiment: keep the value live, skip the store
  return;
#endif
#if DLM_BUF_STORE
  const u2 v = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
#else
  if (voff != OOB) *(double*)(base + (size_t)soff + voff) = x;
#endif
}

// Diagnostic build only (-DDLM_STAMP): s_memtime stamps around the phases of the backward step;
// the sums of series 0 are written into status[1..] (never in the shipped build).
#ifdef DLM_STAMP
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(k) { const unsigned long long _t = stamp(); seg[k] += _t - tlast; tlast = _t; }
#else
#define STAMP(k)
#endif

// Z = T X T^T for symmetric X (std layout), T given by the per-column-lane tables idx/val.
template <int K>
__device__ __forceinline__ d4 congruence(const d4& x, double* imgA, double* imgB, const int (&idx)[K],
                                         const double (&val)[K], int g, int c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) imgA[(4 * r + g) * LD + c] = x[r];
  wave_sync();
  d4 y;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double acc = imgA[(4 * r + g) * LD + idx[0]] * val[0];
#pragma unroll
    for (int s = 1; s < K; ++s) acc = fma(imgA[(4 * r + g) * LD + idx[s]], val[s], acc);
    y[r] = acc;
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) imgB[(4 * r + g) * LD + c] = y[r];
  wave_sync();
  d4 z;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    double acc = imgB[idx[0] * LD + 4 * r + g] * val[0];
#pragma unroll
    for (int s = 1; s < K; ++s) acc = fma(imgB[idx[s] * LD + 4 * r + g], val[s], acc);
    z[r] = acc;
  }
  return z;
}

// ---------------------------------------------------------------------------------------
// forward pass (no MFMA at all: O(K d^2) per step)
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256, DLM_FI_WAVES) void k_filter_sp16(KArgs a, const SparseT* __restrict__ sp,
                                                     double* __restrict__ side) {
  __shared__ __attribute__((aligned(16))) double lds[4 * WAVE_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep it scalar
  const int n = blockIdx.x * 4 + wave;
  if (n >= a.N) return;
  double* imgA = lds + wave * WAVE_LDS;
  double* imgB = imgA + IMG;
  double* vM = imgB + IMG;       // m, column-indexed
  double* vRF = vM + 16;         // R F
  const int d = a.d, T = a.T, rec = d + d * d;
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d;

  const double* W = a.W + (size_t)n * a.w_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double V = a.V[(size_t)n * a.v_stride];
  const double* y = a.y + (size_t)n * T;
  char* bout = (char*)(a.filt + (size_t)n * (T + 1) * rec);
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(bout, (size_t)(T + 1) * rec * 8);
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * 2 : nullptr;
  double* sd = side ? side + (size_t)n * (T + 1) * 2 : nullptr;

  int idx[K];
  double val[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = sp->idx[c][s]; val[s] = sp->val[c][s]; }
  d4 w, cc;
  double Fr[4];
  bool vr[4];
  int offC[4];                                   // byte offset of C[4r+g][c] inside a record
  const int offM = (g == 0 && vc) ? c * 8 : OOB; // byte offset of m[c]
  const int recb = rec * 8;
  const double Fc = vc ? a.F[c] : 0.0;
  double mcol = vc ? m0[c] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    vr[r] = i < d;
    const bool ok = vr[r] && vc;
    offC[r] = ok ? (d + i * d + c) * 8 : OOB;
    w[r] = ok ? W[i * d + c] : 0.0;
    cc[r] = ok ? C0[i * d + c] : 0.0;
    Fr[r] = vr[r] ? a.F[i] : 0.0;
  }
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;

#pragma unroll
  for (int r = 0; r < 4; ++r) buf_store(rout, bout, offC[r], 0, cc[r]);
  buf_store(rout, bout, offM, 0, mcol);
  if (lane == 0) {
    if (fq) { fq[0] = __builtin_nan(""); fq[1] = __builtin_nan(""); }
    if (sd) { sd[0] = __builtin_nan(""); sd[1] = __builtin_nan(""); }
  }

  double ychunk = 0.0;
  for (int t = 0; t < T; ++t) {
    if ((t & 63) == 0) ychunk = (t + lane < T) ? y[t + lane] : 0.0;
    const double yt = uniform_from_lane(ychunk, t & 63);

    // advState: a = G m, R = G C G^T + W
    vM[c] = mcol;
    d4 R = congruence<K>(cc, imgA, imgB, idx, val, g, c);   // first wave_sync also covers vM
    double acol = vM[idx[0]] * val[0];
#pragma unroll
    for (int s = 1; s < K; ++s) acol = fma(vM[idx[s]], val[s], acol);
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] += w[r];

    // f = F^T a ; RF ; Q = F^T R F + V
    const double f = row_sum(Fc * acol);
    double rfc = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) rfc = fma(R[r], Fr[r], rfc);
    rfc = sum_g(rfc);                                        // (R F)[c] in every lane
    vRF[c] = rfc;
    wave_sync();
    double rfr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rfr[r] = vRF[4 * r + g];     // (R F)[4r+g]
    const double Q = row_sum(Fc * rfc) + V;

    if (yt == yt) {
      // Joseph form for p = 1 with K = RF / Q:  R - K RF^T - RF K^T + Q K K^T
      //   = R - (RF_i / Q) * RF_j * (2 - Q * (1/Q))   -- the same expression, factored
      const double e = yt - f, rq = 1.0 / Q;
      const double Kc = rfc * rq;
      const double gam = rfc * (2.0 - Q * rq);
#pragma unroll
      for (int r = 0; r < 4; ++r) cc[r] = fma(-(rfr[r] * rq), gam, R[r]);
      mcol = fma(Kc, e, acol);
      if (sd && lane == 0) { sd[2 * (t + 1)] = e * rq; sd[2 * (t + 1) + 1] = rq; }
    } else {
      cc = R;
      mcol = acol;
      if (sd && lane == 0) { sd[2 * (t + 1)] = __builtin_nan(""); sd[2 * (t + 1) + 1] = __builtin_nan(""); }
    }
    if (fq && lane == 0) { fq[2 * (t + 1)] = f; fq[2 * (t + 1) + 1] = Q; }
    const int so = (t + 1) * recb;
#pragma unroll
    for (int r = 0; r < 4; ++r) buf_store(rout, bout, offC[r], so, cc[r]);
    buf_store(rout, bout, offM, so, mcol);
    wave_sync();   // vM / images are rewritten at the top of the next step
  }
  bool bad = vc && !isfinite(mcol);
#pragma unroll
  for (int r = 0; r < 4; ++r) bad |= vr[r] && vc && !isfinite(cc[r]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// backward pass: MFMA for P C and C (P C); gathers for G^T M G
// ---------------------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256, DLM_SM_WAVES) void k_smoother_sp16(KArgs a, const SparseT* __restrict__ sp,
                                                       const double* __restrict__ side) {
  __shared__ __attribute__((aligned(16))) double lds[4 * WAVE_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep it scalar
  const int n = blockIdx.x * 4 + wave;
  if (n >= a.N) return;
  double* imgA = lds + wave * WAVE_LDS;
  double* imgB = imgA + IMG;
  double* vK = imgB + IMG;       // K_t
  double* vQ = vK + 16;          // q_t
  double* vPK = vQ + 16;         // P K
  double* vCQ = vPK + 16;        // C q
  double* vR = vCQ + 16;         // r
  const int d = a.d, T = a.T, rec = d + d * d;
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d, col15 = (c == 15);

  const double V = a.V[(size_t)n * a.v_stride];
  const double rV = 1.0 / V;
  const char* bin = (const char*)(a.filt_in + (size_t)n * (T + 1) * rec);
  char* bout = (char*)(a.smooth + (size_t)n * (T + 1) * rec);
  const __amdgpu_buffer_rsrc_t rin = make_rsrc(bin, (size_t)(T + 1) * rec * 8);
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(bout, (size_t)(T + 1) * rec * 8);
  const double* sd = side + (size_t)n * (T + 1) * 2;
  const int recb = rec * 8;

  int idx[K];
  double val[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = sp->idx[c][s]; val[s] = sp->val[c][s]; }
  double Fr[4];
  bool vr[4];
  int offC[4];
  const int offM = (g == 0 && vc) ? c * 8 : OOB;     // store of s[c]: one row group only
  const int offMl = vc ? c * 8 : OOB;                // load of m[c]: every row group
  const double Fc = vc ? a.F[c] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    vr[r] = i < d;
    Fr[r] = vr[r] ? a.F[i] : 0.0;
    offC[r] = (vr[r] && vc) ? (d + i * d + c) * 8 : OOB;
  }
  d4 P = {0.0, 0.0, 0.0, 0.0};
  double qcol = 0.0;
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;

  d4 ncc;
  double nm;
#pragma unroll
  for (int r = 0; r < 4; ++r) ncc[r] = buf_load(rin, bin, offC[r], T * recb);
  nm = buf_load(rin, bin, offMl, T * recb);
  double neq = sd[2 * T], niq = sd[2 * T + 1];
  vQ[c] = 0.0;
  d4 Sv = {0.0, 0.0, 0.0, 0.0};
  double scol = 0.0;
  const double m15 = col15 ? 1.0 : 0.0;

#ifdef DLM_STAMP
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp();
#endif
  for (int t = T; t >= 0; --t) {
    STAMP(7)
    const d4 cc = ncc;
    const double mcol = nm;
    // the innovations are per-series scalars: keep them in SGPRs so `observed` is a scalar branch
    const double eq = uniform_from_lane(neq, 0), iq = uniform_from_lane(niq, 0);
    {
      const int tp = t > 0 ? t - 1 : 0;                      // record 0 is re-read harmlessly at the end
#pragma unroll
      for (int r = 0; r < 4; ++r) ncc[r] = buf_load(rin, bin, offC[r], tp * recb);
      nm = buf_load(rin, bin, offMl, tp * recb);
      neq = sd[2 * tp]; niq = sd[2 * tp + 1];
    }
    const bool observed = (iq == iq) && t > 0;

    // K_t = C_t F / V
    double kcol = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) kcol = fma(cc[r], Fr[r], kcol);
    kcol = observed ? sum_g(kcol) * rV : 0.0;
    vK[c] = kcol;
    wave_sync();                                             // also publishes vQ of the last step
    double kr[4], qr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { kr[r] = vK[4 * r + g]; qr[r] = vQ[4 * r + g]; }
    d4 b1;
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[r] = fma(kr[r], m15, cc[r]);   // column 15 of C is zero
    STAMP(0)
    const d4 x1 = mmT(P, b1);                                // [P C | P K]
    d4 b2;
#pragma unroll
    for (int r = 0; r < 4; ++r) b2[r] = col15 ? qr[r] : x1[r];
    // [C P C | C q]: only the OUTPUT (s_t, S_t) needs it, so it is consumed at the very end of the
    // step and its MFMA latency hides behind the recursion work below
#ifdef DLM_STAMP
    asm volatile("" ::"v"(b2[0]), "v"(b2[1]), "v"(b2[2]), "v"(b2[3]));
#endif
    STAMP(1)
    const d4 x2 = mmT(cc, b2);
    STAMP(2)

    if (t > 0) {
      // (q_{t-1}, P_{t-1}) from (q_t, P_t).  Column 15 of x1 is P K: park x1 in the idle image
      // and read that column back in both indexings.
      d4 M = P;
      double rcol = qcol;
      if (observed) {
#pragma unroll
        for (int r = 0; r < 4; ++r) imgA[(4 * r + g) * LD + c] = x1[r];
        wave_sync();
        const double pkc = imgA[c * LD + 15];
        double pkr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) pkr[r] = imgA[(4 * r + g) * LD + 15];
        const double kq = row_sum(kcol * qcol), kpk = row_sum(kcol * pkc);
        const double sc = iq + kpk;
        rcol = fma(Fc, eq - kq, qcol);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          M[r] = fma(-pkr[r], Fc, fma(-Fr[r], pkc, fma(Fr[r] * Fc, sc, P[r])));
      }
      vR[c] = rcol;
      wave_sync();                                           // column-15 reads precede the image rewrite
#ifdef DLM_STAMP
      asm volatile("" ::"v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]));
#endif
      STAMP(3)
      P = congruence<K>(M, imgA, imgB, idx, val, g, c);      // G^T M G (its first sync covers vR)
      qcol = vR[idx[0]] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) qcol = fma(vR[idx[s]], val[s], qcol);
      vQ[c] = qcol;                                          // published by the next wave_sync
      wave_sync();                                           // pass-2 reads of imgB precede its reuse
#ifdef DLM_STAMP
      asm volatile("" ::"v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]));
#endif
      STAMP(4)
    }

    // output: s_t = m_t + C_t q_t (column 15 of x2), S_t = C_t - C_t P_t C_t
#pragma unroll
    for (int r = 0; r < 4; ++r) imgB[(4 * r + g) * LD + c] = x2[r];
    wave_sync();
    scol = mcol + imgB[c * LD + 15];
    const int so = t * recb;
#pragma unroll
    for (int r = 0; r < 4; ++r) { Sv[r] = cc[r] - x2[r]; buf_store(rout, bout, offC[r], so, Sv[r]); }
    buf_store(rout, bout, offM, so, scol);
    STAMP(5)
  }
#ifdef DLM_STAMP
  if (n == 0 && lane == 0 && a.status)
    for (int k = 0; k < 8; ++k) a.status[1 + k] = (int)(seg[k] / (unsigned long long)(T + 1));
#endif
  // P and q carry any non-finite value down to record 0: test the last output
  bool bad = vc && !isfinite(scol);
#pragma unroll
  for (int r = 0; r < 4; ++r) bad |= vr[r] && vc && !isfinite(Sv[r]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// host side: structure detection and launch
// ---------------------------------------------------------------------------------------
// Nonzeros of the rows of G (`rows`, forward pass) and of the columns of G (`cols`, backward
// pass).  Returns the largest count, or 99 if any row/column has more than 4.
int sparse16_analyse(const double* G /* d x d column-major, host */, int d, SparseT* rows, SparseT* cols) {
  int kmax = 1;
  for (int pass = 0; pass < 2; ++pass) {
    SparseT* t = pass ? cols : rows;
    for (int j = 0; j < 16; ++j)
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
  return kmax;
}

template <int K>
static hipError_t launch_f(const KArgs& a, const SparseT* sp, double* side, hipStream_t s) {
  hipLaunchKernelGGL(k_filter_sp16<K>, dim3((a.N + 3) / 4), dim3(256), 0, s, a, sp, side);
  return hipGetLastError();
}
template <int K>
static hipError_t launch_s(const KArgs& a, const SparseT* sp, const double* side, hipStream_t s) {
  hipLaunchKernelGGL(k_smoother_sp16<K>, dim3((a.N + 3) / 4), dim3(256), 0, s, a, sp, side);
  return hipGetLastError();
}

hipError_t launch_sparse16_filter(const KArgs& a, int K, const SparseT* rows_dev, double* side, hipStream_t s) {
  switch (K) {
    case 1: return launch_f<1>(a, rows_dev, side, s);
    case 2: return launch_f<2>(a, rows_dev, side, s);
    case 3: return launch_f<3>(a, rows_dev, side, s);
    case 4: return launch_f<4>(a, rows_dev, side, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_sparse16_smoother(const KArgs& a, int K, const SparseT* cols_dev, const double* side, hipStream_t s) {
  switch (K) {
    case 1: return launch_s<1>(a, cols_dev, side, s);
    case 2: return launch_s<2>(a, cols_dev, side, s);
    case 3: return launch_s<3>(a, cols_dev, side, s);
    case 4: return launch_s<4>(a, cols_dev, side, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace dlm
