// Fast path for structured system matrices: d <= 15, p == 1, any time grid (several G(dt) tables,
// W dt, dt == 0), and at most K <= 4 nonzeros in every row AND every column of every G.  Every model the reference can
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
constexpr int WAVE_LDS = 2 * IMG + 8 * 16 + 2;    // two images + eight 16-vectors + the state of the steady-state test per wave

#ifndef DLM_SM_WAVES
#define DLM_SM_WAVES 4
#endif
#ifndef DLM_FI_WAVES
#define DLM_FI_WAVES 5
#endif
#ifndef DLM_SM_WAVES_K2
#define DLM_SM_WAVES_K2 5
#endif
constexpr int SM_WAVES_K2 = DLM_SM_WAVES_K2;   // backward kernel, at most two nonzeros per row / column of G (C2): five waves fit with one value spilled into the every-8th-step branch
constexpr int SM_WAVES = DLM_SM_WAVES, FI_WAVES = DLM_FI_WAVES;     // waves per SIMD the register allocator must at least allow (backward / forward kernels)

// One dependent chain of four: two chains of two were measured slower (profiles/r01_pmc_notes.md).
__device__ __forceinline__ d4 mmT(const d4& x, const d4& y) {  // X^T * Y
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], acc, 0, 0, 0);
  return acc;
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
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x120 + N, 0xf, 0xf, true);   // bound_ctrl: no "old" value to set up
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x120 + N, 0xf, 0xf, true);
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

// max over the 64 lanes; every lane gets it (the steady-state tests: every fourth step at most)
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, row_ror<8>(v)); v = fmax(v, row_ror<4>(v)); v = fmax(v, row_ror<2>(v)); v = fmax(v, row_ror<1>(v));
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = fmax(__hiloint2double((int)h[0], (int)l[0]), __hiloint2double((int)h[1], (int)l[1]));
  lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return fmax(__hiloint2double((int)h[0], (int)l[0]), __hiloint2double((int)h[1], (int)l[1]));
}

// two float maxima at the price of one 64-bit reduction: a rides in the low, b in the high word through the same shuffles
__device__ __forceinline__ void wave_max2f(float& a, float& b) {
#define DLM_MAX2F_ROW(N) { a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x120 + N, 0xf, 0xf, true))); \
                           b = fmaxf(b, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(b), 0x120 + N, 0xf, 0xf, true))); }
  DLM_MAX2F_ROW(8) DLM_MAX2F_ROW(4) DLM_MAX2F_ROW(2) DLM_MAX2F_ROW(1)
#undef DLM_MAX2F_ROW
  u2 l = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(a), (unsigned)__float_as_int(a), false, false);
  u2 h = __builtin_amdgcn_permlane16_swap((unsigned)__float_as_int(b), (unsigned)__float_as_int(b), false, false);
  a = fmaxf(__int_as_float((int)l[0]), __int_as_float((int)l[1])); b = fmaxf(__int_as_float((int)h[0]), __int_as_float((int)h[1]));
  l = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(a), (unsigned)__float_as_int(a), false, false);
  h = __builtin_amdgcn_permlane32_swap((unsigned)__float_as_int(b), (unsigned)__float_as_int(b), false, false);
  a = fmaxf(__int_as_float((int)l[0]), __int_as_float((int)l[1])); b = fmaxf(__int_as_float((int)h[0]), __int_as_float((int)h[1]));
}
// 1/x from v_rcp_f64 and two Newton steps (about 1 ulp): 5 VALU instructions instead of the 11 of the IEEE
// division expansion.  The forward pass is bound by VALU issue, and every lane computes this scalar.
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
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
// -DDLM_EXP_NOLOAD / -DDLM_EXP_NOSTORE: the two timing probes behind the floors quoted in profiles/r01_pmc_notes.md
// (records re-read from cache / stores skipped with the value kept live); never defined in the shipped build.
__device__ __forceinline__ double buf_load(__amdgpu_buffer_rsrc_t r, const char*, int voff, int soff) {
#ifdef DLM_EXP_NOLOAD
  soff = 0;
#endif
  const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __hiloint2double((int)v[1], (int)v[0]);
}
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, char*, int voff, int soff, double x) {
#ifdef DLM_EXP_NOSTORE
  asm volatile("" ::"v"(x));
  return;
#endif
  const u2 v = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// the forward pass's (e / Q, 1 / Q) pair for the backward pass: one 16-byte store from lane 0 (the others carry OOB)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void side_store(__amdgpu_buffer_rsrc_t r, int voff, int soff, double a, double b) {
  const u4 v = {(unsigned)__double2loint(a), (unsigned)__double2hiint(a), (unsigned)__double2loint(b), (unsigned)__double2hiint(b)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}

// ---- record prefetch by LDS DMA (backward pass) ---------------------------------------------------
// `buffer_load_dwordx4 ... lds` copies 16 B per lane straight from HBM into LDS: no VGPRs are held while
// the load is in flight, so the backward pass can keep TWO records in flight per wave (the loaded HBM
// latency is of the order of one step) without giving up occupancy.  The instruction is issued from inline
// assembly on purpose: the compiler's wait-count insertion treats an LDS-DMA it knows about as aliasing
// every later LDS read and waits vmcnt(0) -- which would also wait for the just-issued record stores.
// The waits are placed by hand instead (vm_wait<N>); vector-memory operations of a wave retire in order.
typedef int i4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i4 rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i4 r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  return r;
}
// copy the record of n16 x 16 bytes at byte offset soff of the buffer to LDS byte address lds_addr
__device__ __forceinline__ void dma_record(const i4& rs, unsigned lds_addr, int soff, int lane, int n16) {
  const int voff = lane * 16;
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);   // wave-uniform by construction
  soff = __builtin_amdgcn_readfirstlane(soff);
  if (lane < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if (lane + 64 < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds"
                 ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {   // at most N vector-memory operations of this wave still in flight
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
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

// LDS reads issued from inline assembly as single ds_read_b64: the compiler would pair them into
// ds_read2_b64, which runs at half the LDS rate (8 LDS cycles per KiB against 4 for two ds_read_b64;
// MI355X_MICROARCH LDS table) -- and both passes of this kernel are bound by LDS cycles.  The compiler
// does not count these reads: lds_fence() waits for them before their results are used.
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
}
template <int OFF>
__device__ __forceinline__ double lds_read64(unsigned addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void lds_fence(d4& a, d4& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}
__device__ __forceinline__ void lds_fence(d4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory"); }

template <int K>
__device__ __forceinline__ d4 congruence_pass2(const d4& y, double* imgB, const int (&idx)[K], const double (&val)[K],
                                               int g, int c, const d4* add);

// Z = T X T^T for symmetric X (std layout), T given by the per-column-lane tables idx/val.
// sym (wave-uniform): replace X by (X + X^T)/2 first.  The backward recursion needs it now and then: its
// rank-2 update treats P as exactly symmetric, and an antisymmetric rounding component would otherwise
// escape the contraction (I - F K^T) . (I - K F^T).
template <int K>
__device__ __forceinline__ d4 congruence_pass1(const d4& x, double* imgA, const int (&idx)[K],
                                               const double (&val)[K], int g, int c, bool sym) {
#pragma unroll
  for (int r = 0; r < 4; ++r) imgA[(4 * r + g) * LD + c] = x[r];
  wave_sync();
  if (sym) {
    d4 xt;
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[r] = imgA[c * LD + 4 * r + g];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) imgA[(4 * r + g) * LD + c] = 0.5 * (x[r] + xt[r]);
    wave_sync();
  }
  d4 y;
  {
    d4 in[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const unsigned base = lds_addr_of(imgA + g * LD + idx[s]);
      in[s][0] = lds_read64<0>(base);
      in[s][1] = lds_read64<4 * LD * 8>(base);
      in[s][2] = lds_read64<8 * LD * 8>(base);
      in[s][3] = lds_read64<12 * LD * 8>(base);
    }
    if constexpr (K == 1) lds_fence(in[0]);
    else if constexpr (K == 2) lds_fence(in[0], in[1]);
    else if constexpr (K == 3) { lds_fence(in[0], in[1]); lds_fence(in[2]); }
    else { lds_fence(in[0], in[1]); lds_fence(in[2], in[3]); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double acc = in[0][r] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) acc = fma(in[s][r], val[s], acc);
      y[r] = acc;
    }
  }
  return y;
}
template <int K>
__device__ __forceinline__ d4 congruence(const d4& x, double* imgA, double* imgB, const int (&idx)[K],
                                         const double (&val)[K], int g, int c, bool sym = false,
                                         const d4* add = nullptr) {   // add: Z = T X T^T + *add for free
  const d4 y = congruence_pass1<K>(x, imgA, idx, val, g, c, sym);
  return congruence_pass2<K>(y, imgB, idx, val, g, c, add);
}

// second pass: Z[i][c] = sum_s Y[idx_c[s]][i] val_c[s] (+ *add) = (T Y)^T = T X T^T for symmetric results
template <int K>
__device__ __forceinline__ d4 congruence_pass2(const d4& y, double* imgB, const int (&idx)[K], const double (&val)[K],
                                               int g, int c, const d4* add) {
#pragma unroll
  for (int r = 0; r < 4; ++r) imgB[(4 * r + g) * LD + c] = y[r];
  wave_sync();
  d4 z;
  {
    d4 in[K];
#pragma unroll
    for (int s = 0; s < K; ++s) {
      const unsigned base = lds_addr_of(imgB + idx[s] * LD + g);
      in[s][0] = lds_read64<0>(base);
      in[s][1] = lds_read64<4 * 8>(base);
      in[s][2] = lds_read64<8 * 8>(base);
      in[s][3] = lds_read64<12 * 8>(base);
    }
    if constexpr (K == 1) lds_fence(in[0]);
    else if constexpr (K == 2) lds_fence(in[0], in[1]);
    else if constexpr (K == 3) { lds_fence(in[0], in[1]); lds_fence(in[2]); }
    else { lds_fence(in[0], in[1]); lds_fence(in[2], in[3]); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double acc = add ? fma(in[0][r], val[0], (*add)[r]) : in[0][r] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) acc = fma(in[s][r], val[s], acc);
      z[r] = acc;
    }
  }
  return z;
}

// ---------------------------------------------------------------------------------------
// forward pass (no MFMA at all: O(K d^2) per step)
// ---------------------------------------------------------------------------------------
// Wave-cooperative lower Cholesky of the symmetric d x d matrix held row-major in an LDS image
// (img[i * LD + j]); one-off per series (simulation smoother set-up), so simplicity over speed.
__device__ void wave_chol(double* img, int d, int g, int c) {
  for (int k = 0; k < d; ++k) {
    wave_sync();
    double akk = img[k * LD + k];
    akk = akk > 0.0 ? akk : 1e-300;
    const double lkk = sqrt(akk), inv = 1.0 / lkk;
    wave_sync();
    if (g == 0 && c >= k && c < d) img[c * LD + k] = (c == k) ? lkk : img[c * LD + k] * inv;
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + g;
      if (i > k && c > k && c <= i && i < d) img[i * LD + c] = fma(-img[i * LD + k], img[c * LD + k], img[i * LD + c]);
    }
  }
  wave_sync();
}

// SIM = true turns the forward pass into the first half of the Durbin-Koopman (2002) simulation
// smoother: it also simulates (x+, y+) from the model, filters y* = y - y+ from a zero prior mean
// and writes x+ to `xplus` [N][T+1][d].  Normal (record t, component i) of series n is
// philox_normal(seed, series_offset + n, t, i), i = 0..d-1 for the state noise (the initial state at
// record 0), i = d for the observation noise; injected normals are z[N][T+1][d+1].
// IRR: irregular time grid (several G tables, W dt, dt == 0) and/or time-varying F; the regular
// instantiation keeps none of that.
// LL: also accumulate the prediction-error log-likelihood (its own instantiation: the expansion of log() in the loop
// would cost the plain filter two waves per SIMD of occupancy).
// COV: the covariance-only run of the shared-covariance path -- this very code on ONE series of zeros with a zero
// prior mean (the covariance entries of the augmented tile never see the mean: bit for bit the C_t, K_t, Q_t of every
// series of a batch with shared parameters and no missing observation); it also leaves the vector of the mean update,
// R_t F (full step) or K_t (steady step), in kftab [T+1][16] for the mean-only kernel.
template <int K, bool SIM, bool IRR, bool LL, bool COV = false>
__device__ __forceinline__ void filter_body(const KArgs& a, const SparseT* __restrict__ sp, double* __restrict__ side,
                                            double* __restrict__ xplus, double* lds /* 4 WAVE_LDS (+ 4 IMG with SIM) doubles */,
                                            double* __restrict__ kftab = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep it scalar
  const int n = blockIdx.x * (int)(blockDim.x >> 6) + wave;   // 4 waves per block, or 1 for small batches (launch)
  if (n >= a.N) return;
  if (!COV && a.route && (a.route[n] != 0) != (a.route_take != 0)) return;   // shared-covariance call: only the series routed here
  double* imgA = lds + wave * WAVE_LDS;
  double* imgB = imgA + IMG;
  double* vRF = imgB + IMG;      // R F  (one 16-vector is spare)
  double* vX = vRF + 32;         // x+ (SIM)
  double* vZ = vX + 16;          // 64 normals = 4 records x 16 components (SIM)
  double* imgW = lds + 4 * WAVE_LDS + wave * IMG;   // chol(W), row-major (SIM, dense W only)
  const int d = a.d, T = a.T, rec = d + d * d;
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d, col15 = (c == 15);

  const double* W = a.W + (size_t)n * a.w_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  double V = a.V[(size_t)n * a.v_stride];   // V_0; reloaded every step when time-varying (IRR instantiation)
  const double* y = COV ? nullptr : a.y + (size_t)n * T;
  // likelihood-only calls pass no record buffer: a zero-sized resource drops every store
  const bool packed = SIM || a.packed;                // records go to an engine-internal workspace: packed (see below)
  const int recb = COV ? rec * 8 + 128 : (packed ? packed_rec_bytes(d) : rec * 8);   // record stride (COV: a table row is the record followed by the 16 doubles of kftab)
  char* bout = a.filt ? (char*)a.filt + (size_t)n * (T + 1) * recb : nullptr;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(bout, a.filt ? (size_t)(T + 1) * recb : 0);
  double ll = 0.0;   // sum_t log N(y_t; f_t, Q_t) (KalmanFilter.conditionalLikelihood, KalmanFilter.scala:138-153)
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * 2 : nullptr;
  double* sd = side ? side + (size_t)n * (T + 1) * 2 : nullptr;
  const __amdgpu_buffer_rsrc_t rside = make_rsrc(sd, sd ? (size_t)(T + 1) * 16 : 0);   // zero-sized without a side buffer: stores dropped
  const int offS = lane == 0 ? 0 : OOB;
  char* bpri = a.prior ? (char*)(a.prior + (size_t)n * (T + 1) * rec) : nullptr;   // optional (a_t, R_t) records
  const __amdgpu_buffer_rsrc_t rpri = make_rsrc(bpri ? bpri : bout, (size_t)(T + 1) * rec * 8);

  int idx[K];
  double val[K];
  int gcur = 0;    // table currently held in idx/val (rows of G_gcur); irregular grids switch per step
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = sp[0].idx[c][s]; val[s] = sp[0].val[c][s]; }
  // The state rides in the registers as the symmetric augmented tile [[C, m], [m^T, 0]] (row and column 15
  // are free for d <= 15): with row/column 15 of the transition set to the unit vector, the congruence
  // advances mean and covariance together, (R F)[15] is the forecast f = F.a, and the Joseph update of
  // column 15 is the mean update.  A record is stored by the same 4 instructions: lanes c == 15 of register r
  // carry m[4r+g].  Row 15 (lanes g == 3 of register 3, column-indexed mean) is never stored.
  d4 w, cc;
  double Fr[4];
  bool vr[4], va[4];
  int offA[4];                                   // byte offset of this lane's element of register r inside a record
  // packed (SIM, or a fused filter + smoother call that does not want the filtered records): the records only feed the
  // backward kernel, so the symmetric C_t is stored as [m (d) | lower triangle by rows, d (d + 1) / 2] -- 832 B
  // instead of 1456 B per step at d = 13; these kernels are bound by exactly this stream.
  double Fc = vc ? a.F[c] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    vr[r] = i < d;
    const bool ok = vr[r] && vc;
    va[r] = vr[r] && (vc || col15);
    offA[r] = packed ? ((ok && c <= i) ? (d + i * (i + 1) / 2 + c) * 8 : (vr[r] && col15 ? i * 8 : OOB))
                     : (ok ? (d + i * d + c) * 8 : (vr[r] && col15 ? i * 8 : OOB));
    w[r] = ok ? W[i * d + c] : 0.0;
    cc[r] = ok ? C0[i * d + c] : 0.0;
    if (!SIM && !COV) {                          // SIM: y* is filtered from a zero prior mean
      if (vr[r] && col15) cc[r] = m0[i];
      if (i == 15 && vc) cc[r] = m0[c];
    }
    Fr[r] = vr[r] ? a.F[i] : 0.0;
  }
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;

  // ---- simulation-smoother set-up: factors of C0 and W, x+_0, zero prior mean -------------
  double xcol = 0.0, wsd = 0.0;
  bool wdiag = true;
  const double sqV = sqrt(V);
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  const double* zin = (SIM && a.z) ? a.z + (size_t)n * (T + 1) * (d + 1) : nullptr;
  double* xp = SIM ? xplus + (size_t)n * (T + 1) * d : nullptr;
  if (SIM) {
    bool offd = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) offd |= (4 * r + g != c) && (w[r] != 0.0 || cc[r] != 0.0);
    wdiag = (__ballot(offd) == 0ull) && !(IRR && a.w_tstride);   // W and C0 both diagonal: no factorisation (a W_t stream is factored at every step)
    const double z0 = (c < d) ? (zin ? zin[c] : philox_normal(a.seed, series, 0u, (unsigned)c)) : 0.0;
    if (wdiag) {
      wsd = vc ? sqrt(W[c * d + c]) : 0.0;
      xcol = vc ? fma(sqrt(C0[c * d + c]), z0, m0[c]) : 0.0;
    } else {
      vZ[c] = z0;
#pragma unroll
      for (int r = 0; r < 4; ++r) { imgW[(4 * r + g) * LD + c] = w[r]; imgB[(4 * r + g) * LD + c] = cc[r]; }
      wave_chol(imgW, d, g, c);
      wave_chol(imgB, d, g, c);
      double x0 = vc ? m0[c] : 0.0;
      for (int k = 0; k <= c && k < d; ++k) x0 = fma(imgB[c * LD + k], vZ[k], x0);
      xcol = vc ? x0 : 0.0;
      wave_sync();
    }
    if (g == 0 && vc) xp[c] = xcol;
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) buf_store(rout, bout, offA[r], 0, cc[r]);
  if (bpri) {   // record 0: a = m0, R = C0 (KalmanFilter.scala:117)
#pragma unroll
    for (int r = 0; r < 4; ++r) buf_store(rpri, bpri, offA[r], 0, cc[r]);
  }
  if (lane == 0) {
    if (fq) { fq[0] = __builtin_nan(""); fq[1] = __builtin_nan(""); }
    if (sd) { sd[0] = __builtin_nan(""); sd[1] = __builtin_nan(""); }
  }

  // Steady state (regular grid, time-invariant model: the !IRR instantiation).  The covariance recursion of a
  // time-invariant filter converges: once C is within DLM_SETTLE_TOL max|C| of its limit (settle_test, dlm_internal.h: the
  // geometric tail of the one-step changes, tested every fourth step -- NOT a one-step change alone, which says nothing when
  // the recursion contracts slowly), C_t, R_t, K_t and Q_t are those of the step before and only the mean moves -- a = G m, e = y - F.a,
  // m = a + K e: a gather on one 16-vector instead of the congruence, the products with F and the rank-one update.  The
  // record (the same C, the new m) is stored as always.  A missing observation takes the full step again (C changes),
  // and the test starts over.  The backward pass learns from the sign of the side record's 1/Q that C_t is C_{t-1}.
  const bool may_settle = !IRR && !(a.flags & DLM_OPT_NO_STEADY) && !(a.plain && a.plain[n]);   // (the simulation smoother's pass too: y* has the covariance recursion of y)
  bool steady = false;
  float* settle = (float*)(imgA + 2 * IMG + 8 * 16);   // state of the steady-state test (dlm_internal.h)
  settle_reset(settle);
  unsigned nsteady = 0;   // steady steps taken (KArgs::counters[0])
  double Kst = 0.0, rq_st = 0.0, Q_st = 0.0, lq_st = 0.0;
  double ychunk = 0.0;
  for (int t = 0; t < T; ++t) {
    if (!COV && (t & 63) == 0) {
      ychunk = (t + lane < T) ? y[t + lane] : 0.0;
      // consume the load inside the branch: otherwise the wait for it lands on the common path as
      // vmcnt(0), which every step would also wait for the record stores of the step before
      asm volatile("" ::"v"(ychunk));
    }
    double yt = uniform_from_lane(ychunk, t & 63);
    const int gi = (IRR && a.g_index) ? a.g_index[t] : 0;   // uniform: scalar loads
    const double dt = (IRR && a.dt) ? a.dt[t] : 1.0;
    if (IRR && gi != gcur) {
#pragma unroll
      for (int s = 0; s < K; ++s) { idx[s] = sp[2 * gi].idx[c][s]; val[s] = sp[2 * gi].val[c][s]; }
      gcur = gi;
    }
    if (IRR && a.f_stride) {                                 // time-varying F (Dlm.regression): F_t = f(time_t)
      const double* Ft = a.F + (size_t)t * a.f_stride;
      Fc = vc ? Ft[c] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) Fr[r] = vr[r] ? Ft[4 * r + g] : 0.0;
    }
    if (IRR && a.v_tstride) {                                // time-varying V_t (StudentTGibbs.scala:100-136)
      V = a.V[(size_t)n * a.v_stride + (size_t)t * a.v_tstride];
      if (!(V > 0.0)) st |= DLM_ST_NOT_PD;
    }
    if (IRR && a.w_tstride) {                                // time-varying W_t (DlmFsvSystem.scala:137-208)
      const double* Wt = W + (size_t)t * a.w_tstride;
#pragma unroll
      for (int r = 0; r < 4; ++r) w[r] = (vr[r] && vc) ? Wt[(4 * r + g) * d + c] : 0.0;
    }

    if (may_settle && steady && yt == yt) {
      ++nsteady;
      // the mean rides in row 15 of the tile (lanes g == 3 of register 3) and in column 15
      if (g == 3) vRF[c] = cc[3];
      if (SIM) {   // x+ and the normals of the next four records, as in the full step
        vX[c] = xcol;
        if ((t & 3) == 0) {
          const int tr = t + 1 + g;
          double zz = 0.0;
          if (c <= d && tr <= T) zz = zin ? zin[(size_t)tr * (d + 1) + c] : philox_normal(a.seed, series, (unsigned)tr, (unsigned)c);
          vZ[lane] = zz;
        }
      }
      wave_sync();
      if (SIM) {   // x+_t = G x+_{t-1} + L_W z ;  y*_t = y_t - F^T x+_t - sqrt(V) z_v   (regular grid: dt = 1)
        const double* zr = vZ + 16 * (t & 3);
        double xg = vX[idx[0]] * val[0];
#pragma unroll
        for (int s = 1; s < K; ++s) xg = fma(vX[idx[s]], val[s], xg);
        double wl;
        if (wdiag) wl = wsd * zr[c];
        else { wl = 0.0; for (int k = 0; k <= c && k < d; ++k) wl = fma(imgW[c * LD + k], zr[k], wl); }
        xcol = vc ? wl + xg : 0.0;
        yt = yt - fma(sqV, zr[d], row_sum(Fc * xcol));
        if (g == 0 && vc) xp[(size_t)(t + 1) * d + c] = xcol;
      }
      double ac = vRF[idx[0]] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) ac = fma(vRF[idx[s]], val[s], ac);      // a = G m (row 15 of the table: the unit row)
      const double f = uniform_from_lane(row_sum(Fc * ac), 0);
      const double e = yt - f, erq = e * rq_st;
      const double mn = fma(Kst, e, ac);                                   // m = a + K e
      if (COV && g == 0) ((double*)((char*)kftab + (size_t)(t + 1) * recb))[c] = vc ? Kst : (c == 15 ? -rq_st : 0.0);   // [15]: 1 / Q negated = "steady step"

      wave_sync();                                                         // the reads of vRF above precede its rewrite
      vRF[c] = mn;
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) { const double mr = vRF[4 * r + g]; cc[r] = col15 ? mr : cc[r]; }
      cc[3] = (g == 3 && !col15) ? mn : cc[3];
      if (LL) ll -= 0.5 * (lq_st + e * erq);                                 // log(2 pi) + log Q of the settled forecast variance: computed once
      side_store(rside, offS, (t + 1) * 16, erq, -rq_st);                  // 1/Q negated: "C_t is C_{t-1}" for the backward pass
      if (fq && lane == 0) { fq[2 * (t + 1)] = f; fq[2 * (t + 1) + 1] = Q_st; }
      const int so = (t + 1) * recb;
#pragma unroll
      for (int r = 0; r < 4; ++r) buf_store(rout, bout, offA[r], so, cc[r]);
      wave_sync();
      if (COV && a.settle_step) {   // the covariance-only run: every later row repeats this one (the series is all zeros) -- k_cov_fill_sp16 copies it
        if (lane == 0) *a.settle_step = t + 1;
        break;
      }
      continue;
    }
    // advState: a = G m, R = G C G^T + W dt   (dt == 0: a = m, R = C, KalmanFilter.scala:279-280)
    if (SIM) {
      vX[c] = xcol;
      if ((t & 3) == 0) {   // 64 normals: records t+1 .. t+4, components 0..15 (0..d used)
        const int tr = t + 1 + g;
        double zz = 0.0;
        if (c <= d && tr <= T) zz = zin ? zin[(size_t)tr * (d + 1) + c] : philox_normal(a.seed, series, (unsigned)tr, (unsigned)c);
        vZ[lane] = zz;
      }
    }
    d4 R;
    if (dt == 0.0) {
      wave_sync();
      R = cc;
    } else {
      d4 wdt = w;
      if (IRR) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wdt[r] = w[r] * dt;
      }
      R = congruence<K>(cc, imgA, imgB, idx, val, g, c, false, &wdt);   // first wave_sync also covers vX, vZ
    }
    if (SIM) {
      // x+_t = G x+_{t-1} + L_W z ;  y+_t = F^T x+_t + sqrt(V) z_v ;  y*_t = y_t - y+_t
      const double* zr = vZ + 16 * (t & 3);
      if (IRR && a.w_tstride && dt != 0.0) {   // W_t of this step (DlmFsvSystem.scala:137-208): its Cholesky factor, 13 pivots in LDS
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) imgW[(4 * r + g) * LD + c] = w[r];
        wave_chol(imgW, d, g, c);
      }
      if (dt != 0.0) {
        double xg = vX[idx[0]] * val[0];
#pragma unroll
        for (int s = 1; s < K; ++s) xg = fma(vX[idx[s]], val[s], xg);
        double wl;
        if (wdiag) wl = wsd * zr[c];
        else { wl = 0.0; for (int k = 0; k <= c && k < d; ++k) wl = fma(imgW[c * LD + k], zr[k], wl); }
        xcol = vc ? fma(wl, sqrt(dt), xg) : 0.0;
      }
      const double yplus = fma((IRR && a.v_tstride) ? sqrt(V) : sqV, zr[d], row_sum(Fc * xcol));   // V_t of this step when time-varying
      yt = yt - yplus;                                       // NaN (missing) stays NaN
      if (g == 0 && vc) xp[(size_t)(t + 1) * d + c] = xcol;
    }
    if (bpri) {
#pragma unroll
      for (int r = 0; r < 4; ++r) buf_store(rpri, bpri, offA[r], (t + 1) * recb, R[r]);
    }
    // RF (element 15 is the forecast f = F^T a) ; Q = F^T R F + V
    double rfc = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) rfc = fma(R[r], Fr[r], rfc);
    rfc = sum_g(rfc);                                        // (R F)[c] in every lane
    vRF[c] = rfc;
    wave_sync();
    double rfr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rfr[r] = vRF[4 * r + g];     // (R F)[4r+g]
    const double f = uniform_from_lane(rfc, 15);
    const double Q = uniform_from_lane(row_sum(Fc * rfc), 0) + V;

    if (yt == yt) {
      // Joseph form for p = 1 with K = RF / Q:  R - K RF^T - RF K^T + Q K K^T
      //   = R - RF_i * (RF_j / Q) * (2 - Q * (1/Q))   -- the same expression, factored
      //   column 15: a + RF e / Q = m ;  row 15 (lanes g == 3 of register 3): a[c] + K[c] e
      const double e = yt - f, rq = fast_rcp(Q), erq = e * rq;
      if (COV && g == 0) ((double*)((char*)kftab + (size_t)(t + 1) * recb))[c] = vc ? rfc : (c == 15 ? rq : 0.0);   // R_t F (the mean-only kernel's m = a + (R F)(e / Q)); [15]: 1 / Q
      const double Kc = (col15 ? 0.0 : rfc) * rq;
      const double ngam = col15 ? erq : -Kc * fma(-Q, rq, 2.0);
      const bool check = may_settle && bpri == nullptr && (t & 3) == 3;
      d4 old = cc;
#pragma unroll
      for (int r = 0; r < 3; ++r) cc[r] = fma(rfr[r], ngam, R[r]);
      cc[3] = (g == 3) ? fma(Kc, e, R[3]) : fma(rfr[3], ngam, R[3]);
      if (check) {   // has the covariance reached its limit?  (column 15 and row 15 carry the mean: not compared)
        const double sc = settle_pow2_inverse_of(Q);        // single precision after scaling by a power of two near 1 / Q (C is of the order of Q at most)
        float dl = 0.f, mx = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (vc && !(r == 3 && g == 3)) { dl = fmaxf(dl, (float)(settle_absdiff(cc[r], old[r]) * sc)); mx = fmaxf(mx, (float)(fabs(cc[r]) * sc)); }
        wave_max2f(dl, mx);
        if (settle_test(settle, dl, mx, 4)) { steady = true; Kst = Kc; rq_st = rq; Q_st = Q; if (LL) lq_st = 1.8378770664093453 + log(Q); }
      }
      if (LL) ll -= 0.5 * (1.8378770664093453 + log(Q) + e * erq);   // -log N(y; f, Q); log(2 pi) = 1.83787...
      side_store(rside, offS, (t + 1) * 16, erq, rq);
    } else {
      cc = R;
      steady = false;
      settle_reset(settle);
      side_store(rside, offS, (t + 1) * 16, __builtin_nan(""), __builtin_nan(""));
    }
    if (fq && lane == 0) { fq[2 * (t + 1)] = f; fq[2 * (t + 1) + 1] = Q; }
    const int so = (t + 1) * recb;
#pragma unroll
    for (int r = 0; r < 4; ++r) buf_store(rout, bout, offA[r], so, cc[r]);
    wave_sync();   // the images are rewritten at the top of the next step
  }
  if (LL && a.loglik && lane == 0) a.loglik[n] = ll;
  if (!COV && a.counters && lane == 0) {
    if (nsteady) atomicAdd(&a.counters[0], (unsigned long long)nsteady);
    if (a.route) atomicAdd(&a.counters[3], 1ull);          // a series of a shared-covariance call that ran its own recursion
  }
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) bad |= va[r] && !isfinite(cc[r]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

template <int K, bool SIM, bool IRR, bool LL = false>
__global__ __launch_bounds__(256, SIM ? 4 : FI_WAVES) void k_filter_sp16(KArgs a, const SparseT* __restrict__ sp,
                                                     double* __restrict__ side, double* __restrict__ xplus) {
  __shared__ __attribute__((aligned(16))) double lds[4 * WAVE_LDS + (SIM ? 4 * IMG : 0)];
  filter_body<K, SIM, IRR, LL>(a, sp, side, xplus, lds);
}
// the covariance-only run: one wave (a: N = 1, filt = the C_t table)
template <int K>
__global__ __launch_bounds__(64) void k_cov_filter_sp16(KArgs a, const SparseT* __restrict__ sp, double* __restrict__ side,
                                                        double* __restrict__ kftab, const int* __restrict__ skip) {
  if (skip && *skip) return;   // (the call found that it does not want the tables)
  __shared__ __attribute__((aligned(16))) double lds[4 * WAVE_LDS];
  __builtin_amdgcn_s_setprio(3);   // one wave that every mean kernel of the call waits for, possibly beside a kernel that fills the device
  filter_body<K, false, false, false, true>(a, sp, side, nullptr, lds, kftab);
}

// Rows settle[0] + 1 .. T of the forward table (and of its side records) are row settle[0]: the covariance-only wave stops at the first
// steady row it wrote -- 0.42 -> 0.18 ms of strictly sequential work in front of every kernel that reads the table -- and this
// kernel copies it, one workgroup per row (rowd doubles of table, two of side records).
__global__ __launch_bounds__(64) void k_cov_fill_sp16(double* __restrict__ ftab, int rowd, double* __restrict__ cside, const int* __restrict__ settle, int T) {
  const int t = blockIdx.x, ts = settle[0];
  if (t <= ts || t > T) return;
  const double* src = ftab + (size_t)ts * rowd;
  double* dst = ftab + (size_t)t * rowd;
  for (int i = threadIdx.x; i < rowd; i += 64) dst[i] = src[i];
  if (threadIdx.x < 2) cside[2 * (size_t)t + threadIdx.x] = cside[2 * (size_t)ts + threadIdx.x];
}

// ---------------------------------------------------------------------------------------
// backward pass: MFMA for P C and C (P C); gathers for G^T M G
//
// The record rides in the registers AUGMENTED: lanes c < d of register r hold C[4r+g][c], lanes c == 15
// hold m[4r+g] (column 15 of a 16 x 16 tile is free for d <= 15).  With -q in column 15 of the second
// MFMA's B operand,  [C | m] - C [P C | -q] = [C - C P C | m + C q] = [S | s]:  the smoothed mean needs no
// extraction from the product, and mean and covariance are read and stored by the same 4 instructions.
// ---------------------------------------------------------------------------------------
constexpr int SM_LDS = 2 * IMG + 3 * 16 + 2;   // backward pass: two images + three 16-vectors + the state of the steady-state test per wave
// COV: the covariance-only run of the shared-covariance path -- this code on the C_t table (means zero): S_t, bit for bit that of
// every series of the batch; K_t = C_t F / V and the kind of each step (steady: 1) go to kbtab [T+1][16] for the mean-only kernel.
template <int K, bool IRR, bool PIPE = false, bool COV = false, bool PLAIN = false>
__device__ __forceinline__ void smoother_body(const KArgs& a, const SparseT* __restrict__ sp, const double* __restrict__ side,
                                              double* lds /* 4 SM_LDS doubles */, char* ring_all, double* __restrict__ kbtab = nullptr) {
  // ring_all: two-slot ring per wave for the LDS-DMA prefetch; a slot is a raw record followed by one
  // zero double, which the padded lanes read.  Record t lives in slot t & 1 and is requested two steps ahead.
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: keep it scalar
  const int n = blockIdx.x * (int)(blockDim.x >> 6) + wave;   // 4 waves per block, or 1 for small batches (launch)
  if (n >= a.N) return;
  if (!COV && a.route && (a.route[n] != 0) != (a.route_take != 0)) return;   // shared-covariance call: only the series routed here
  // PLAIN (DLM_OPT_NO_STEADY on a regular grid): no steady-state machinery at all and F, 1 / V in registers -- the kernel of round 1, for the
  // calls whose every step is a full step (what the machinery costs them: profiles/r03_notes.md section 3)
  constexpr bool ST = !IRR && !PLAIN;      // steady-state steps possible
  constexpr bool FREG = IRR || PLAIN;      // F[4r+g], F[c] and 1 / V live in registers (otherwise in spare LDS columns)
  double* imgA = lds + wave * SM_LDS;
  double* imgB = imgA + IMG;
  double* vK = imgB + IMG;       // K_t
  double* vQ = vK + 16;          // -q_t
  double* vR = vQ + 16;          // r
  const int d = a.d, T = a.T, rec = d + d * d;
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d, col15 = (c == 15);

  const double V = a.V[(size_t)n * a.v_stride];
  double rV = 1.0 / V;   // regular instantiation: parked in LDS below (a uniform value the compiler keeps in a vector register).  1 / V_t of the record's observation when time-varying (IRR instantiation)
  const bool pout = (a.packed & 2) != 0;                          // smoothed records (output): packed with DLM_OPT_PACKED_SYM, dense otherwise
  const int recb = pout ? packed_rec_bytes(d) : rec * 8;
  const int rinb = (a.packed & 1) ? packed_rec_bytes(d) : rec * 8; // filtered records (input): packed when engine-internal or DLM_OPT_PACKED_SYM
  // COV: the input is the forward table (rows of rinb + 128 bytes), the output the backward table, whose rows are
  // [S_t record | kbtab (16 doubles) | C_t record]
  const int rins = COV ? rinb + 128 : rinb, recs = COV ? 2 * recb + 128 : recb;   // strides
  const char* bin = (const char*)a.filt_in + (size_t)n * (T + 1) * rins;
  char* bout = (char*)a.smooth + (size_t)n * (T + 1) * recs;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(bout, (size_t)(T + 1) * recs);
  const double* sd = side + (size_t)n * (T + 1) * 2;

  int idx[K];
  double val[K];
  int gcur = 0;    // idx/val hold the COLUMNS of G_gcur (the transition into the current record)
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = sp[1].idx[c][s]; val[s] = sp[1].val[c][s]; }
  double Fr[4];
  bool va[4];      // this lane's register r carries an element of the augmented record
  int offA[4];     // its byte offset inside a record (OOB otherwise: loads give 0, stores are dropped)
  int ldsA[4];     // the same inside a ring slot (padded lanes read the slot's zero double)
  double Fcr = vc ? a.F[c] : 0.0;   // F[c] (regular instantiation: read from column 16 of image A where needed)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    Fr[r] = i < d ? a.F[i] : 0.0;
    va[r] = i < d && (vc || col15);
    const int offD = i < d ? (vc ? (d + i * d + c) * 8 : (col15 ? i * 8 : OOB)) : OOB;   // dense record
    offA[r] = !pout ? offD : (i < d ? ((vc && c <= i) ? (d + i * (i + 1) / 2 + c) * 8 : (col15 ? i * 8 : OOB)) : OOB);
    if (a.packed & 1) { const int hi = i > c ? i : c, lo = i > c ? c : i; ldsA[r] = va[r] ? (vc ? (d + hi * (hi + 1) / 2 + lo) * 8 : i * 8) : rinb; }
    else ldsA[r] = va[r] ? offD : rinb;
  }
  // Regular instantiation: F[4r+g] waits in the spare column 16 of image A (the images have a leading dimension of 17) and is
  // read where a step needs it -- eight registers that decide whether five waves fit a SIMD.
  if (!FREG) {
#pragma unroll
    for (int r = 0; r < 4; ++r) imgA[(4 * r + g) * LD + 16] = Fr[r];
    imgB[16] = rV;
  }
  const unsigned fr_lds = lds_addr_of(imgA + g * LD + 16);
  d4 P = {0.0, 0.0, 0.0, 0.0};
  double qcol = 0.0;
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;
  const bool may_settle = !(a.flags & DLM_OPT_NO_STEADY);
  bool psteady = false, same_next = false, was_steady = false, chk = false;
  float chk_dl = 0.f, chk_mx = 0.f;
  float* settle = (float*)(imgA + 2 * IMG + 3 * 16);   // state of the steady-state test (dlm_internal.h)
  settle_reset(settle);
  unsigned nsteady = 0;   // steady steps taken (KArgs::counters[1])

  const int slotb = rinb + 16;
  char* ring = ring_all + wave * 2 * slotb;
  const unsigned ring_lds = lds_addr_of(ring);
  const i4 rdma = rsrc_words(bin, (unsigned)((size_t)(T + 1) * rins));
  const int n16 = rinb / 16;                          // d (d + 1) is even (and packed records are padded): whole 16 B pieces
  const int n16m = (d * 8 + 15) / 16;                 // the mean alone (it leads the record in both layouts)
  if (lane < 2) *(double*)(ring + lane * slotb + rinb) = 0.0;
  // The side records run two steps ahead of the recursion: the forward pass's mark on record t-1 (1/Q negated: C_{t-1} is
  // C_{t-2}) decides at step t how much of record t-2 is requested -- the whole record, or only its mean when the covariance is
  // the one already in the registers (64 % of the records of the C2 bench: 104 instead of 1456 bytes read).
  double ceq = sd[2 * T], ciq = sd[2 * T + 1];
  double neq, niq;
  { const int t1 = T > 0 ? T - 1 : 0; neq = sd[2 * t1]; niq = sd[2 * t1 + 1]; }
  dma_record(rdma, ring_lds + (T & 1) * slotb, T * rins, lane, n16);
  { const int t1 = T > 0 ? T - 1 : 0;
    dma_record(rdma, ring_lds + ((T - 1) & 1) * slotb, t1 * rins, lane, (ST && uniform_from_lane(ciq, 0) < 0.0) ? n16m : n16); }
  vQ[c] = 0.0;
  d4 out = {0.0, 0.0, 0.0, 0.0};                     // the record stored last (assigned in every step before its store)
  d4 cc = {0.0, 0.0, 0.0, 0.0};

#ifdef DLM_STAMP
  unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = stamp();
#endif
  for (int t = T; t >= 0; --t) {
    STAMP(7)
    // Operations issued after the request for record t: the 4 stores of step t+2, the request for t-1
    // (1 or 2 instructions), the 4 stores of step t+1.  Counting the requests as one instruction only
    // over-waits by one long-finished store.  The first two steps have fewer operations behind them.
    if (t == T) vm_wait<1>();
    else if (t == T - 1) vm_wait<5>();
    else vm_wait<9>();
    // C_t is C_{t+1} (the mark on record t+1): only the mean of record t was requested, the covariance stays in the registers
    const bool inherit = ST && same_next;
    {
      const unsigned slot = ring_lds + (t & 1) * slotb;
      d4 nr;                                                 // the record as fetched: [C_t | m_t], or only m_t in column 15
      nr[0] = lds_read64<0>(slot + ldsA[0]);
      nr[1] = lds_read64<0>(slot + ldsA[1]);
      nr[2] = lds_read64<0>(slot + ldsA[2]);
      nr[3] = lds_read64<0>(slot + ldsA[3]);
      lds_fence(nr);
#pragma unroll
      for (int r = 0; r < 4; ++r) cc[r] = (inherit && !col15) ? cc[r] : nr[r];   // (holding nr for the steady step instead costs 8 VGPRs: 4 waves per SIMD)
    }
    // the innovations are per-series scalars: keep them in SGPRs so `observed` is a scalar branch
    const double eq = uniform_from_lane(ceq, 0), iqraw = uniform_from_lane(ciq, 0);
    const double iq = fabs(iqraw);                           // the forward pass negates 1/Q where C_t is C_{t-1} (its steady state)
    const bool same_c = iqraw < 0.0;
    const bool mean_only = ST && uniform_from_lane(niq, 0) < 0.0;   // C_{t-1} is C_{t-2}: record t-2 needs only its mean
    ceq = neq; ciq = niq;
    {
      const int tp = t > 1 ? t - 2 : 0;                      // record 0 is re-read harmlessly at the end
      neq = sd[2 * tp]; niq = sd[2 * tp + 1];
    }
    const bool observed = (iq == iq) && t > 0;
    if (IRR && a.f_stride && t > 0) {                        // F of the observation at record t
      const double* Ft = a.F + (size_t)(t - 1) * a.f_stride;
      Fcr = vc ? Ft[c] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) Fr[r] = (4 * r + g < d) ? Ft[4 * r + g] : 0.0;
    }
    if (IRR && a.v_tstride && t > 0) {                       // V of the observation at record t
      const double Vt = a.V[(size_t)n * a.v_stride + (size_t)(t - 1) * a.v_tstride];
      if (!(Vt > 0.0)) st |= DLM_ST_NOT_PD;
      rV = 1.0 / Vt;
    }

    // Steady state of the whole backward step: C_t = C_{t+1} (the forward pass's mark) and P_t = P_{t+1} (found below) give
    // P_{t-1} = P_t and S_t = S_{t+1}.  What is left of the step is the mean: s_t = m_t + C_t q_t (four FMAs per lane on the
    // symmetric C, one cross-row sum, a transposition through LDS), q_{t-1} = G^T [q_t + F (e_t/Q_t - K.q_t)] -- no MFMA, no
    // rank-two update, no congruence -- and the record store (the covariance registers of the step before, the new mean).
    if (ST && psteady && same_next && observed) {
      ++nsteady;
      was_steady = true;
      if (COV && g == 0) ((double*)(bout + (size_t)t * recs + recb))[c] = col15 ? 1.0 : vK[c];   // K_t (that of the step before) and the mark "steady step"
      { const int t2 = t > 1 ? t - 2 : 0; dma_record(rdma, ring_lds + (t & 1) * slotb, t2 * rins, lane, mean_only ? n16m : n16); }
      d4 nqr;
      {
        const unsigned bq = lds_addr_of(vQ + g);             // -q_t, published by the last step's closing wave_sync
        nqr[0] = lds_read64<0>(bq); nqr[1] = lds_read64<32>(bq); nqr[2] = lds_read64<64>(bq); nqr[3] = lds_read64<96>(bq);
        lds_fence(nqr);
      }
      double ncq = 0.0;                                      // -(C q)[c]: C symmetric, so the column sum serves (lanes c == 15 sum the mean: unused)
#pragma unroll
      for (int r = 0; r < 4; ++r) ncq = fma(cc[r], nqr[r], ncq);
      ncq = sum_g(ncq);
      const double kq = uniform_from_lane(row_sum(vK[c] * qcol), 0);   // K_t is K_{t+1}: still in vK
      vR[c] = fma(FREG ? Fcr : imgA[c * LD + 16], eq - kq, qcol);
      imgB[c * LD + 15] = ncq;                               // column 15 of the parked S_t: one read per register fetches [S | -C q]
      wave_sync();
      qcol = vR[idx[0]] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) qcol = fma(vR[idx[s]], val[s], qcol);
      d4 ns;
      {
        const unsigned bs = lds_addr_of(imgB + g * LD + c);
        ns[0] = lds_read64<0>(bs); ns[1] = lds_read64<4 * LD * 8>(bs); ns[2] = lds_read64<8 * LD * 8>(bs); ns[3] = lds_read64<12 * LD * 8>(bs);
        lds_fence(ns);
      }
      vQ[c] = -qcol;
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) out[r] = col15 ? cc[r] - ns[r] : ns[r];
    } else {

    // K_t = C_t F / V  (column 15 would give F.m: masked); it is K_{t+1} when the covariance was inherited
    if (!inherit) {
      d4 fr;
      if (FREG) { fr[0] = Fr[0]; fr[1] = Fr[1]; fr[2] = Fr[2]; fr[3] = Fr[3]; }
      else {
        fr[0] = lds_read64<0>(fr_lds); fr[1] = lds_read64<4 * LD * 8>(fr_lds); fr[2] = lds_read64<8 * LD * 8>(fr_lds); fr[3] = lds_read64<12 * LD * 8>(fr_lds);
        lds_fence(fr);
      }
      double ks = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) ks = fma(cc[r], fr[r], ks);
      vK[c] = (observed && vc) ? sum_g(ks) * (FREG ? rV : imgB[16]) : 0.0;
    }
    wave_sync();                                             // also publishes vQ of the last step
    if (COV && g == 0) ((double*)(bout + (size_t)t * recs + recb))[c] = col15 ? 0.0 : vK[c];
    // the slot just read is free again: request record t-2 into it (always issued, so that the operation
    // count behind every request is the same; below record 0 it re-reads record 0, which nobody uses)
    { const int t2 = t > 1 ? t - 2 : 0; dma_record(rdma, ring_lds + (t & 1) * slotb, t2 * rins, lane, mean_only ? n16m : n16); }
    d4 kr, nqr;
    {
      const unsigned bk = lds_addr_of(vK + g), bq = lds_addr_of(vQ + g);
      kr[0] = lds_read64<0>(bk); kr[1] = lds_read64<32>(bk); kr[2] = lds_read64<64>(bk); kr[3] = lds_read64<96>(bk);
      nqr[0] = lds_read64<0>(bq); nqr[1] = lds_read64<32>(bq); nqr[2] = lds_read64<64>(bq); nqr[3] = lds_read64<96>(bq);
      lds_fence(kr, nqr);
    }
    d4 b1;
#pragma unroll
    for (int r = 0; r < 4; ++r) b1[r] = col15 ? kr[r] : cc[r];
    STAMP(0)
    const d4 x1 = mmT(P, b1);                                // [P C | P K]  (two chains of two for small batches: measured slower)
    d4 b2;
#pragma unroll
    for (int r = 0; r < 4; ++r) b2[r] = col15 ? nqr[r] : x1[r];
    // C [P C | -q]: only the OUTPUT (S_t, s_t) needs it, so it is consumed at the very end of the
    // step and its MFMA latency hides behind the recursion work below
#ifdef DLM_STAMP
    asm volatile("" ::"v"(b2[0]), "v"(b2[1]), "v"(b2[2]), "v"(b2[3]));
#endif
    STAMP(1)
    // A operand = [C | m]^T: row 15 of the product is never stored.  PIPE (batches that leave a wave alone on its SIMD): the four
    // MFMAs are spread over the recursion below -- a wave issues in order, and a dependent fp64 MFMA holds it for ~200 cycles.
    d4 x2;
#define X2_STEP(k) { x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cc[k], b2[k], x2, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    if constexpr (PIPE) { x2[0] = 0.0; x2[1] = 0.0; x2[2] = 0.0; x2[3] = 0.0; __builtin_amdgcn_sched_barrier(0); X2_STEP(0) } else x2 = mmT(cc, b2);
    STAMP(2)

    if (t > 0) {
      // (q_{t-1}, P_{t-1}) from (q_t, P_t).  Column 15 of x1 is P K: park x1 in the idle image
      // and read that column back in both indexings.
      const int gi = (IRR && a.g_index) ? a.g_index[t - 1] : 0;   // G of the step into record t
      if (IRR && gi != gcur) {
#pragma unroll
        for (int s = 0; s < K; ++s) { idx[s] = sp[2 * gi + 1].idx[c][s]; val[s] = sp[2 * gi + 1].val[c][s]; }
        gcur = gi;
      }
      {
      psteady = false;
      d4 M = P;
      double rcol = qcol;
      if constexpr (PIPE) X2_STEP(1)
      if (observed) {
#pragma unroll
        for (int r = 0; r < 4; ++r) imgA[(4 * r + g) * LD + c] = x1[r];
        wave_sync();
        d4 pk;                                               // (P K)[4r+g]
        double pkc;                                          // (P K)[c]
        d4 fr;                                               // F[4r+g]
        {
          const unsigned br = lds_addr_of(imgA + g * LD + 15), bc = lds_addr_of(imgA + c * LD + 15);
          pk[0] = lds_read64<0>(br); pk[1] = lds_read64<4 * LD * 8>(br);
          pk[2] = lds_read64<8 * LD * 8>(br); pk[3] = lds_read64<12 * LD * 8>(br);
          pkc = lds_read64<0>(bc);
          if (FREG) { fr[0] = Fr[0]; fr[1] = Fr[1]; fr[2] = Fr[2]; fr[3] = Fr[3]; }
          else { fr[0] = lds_read64<8>(br); fr[1] = lds_read64<4 * LD * 8 + 8>(br); fr[2] = lds_read64<8 * LD * 8 + 8>(br); fr[3] = lds_read64<12 * LD * 8 + 8>(br); }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pk), "+v"(pkc), "+v"(fr)::"memory");
        }
        const double kk = row_sum(vK[c] * (g < 2 ? qcol : pkc));  // rows 0-1: K.q, rows 2-3: K.(P K)
        const double kq = uniform_from_lane(kk, 0), kpk = uniform_from_lane(kk, 32);
        const double sc = iq + kpk;
        const double Fc = FREG ? Fcr : imgA[c * LD + 16];
        rcol = fma(Fc, eq - kq, qcol);
        const double u = fma(Fc, sc, -pkc);                  // F_i F_c (1/Q + K'PK) - F_i (PK)_c - (PK)_i F_c = F_i u_c - (PK)_i F_c
#pragma unroll
        for (int r = 0; r < 4; ++r) M[r] = fma(-pk[r], Fc, fma(fr[r], u, P[r]));
      }
      if constexpr (PIPE) X2_STEP(2)
      vR[c] = rcol;
      wave_sync();                                           // column-15 reads precede the image rewrite
#ifdef DLM_STAMP
      asm volatile("" ::"v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]));
#endif
      STAMP(3)
      // In this kernel both cross terms use the same vector (P^T K), so the antisymmetric rounding part of
      // P is only rotated by G from step to step (polynomial growth at worst for unit-root G), never
      // amplified by the update: removing it every 8th step keeps it at rounding level.
      // Like Smoothing.smoothStep (Smoothing.scala:41) this always uses the table entry g(dt), also for
      // dt == 0 where the filter made an identity advance; only polynomial-type g ignore dt, see DESIGN.md.
      d4 Pn;                                                                       // G^T M G (first sync covers vR)
      if constexpr (PIPE) {
        const d4 yc = congruence_pass1<K>(M, imgA, idx, val, g, c, (t & 7) == 0);
        X2_STEP(3)
        Pn = congruence_pass2<K>(yc, imgB, idx, val, g, c, nullptr);
      } else Pn = congruence<K>(M, imgA, imgB, idx, val, g, c, (t & 7) == 0);
      // has P reached its limit (and will the next C be this one)?  The same geometric-tail test as in the forward pass,
      // relative to max|P|; any step off the settled, observed stretch starts it over.
      if (ST && may_settle) {
        if (!(observed && same_c) || was_steady) { settle_reset(settle); was_steady = false; }   // off the settled, observed stretch, or back from steady steps: start over
        else if ((t & 3) == 2) {
          // (in single precision after scaling by a power of two near Q -- P is of the order of 1 / Q --: the kernel has no
          // vector register to spare, and the test only needs the ratio of two maxima to a few digits)
          const double sc = settle_pow2_inverse_of(iq);
          float dl = 0.f, mx = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) { dl = fmaxf(dl, (float)(settle_absdiff(Pn[r], P[r]) * sc)); mx = fmaxf(mx, (float)(fabs(Pn[r]) * sc)); }
          wave_max2f(dl, mx);
          chk_dl = settle_uniform(dl); chk_mx = settle_uniform(mx); chk = true;   // judged at the end of the step, where fewer values are live
        }
      }
      P = Pn;
      qcol = vR[idx[0]] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) qcol = fma(vR[idx[s]], val[s], qcol);
      vQ[c] = -qcol;                                         // published by the next wave_sync
      wave_sync();                                           // pass-2 reads of imgB precede its reuse
#ifdef DLM_STAMP
      asm volatile("" ::"v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]));
#endif
      STAMP(4)
      }
    } else if constexpr (PIPE) { X2_STEP(1) X2_STEP(2) X2_STEP(3) }
#undef X2_STEP
    // output: [S_t | s_t] = [C_t | m_t] - C_t [P_t C_t | -q_t]
#pragma unroll
    for (int r = 0; r < 4; ++r) out[r] = cc[r] - x2[r];
    if (ST && chk) { psteady = settle_test(settle, chk_dl, chk_mx, 4); chk = false; }
    if (ST && psteady) {                                   // the steady steps may begin: S_t waits for them in the idle image
#pragma unroll
      for (int r = 0; r < 4; ++r) imgB[(4 * r + g) * LD + c] = out[r];
      wave_sync();
    }
    }
    same_next = same_c;
    const int so = t * recs;
#pragma unroll
    for (int r = 0; r < 4; ++r) buf_store(rout, bout, offA[r], so, out[r]);
    if (COV) {   // C_t behind S_t and the K row: one table row serves a whole backward step of the mean-only kernel
#pragma unroll
      for (int r = 0; r < 4; ++r) buf_store(rout, bout, offA[r], so + recb + 128, cc[r]);
    }
    STAMP(5)
  }
#ifdef DLM_STAMP
  if (n == 0 && lane == 0 && a.status)
    for (int k = 0; k < 8; ++k) a.status[1 + k] = (int)(seg[k] / (unsigned long long)(T + 1));
#endif
  vm_wait<0>();   // no DMA may still be writing this block's LDS when the wave ends
  if (!COV && a.counters && lane == 0 && nsteady) atomicAdd(&a.counters[1], (unsigned long long)nsteady);
  // P and q carry any non-finite value down to record 0: test the last output
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) bad |= va[r] && !isfinite(out[r]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

template <int K, bool IRR, bool PIPE = false, bool PLAIN = false>
__global__ __launch_bounds__(256, PIPE ? 2 : (PLAIN ? SM_WAVES : (K <= 2 ? SM_WAVES_K2 : SM_WAVES))) void k_smoother_sp16(KArgs a, const SparseT* __restrict__ sp,
                                                       const double* __restrict__ side) {
  __shared__ __attribute__((aligned(16))) double lds[4 * SM_LDS];
  extern __shared__ __attribute__((aligned(16))) char ring_all[];
#ifndef DLM_NO_PLAIN_MIX   // (A/B builds only)
  if constexpr (!IRR && !PLAIN) {   // a series marked by its gaps (KArgs::plain) takes the body without the shortcut's machinery
    const int n = blockIdx.x * (int)(blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (a.plain && n < a.N && a.plain[n]) { smoother_body<K, false, PIPE, false, true>(a, sp, side, lds, ring_all); return; }
  }
#endif
  smoother_body<K, IRR, PIPE, false, PLAIN>(a, sp, side, lds, ring_all);
}
// plain[n] = 1 where series n misses more than T / 256 of its observations (p = 1)
__global__ __launch_bounds__(256) void k_count_gaps(const double* __restrict__ y, int N, int T, unsigned char* __restrict__ plain) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  const double* yn = y + (size_t)n * T;
  int cnt = 0;
  for (int t = lane; t < T; t += 64) { const double v = yn[t]; cnt += (v == v) ? 0 : 1; }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) plain[n] = (cnt * 256 > T) ? 1 : 0;
}
// the covariance-only run: one wave alone on its SIMD (the variant whose output-product MFMAs are spread over the recursion)
template <int K>
__global__ __launch_bounds__(64) void k_cov_smoother_sp16(KArgs a, const SparseT* __restrict__ sp, const double* __restrict__ side,
                                                          double* __restrict__ kbtab) {
  __shared__ __attribute__((aligned(16))) double lds[4 * SM_LDS];
  extern __shared__ __attribute__((aligned(16))) char ring_all[];
  smoother_body<K, false, true, true>(a, sp, side, lds, ring_all, kbtab);
}

// ---------------------------------------------------------------------------------------
// second half of the simulation smoother: mean-only backward pass on y*, theta = s* + x+,
// Gibbs sufficient statistics (Gibbs.scala:23-78, GibbsWishart.scala:16-35) on the fly.
//   s*_t = m*_t + C_t q_t ,  q_{t-1} = G^T [ q_t + F (e_t/Q_t - K_t^T q_t) ] ,  K_t = C_t F / V
// No covariance recursion and no MFMA: O(K d + d^2) work per step.
// ---------------------------------------------------------------------------------------
template <int K, bool IRR>
__global__ __launch_bounds__(256, FI_WAVES) void k_simsmooth_sp16(KArgs a, const SparseT* __restrict__ sp,
                                                        const double* __restrict__ side,
                                                        const double* __restrict__ xplus) {
  __shared__ __attribute__((aligned(16))) double lds[4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x * (int)(blockDim.x >> 6) + wave;   // 4 waves per block, or 1 for small batches (launch)
  if (n >= a.N) return;
  double* vQ = lds + wave * 64;   // q_t
  double* vR = vQ + 16;           // r
  double* vT = vR + 16;           // theta_t
  double* vD = vT + 16;           // theta_{t+1} - G theta_t
  const int d = a.d, T = a.T, recb = packed_rec_bytes(d);   // packed records of the SIM forward pass
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;

  const double V = a.V[(size_t)n * a.v_stride];
  double rV = 1.0 / V;   // 1 / V_t of the record's observation when time-varying (IRR instantiation)
  const char* bin = (const char*)a.filt_in + (size_t)n * (T + 1) * recb;
  const __amdgpu_buffer_rsrc_t rin = make_rsrc(bin, (size_t)(T + 1) * recb);
  const double* sd = side + (size_t)n * (T + 1) * 2;
  const double* xp = xplus + (size_t)n * (T + 1) * d;
  const double* y = a.y ? a.y + (size_t)n * T : nullptr;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;

  // rows of G (for G theta) and columns of G (for G^T r)
  int ridx[K], cidx[K];
  double rval[K], cval[K];
  int grow = 0, gcol = 0;     // tables held: rows of G_grow (step t -> t+1), columns of G_gcol (step t-1 -> t)
#pragma unroll
  for (int s = 0; s < K; ++s) { ridx[s] = sp[0].idx[c][s]; rval[s] = sp[0].val[c][s]; cidx[s] = sp[1].idx[c][s]; cval[s] = sp[1].val[c][s]; }
  double Fr[4];
  int offC[4];
  const int offMl = vc ? c * 8 : OOB;
  double Fc = vc ? a.F[c] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    Fr[r] = i < d ? a.F[i] : 0.0;
    const int hi = i > c ? i : c, lo = i > c ? c : i;   // C is symmetric: element (i, c) of the packed lower triangle
    offC[r] = (i < d && vc) ? (d + hi * (hi + 1) / 2 + lo) * 8 : OOB;
  }
  double qcol = 0.0, thn = 0.0, ssy = 0.0, nob = 0.0, ssc = 0.0;
  d4 so = {0.0, 0.0, 0.0, 0.0};
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;

  d4 ncc;
  double nm, nx;
#pragma unroll
  for (int r = 0; r < 4; ++r) ncc[r] = buf_load(rin, bin, offC[r], T * recb);
  nm = buf_load(rin, bin, offMl, T * recb);
  nx = vc ? xp[(size_t)T * d + c] : 0.0;
  double neq = sd[2 * T], niq = sd[2 * T + 1];
  double ychunk = 0.0;

  bool inh = false;       // C_t is C_{t+1} (the forward pass's mark on record t + 1: 1 / Q negated): K_t is K_{t+1}
  double kcs = 0.0;
  for (int t = T; t >= 0; --t) {
    const d4 cc = ncc;
    const double mcol = nm, xcol = nx;
    const double eq = uniform_from_lane(neq, 0), iq = uniform_from_lane(niq, 0);
    const bool same_prev = !IRR && iq < 0.0;                 // C_{t-1} is C_t: the next record is fetched as its mean alone
    {
      const int tp = t > 0 ? t - 1 : 0;
      if (!same_prev) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ncc[r] = buf_load(rin, bin, offC[r], tp * recb);
      }
      nm = buf_load(rin, bin, offMl, tp * recb);
      nx = vc ? xp[(size_t)tp * d + c] : 0.0;
      neq = sd[2 * tp]; niq = sd[2 * tp + 1];
    }
    const bool observed = (iq == iq) && t > 0;
    if (IRR && a.f_stride && t > 0) {                        // F of the observation at record t
      const double* Ft = a.F + (size_t)(t - 1) * a.f_stride;
      Fc = vc ? Ft[c] : 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) Fr[r] = (4 * r + g < d) ? Ft[4 * r + g] : 0.0;
    }

    // theta_t = m*_t + C_t q_t + x+_t
    vQ[c] = qcol;
    wave_sync();
    double cq = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) cq = fma(cc[r], vQ[4 * r + g], cq);   // C symmetric: column sums
    const double th = mcol + sum_g(cq) + xcol;
    if (thout && g == 0 && vc) thout[(size_t)t * d + c] = th;

    if (a.stats) {
      vT[c] = th;
      wave_sync();
      if (t < T) {   // system innovation (theta_{t+1} - G_{t+1} theta_t) / sqrt(dt_{t+1})
        const int gn = (IRR && a.g_index) ? a.g_index[t] : 0;
        const double dtn = (IRR && a.dt) ? a.dt[t] : 1.0;
        if (IRR && gn != grow) {
#pragma unroll
          for (int s = 0; s < K; ++s) { ridx[s] = sp[2 * gn].idx[c][s]; rval[s] = sp[2 * gn].val[c][s]; }
          grow = gn;
        }
        double gth = th;
        if (dtn != 0.0) {
          gth = vT[ridx[0]] * rval[0];
#pragma unroll
          for (int s = 1; s < K; ++s) gth = fma(vT[ridx[s]], rval[s], gth);
        }
        const double df = vc ? (thn - gth) / sqrt(dtn == 0.0 ? 1.0 : dtn) : 0.0;
        ssc = fma(df, df, ssc);
        if (outer) {
          vD[c] = df;
          wave_sync();
#pragma unroll
          for (int r = 0; r < 4; ++r) so[r] = fma(vD[4 * r + g], df, so[r]);
        }
      }
      if (t > 0 && y) {   // observation residual of theta_t against y_t
        const int ti = t - 1;
        if (t == T || (ti & 63) == 63) ychunk = ((ti & ~63) + lane < T) ? y[(ti & ~63) + lane] : 0.0;
        const double yv = uniform_from_lane(ychunk, ti & 63);
        if (yv == yv) { const double res = yv - row_sum(Fc * th); ssy = fma(res, res, ssy); nob += 1.0; }
      }
    }
    thn = th;
    if (t == 0) break;

    double rcol = qcol;
    if (IRR && a.v_tstride) {                                // V of the observation at record t (StudentTGibbs.scala:100-136)
      const double Vt = a.V[(size_t)n * a.v_stride + (size_t)(t - 1) * a.v_tstride];
      if (!(Vt > 0.0)) st |= DLM_ST_NOT_PD;
      rV = 1.0 / Vt;
    }
    if (observed) {
      if (!inh) {
        double kc = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) kc = fma(cc[r], Fr[r], kc);
        kcs = sum_g(kc) * rV;
      }
      rcol = fma(Fc, eq - row_sum(kcs * qcol), qcol);
    }
    inh = same_prev;
    const int gi = (IRR && a.g_index) ? a.g_index[t - 1] : 0;   // G of the step into record t
    const double dtt = (IRR && a.dt) ? a.dt[t - 1] : 1.0;
    if (IRR && gi != gcol) {
#pragma unroll
      for (int s = 0; s < K; ++s) { cidx[s] = sp[2 * gi + 1].idx[c][s]; cval[s] = sp[2 * gi + 1].val[c][s]; }
      gcol = gi;
    }
    vR[c] = rcol;
    wave_sync();
    if (dtt == 0.0) qcol = rcol;
    else {
      qcol = vR[cidx[0]] * cval[0];
#pragma unroll
      for (int s = 1; s < K; ++s) qcol = fma(vR[cidx[s]], cval[s], qcol);
    }
    wave_sync();
  }
  if (__ballot(vc && !isfinite(thn)) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.stats) {
    const int L = stats_len(d, 1, a.flags);
    double* sout = a.stats + (size_t)n * L;
    if (lane == 0) { sout[0] = ssy; sout[1] = nob; sout[L - 1] = (double)T; }
    if (outer) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int i = 4 * r + g; if (i < d && vc) sout[2 + i + c * d] = so[r]; }
    } else if (g == 0 && vc) sout[2 + c] = ssc;
  }
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// Shared covariance sequence.
//
// With V, W and C0 shared by the batch and no missing observation, C_t, R_t, K_t, Q_t (KalmanFilter.scala:64-107) and the
// smoother's P_t, S_t (Smoothing.scala:31-47, information form above) do not depend on the data: the reference recomputes them
// for every series because it has no batch.  Here ONE wave runs the two covariance recursions -- k_cov_filter_sp16 /
// k_cov_smoother_sp16: the per-series kernels' own code on a series of zeros -- into tables that stay in L2, and every series runs
// only the mean recursions against them (k_mean_filter_sp16 / k_mean_smoother_sp16).  The arithmetic of every output element is
// that of the per-series kernels, operation for operation (the same FMA chains, the same reduction orders, the MFMA for the
// smoothed mean of a full step), including which steps they take in their steady-state form: the results are bit for bit
// those of k_filter_sp16 / k_smoother_sp16 (tests/test_shared_cov_gpu.py).  A series that meets a missing observation is
// marked in KArgs::route by the forward kernel and served by the per-series kernels launched behind.
//
// What a series-step costs is then its records: the state record is the table's C_t (S_t) with the mean patched into its
// first d doubles, written as 16-byte pieces -- two store instructions per record.
// ---------------------------------------------------------------------------------------
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ d2 buf_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  d2 o = {__hiloint2double((int)v[1], (int)v[0]), __hiloint2double((int)v[3], (int)v[2])};
  return o;
}
__device__ __forceinline__ void buf_store2(__amdgpu_buffer_rsrc_t r, int voff, int soff, d2 x) {
  const u4 v = {(unsigned)__double2loint(x[0]), (unsigned)__double2hiint(x[0]), (unsigned)__double2loint(x[1]), (unsigned)__double2hiint(x[1])};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
}


// n16 <= 256 pieces of 16 bytes from byte offset soff of the buffer to LDS byte address lds_addr (see dma_record)
__device__ __forceinline__ void dma_pieces(const i4& rs, unsigned lds_addr, int soff, int lane, int n16) {
  const int voff = lane * 16;
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  if (lane < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if (lane + 64 < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if (lane + 128 < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if (lane + 192 < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ d2 lds_read128(unsigned addr) {
  d2 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// the reads above are not counted by the compiler: these waits also tie the loaded registers to the wait, so that no use of
// them can be scheduled ahead of it
__device__ __forceinline__ void lds_wait(d2& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a)::"memory"); }
__device__ __forceinline__ void lds_wait(d2& a, d2& b, d2& c) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory"); }
__device__ __forceinline__ void lds_wait(d2& a, d2& b, double& c) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c)::"memory"); }
__device__ __forceinline__ void lds_wait(d4& a, d4& b, d2& c, d2& e, double& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(e), "+v"(f)::"memory");
}

// ---- four series per wave ------------------------------------------------------------------------------------------
// A mean recursion is a d-vector: sixteen lanes.  Lane 16 j + c of a wave holds component c of series j (j = 0..3) of the wave's
// four consecutive series; the row-wise DPP reductions and the gathers through LDS advance four series with the instructions of
// one, one table row (LDS-DMA, two steps ahead) serves all four, and their four records leave as 16-byte pieces, six store
// instructions for 4 x d (d + 1) / 2 pieces.
__device__ __forceinline__ double row_lane0(double v, int lane) {   // every lane of a 16-lane row gets the value of the row's lane 0
  const int a = (lane & 48) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_pick(double v, int lane, int src) {   // ... of the row's lane src (0..15, wave-uniform)
  const int a = ((lane & 48) + src) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(a, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(a, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int MEAN4_LDS = 6 * 64;   // per wave: six vectors of 4 x 16 doubles
constexpr int MEAN_AHEAD = 8;       // backward: the filtered means are requested this many steps ahead (512-byte slots)
template <int NP>
__device__ __forceinline__ void lds_wait_pieces(d2 (&pc)[NP]) {
  if constexpr (NP == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3])::"memory");
  else if constexpr (NP == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]), "+v"(pc[4]), "+v"(pc[5])::"memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]), "+v"(pc[4]), "+v"(pc[5]), "+v"(pc[6]), "+v"(pc[7])::"memory");
}

// Forward.  Table row r (tb.ftab, tb.frow bytes) = [C_r record | R_r F or K_r (16 doubles; [15] = +-1/Q_r)].
// NP: store instructions per step = 64-lane groups of 16-byte pieces covering the four records (4, 6 or 8 for d <= 10, 13, 15)
// REC: the filtered records are written here (dlm_filter_batch).  In the fused call (REC = false) only the means leave, compact --
// tb.mc, 512 bytes per step and wave: [group of four series][T+1][4][16] -- and the backward kernel writes BOTH record streams
// (it holds C_t anyway): the two passes' worth of stores in flight behind one wave, and no scattered re-read of the means.
template <int K, int NP, int MODE>   // MODE 0: compact means only, 1: records only (dlm_filter_batch), 2: records and compact means (fused call)
__global__ __launch_bounds__(256) void k_mean_filter_sp16(KArgs a, const SparseT* __restrict__ sp, CovTabs tb) {
  constexpr bool REC = MODE != 0, CMP = MODE != 1;
  constexpr int NS = (REC ? NP : 0) + (CMP ? 1 : 0);   // store instructions per step
  __shared__ __attribute__((aligned(16))) double lds[4 * MEAN4_LDS];
  extern __shared__ __attribute__((aligned(16))) char ring_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = 4 * (blockIdx.x * (int)(blockDim.x >> 6) + wave);   // first of this wave's four series
  if (n0 >= a.N) return;
  const int j = lane >> 4, c = lane & 15;
  const int nser = a.N - n0 < 4 ? a.N - n0 : 4;
  const bool have = j < nser;                       // this row has a series
  const int n = have ? n0 + j : n0;                 // (a row without one shadows the first: loads stay in bounds, stores are dropped)
  double* vM = lds + wave * MEAN4_LDS;   // m_col [4][16]
  double* vW = vM + 64;                  // m_row
  double* vA = vW + 64;                  // a_col (full steps)
  double* vH = vA + 64;                  // the first 16 doubles of each series' record
  const int d = a.d, T = a.T, rec = d + d * d, recb = rec * 8, frow = recb + 128;
  const bool vc = c < d;
  int idx[K];
  double val[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = 16 * j + sp[0].idx[c][s]; val[s] = sp[0].val[c][s]; }
  const double V = a.V[0];
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double* y = a.y + (size_t)n * T;
  const size_t sbytes = (size_t)(T + 1) * recb;                      // one series' records
  char* bout = REC ? (char*)a.filt + (size_t)n0 * sbytes : nullptr;
  const __amdgpu_buffer_rsrc_t rout = make_rsrc(bout, REC ? (size_t)nser * sbytes : 0);
  const __amdgpu_buffer_rsrc_t rcmp = make_rsrc(CMP ? (char*)(tb.mc + (size_t)(n0 / 4) * (T + 1) * 64) : nullptr, CMP ? (size_t)(T + 1) * 512 : 0);
  const int offmc = have ? lane * 8 : OOB;           // this lane's double of the wave's 512 compact bytes of a step
  double* eqn = tb.eq + (size_t)n * (T + 1);
  const double Fc = vc ? a.F[c] : 0.0;
  double F4[4];                                      // F[(c & 3) + 4 k]: the full step's forecast sums rows 4 k + g' in this order
#pragma unroll
  for (int k = 0; k < 4; ++k) F4[k] = ((c & 3) + 4 * k < d) ? a.F[(c & 3) + 4 * k] : 0.0;
  // the four records as 16-byte pieces: piece q = 64 k + lane is piece q % npc of series q / npc
  const int npc = rec / 2;
  unsigned psrc[NP];       // LDS byte offset of the piece inside a slot; bit 31: it lies in the record's first 16 doubles (taken from vH)
  int pdst[NP], pser[NP];   // byte offset in the output, series
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int q = 64 * k + lane, sj = q / npc, pp = q - sj * npc;
    pser[k] = sj;
    pdst[k] = sj < nser ? (int)((size_t)sj * sbytes) + pp * 16 : OOB;
    psrc[k] = pp < 8 ? (0x80000000u | (unsigned)((16 * (sj & 3) + 2 * pp) * 8)) : (unsigned)(pp * 16);
  }
  const unsigned vH_lds = lds_addr_of(vH);
  char* ring = ring_all + wave * 2 * frow;
  const unsigned ring_lds = lds_addr_of(ring);
  const unsigned pkf = recb + c * 8, ptd = c * 8;   // K row entry, table double c (the record's first 16 doubles beyond the mean)
  const i4 rtab = rsrc_words(tb.ftab, (unsigned)((size_t)(T + 1) * frow));
  const int nrow = frow / 16;
  dma_pieces(rtab, ring_lds, 0, lane, nrow);                    // row 0 -> slot 0
  dma_pieces(rtab, ring_lds + frow, frow, lane, nrow);          // row 1 -> slot 1
  // k_filter_sp16 carries the mean twice, as column 15 and as row 15 of its augmented tile, and a full step updates the two
  // copies with differently associated products (m_col = a + (R F)(e / Q), m_row = a + (R F / Q) e) from each other's advance:
  // a_col = G m_row, a_row = G m_col.  The record holds m_col.  Both are followed here (a steady step sets them equal).
  double mcol = vc ? m0[c] : 0.0, mrow = mcol;
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;
  unsigned nsteady = 0;
  bool dead = !have;                                 // row-uniform: the series met a missing observation (or there is none)
  unsigned deadmask = (unsigned)(0xf << nser) & 0xf; // the same for all four rows, wave-uniform
  vM[lane] = mcol; vW[lane] = mrow;
  // observations: lane 16 j + c holds y_j[64 b + 16 k + c] in yk[k], refreshed every 64 steps
  double yk[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) yk[k] = (16 * k + c < T) ? y[16 * k + c] : 0.0;
  asm volatile("" ::"v"(yk[0]), "v"(yk[1]), "v"(yk[2]), "v"(yk[3]));
  double ek[4] = {__builtin_nan(""), 0.0, 0.0, 0.0};   // e / Q of records 64 b + 16 k + c (record 0: NaN), stored every 64 records
  wave_sync();
  vm_wait<1>();                                                 // row 0 (only the requests for row 1 may still be on their way)

  auto store = [&](const d2 (&pc)[NP], int so) {
#pragma unroll
    for (int k = 0; k < NP; ++k) buf_store2(rout, ((deadmask >> pser[k]) & 1u) ? OOB : pdst[k], so, pc[k]);
  };
  if constexpr (REC) {   // record 0: [m0 | C0]
    d2 pc[NP];
    {
      double td = lds_read64<0>(ring_lds + ptd);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(td)::"memory");
      vH[lane] = vc ? mcol : td;
      wave_sync();
#pragma unroll
      for (int k = 0; k < NP; ++k) pc[k] = lds_read128((psrc[k] & 0x80000000u) ? vH_lds + (psrc[k] & 0x7fffffffu) : ring_lds + psrc[k]);
      lds_wait_pieces<NP>(pc);
    }
    dma_pieces(rtab, ring_lds, (T >= 2 ? 2 : T) * frow, lane, nrow);   // row 2 -> slot 0
    store(pc, 0);
  } else dma_pieces(rtab, ring_lds, (T >= 2 ? 2 : T) * frow, lane, nrow);   // row 2 -> slot 0
  if constexpr (CMP) buf_store(rcmp, nullptr, offmc, 0, vc ? mcol : 0.0);
  for (int t = 0; t < T; ++t) {
    if (t > 0 && (t & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) yk[k] = (t + 16 * k + c < T) ? y[t + 16 * k + c] : 0.0;
      asm volatile("" ::"v"(yk[0]), "v"(yk[1]), "v"(yk[2]), "v"(yk[3]));     // consume the loads inside the branch (see k_filter_sp16)
    }
    const int kk = (t >> 4) & 3;
    const double yt = row_pick(kk == 0 ? yk[0] : kk == 1 ? yk[1] : kk == 2 ? yk[2] : yk[3], lane, t & 15);
    if (!dead && !(yt == yt)) {   // a missing observation: this series' covariances are its own -- the per-series kernels take it (all of it)
      dead = true;
      if (c == 0) a.route[n] = 1;
    }
    deadmask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) deadmask |= (__builtin_amdgcn_readlane((int)dead, 16 * q) & 1) << q;
    if (deadmask == 0xfu) break;
    // Operations issued after the request for row t + 1: the NP stores of record t - 1, the request for row t + 2 (>= 1
    // instruction), the NP stores of record t.  In the first step only the request for row 2 and the stores of record 0.
    if (t == 0) vm_wait<NS + 1>(); else vm_wait<2 * NS + 1>();
    const unsigned slot = ring_lds + ((t + 1) & 1) * frow;
    double kfc = lds_read64<0>(slot + pkf);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kfc)::"memory");
    const double rqs = uniform_from_lane(kfc, 15);        // +-1/Q_t rides in the row's last slot; negative: the step is a steady one
    const double rq = fabs(rqs);
    double ac = vW[idx[0]] * val[0];
#pragma unroll
    for (int s = 1; s < K; ++s) ac = fma(vW[idx[s]], val[s], ac);          // a_col = G m_row
    double e, erq;
    if (rqs < 0.0) {                                                       // steady step of k_filter_sp16: f by the row sum, m = a + K e
      nsteady += 4 - __builtin_popcount(deadmask);
      const double f = row_lane0(row_sum(Fc * ac), lane);
      e = yt - f; erq = e * rq;
      mcol = fma(kfc, e, ac);
      mrow = mcol;
    } else {                                                               // full step: f = (R F)[15] in ITS summation order
      double ar = vM[idx[0]] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) ar = fma(vM[idx[s]], val[s], ar);        // a_row = G m_col
      vA[lane] = ac;
      wave_sync();
      double pf = 0.0;                                                     // rows g', g' + 4, g' + 8, g' + 12 (g' = c & 3) as one FMA chain ...
#pragma unroll
      for (int k = 0; k < 4; ++k) pf = fma(vA[16 * j + (c & 3) + 4 * k], F4[k], pf);
      pf = pf + quad_perm<0xB1>(pf);                                       // ... then (p0 + p1) + (p2 + p3), as the cross-row sum of k_filter_sp16 adds them
      const double f = pf + quad_perm<0x4E>(pf);
      e = yt - f; erq = e * rq;
      mcol = fma(kfc, erq, ac);                                            // m_col = a_col + (R F)(e / Q)
      mrow = fma(kfc * rq, e, ar);                                         // m_row = a_row + K e,  K = (R F) / Q
    }
    {   // e / Q of record t + 1 into its place of the 64-record chunk
      const int k1 = ((t + 1) >> 4) & 3;
      const bool mine = c == ((t + 1) & 15);
      ek[0] = (mine && k1 == 0) ? erq : ek[0]; ek[1] = (mine && k1 == 1) ? erq : ek[1];
      ek[2] = (mine && k1 == 2) ? erq : ek[2]; ek[3] = (mine && k1 == 3) ? erq : ek[3];
      if (((t + 1) & 63) == 63 || t + 1 == T) {
        const int base = (t + 1) & ~63;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (!dead && base + 16 * k + c <= t + 1) eqn[base + 16 * k + c] = ek[k];
      }
    }
    wave_sync();                                                           // the reads of vM, vW (and vA) precede their rewrite
    vM[lane] = vc ? mcol : 0.0; vW[lane] = vc ? mrow : 0.0;
    wave_sync();
    if constexpr (REC) {
      d2 pc[NP];
      {
        double td = lds_read64<0>(slot + ptd);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(td)::"memory");
        vH[lane] = vc ? mcol : td;
        wave_sync();
#pragma unroll
        for (int k = 0; k < NP; ++k) pc[k] = lds_read128((psrc[k] & 0x80000000u) ? vH_lds + (psrc[k] & 0x7fffffffu) : slot + psrc[k]);
        lds_wait_pieces<NP>(pc);
      }
      { const int tn = t + 3 <= T ? t + 3 : T; dma_pieces(rtab, slot, tn * frow, lane, nrow); }   // the slot is free again: row t + 3
      store(pc, (t + 1) * recb);
    } else { const int tn = t + 3 <= T ? t + 3 : T; dma_pieces(rtab, slot, tn * frow, lane, nrow); }
    if constexpr (CMP) buf_store(rcmp, nullptr, dead ? OOB : offmc, (t + 1) * 512, vc ? mcol : 0.0);
  }
  vm_wait<0>();   // no DMA may still be writing this block's LDS when the wave ends
  if (have && !dead && c == 0) a.route[n] = 0;
  if (a.counters && lane == 0) { atomicAdd(&a.counters[2], (unsigned long long)(4 - __builtin_popcount(deadmask))); if (nsteady) atomicAdd(&a.counters[0], (unsigned long long)nsteady); }
  const unsigned long long badl = __ballot(!dead && vc && !isfinite(mcol));
  if (have && !dead && c == 0) {
    const int sj = st | (((badl >> (16 * j)) & 0xffffull) ? DLM_ST_NONFINITE : 0);
    if (a.status && sj) atomicOr(&a.status[n], sj);
  }
}

// Backward.  Of table row t (tb.btab, tb.brow bytes: [S_t record | K_t (16 doubles; [15] = 1: steady step) | C_t record]) the K row
// and C_t travel two steps ahead into a two-slot ring (they come from L2); the four series' filtered means m_t -- the forward
// kernel's compact 512 bytes per step, from HBM -- MEAN_AHEAD steps ahead into a ring of their own.  Only the smoothed MEANS
//
// REC: the smoothed records are written here (the whole row travels, S_t leaves as 16-byte pieces with s_t patched in).  (REC = false
// -- only the smoothed means leave, compact, for a separate fully parallel record-writer kernel -- was built and measured: the
// writer reached 3.3 TB/s, no better than the stores of this kernel; profiles/r03_notes.md.)
template <int K, int NP, bool REC>
__global__ __launch_bounds__(256) void k_mean_smoother_sp16(KArgs a, const SparseT* __restrict__ sp, CovTabs tb) {
  constexpr int NS = REC ? NP : 1;   // store instructions per step
  __shared__ __attribute__((aligned(16))) double lds[4 * MEAN4_LDS];
  extern __shared__ __attribute__((aligned(16))) char ring_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n0 = 4 * (blockIdx.x * (int)(blockDim.x >> 6) + wave);
  if (n0 >= a.N) return;
  const int j = lane >> 4, c = lane & 15, g = j;
  const int nser = a.N - n0 < 4 ? a.N - n0 : 4;
  const bool have = j < nser;
  const int n = have ? n0 + j : n0;
  const bool dead = !have || a.route[n] != 0;       // a series with a missing observation: k_smoother_sp16 serves it
  unsigned deadmask = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) deadmask |= (__builtin_amdgcn_readlane((int)dead, 16 * q) & 1) << q;
  if (deadmask == 0xfu) return;
  double* vQ = lds + wave * MEAN4_LDS;   // -q_t [4][16]
  double* vR = vQ + 64;                  // r
  double* vX = vR + 64;                  // C q products on their way from the column lanes to the series rows
  double* vH = vX + 64;                  // REC: the first 16 doubles of each series' smoothed record
  const int d = a.d, T = a.T, rec = d + d * d, recb = rec * 8, brow = 2 * recb + 128;
  const int skip = REC ? 0 : recb;                  // bytes of a table row that do not travel (S_t when only the means leave)
  const int part = brow - skip, slotb = part + 16;
  const bool vc = c < d;
  int idx[K];
  double val[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = 16 * j + sp[1].idx[c][s]; val[s] = sp[1].val[c][s]; }
  const double V = a.V[0];
  const char* bin = (const char*)(tb.mc + (size_t)(n0 / 4) * (T + 1) * 64);    // the wave's compact filtered means: 512 bytes per step
  const size_t sbytes = (size_t)(T + 1) * recb;
  const __amdgpu_buffer_rsrc_t rout = REC ? make_rsrc((char*)a.smooth + (size_t)n0 * sbytes, (size_t)nser * sbytes)
                                          : make_rsrc(tb.sc + (size_t)(n0 / 4) * (T + 1) * 64, (size_t)(T + 1) * 512);
  const int offsc = dead ? OOB : lane * 8;
  const int npc = rec / 2;
  unsigned psrc[NP];
  int pdst[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int q = 64 * k + lane, sj = q / npc, pp = q - sj * npc;
    pdst[k] = (sj < nser && !((deadmask >> (sj & 3)) & 1u)) ? (int)((size_t)sj * sbytes) + pp * 16 : OOB;
    psrc[k] = pp < 8 ? (0x80000000u | (unsigned)((16 * (sj & 3) + 2 * pp) * 8)) : (unsigned)(pp * 16);
  }
  const unsigned vH_lds = lds_addr_of(vH);
  const double* eqn = tb.eq + (size_t)n * (T + 1);
  const double Fc = vc ? a.F[c] : 0.0;
  char* ring = ring_all + wave * (2 * slotb + MEAN_AHEAD * 512);
  const unsigned ring_lds = lds_addr_of(ring), mring_lds = ring_lds + 2 * slotb;   // table slots, then the slots of the means
  const unsigned pzero = part;                      // the zero double behind the slot's data (what the padded lanes read)
  const unsigned pkb = recb - skip + c * 8, pmean = lane * 8, ptd = c * 8;     // a slot holds [(S_t record) | K row | C_t record]
  unsigned pC[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int i = 4 * r + g; pC[r] = (i < d && vc) ? (unsigned)(recb - skip + 128 + (d + i * d + c) * 8) : pzero; }   // C_t[4r+g][c]
  const i4 rtab = rsrc_words((const char*)tb.btab + skip, (unsigned)((size_t)(T + 1) * brow - skip));
  const i4 rmean = rsrc_words(bin, (unsigned)((size_t)(T + 1) * 512));
  const int nrow = part / 16;
  const int mvoff = lane * 16;
  const bool mact = lane < 32;
  auto dma_means = [&](unsigned lds_addr, int soff) {
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    if (mact) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(mvoff), "s"(rmean), "s"(soff) : "memory");
  };
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;
  unsigned nsteady = 0;
  double qcol = 0.0;
  vQ[lane] = 0.0;
  if (lane < 2) *(double*)(ring + lane * slotb + part) = 0.0;
  wave_sync();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
    // the means of steps T .. T - MEAN_AHEAD + 1 first (the oldest requests), then the table rows T and T - 1
    for (int k = 0; k < MEAN_AHEAD; ++k) { const int tk = T - k > 0 ? T - k : 0; dma_means(mring_lds + ((T - k) & (MEAN_AHEAD - 1)) * 512, tk * 512); }
    const int t1 = T > 0 ? T - 1 : 0;
    dma_pieces(rtab, ring_lds + (T & 1) * slotb, T * brow, lane, nrow);
    dma_pieces(rtab, ring_lds + (t1 & 1) * slotb, t1 * brow, lane, nrow);
  }
  double ek[4] = {0.0, 0.0, 0.0, 0.0};   // e / Q of records 64 b + 16 k + c of this row's series
  double sm = 0.0;
  for (int t = T; t >= 0; --t) {
    if (t == T || (t & 63) == 63) {
      const int base = t & ~63;
#pragma unroll
      for (int k = 0; k < 4; ++k) ek[k] = (base + 16 * k + c <= T) ? eqn[base + 16 * k + c] : 0.0;
      asm volatile("" ::"v"(ek[0]), "v"(ek[1]), "v"(ek[2]), "v"(ek[3]));
    }
    const int kk = (t >> 4) & 3;
    const double eq = row_pick(kk == 0 ? ek[0] : kk == 1 ? ek[1] : kk == 2 ? ek[2] : ek[3], lane, t & 15);
    // Operations issued after the request for table row t: the means request and the store of step t + 2, the two requests of
    // step t + 1 (>= 2 instructions) and its store.  The first two steps have fewer operations behind them.  The means of step t
    // were requested MEAN_AHEAD steps ago: older than row t, complete with it.
    if (t == T) vm_wait<1>(); else if (t == T - 1) vm_wait<NS + 2>(); else vm_wait<2 * NS + 3>();
    const unsigned slot = ring_lds + (t & 1) * slotb, mslot = mring_lds + (t & (MEAN_AHEAD - 1)) * 512;
    d4 cc;
    cc[0] = lds_read64<0>(slot + pC[0]); cc[1] = lds_read64<0>(slot + pC[1]); cc[2] = lds_read64<0>(slot + pC[2]); cc[3] = lds_read64<0>(slot + pC[3]);
    double mr = lds_read64<0>(mslot + pmean), kb = lds_read64<0>(slot + pkb), td = REC ? lds_read64<0>(slot + ptd) : 0.0;
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cc), "+v"(mr), "+v"(kb), "+v"(td)::"memory");
    if constexpr (!REC) {   // the slots are free again: table row t - 2, means t - MEAN_AHEAD (always issued: the operation count behind every request stays the same)
      const int t2 = t > 1 ? t - 2 : 0, tm = t > MEAN_AHEAD ? t - MEAN_AHEAD : 0;
      dma_pieces(rtab, slot, t2 * brow, lane, nrow);
      dma_means(mslot, tm * 512);
    }
    const bool steady = uniform_from_lane(kb, 15) != 0.0;    // the step k_smoother_sp16 takes in its steady form
    if (steady) {
      // s_t = m_t + C_t q_t by column sums on the symmetric C_t (k_smoother_sp16's steady step), once per series
      nsteady += 4 - __builtin_popcount(deadmask);
      double pick = 0.0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        double ncq = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) ncq = fma(cc[r], vQ[16 * q + 4 * r + g], ncq);
        ncq = sum_g(ncq);
        pick = (j == q) ? ncq : pick;
      }
      sm = mr - pick;
    } else {
      // full step: the columns [ -q_0 | -q_1 | -q_2 | -q_3 ] of C_t [ . ] on the matrix pipe, as column 15 of [S | s] = [C | m] - C [P C | -q]
      d4 b2;
#pragma unroll
      for (int r = 0; r < 4; ++r) b2[r] = (c < 4) ? vQ[16 * c + 4 * r + g] : 0.0;
      const d4 x2 = mmT(cc, b2);
      wave_sync();                                           // (the reads of vX of the step before are done)
      if (c < 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) vX[16 * c + 4 * r + g] = x2[r];
      }
      wave_sync();
      sm = mr - vX[lane];
    }
    if (t > 0) {   // q_{t-1} = G^T [ q_t + F (e_t / Q_t - K_t . q_t) ]
      const double kq = row_lane0(row_sum(kb * qcol), lane);
      wave_sync();                                           // the reads of vQ above precede its rewrite below; vR's of the step before too
      vR[lane] = fma(Fc, eq - kq, qcol);
      wave_sync();
      qcol = vR[idx[0]] * val[0];
#pragma unroll
      for (int s = 1; s < K; ++s) qcol = fma(vR[idx[s]], val[s], qcol);
    } else wave_sync();
    vQ[lane] = -qcol;
    if constexpr (REC) {   // the four records: S_t of the table with s_t in the first d doubles
      vH[lane] = vc ? sm : td;
      wave_sync();
      d2 pc[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) pc[k] = lds_read128((psrc[k] & 0x80000000u) ? vH_lds + (psrc[k] & 0x7fffffffu) : slot + psrc[k]);
      lds_wait_pieces<NP>(pc);
      {
        const int t2 = t > 1 ? t - 2 : 0, tm = t > MEAN_AHEAD ? t - MEAN_AHEAD : 0;
        dma_pieces(rtab, slot, t2 * brow, lane, nrow);
        dma_means(mslot, tm * 512);
      }
      const int so = t * recb;
#pragma unroll
      for (int k = 0; k < NP; ++k) buf_store2(rout, pdst[k], so, pc[k]);
    } else {
      wave_sync();
      buf_store(rout, nullptr, offsc, t * 512, vc ? sm : 0.0);
    }
  }
  vm_wait<0>();   // no DMA may still be writing this block's LDS when the wave ends
  if (a.counters && lane == 0 && nsteady) atomicAdd(&a.counters[1], (unsigned long long)nsteady);
  const unsigned long long badl = __ballot(!dead && vc && !isfinite(sm));
  if (!dead && c == 0) {
    const int sj = st | (((badl >> (16 * j)) & 0xffffull) ? DLM_ST_NONFINITE : 0);
    if (a.status && sj) atomicOr(&a.status[n], sj);
  }
}

size_t covtabs_doubles(int d, int T) { return (size_t)(T + 1) * (3 * (size_t)(d + d * d) + 16 + 16 + 2) + 64; }
void covtabs_carve(double* base, int d, int T, CovTabs& t) {
  const size_t rec = (size_t)(d + d * d), n1 = (size_t)T + 1;
  t.frow = (int)(rec * 8 + 128); t.brow = (int)(2 * rec * 8 + 128);
  t.ftab = base; t.btab = t.ftab + n1 * (rec + 16); t.cside = t.btab + n1 * (2 * rec + 16);
  t.skip = nullptr;
}
bool shared_cov_eligible(const KArgs& a) {
  return a.d <= 15 && a.p == 1 && !a.g_index && !a.dt && !a.f_stride && !a.v_tstride && !a.w_tstride && !a.v_stride && !a.w_stride &&
         !a.c0_stride && !a.packed && !a.prior && !a.fq && !a.loglik && a.filt && (a.flags & DLM_OPT_SHARED_COV) && !(a.flags & DLM_OPT_FORCE_GENERIC) &&
         4 * ((size_t)a.T + 1) * (size_t)(a.d + a.d * a.d) * 8 < ((size_t)1 << 31);   // four series' records under one buffer resource
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
  // row / column 15 of the augmented transition [[G, 0], [0, 1]] (the kernels carry the mean there)
  rows->idx[15][0] = cols->idx[15][0] = 15;
  rows->val[15][0] = cols->val[15][0] = 1.0;
  rows->K = cols->K = kmax;
  return kmax;
}

// Batches that cannot fill the chip's SIMDs twice over go out as one-wave workgroups: the dispatcher then spreads the waves
// over all CUs one by one instead of four at a time (at 1250 series: 5 waves on almost every CU instead of 8 on 57 of them).
static int waves_per_block(const KArgs& a) { return (a.N < 2048 && !(a.flags & DLM_OPT_NO_SMALL_BATCH)) ? 1 : 4; }

template <int K>
static hipError_t launch_f(const KArgs& a, const SparseT* sp, double* side, double* xplus, hipStream_t s) {
  const bool irr = a.g_index || a.dt || a.f_stride || a.v_tstride || a.w_tstride;
  const int wpb = waves_per_block(a);
  const dim3 grid((a.N + wpb - 1) / wpb), blk(64 * wpb);
  if (a.loglik && !xplus) {
    if (irr) hipLaunchKernelGGL((k_filter_sp16<K, false, true, true>), grid, blk, 0, s, a, sp, side, xplus);
    else hipLaunchKernelGGL((k_filter_sp16<K, false, false, true>), grid, blk, 0, s, a, sp, side, xplus);
  } else if (xplus && irr) hipLaunchKernelGGL((k_filter_sp16<K, true, true>), grid, blk, 0, s, a, sp, side, xplus);
  else if (xplus) hipLaunchKernelGGL((k_filter_sp16<K, true, false>), grid, blk, 0, s, a, sp, side, xplus);
  else if (irr) hipLaunchKernelGGL((k_filter_sp16<K, false, true>), grid, blk, 0, s, a, sp, side, xplus);
  else hipLaunchKernelGGL((k_filter_sp16<K, false, false>), grid, blk, 0, s, a, sp, side, xplus);
  return hipGetLastError();
}
template <int K>
static hipError_t launch_ss(const KArgs& a, const SparseT* sp, const double* side, const double* xplus, hipStream_t s) {
  const int wpb = waves_per_block(a);
  const dim3 grid((a.N + wpb - 1) / wpb), blk(64 * wpb);
  if (a.g_index || a.dt || a.f_stride || a.v_tstride) hipLaunchKernelGGL((k_simsmooth_sp16<K, true>), grid, blk, 0, s, a, sp, side, xplus);
  else hipLaunchKernelGGL((k_simsmooth_sp16<K, false>), grid, blk, 0, s, a, sp, side, xplus);
  return hipGetLastError();
}
template <int K>
static hipError_t launch_s(const KArgs& a, const SparseT* sp, const double* side, hipStream_t s) {
  const int wpb = waves_per_block(a);
  const size_t ring = (size_t)wpb * 2 * (((a.packed & 1) ? packed_rec_bytes(a.d) : (a.d + a.d * a.d) * 8) + 16);   // dynamic LDS: DMA ring, 2 slots per wave
  const dim3 grid((a.N + wpb - 1) / wpb), blk(64 * wpb);
  if (a.g_index || a.dt || a.f_stride || a.v_tstride || a.w_tstride) hipLaunchKernelGGL((k_smoother_sp16<K, true>), grid, blk, ring, s, a, sp, side);
#ifndef DLM_PIPE_MAX
#define DLM_PIPE_MAX 3072   // up to three waves per SIMD (measured: 1.86 -> 1.78 ms at 2500 series, 3.00 -> 3.17 at 5000)
#endif
  else if ((a.flags & DLM_OPT_NO_STEADY) && a.N > DLM_PIPE_MAX) hipLaunchKernelGGL((k_smoother_sp16<K, false, false, true>), grid, blk, ring, s, a, sp, side);   // every step a full step: the kernel without the machinery
  else if (a.N <= DLM_PIPE_MAX && !(a.flags & DLM_OPT_NO_PIPE)) hipLaunchKernelGGL((k_smoother_sp16<K, false, true>), grid, blk, ring, s, a, sp, side);
  else hipLaunchKernelGGL((k_smoother_sp16<K, false>), grid, blk, ring, s, a, sp, side);
  return hipGetLastError();
}

hipError_t launch_sparse16_filter(const KArgs& a, int K, const SparseT* rows_dev, double* side, double* xplus,
                                  hipStream_t s) {
  switch (K) {
    case 1: return launch_f<1>(a, rows_dev, side, xplus, s);
    case 2: return launch_f<2>(a, rows_dev, side, xplus, s);
    case 3: return launch_f<3>(a, rows_dev, side, xplus, s);
    case 4: return launch_f<4>(a, rows_dev, side, xplus, s);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_sparse16_simsmooth(const KArgs& a, int K, const SparseT* tabs_dev, const double* side,
                                     const double* xplus, hipStream_t s) {
  switch (K) {
    case 1: return launch_ss<1>(a, tabs_dev, side, xplus, s);
    case 2: return launch_ss<2>(a, tabs_dev, side, xplus, s);
    case 3: return launch_ss<3>(a, tabs_dev, side, xplus, s);
    case 4: return launch_ss<4>(a, tabs_dev, side, xplus, s);
  }
  return hipErrorInvalidValue;
}

// ---- shared-covariance launches: the covariance-only runs take `a` with N = 1 and the tables as their record buffers ----------
static KArgs cov_args(const KArgs& a, const CovTabs& tb) {
  KArgs k = a;
  k.N = 1; k.y = nullptr; k.status = nullptr; k.route = nullptr; k.counters = nullptr; k.loglik = nullptr; k.prior = nullptr; k.fq = nullptr;
  k.filt = tb.ftab; k.filt_in = tb.ftab; k.smooth = tb.btab; k.packed = 0; k.m0_stride = 0; k.plain = nullptr;
  return k;
}
template <int K>
static hipError_t launch_cf(const KArgs& a, const SparseT* sp, const CovTabs& tb, hipStream_t s) {
  int* settle = (int*)(tb.cside + 2 * ((size_t)a.T + 1));          // (in the 64 spare doubles behind the side records, covtabs_doubles)
  hipError_t err = hipMemsetD32Async((hipDeviceptr_t)settle, a.T, 1, s);   // T: nothing to fill
  if (err != hipSuccess) return err;
  KArgs k = cov_args(a, tb);
  k.settle_step = (a.flags & DLM_OPT_NO_STEADY) ? nullptr : settle;
  hipLaunchKernelGGL((k_cov_filter_sp16<K>), dim3(1), dim3(64), 0, s, k, sp, tb.cside, tb.ftab + (a.d + a.d * a.d), tb.skip);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  hipLaunchKernelGGL(k_cov_fill_sp16, dim3(a.T + 1), dim3(64), 0, s, tb.ftab, tb.frow / 8, tb.cside, (const int*)settle, a.T);
  return hipGetLastError();
}
template <int K>
static hipError_t launch_cs(const KArgs& a, const SparseT* sp, const CovTabs& tb, hipStream_t s) {
  const size_t ring = 2 * ((size_t)(a.d + a.d * a.d) * 8 + 16);
  hipLaunchKernelGGL((k_cov_smoother_sp16<K>), dim3(1), dim3(64), ring, s, cov_args(a, tb), sp, (const double*)tb.cside, tb.btab);
  return hipGetLastError();
}
template <int K>
static hipError_t launch_mf(const KArgs& a, const SparseT* sp, const CovTabs& tb, hipStream_t s) {
  const int wpb = a.N < 8192 ? 1 : 4, nw = (a.N + 3) / 4;   // four series per wave
  const dim3 grid((nw + wpb - 1) / wpb), blk(64 * wpb);
  const size_t ring = (size_t)wpb * 2 * tb.frow;
#define DLM_MF(MODE) { if (a.d <= 10) hipLaunchKernelGGL((k_mean_filter_sp16<K, 4, MODE>), grid, blk, ring, s, a, sp, tb); \
                      else if (a.d <= 13) hipLaunchKernelGGL((k_mean_filter_sp16<K, 6, MODE>), grid, blk, ring, s, a, sp, tb); \
                      else hipLaunchKernelGGL((k_mean_filter_sp16<K, 8, MODE>), grid, blk, ring, s, a, sp, tb); }
  if (tb.mc && !a.filt) DLM_MF(0)        // dlm_ffbs_batch that keeps no records: the compact means alone (for the shared-factor draw kernel)
  else if (tb.mc) DLM_MF(2)              // fused call: records, and compact means for the backward kernel
  else DLM_MF(1)                         // dlm_filter_batch: records only
#undef DLM_MF
  return hipGetLastError();
}
template <int K>
static hipError_t launch_ms(const KArgs& a, const SparseT* sp, const CovTabs& tb, hipStream_t s) {
  const int wpb = a.N < 8192 ? 1 : 4, nw = (a.N + 3) / 4;
  const dim3 grid((nw + wpb - 1) / wpb), blk(64 * wpb);
  const size_t ring = (size_t)wpb * (2 * ((size_t)tb.brow + 16) + MEAN_AHEAD * 512);
  if (a.d <= 10) hipLaunchKernelGGL((k_mean_smoother_sp16<K, 4, true>), grid, blk, ring, s, a, sp, tb);
  else if (a.d <= 13) hipLaunchKernelGGL((k_mean_smoother_sp16<K, 6, true>), grid, blk, ring, s, a, sp, tb);
  else hipLaunchKernelGGL((k_mean_smoother_sp16<K, 8, true>), grid, blk, ring, s, a, sp, tb);
  return hipGetLastError();
}
#define DLM_K_SWITCH(fn, ...) switch (K) { case 1: return fn<1>(__VA_ARGS__); case 2: return fn<2>(__VA_ARGS__); case 3: return fn<3>(__VA_ARGS__); case 4: return fn<4>(__VA_ARGS__); } return hipErrorInvalidValue;
hipError_t launch_sparse16_cov_filter(const KArgs& a, int K, const SparseT* rows_dev, const CovTabs& tabs, hipStream_t s) { DLM_K_SWITCH(launch_cf, a, rows_dev, tabs, s) }
hipError_t launch_sparse16_cov_smoother(const KArgs& a, int K, const SparseT* cols_dev, const CovTabs& tabs, hipStream_t s) { DLM_K_SWITCH(launch_cs, a, cols_dev, tabs, s) }
hipError_t launch_sparse16_mean_filter(const KArgs& a, int K, const SparseT* rows_dev, const CovTabs& tabs, hipStream_t s) { DLM_K_SWITCH(launch_mf, a, rows_dev, tabs, s) }
hipError_t launch_sparse16_mean_smoother(const KArgs& a, int K, const SparseT* cols_dev, const CovTabs& tabs, hipStream_t s) { DLM_K_SWITCH(launch_ms, a, cols_dev, tabs, s) }
#undef DLM_K_SWITCH

hipError_t launch_sparse16_count_gaps(const KArgs& a, unsigned char* plain, hipStream_t s) {
  hipLaunchKernelGGL(k_count_gaps, dim3((a.N + 3) / 4), dim3(256), 0, s, a.y, a.N, a.T, plain);
  return hipGetLastError();
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
