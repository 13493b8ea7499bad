// One WAVEFRONT per series for multivariate models (16 <= d <= 48, or d <= 15 with p >= 2) with a structured G: the whole
// step lives in the registers of one wave, in the fp64 MFMA accumulator layout, tile by tile.
//
// dlm_tiled.hip gives a series a workgroup of 8 waves that cooperate through LDS operands: 10-14 barrier-separated
// phases per step, each as long as its busiest SIMD (19 % of the fp64 MFMA peak at d = 40, p = 20,
// profiles/r01_pmc_notes.md).  Here a matrix is an array of 16 x 16 tiles held by ONE wave (a 48 x 48 matrix is 72
// VGPRs), and every product is X^T Y over tiles:  Z[a][b] = sum_k X[k][a]^T Y[k][b], where one
// v_mfma_f64_16x16x4_f64 takes register r of an X tile as its A operand and register r of a Y tile as its B operand
// (lane 16 g + c, register r holds element (4 r + g, c): as an A operand that is (rows 4r..4r+3)^T, as a B operand
// rows 4r..4r+3).  No operand ever goes through LDS and no barrier is needed: four series per CU advance independently,
// one per SIMD, and the matrix pipe sees back-to-back independent accumulators.  All matrices of the recursion are
// symmetric or appear next to a symmetric factor, so X^T Y covers every product; the few genuine transposes (mirroring
// the upper tiles of a symmetric result, S = R F -> S^T) and the gather congruence G C G^T go through one wave-private
// LDS image with an odd leading dimension.
//
// Kernels: k_filter_w48 (below), k_smoother_w48 (information-form backward pass, fused call), k_sim_prologue_w48 +
// k_simsmooth_w48 (Durbin-Koopman simulation-smoother FFBS with the Gibbs statistics).  Shapes: 16 <= d <= 48, p <= 32,
// and d <= 15 with 2 <= p <= 32 (one tile per dimension); every LDS region is sized by the instantiation's tile counts.
// With a structured F (SparseF) the products with F are gathers through the image as well.
//
// Forward recursion (KalmanFilter.scala:64-118, 262-294), per step, with Fm = F with the columns of missing observations zeroed:
//   a = G m, R = G C G^T + W dt        two gather passes with the row tables of G (sparse48_analyse)
//   S = R Fm, f = F^T a, Q = Fm^T S + Vm   (Vm: V with unit rows/columns for the missing observations)
//   Qi = Qm^-1                          Newton-Schulz from the previous step's inverse, direct Cholesky otherwise
//   K^T = Qi S^T,  m' = a + K e,  C' = R - K S^T   (upper tiles, mirrored: exactly symmetric)
#include <cstdlib>
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {
namespace w48 {
// The convergence test of the steady-state shortcut (dlm_internal.h: settle_test) as a REAL call: these kernels live at the
// 512-register limit, and inlined the few dozen instructions of the test changed the allocator's choices for the whole kernel
// (k_smoother_w48<3, 2, 2, 1>: 6 -> 296 spilled registers, C4 18.7 -> 23.2 ms, 64 -> 121 ms without the shortcut).
__device__ __attribute__((noinline)) bool settle_eval(float* st, double dmax, double smax, bool reset) {
  if (reset) settle_reset(st);
  const double sc = settle_pow2_inverse_of(smax);
  return settle_test(st, (float)(dmax * sc), (float)(smax * sc), 4);
}


typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int OOB = 0x7ffffff0;
// LDS of one wave, sized by the tile counts of the instantiation (DT tiles for d, PT for p, MX the larger): the image
// (16 MX rows, odd leading dimension: row-wise and transposed reads both spread over the banks), the copy of F or the
// tables of a structured F, the scratch of the direct inverse, ten vectors.  8 KB at one tile per dimension, 35 KB at
// d = 48 / p = 32: the small shapes keep several waves per SIMD resident.
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int il_of(int DT, int PT) { return 16 * cmax(DT, PT) + 1; }
constexpr int img_of(int DT, int PT) { return 16 * cmax(DT, PT) * il_of(DT, PT); }
constexpr int fld_of(int PT) { return 16 * PT + 1; }
constexpr int fimg_of(int DT, int PT) { return cmax(16 * DT * fld_of(PT), 192); }
constexpr int inv_of(int PT) { return 16 * PT * (16 * PT + 1); }
constexpr int vl_of(int DT, int PT) { return 16 * cmax(DT, PT); }
// the scratch of the direct inverse (its second half: the first is the image) sits in the image too when that is large enough
constexpr bool inv_in_img(int DT, int PT) { return img_of(DT, PT) >= 2 * inv_of(PT); }
constexpr int inv_extra(int DT, int PT) { return inv_in_img(DT, PT) ? 0 : inv_of(PT); }
constexpr int lds_doubles(int DT, int PT) { return img_of(DT, PT) + fimg_of(DT, PT) + inv_extra(DT, PT) + 10 * vl_of(DT, PT); }
constexpr int FIMG = 48 * 33;   // (the prologue kernel's own fixed layout)
constexpr int FLD = 33;

__device__ __forceinline__ void wave_sync() {   // LDS hand-off between lanes of one wave (in-order LDS queue)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double sum_g(double v) {   // sum over lanes c, c+16, c+32, c+48; all get it
  unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
  u2 l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  u2 h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  v = __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
  lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
  l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* p, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ double bld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return __hiloint2double((int)v[1], (int)v[0]);
}
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x) {
  const u2 v = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}

// 16-byte stores.  One wave per SIMD can keep at most 63 vector-memory operations in flight: with 8 bytes per lane a step's 39
// record stores are all the bandwidth a wave can ask for.  A lane of the tile layout holds rows 4 r + g of its column;
// v_permlane16_swap of registers r and r + 1 leaves it with two ADJACENT rows of one of them (even d: 16 contiguous bytes).
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bst4(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x, double y) {
  const u4 v = {(unsigned)__double2loint(x), (unsigned)__double2hiint(x), (unsigned)__double2loint(y), (unsigned)__double2hiint(y)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);   // (nontemporal stores measured slower: 22.1 against 18.7 ms at C4)
}
// registers (x: rows 4 r + g, y: rows 4 (r + 1) + g)  ->  (lo, hi) = rows (base, base + 1), base = 4 (r + (g & 1)) + (g & 2)
__device__ __forceinline__ void pair_rows(double x, double y, double& lo, double& hi) {
  const u2 l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const u2 h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  lo = __hiloint2double((int)h[0], (int)l[0]);
  hi = __hiloint2double((int)h[1], (int)l[1]);
}

// Z[a][b] = sum_k X[k][a]^T Y[k][b] over the first `rows` rows of the operands (MFMAs of all-padding k-blocks are
// skipped).  UP: only the tiles a <= b.  Consecutive MFMAs go to different accumulators.
template <int KT, int MT, int NT, bool UP>
__device__ __forceinline__ void mmT(const d4 (&X)[KT][MT], const d4 (&Y)[KT][NT], d4 (&Z)[MT][NT], int rows) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) Z[a][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * k + 4 * r < rows) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b)
            if (!UP || a <= b) Z[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[k][a][r], Y[k][b][r], Z[a][b], 0, 0, 0);
      }
}

// Complete a symmetric matrix from its upper tiles: lower tiles by transposition through the image, diagonal tiles
// averaged with their own transposes (exactly symmetric result).  FULL: all tiles are given; every tile is averaged
// with the transpose of its mirror tile instead.
template <int IL, int MT, bool FULL, bool KEEP = false>   // KEEP: leave the completed matrix in the image
__device__ __forceinline__ void mirror(d4 (&Z)[MT][MT], double* img, int g, int c) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b)
      if (FULL || a <= b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) img[(16 * a + 4 * r + g) * IL + 16 * b + c] = Z[a][b][r];
      }
  wave_sync();
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      // element (4r+g, c) of tile (b, a) of the transpose = element (c, 4r+g) of tile (a, b)
      if (FULL || a == b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[b][a][r] = 0.5 * (Z[b][a][r] + img[(16 * a + c) * IL + 16 * b + 4 * r + g]);
      } else if (a < b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Z[b][a][r] = img[(16 * a + c) * IL + 16 * b + 4 * r + g];
      }
    }
  wave_sync();
  if (KEEP) {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b <= a; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) img[(16 * a + 4 * r + g) * IL + 16 * b + c] = Z[a][b][r];
    wave_sync();
  }
}

// ST = S^T for an (MT x NT)-tile matrix
template <int IL, int MT, int NT>
__device__ __forceinline__ void transpose(const d4 (&S)[MT][NT], d4 (&ST)[NT][MT], double* img, int g, int c) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(16 * a + 4 * r + g) * IL + 16 * b + c] = S[a][b][r];
  wave_sync();
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) ST[b][a][r] = img[(16 * a + c) * IL + 16 * b + 4 * r + g];
  wave_sync();
}

// y[j] = sum_i M[i][j] x[i] for the lanes' columns j = 16 b + c (every g gets it); x is an LDS vector, zero beyond its length
template <int KT, int NT>
__device__ __forceinline__ void matTvec(const d4 (&M)[KT][NT], const double* x, int g, double (&y)[NT]) {
  double xr[KT][4];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[k][r] = x[16 * k + 4 * r + g];
#pragma unroll
  for (int b = 0; b < NT; ++b) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = fma(M[k][b][r], xr[k][r], acc);
    y[b] = sum_g(acc);
  }
}


// F (d x p) is kept in LDS, row-major with leading dimension FLD; both operand forms are read from there when a product
// needs them (they would otherwise hold 96 registers for the whole step)
template <int DT, int PT>
__device__ __forceinline__ void f_tiles(const double* Fl, d4 (&Ft)[DT][PT], int g, int c) {      // F as [d-tiles][p-tiles]
#pragma unroll
  for (int a = 0; a < DT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Ft[a][b][r] = Fl[(16 * a + 4 * r + g) * fld_of(PT) + 16 * b + c];
}
template <int DT, int PT>
__device__ __forceinline__ void ft_tiles(const double* Fl, d4 (&FT)[PT][DT], int g, int c) {     // F^T as [p-tiles][d-tiles]
#pragma unroll
  for (int a = 0; a < PT; ++a)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) FT[a][b][r] = Fl[(16 * b + c) * fld_of(PT) + 16 * a + 4 * r + g];
}
template <int DT, int PT>
__device__ __forceinline__ void load_f_lds(double* Fl, const double* F, int d, int p, int lane) {   // F column-major d x p in global memory
  for (int idx = lane; idx < 16 * DT * 16 * PT; idx += 64) {
    const int i = idx % (16 * DT), j = idx / (16 * DT);
    Fl[i * fld_of(PT) + j] = (i < d && j < p) ? F[i + j * d] : 0.0;
  }
}

// Z = T X T^T (+ Wt dt on the upper tiles) for symmetric X, T given by the row tables (tix, tvl) of the lanes' columns:
//   pass 1   Y[i][j] = sum_s X[i][idx_s(j)] val_s(j)          (= X T^T)
//   pass 2   Z[i][j] = sum_s Y[idx_s(j)][i] val_s(j)          (= (T Y)^T = T X T^T), upper tiles, then mirrored
template <int IL, int DT, int K, bool ADDW, bool KEEP = false>
__device__ __forceinline__ void congruence(d4 (&C)[DT][DT], const d4 (&Wt)[DT][DT], double dt, const int (&tix)[DT][K],
                                           const double (&tvl)[DT][K], double* img, int g, int c) {
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(16 * aa + 4 * r + g) * IL + 16 * b + c] = C[aa][b][r];
  wave_sync();
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* row = img + (16 * aa + 4 * r + g) * IL;
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < K; ++s) s_ = fma(row[tix[b][s]], tvl[b][s], s_);
        C[aa][b][r] = s_;
      }
  wave_sync();
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(16 * aa + 4 * r + g) * IL + 16 * b + c] = C[aa][b][r];
  wave_sync();
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = aa; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * aa + 4 * r + g;
        double s_ = ADDW ? Wt[aa][b][r] * dt : 0.0;
#pragma unroll
        for (int s = 0; s < K; ++s) s_ = fma(img[tix[b][s] * IL + i], tvl[b][s], s_);
        C[aa][b][r] = s_;
      }
  wave_sync();
  mirror<IL, DT, false, KEEP>(C, img, g, c);
}
// y[j] = sum_s x[idx_s(j)] val_s(j) for the lanes' columns (x: LDS vector)
template <int DT, int K>
__device__ __forceinline__ void gather_vec(const double* x, const int (&tix)[DT][K], const double (&tvl)[DT][K], double (&y)[DT]) {
#pragma unroll
  for (int b = 0; b < DT; ++b) {
    double s_ = 0.0;
#pragma unroll
    for (int s = 0; s < K; ++s) s_ = fma(x[tix[b][s]], tvl[b][s], s_);
    y[b] = s_;
  }
}
// Z[a][b] += sum_k X[k][a]^T Y[k][b] on the upper tiles
template <int KT, int MT>
__device__ __forceinline__ void mmT_acc_up(const d4 (&X)[KT][MT], const d4 (&Y)[KT][MT], d4 (&Z)[MT][MT], int rows) {
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (16 * k + 4 * r < rows) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
          for (int b = a; b < MT; ++b) Z[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[k][a][r], Y[k][b][r], Z[a][b], 0, 0, 0);
      }
}

// Products with a structured F as gathers through the image (tables: SparseF).
// Z[j][i] = sum_s val_s(j) X[idx_s(j)][i] (= F^T X) for the rows j = 16 a + 4 r + g of a p-row result; the column
// tables of F are read from LDS (ci: [32][4] ints, cv: [32][4] doubles), X is in the image.
template <int IL, int PT, int NT, int KF>
__device__ __forceinline__ void ft_times_image(d4 (&Z)[PT][NT], const double* img, const int* ci, const double* cv, int g, int c) {
#pragma unroll
  for (int a = 0; a < PT; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * a + 4 * r + g;
      double acc[NT];
#pragma unroll
      for (int b = 0; b < NT; ++b) acc[b] = 0.0;
#pragma unroll
      for (int s = 0; s < KF; ++s) {
        const double* row = img + ci[4 * j + s] * IL;
        const double v = cv[4 * j + s];
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[b] = fma(row[16 * b + c], v, acc[b]);
      }
#pragma unroll
      for (int b = 0; b < NT; ++b) Z[a][b][r] = acc[b];
    }
}
// Z[i][i'] = sum_s X[i][idx_s(i')] val_s(i') (= X F^T) for an (MT x .)-tile X in the image; (ri, rv): the row tables of F
// of the lanes' columns i' = 16 b + c.
template <int IL, int MT, int DT, int KF>
__device__ __forceinline__ void image_times_ft(d4 (&Z)[MT][DT], const double* img, const int (&ri)[DT][KF], const double (&rv)[DT][KF], int g) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* row = img + (16 * a + 4 * r + g) * IL;
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < KF; ++s) s_ = fma(row[ri[b][s]], rv[b][s], s_);
        Z[a][b][r] = s_;
      }
}
template <int IL, int MT, int NT>
__device__ __forceinline__ void to_image(const d4 (&X)[MT][NT], double* img, int g, int c) {
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) img[(16 * a + 4 * r + g) * IL + 16 * b + c] = X[a][b][r];
}
// The tables of a structured F: column tables into LDS (for the row-indexed gathers), row tables of the lanes' columns into registers
template <int DT, int KF>
__device__ __forceinline__ void load_f_tables(const SparseF* spf, int* ci, double* cv, int (&ri)[DT][KF], double (&rv)[DT][KF], int lane, int c) {
  for (int idx = lane; idx < 32 * 4; idx += 64) { ci[idx] = spf->cidx[idx >> 2][idx & 3]; cv[idx] = spf->cval[idx >> 2][idx & 3]; }
#pragma unroll
  for (int b = 0; b < DT; ++b)
#pragma unroll
    for (int s = 0; s < KF; ++s) { ri[b][s] = spf->ridx[16 * b + c][s]; rv[b][s] = spf->rval[16 * b + c][s]; }
}

// Direct inverse of the SPD n x n matrix given as tiles (n <= 16 PT): Cholesky and n triangular solves by ONE wave in
// the LDS image (fallback of the Newton-Schulz refinement: first step, changed missingness pattern).  Returns whether
// a non-positive pivot was met.
template <int PT>
__device__ __forceinline__ bool direct_inverse(const d4 (&Q)[PT][PT], d4 (&X)[PT][PT], int n, double* img, double* B, int lane, int g, int c) {
  constexpr int QL = 16 * PT + 1;
  double* L = img;              // n x n, row-major, leading dimension QL; B: the inverse, column j by lane j
#pragma unroll
  for (int a = 0; a < PT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) L[(16 * a + 4 * r + g) * QL + 16 * b + c] = Q[a][b][r];
  wave_sync();
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    double akk = L[k * QL + k];
    if (!(akk > 0.0)) { bad = true; akk = 1e-300; }
    const double lkk = sqrt(akk), inv = 1.0 / lkk;
    wave_sync();
    if (lane >= k && lane < n) L[lane * QL + k] = (lane == k) ? lkk : L[lane * QL + k] * inv;
    wave_sync();
    const int rr = n - k - 1;
    for (int idx = lane; idx < rr * rr; idx += 64) {
      const int i = k + 1 + idx % rr, j = k + 1 + idx / rr;
      if (i >= j) L[i * QL + j] = fma(-L[i * QL + k], L[j * QL + k], L[i * QL + j]);
    }
    wave_sync();
  }
  if (lane < n) {   // (L L^T) x = e_lane
    double* x = B + lane;   // x[i] at B[i * QL + lane]
    for (int i = 0; i < n; ++i) {
      double s = (i == lane) ? 1.0 : 0.0;
      for (int l = 0; l < i; ++l) s = fma(-L[i * QL + l], x[l * QL], s);
      x[i * QL] = s / L[i * QL + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i * QL];
      for (int l = i + 1; l < n; ++l) s = fma(-L[l * QL + i], x[l * QL], s);
      x[i * QL] = s / L[i * QL + i];
    }
  }
  wave_sync();
#pragma unroll
  for (int a = 0; a < PT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * a + 4 * r + g, j = 16 * b + c;
        // symmetric by construction up to rounding: average the two triangles
        X[a][b][r] = (i < n && j < n) ? 0.5 * (B[i * QL + j] + B[j * QL + i]) : 0.0;
      }
  wave_sync();
  return bad;
}

// X <- (SPD Q)^-1 by Newton-Schulz refinement of the warm start X (E = I - Q X, X <- X + X E, kept exactly symmetric:
// the antisymmetric part of the iterate is a neutral mode, see dlm_tiled.hip), direct inverse when the start is too far
// off or 6 iterations do not reach max|E| <= 2e-10.  Returns whether the direct path met a non-positive pivot.
template <int IL, int PT>
__device__ __forceinline__ bool spd_inverse_warm(const d4 (&Q)[PT][PT], d4 (&X)[PT][PT], int n, bool warm, double* img, double* inv,
                                                 int lane, int g, int c) {
  const double tol = 2e-10;
  bool done = false;
  if (warm) {
    for (int it = 0; it < 6 && !done; ++it) {
      d4 E[PT][PT];
      mmT<PT, PT, PT, false>(Q, X, E, n);
      bool big = false, far = false;
#pragma unroll
      for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * a + 4 * r + g, j = 16 * b + c;
            const double e = ((i == j) ? 1.0 : 0.0) - E[a][b][r];
            E[a][b][r] = e;
            if (i < n && j < n) { big |= !(fabs(e) <= tol); far |= !(fabs(e) * n < 0.5); }
          }
      const bool any_big = __ballot(big) != 0ull, any_far = __ballot(far) != 0ull;
      if (any_far && any_big) break;
      if (!any_big) done = true;   // the update below squares the residual: it lands on the fp64 floor
      d4 D[PT][PT];
      mmT<PT, PT, PT, false>(X, E, D, n);
#pragma unroll
      for (int a = 0; a < PT; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) X[a][b] += D[a][b];
      mirror<IL, PT, true>(X, img, g, c);
    }
  }
  if (done) return false;
  return direct_inverse<PT>(Q, X, n, img, inv, lane, g, c);
}

__device__ __forceinline__ bool chol_rows(double* L, int n, int ld, int lane) {   // in-place lower Cholesky, row-major
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    double akk = L[k * ld + k];
    if (!(akk > 0.0)) { bad = true; akk = 1e-300; }
    const double lkk = sqrt(akk), inv = 1.0 / lkk;
    wave_sync();
    if (lane >= k && lane < n) L[lane * ld + k] = (lane == k) ? lkk : L[lane * ld + k] * inv;
    wave_sync();
    const int rr = n - k - 1;
    for (int idx = lane; idx < rr * rr; idx += 64) {
      const int i = k + 1 + idx % rr, j = k + 1 + idx / rr;
      if (i >= j) L[i * ld + j] = fma(-L[i * ld + k], L[j * ld + k], L[i * ld + j]);
    }
    wave_sync();
  }
  return bad;
}


// ---------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------
// KF > 0: F is structured (at most KF nonzeros per column, SparseF): S = R F, f = F^T a and Q = F^T S are gathers through
// the image instead of MFMA products (90 of the 190 MFMAs of a step at d = 40, p = 20).
template <int DT, int PT, int K, int KF>
__global__ __launch_bounds__(64, (DT <= 2 && PT == 1 && KF <= 1) ? 2 : 1) void k_filter_w48(KArgs a, double* innov, int zero_m0) {
  if (a.settle_step) __builtin_amdgcn_s_setprio(3);   // the series of zeros of a shared-factor table: one wave beside the batch's filter, and the table waits for it
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int IL = il_of(DT, PT), IMG = img_of(DT, PT), FIMG = fimg_of(DT, PT), VL = vl_of(DT, PT), QL = 16 * PT + 1;
  double* img = sm;        double* Fl = sm + IMG;  double* vec0 = Fl + FIMG + inv_extra(DT, PT);   double* inv = inv_in_img(DT, PT) ? img + inv_of(PT) : Fl + FIMG;
  double* mv = vec0;  double* av = mv + VL;   double* ev = av + VL;   double* ob = ev + VL;
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8;
  int st = 0;
  for (int i = lane; i < 10 * VL; i += 64) mv[i] = 0.0;

  bool jd[DT], jp[PT];      // this lane's column 16 b + c lies inside d / p
  int cpart[DT];            // byte offset of C[0][16 b + c] inside a record
#pragma unroll
  for (int b = 0; b < DT; ++b) { jd[b] = 16 * b + c < d; cpart[b] = (d + (16 * b + c) * d) * 8; }
#pragma unroll
  for (int b = 0; b < PT; ++b) jp[b] = 16 * b + c < p;

  const double* V = a.V + (size_t)n * a.v_stride;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  d4 C[DT][DT], Wt[DT][DT], Vt[PT][PT], Qi[PT][PT];
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
        C[aa][b][r] = (i < d && jd[b]) ? C0[i + j * d] : 0.0;
      }
  auto load_W = [&](const double* Wp, int g, int c) {   // upper tiles of W (the rest is never read)
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
          Wt[aa][b][r] = (i < d && jd[b] && aa <= b) ? Wp[i + j * d] : 0.0;
        }
  };
  auto load_V = [&](const double* Vp, int g, int c) {
#pragma unroll
    for (int aa = 0; aa < PT; ++aa)
#pragma unroll
      for (int b = 0; b < PT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
          Vt[aa][b][r] = (i < p && jp[b] && aa <= b) ? Vp[i + j * p] : 0.0;
        }
  };
  load_W(W, g, c);
  load_V(V, g, c);
#pragma unroll
  for (int aa = 0; aa < PT; ++aa)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) Qi[aa][b][r] = 0.0;
  load_f_lds<DT, PT>(Fl, a.F, d, p, lane);
  int tix[DT][K];
  double tvl[DT][K];
  int gcur = -1;
  auto load_tables = [&](int gi) {
    const SparseBig* tab = a.spb + 2 * gi;     // the ROWS of G
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) { tix[b][s] = tab->idx[16 * b + c][s]; tvl[b][s] = tab->val[16 * b + c][s]; }
    gcur = gi;
  };
  load_tables(a.g_index ? a.g_index[0] : 0);
  constexpr int KFA = KF > 0 ? KF : 1;
  int fix[PT][KFA];          // nonzeros of the lanes' columns of F
  double fvl[PT][KFA];
#pragma unroll
  for (int b = 0; b < PT; ++b)
#pragma unroll
    for (int s = 0; s < KFA; ++s) {
      fix[b][s] = KF > 0 ? a.spf->cidx[16 * b + c][s] : 0;
      fvl[b][s] = KF > 0 ? a.spf->cval[16 * b + c][s] : 0.0;
    }
  if (lane < d) mv[lane] = zero_m0 ? 0.0 : (a.m0 + (size_t)n * a.m0_stride)[lane];   // zero_m0: the simulation smoother filters y* from a zero prior mean

  double* out = a.filt ? a.filt + (size_t)n * (T + 1) * rec : nullptr;
  const __amdgpu_buffer_rsrc_t rfo = mk_rsrc(out, out ? (size_t)(T + 1) * recb : 0);   // zero-sized: stores are dropped
  double* pri = a.prior ? a.prior + (size_t)n * (T + 1) * rec : nullptr;               // optional (a_t, R_t) records
  const __amdgpu_buffer_rsrc_t rpr = mk_rsrc(pri, pri ? (size_t)(T + 1) * recb : 0);
  const int frec = p + p * p;
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * frec : nullptr;                     // optional (f_t, Q_t) records
  const __amdgpu_buffer_rsrc_t rfq = mk_rsrc(fq, fq ? (size_t)(T + 1) * frec * 8 : 0);
  if (fq) for (int i = lane; i < frec; i += 64) fq[i] = __builtin_nan("");
  const double* y = a.y + (size_t)n * T * p;
  const __amdgpu_buffer_rsrc_t ry = mk_rsrc(y, (size_t)T * p * 8);
  double* es = innov ? innov + (size_t)n * T * p : nullptr;
  const __amdgpu_buffer_rsrc_t res = mk_rsrc(es, es ? (size_t)T * p * 8 : 0);
  int yoff[PT], moff[DT];
#pragma unroll
  for (int b = 0; b < PT; ++b) yoff[b] = jp[b] ? (16 * b + c) * 8 : OOB;
#pragma unroll
  for (int b = 0; b < DT; ++b) moff[b] = (jd[b] && g == 0) ? (16 * b + c) * 8 : OOB;
  wave_sync();

  const bool d_even = (d & 1) == 0;
  auto store_record = [&](const __amdgpu_buffer_rsrc_t& rfo, const double* mv, int t, int g, int c) {
    const int so = t * recb;
    if (d_even) {
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            double lo, hi;
            pair_rows(C[aa][b][r], C[aa][b][r + 1], lo, hi);
            const int i = 16 * aa + 4 * (r + (g & 1)) + (g & 2);
            bst4(rfo, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, lo, hi);
          }
    } else {
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g;
          bst(rfo, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, C[aa][b][r]);
        }
    }
#pragma unroll
    for (int b = 0; b < DT; ++b) bst(rfo, moff[b], so, mv[16 * b + c]);
  };
  auto store_mean = [&](const __amdgpu_buffer_rsrc_t& rfo, const double* mv, int t, int c) {
    const int so = t * recb;
#pragma unroll
    for (int b = 0; b < DT; ++b) bst(rfo, moff[b], so, mv[16 * b + c]);
  };
  // a call whose records nobody reads beyond the means (KArgs::keep_cov): steady steps then store the mean alone
  const bool mean_only_out = __builtin_amdgcn_readfirstlane((int)(a.keep_cov != nullptr && a.keep_cov[n] == 0)) != 0;
  store_record(rfo, mv, 0, g, c);
  if (pri) store_record(rpr, mv, 0, g, c);

  double ynext[PT];
#pragma unroll
  for (int b = 0; b < PT; ++b) ynext[b] = bld(ry, T > 0 ? yoff[b] : OOB, 0);
  bool warm = false;
  double ll = 0.0;   // prediction-error log-likelihood of the series (every lane holds it)
  // Steady state (time-invariant model, regular grid, every component observed): the covariance recursion converges -- within 30
  // steps for the C4 model -- and from then on R, Q and the gain are those of the step before: a step is a = G m, e = y - F^T a,
  // m = a + K e (K^T parked in the image, which a steady step does not use otherwise) and the record store, C from the registers.
  // Tested every fourth step against the record written the step before (re-read from L2: a copy of C would cost 72 registers
  // this kernel does not have); a missing component takes the full path again.  marks[t] = 1 tells the backward pass that
  // C_t is C_{t-1} (one byte per record behind the innovations).
  const bool may_settle = out && !a.g_index && !a.dt && !a.f_stride && !a.v_tstride && !a.w_tstride && !pri && !fq && !a.loglik &&
                          !(a.flags & DLM_OPT_NO_STEADY);
  unsigned char* marks = innov ? (unsigned char*)(innov + (size_t)a.N * T * p) + (size_t)n * (T + 1) : nullptr;
  if (marks && lane == 0) marks[0] = 0;
  bool steady = false;
  float* settle = (float*)(vec0 + 9 * VL);   // state of the steady-state test (dlm_internal.h: settle_test), in a vector slot this kernel does not use
  wave_sync();
  settle_reset(settle);
  unsigned nsteady = 0;                       // steady steps taken (KArgs::counters[0])
  bool sreset = false;                        // the convergence test starts over at its next evaluation
  int tleave = T;                             // the step at which a series of a records-free call left for k_steady_filter_w48 (KArgs::leave_step; written after
                                              // the loop: a store inside it cost this kernel 400 spilled registers)

  for (int t = 0; t < T; ++t) {
    // opaque copies of the lane coordinates: the compiler would otherwise hoist the few dozen address computations of
    // the step out of the time loop and hold them in registers the matrices need
    int g_ = g, c_ = c;
    asm volatile("" : "+v"(g_), "+v"(c_));
    {
    const int g = g_, c = c_;
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) load_tables(gi);
    if (a.f_stride) { wave_sync(); load_f_lds<DT, PT>(Fl, a.F + (size_t)t * a.f_stride, d, p, lane); wave_sync(); }
    if (a.v_tstride) load_V(V + (size_t)t * a.v_tstride, g, c);   // time-varying variances (StudentTGibbs.scala:100-136, DlmFsvSystem.scala:137-208)
    if (a.w_tstride) load_W(W + (size_t)t * a.w_tstride, g, c);
    double ycur[PT];
#pragma unroll
    for (int b = 0; b < PT; ++b) { ycur[b] = ynext[b]; ynext[b] = bld(ry, t + 1 < T ? yoff[b] : OOB, (t + 1 < T ? t + 1 : 0) * p * 8); }
    bool missing_here = false;
#pragma unroll
    for (int b = 0; b < PT; ++b) missing_here |= jp[b] && !(ycur[b] == ycur[b]);
    bool all = __ballot(missing_here) == 0ull;             // every component of y_t observed
    if (__builtin_amdgcn_readfirstlane((int)(steady && all))) {
      if (a.settle_step) {   // the series of zeros of a shared-factor table: record t is the last one anyone needs
        if (lane == 0) *a.settle_step = t;
        if (a.ktab) for (int i = lane; i < 16 * PT * IL; i += 64) a.ktab[i] = img[i];   // K^T of the steady steps (k_steady_filter_w48)
        break;
      }
      if (mean_only_out && a.ktab) { tleave = t; break; }   // a series without a gap of a records-free call: k_steady_filter_w48 takes its means from record t on
      // a stretch of steady steps as a loop of its own: one back edge, so that the wait for the next observation counts the
      // stores behind it (vmcnt(42)) -- at the head of the big loop, where two paths meet, it would be vmcnt(0): every step
      // would sit out the latency of its 39 record stores.  (The flags are read through readfirstlane: a branch the compiler
      // takes for divergent would make t a vector register and every buffer store a waterfall loop.)
      do {
      double an[DT];
      gather_vec<DT, K>(mv, tix, tvl, an);                          // a = G m
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) av[16 * b + c] = an[b];
      wave_sync();
      double fcol[PT];
      if (KF > 0) {
#pragma unroll
        for (int b = 0; b < PT; ++b) {
          double s_ = 0.0;
#pragma unroll
          for (int s = 0; s < KFA; ++s) s_ = fma(av[fix[b][s]], fvl[b][s], s_);
          fcol[b] = s_;
        }
      } else {
        d4 Fm[DT][PT];
        f_tiles<DT, PT>(Fl, Fm, g, c);
        matTvec<DT, PT>(Fm, av, g, fcol);
      }
#pragma unroll
      for (int b = 0; b < PT; ++b) {
        const double e = ycur[b] - fcol[b];
        bst(res, g == 0 ? yoff[b] : OOB, t * p * 8, e);
        if (g == 0) ev[16 * b + c] = jp[b] ? e : 0.0;
      }
      wave_sync();
#pragma unroll
      for (int b = 0; b < DT; ++b) {                                // K e: K^T[j][i] waits at img[j * IL + i]; this lane takes j = g, g + 4, ...
        double s_ = 0.0;
#pragma unroll
        for (int jj = 0; jj < 4 * PT; ++jj) s_ = fma(img[(4 * jj + g) * IL + 16 * b + c], ev[4 * jj + g], s_);   // (rows beyond p are zero padding: a compile-time trip count lets the LDS reads go out together)
        s_ = sum_g(s_);
        if (g == 0 && jd[b]) mv[16 * b + c] = av[16 * b + c] + s_;
      }
      wave_sync();
      if (mean_only_out) store_mean(rfo, mv, t + 1, c); else store_record(rfo, mv, t + 1, g, c);
      if (marks && lane == 0) marks[t + 1] = 1;
      ++nsteady;
      if (++t >= T) break;
      missing_here = false;
#pragma unroll
      for (int b = 0; b < PT; ++b) {
        ycur[b] = ynext[b]; ynext[b] = bld(ry, t + 1 < T ? yoff[b] : OOB, (t + 1 < T ? t + 1 : 0) * p * 8);
        missing_here |= jp[b] && !(ycur[b] == ycur[b]);
      }
      all = __ballot(missing_here) == 0ull;
      } while (all);
      if (t >= T) break;     // (otherwise: step t, whose observation is in ycur, takes the full path below)
    }
    steady = false;
    if (!all) sreset = true;                    // a missing component disturbs the covariance: the convergence test starts over
    if (marks && lane == 0) marks[t + 1] = 0;

    // ---- advance: a = G m, R = G C G^T + W dt (into C's registers), dt == 0: identity
    if (dt != 0.0) {
      double an[DT];
      gather_vec<DT, K>(mv, tix, tvl, an);                         // a = G m
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) av[16 * b + c] = an[b];
      congruence<IL, DT, K, true, (KF > 0)>(C, Wt, dt, tix, tvl, img, g, c);    // R = G C G^T + W dt (KF: R stays in the image)
    } else {
      if (lane < d) av[lane] = mv[lane];
      if (KF > 0) {
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * aa + 4 * r + g) * IL + 16 * b + c] = C[aa][b][r];
      }
      wave_sync();
    }
    // (C now holds R)
    if (pri) store_record(rpr, av, t + 1, g, c);

    // ---- forecast
    double obs[PT];   // 1.0: component 16 b + c observed
    bool anyobs = false;
#pragma unroll
    for (int b = 0; b < PT; ++b) { const bool o = jp[b] && (ycur[b] == ycur[b]); obs[b] = o ? 1.0 : 0.0; anyobs |= o; }
    const bool any = __ballot(anyobs) != 0ull;
    d4 Fm[DT][PT];
    double fcol[PT];
    if (KF > 0) {
#pragma unroll
      for (int b = 0; b < PT; ++b) {
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < KFA; ++s) s_ = fma(av[fix[b][s]], fvl[b][s], s_);
        fcol[b] = s_;
      }
    } else {
      f_tiles<DT, PT>(Fl, Fm, g, c);
      matTvec<DT, PT>(Fm, av, g, fcol);
    }
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const double e = ycur[b] - fcol[b];                       // NaN = missing
      bst(res, g == 0 ? yoff[b] : OOB, t * p * 8, e);
      if (g == 0) { ev[16 * b + c] = obs[b] != 0.0 ? e : 0.0; ob[16 * b + c] = obs[b]; }
    }
    wave_sync();

    d4 S[DT][PT], Q[PT][PT], ST[PT][DT];
    if (any || fq) {   // S = R F and Q = F^T R F + V with the FULL F (the forecast record wants them unmasked)
      if (KF > 0) {
        // S[i][j] = sum_s R[i][idx_s(j)] val_s(j); R is in the image
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double* row = img + (16 * aa + 4 * r + g) * IL;
              double s_ = 0.0;
#pragma unroll
              for (int s = 0; s < KFA; ++s) s_ = fma(row[fix[b][s]], fvl[b][s], s_);
              S[aa][b][r] = s_;
            }
        wave_sync();
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * aa + 4 * r + g) * IL + 16 * b + c] = S[aa][b][r];
        wave_sync();
        // Q[l][j] = sum_s S[idx_s(j)][l] val_s(j) (= (S^T F)[l][j], symmetric), upper tiles; S^T read from the same image
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = aa; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int l = 16 * aa + 4 * r + g;
              double s_ = 0.0;
#pragma unroll
              for (int s = 0; s < KFA; ++s) s_ = fma(img[fix[b][s] * IL + l], fvl[b][s], s_);
              Q[aa][b][r] = s_;
            }
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) ST[b][aa][r] = img[(16 * aa + c) * IL + 16 * b + 4 * r + g] * __shfl(obs[b], 4 * r + g);   // row 16 b + 4 r + g of S^T: the mask of that observation sits in lane 4 r + g
        wave_sync();
      } else {
        mmT<DT, DT, PT, false>(C, Fm, S, d);                     // R F (R symmetric)
        mmT<DT, PT, PT, true>(Fm, S, Q, d);                      // F^T R F, upper tiles
      }
#pragma unroll
      for (int aa = 0; aa < PT; ++aa)
#pragma unroll
        for (int b = aa; b < PT; ++b) Q[aa][b] += Vt[aa][b];
      mirror<IL, PT, false>(Q, img, g, c);
      if (fq) {
        const int so = (t + 1) * frec * 8;
#pragma unroll
        for (int b = 0; b < PT; ++b) bst(rfq, g == 0 ? yoff[b] : OOB, so, fcol[b]);
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g;
              bst(rfq, (i < p && jp[b]) ? (p + i + (16 * b + c) * p) * 8 : OOB, so, Q[aa][b][r]);
            }
      }
    }
    if (!any) {   // updateState without an observation: m = a, C = R
      if (lane < d) mv[lane] = av[lane];
    } else {
      // missing components: their columns of S and rows / columns of Q leave the update (unit diagonal in Q)
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < PT; ++b) S[aa][b] *= obs[b];
#pragma unroll
      for (int aa = 0; aa < PT; ++aa)
#pragma unroll
        for (int b = 0; b < PT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
            Q[aa][b][r] = (__shfl(obs[aa], 4 * r + g) != 0.0 && obs[b] != 0.0) ? Q[aa][b][r] : ((i == j && jp[b]) ? 1.0 : 0.0);
          }
      // a warm start from a different missingness pattern is too far off anyway: the residual test sends it to the
      // direct inverse
      if (spd_inverse_warm<IL, PT>(Q, Qi, p, warm, img, inv, lane, g, c)) st |= DLM_ST_NOT_PD;
      warm = true;
      if (a.loglik) {   // -1/2 (n_obs log 2 pi + log det Qm + e^T Qm^-1 e) (KalmanFilter.scala:138-153); det from a one-wave Cholesky of Qm
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * aa + 4 * r + g) * QL + 16 * b + c] = Q[aa][b][r];
        wave_sync();
        if (chol_rows(img, p, QL, lane)) st |= DLM_ST_NOT_PD;
        double ucol[PT];
        matTvec<PT, PT>(Qi, ev, g, ucol);                        // Qm^-1 e
        double part = 0.0;
#pragma unroll
        for (int b = 0; b < PT; ++b)
          if (g == 0 && jp[b]) part += 2.0 * log(img[(16 * b + c) * QL + 16 * b + c]) + ev[16 * b + c] * ucol[b] + 1.8378770664093453 * obs[b];
        for (int o_ = 32; o_ > 0; o_ >>= 1) part += __shfl_xor(part, o_);
        ll -= 0.5 * part;
        wave_sync();
      }
      if (KF == 0) transpose<IL, DT, PT>(S, ST, img, g, c);
      d4 KT[PT][DT];
      mmT<PT, PT, DT, false>(Qi, ST, KT, p);                     // K^T = Qi S^T (Qi symmetric)
      double kcol[DT];
      matTvec<PT, DT>(KT, ev, g, kcol);                          // K e
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) mv[16 * b + c] = av[16 * b + c] + kcol[b];
      d4 U[DT][DT];
      mmT<PT, DT, DT, true>(KT, ST, U, p);                       // K S^T, upper tiles
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = aa; b < DT; ++b) C[aa][b] -= U[aa][b];
      mirror<IL, DT, false>(C, img, g, c);
      if (may_settle && all && (t & 3) == 3) {   // has the covariance stopped moving?  record t holds C_{t-1}
        double dmax = 0.0, cmax = 0.0;
        const int so = t * recb;
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g;
              const double old = bld(rfo, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so);
              dmax = fmax(dmax, settle_absdiff(C[aa][b][r], old)); cmax = fmax(cmax, fabs(C[aa][b][r]));
            }
        for (int o_ = 32; o_ > 0; o_ >>= 1) { dmax = fmax(dmax, __shfl_xor(dmax, o_)); cmax = fmax(cmax, __shfl_xor(cmax, o_)); }
        const bool settled = settle_eval(settle, dmax, cmax, sreset);   // the geometric tail of the changes within DLM_SETTLE_TOL max|C| (a scalar flag)
        sreset = false;
        if (settled) {
          steady = true;
          wave_sync();
          to_image<IL, PT, DT>(KT, img, g, c);                      // K^T for the steady steps
        }
      }
    }
    wave_sync();
    store_record(rfo, mv, t + 1, g, c);
    }
  }
  if (a.loglik && lane == 0) a.loglik[n] = ll;
  if (a.leave_step && lane == 0) a.leave_step[n] = tleave;
  if (a.counters && lane == 0 && nsteady) atomicAdd(&a.counters[0], (unsigned long long)nsteady);
  bool bad = false;
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) bad |= !isfinite(C[aa][b][r]);
  if (lane < d) bad |= !isfinite(mv[lane]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}


// ---------------------------------------------------------------------------------------
// backward pass (information form, general p; Smoothing.scala:31-64 restated as in dlm_tiled.hip k_smoother_tiled):
//   K = C F Vm^-1,  Qm^-1 = Vm^-1 - Vm^-1 F^T K                         (Vm^-1 cached while the missingness pattern repeats)
//   s_t = m + C q,  S_t = C - C P C
//   r = q + F (Qm^-1 e - K^T q),  M = P + F X F^T - F (P K)^T - (P K) F^T,  X = Qm^-1 + K^T P K
//   q_{t-1} = G^T r,  P_{t-1} = G^T M G                                   (gathers with the COLUMN tables of G)
// innov: e_t = y_t - f_t of the forward pass (NaN = missing), the observation of record t at innov[(t-1) p ..].
// ---------------------------------------------------------------------------------------
// KF > 0: structured F -- F^T C, F^T K, X F^T and the three F-products of the update of P are gathers (220 of the 505 MFMAs
// of a step at d = 40, p = 20).
template <int DT, int PT, int K, int KF>
__global__ __launch_bounds__(64, (DT <= 2 && PT == 1 && KF <= 1) ? 2 : 1) void k_smoother_w48(KArgs a, const double* __restrict__ innov) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int IL = il_of(DT, PT), IMG = img_of(DT, PT), FIMG = fimg_of(DT, PT), VL = vl_of(DT, PT);
  double* img = sm;        double* Fl = sm + IMG;  double* vec0 = Fl + FIMG + inv_extra(DT, PT);   double* inv = inv_in_img(DT, PT) ? img + inv_of(PT) : Fl + FIMG;
  double* mv = vec0;  double* qv = mv + VL;   double* rv = qv + VL;   double* ev = rv + VL;   double* ob = ev + VL;
  double* tv = ob + VL;
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8;
  int st = 0;
  for (int i = lane; i < 10 * VL; i += 64) mv[i] = 0.0;

  bool jd[DT], jp[PT];
  int cpart[DT];
#pragma unroll
  for (int b = 0; b < DT; ++b) { jd[b] = 16 * b + c < d; cpart[b] = (d + (16 * b + c) * d) * 8; }
#pragma unroll
  for (int b = 0; b < PT; ++b) jp[b] = 16 * b + c < p;
  const double* V = a.V + (size_t)n * a.v_stride;
  constexpr int KFA = KF > 0 ? KF : 1;
  int* fci = (int*)Fl;            // structured F: its column tables live where the dense copy of F would
  double* fcv = Fl + 64;
  int frix[DT][KFA];
  double frvl[DT][KFA];
  if (KF > 0) load_f_tables<DT, KFA>(a.spf, fci, fcv, frix, frvl, lane, c);
  else load_f_lds<DT, PT>(Fl, a.F, d, p, lane);

  int tix[DT][K];
  double tvl[DT][K];
  int gcur = -1;
  auto load_tables = [&](int gi) {
    const SparseBig* tab = a.spb + 2 * gi + 1;     // the COLUMNS of G: rows of G^T
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) { tix[b][s] = tab->idx[16 * b + c][s]; tvl[b][s] = tab->val[16 * b + c][s]; }
    gcur = gi;
  };

  const __amdgpu_buffer_rsrc_t rin = mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const __amdgpu_buffer_rsrc_t rout = mk_rsrc(a.smooth + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const __amdgpu_buffer_rsrc_t rinn = mk_rsrc(innov + (size_t)n * T * p, (size_t)T * p * 8);
  int eoff[PT], moff[DT];
#pragma unroll
  for (int b = 0; b < PT; ++b) eoff[b] = jp[b] ? (16 * b + c) * 8 : OOB;
#pragma unroll
  for (int b = 0; b < DT; ++b) moff[b] = jd[b] ? (16 * b + c) * 8 : OOB;

  // Vm^-1 (cached while the missingness pattern repeats): with a structured F its home is the LDS behind F's tables -- 32 registers
  // the full step does not have to spare (a dense F fills that LDS with its own copy: Vm^-1 stays in registers there)
  constexpr int VS = 16 * PT + 1;
  double* vimg = Fl + 192;
  constexpr bool VI_LDS = KF > 0 && 16 * PT * VS <= FIMG - 192;
  d4 C[DT][DT], P[DT][DT], Vi[PT][PT];
  auto vi_load = [&](d4 (&X)[PT][PT], int g, int c) {
#pragma unroll
    for (int aa = 0; aa < PT; ++aa)
#pragma unroll
      for (int b = 0; b < PT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[aa][b][r] = vimg[(16 * aa + 4 * r + g) * VS + 16 * b + c];
  };
  auto vi_matvec = [&](const double* x, int g, int c, double (&y)[PT]) {   // y = Vm^-1 x for the lanes' columns (symmetric)
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      double acc = 0.0;
#pragma unroll
      for (int i = 0; i < 4 * PT; ++i) acc = fma(vimg[(4 * i + g) * VS + 16 * b + c], x[4 * i + g], acc);
      y[b] = sum_g(acc);
    }
  };
  double mcol[DT], ecol[PT], obsP[PT];
  auto request = [&](int t, int g, int c, bool cov) {   // record t into C / mcol (cov = false: its mean alone, C stays), the innovation of record t into ecol
    const int so = t * recb;
    if (cov) {
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g;
          C[aa][b][r] = bld(rin, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so);
        }
    }
#pragma unroll
    for (int b = 0; b < DT; ++b) mcol[b] = bld(rin, moff[b], so);
#pragma unroll
    for (int b = 0; b < PT; ++b) ecol[b] = bld(rinn, t > 0 ? eoff[b] : OOB, (t > 0 ? t - 1 : 0) * p * 8);
  };
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b) P[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int aa = 0; aa < PT; ++aa)
#pragma unroll
    for (int b = 0; b < PT; ++b) Vi[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
  if (VI_LDS) for (int i = lane; i < 16 * PT * VS; i += 64) vimg[i] = 0.0;
#pragma unroll
  for (int b = 0; b < PT; ++b) obsP[b] = -1.0;
  request(T, g, c, true);
  wave_sync();

  // Steady state (structured F, time-invariant model).  The forward pass marks the records whose covariance is the one before
  // (marks[t] = 1: C_t is C_{t-1}, every component observed).  Such a record is requested as its mean alone (320 B instead of
  // 13 KB at d = 40) and C stays in the registers.  Once P has stopped moving as well (tested every fourth step against a copy
  // parked in the not yet written output record t - 1: there are no 72 spare registers), S_t = S_{t+1} waits in the image, and a
  // step is s = m + C q, the record store, and the q recursion without K or Qm^-1:
  //   Qm^-1 e - K^T q = v - Vm^-1 F^T C (F v + q),  v = Vm^-1 e        (K = C F Vm^-1, Qm^-1 = Vm^-1 - Vm^-1 F^T K)
  // -- matrix-vector products with the tiles that are in the registers anyway (C, Vm^-1) and gathers through the tables of F and G.
  const unsigned char* marks = (const unsigned char*)(innov + (size_t)a.N * T * p) + (size_t)n * (T + 1);
  double* xv = tv + VL;   double* zv = xv + VL;
  const bool can_steady = KF > 0 && !a.g_index && !a.f_stride && !a.v_tstride && !(a.flags & DLM_OPT_NO_STEADY);
  bool smode = false, psteady = false, inh_next = false;
  float* settle = (float*)(vec0 + 9 * VL);   // state of the steady-state test (dlm_internal.h: settle_test), in a vector slot this kernel does not use
  wave_sync();
  settle_reset(settle);
  unsigned nsteady = 0;                       // steady steps taken (KArgs::counters[1])
  bool sreset = false;                        // the convergence test starts over at its next evaluation
  int mk_cur = can_steady ? marks[T] : 0;

  for (int t = T; t >= 0; --t) {
    int g_ = g, c_ = c;   // opaque copies, see k_filter_w48
    asm volatile("" : "+v"(g_), "+v"(c_));
    {
    const int g = g_, c = c_;
    if (a.f_stride && t > 0) { wave_sync(); load_f_lds<DT, PT>(Fl, a.F + (size_t)(t - 1) * a.f_stride, d, p, lane); }
    const bool inherit = inh_next;                                   // C_t is C_{t+1}: record t came as its mean alone
    const bool mk = __builtin_amdgcn_readfirstlane(mk_cur) != 0;     // C_{t-1} is C_t
    inh_next = mk;
    mk_cur = (can_steady && t >= 2) ? marks[t - 1] : 0;              // for the next step: travels during this one
    double obs[PT];
    bool anyobs = false, changed = false, miss = false;
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const bool o = t > 0 && jp[b] && (ecol[b] == ecol[b]);
      obs[b] = o ? 1.0 : 0.0; anyobs |= o; changed |= jp[b] && obs[b] != obsP[b]; miss |= jp[b] && !o;
      if (g == 0) { ev[16 * b + c] = o ? ecol[b] : 0.0; ob[16 * b + c] = obs[b]; }
    }
#pragma unroll
    for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) mv[16 * b + c] = mcol[b];
    const bool any = __ballot(anyobs) != 0ull;
    const bool allobs = __ballot(miss) == 0ull;
    wave_sync();

    if constexpr (KF > 0) {
    if (__builtin_expect(smode && inherit && allobs, 1)) {
      // ---- steady step
      double cq[DT], vcol[PT];
      matTvec<DT, DT>(C, qv, g, cq);                              // C q
      if constexpr (VI_LDS) vi_matvec(ev, g, c, vcol); else matTvec<PT, PT>(Vi, ev, g, vcol);   // v = Vm^-1 e
      const int so = t * recb;
      if ((d & 1) == 0) {   // 16 bytes per lane: rows (i, i + 1) of the lane's column, i = 16 aa + 8 h + 2 g
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int i = 16 * aa + 8 * h + 2 * g;
              bst4(rout, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, img[i * IL + 16 * b + c], img[(i + 1) * IL + 16 * b + c]);
            }
      } else {
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g;
            bst(rout, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, img[i * IL + 16 * b + c]);
          }
      }
#pragma unroll
      for (int b = 0; b < DT; ++b) bst(rout, g == 0 ? moff[b] : OOB, so, mcol[b] + cq[b]);
      request(t - 1, g, c, !mk);
#pragma unroll
      for (int b = 0; b < PT; ++b) if (g == 0) xv[16 * b + c] = jp[b] ? vcol[b] : 0.0;
      wave_sync();
      double fv[DT];
      gather_vec<DT, KFA>(xv, frix, frvl, fv);                    // F v
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0) zv[16 * b + c] = jd[b] ? qv[16 * b + c] + fv[b] : 0.0;
      wave_sync();
      double zc[DT];
      matTvec<DT, DT>(C, zv, g, zc);                              // C (F v + q)
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0) rv[16 * b + c] = jd[b] ? zc[b] : 0.0;
      wave_sync();
#pragma unroll
      for (int b = 0; b < PT; ++b) {                              // F^T (.) through the column tables of F
        const int j = 16 * b + c;
        double s_ = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < KFA; ++s2) s_ = fma(rv[fci[4 * j + s2]], fcv[4 * j + s2], s_);
        if (g == 0) xv[j] = jp[b] ? s_ : 0.0;
      }
      wave_sync();
      double t2[PT];
      if constexpr (VI_LDS) vi_matvec(xv, g, c, t2); else matTvec<PT, PT>(Vi, xv, g, t2);       // Vm^-1 F^T C (F v + q)
#pragma unroll
      for (int b = 0; b < PT; ++b) if (g == 0) tv[16 * b + c] = jp[b] ? vcol[b] - t2[b] : 0.0;
      wave_sync();
      double ftv[DT];
      gather_vec<DT, KFA>(tv, frix, frvl, ftv);                   // F (Qm^-1 e - K^T q)
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) rv[16 * b + c] = qv[16 * b + c] + ftv[b];
      wave_sync();
      double qn[DT];
      gather_vec<DT, K>(rv, tix, tvl, qn);                        // q = G^T r
      wave_sync();
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) qv[16 * b + c] = qn[b];
      wave_sync();
      ++nsteady;
      continue;
    }
    }
    smode = false;
    if (!(inherit && allobs)) { psteady = false; sreset = true; }

    d4 Kg[DT][PT];
    if (any) {
      if (__ballot(changed) != 0ull || a.v_tstride) {   // Vm^-1 for this missingness pattern (and this step's V_t)
        const double* Vr = V + (size_t)(t - 1) * a.v_tstride;   // V of the observation at record t
        d4 Vm[PT][PT];
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
              Vm[aa][b][r] = (i < p && jp[b]) ? ((__shfl(obs[aa], 4 * r + g) != 0.0 && obs[b] != 0.0) ? Vr[i + j * p] : (i == j ? 1.0 : 0.0)) : 0.0;
            }
        // a diagonal V_m (the usual case: independent measurement errors) inverts entry by entry -- with partially missing
        // observations the pattern changes at nearly every step, and the direct inverse is p sequential pivots each time
        bool offd = false;
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) offd |= (16 * aa + 4 * r + g != 16 * b + c) && Vm[aa][b][r] != 0.0;
        if (__ballot(offd) == 0ull) {
#pragma unroll
          for (int aa = 0; aa < PT; ++aa)
#pragma unroll
            for (int b = 0; b < PT; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool dg = (16 * aa + 4 * r + g == 16 * b + c) && jp[b];
                if (dg && !(Vm[aa][b][r] > 0.0)) st |= DLM_ST_NOT_PD;
                Vi[aa][b][r] = dg ? 1.0 / Vm[aa][b][r] : 0.0;
              }
        } else if (direct_inverse<PT>(Vm, Vi, p, img, inv, lane, g, c)) st |= DLM_ST_NOT_PD;
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (!(__shfl(obs[aa], 4 * r + g) != 0.0 && obs[b] != 0.0)) Vi[aa][b][r] = 0.0;
        if constexpr (VI_LDS) {
#pragma unroll
          for (int aa = 0; aa < PT; ++aa)
#pragma unroll
            for (int b = 0; b < PT; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) vimg[(16 * aa + 4 * r + g) * VS + 16 * b + c] = Vi[aa][b][r];
          wave_sync();
        }
#pragma unroll
        for (int b = 0; b < PT; ++b) obsP[b] = obs[b];
      }
      d4 CFT[PT][DT];
      if (KF > 0) {
        to_image<IL, DT, DT>(C, img, g, c);
        wave_sync();
        ft_times_image<IL, PT, DT, KFA>(CFT, img, fci, fcv, g, c);  // F^T C
        wave_sync();
      } else {
        d4 Ft[DT][PT];
        f_tiles<DT, PT>(Fl, Ft, g, c);
        mmT<DT, PT, DT, false>(Ft, C, CFT, d);                  // F^T C
      }
      if constexpr (VI_LDS) {
        d4 Vt[PT][PT];
        vi_load(Vt, g, c);
        mmT<PT, DT, PT, false>(CFT, Vt, Kg, p);                 // K = C F Vm^-1
      } else
      mmT<PT, DT, PT, false>(CFT, Vi, Kg, p);                   // K = C F Vm^-1
    }
    // the products that need C come first: C, x1 and x2 are gone before the p-sized quantities are built
    double cq[DT];
    matTvec<DT, DT>(C, qv, g, cq);                              // C q
    {
      d4 x1[DT][DT], x2[DT][DT];
      mmT<DT, DT, DT, false>(P, C, x1, d);                      // P C
      mmT<DT, DT, DT, true>(C, x1, x2, d);                      // C P C, upper tiles
      // s_t = m + C q ; S_t = C - C P C: an upper tile goes to both triangles of the record
      const int so = t * recb;
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = aa; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
            const double v = C[aa][b][r] - x2[aa][b][r];
            const bool in = i < d && jd[b];
            bst(rout, in ? cpart[b] + i * 8 : OOB, so, v);
            if (aa != b) bst(rout, in ? (d + j + i * d) * 8 : OOB, so, v);
          }
#pragma unroll
      for (int b = 0; b < DT; ++b) bst(rout, g == 0 ? moff[b] : OOB, so, mcol[b] + cq[b]);
    }
    if (t == 0) break;
    request(t - 1, g, c, true);   // C is free: the next record travels during the rest of the step (a full step always takes the whole record:
                                  // keeping C through the recursion below would cost it 72 registers)
    const bool pcheck = can_steady && inherit && allobs && !psteady && (t & 3) == 1;
    if (pcheck) {   // P_t parked in the output record t - 1 (written for real in the next step)
      const int so = (t - 1) * recb;
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g;
            bst(rout, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, P[aa][b][r]);
          }
    }

    const int gi = a.g_index ? a.g_index[t - 1] : 0;   // G of the step INTO record t
    if (gi != gcur) load_tables(gi);
    if (any) {
      d4 X[PT][PT], NPKT[PT][DT];
      {
        d4 Qi[PT][PT];
        {
          d4 X0[PT][PT];
          if (KF > 0) {
            to_image<IL, DT, PT>(Kg, img, g, c);
            wave_sync();
            ft_times_image<IL, PT, PT, KFA>(X0, img, fci, fcv, g, c);   // F^T K
            wave_sync();
          } else {
            d4 Ft[DT][PT];
            f_tiles<DT, PT>(Fl, Ft, g, c);
            mmT<DT, PT, PT, false>(Ft, Kg, X0, d);              // F^T K
          }
          if constexpr (VI_LDS) {
            d4 Vt[PT][PT];
            vi_load(Vt, g, c);
            mmT<PT, PT, PT, true>(Vt, X0, Qi, p);               // Vm^-1 F^T K, upper tiles
#pragma unroll
            for (int aa = 0; aa < PT; ++aa)
#pragma unroll
              for (int b = aa; b < PT; ++b) Qi[aa][b] = Vt[aa][b] - Qi[aa][b];
          } else {
          mmT<PT, PT, PT, true>(Vi, X0, Qi, p);                 // Vm^-1 F^T K, upper tiles
#pragma unroll
          for (int aa = 0; aa < PT; ++aa)
#pragma unroll
            for (int b = aa; b < PT; ++b) Qi[aa][b] = Vi[aa][b] - Qi[aa][b];
          }
        }
        mirror<IL, PT, false>(Qi, img, g, c);                       // Qm^-1
        double ucol[PT], ktq[PT];
        matTvec<PT, PT>(Qi, ev, g, ucol);                       // u = Qm^-1 e
        matTvec<DT, PT>(Kg, qv, g, ktq);                        // K^T q
#pragma unroll
        for (int b = 0; b < PT; ++b) if (g == 0) tv[16 * b + c] = jp[b] ? ucol[b] - ktq[b] : 0.0;
        d4 PK[DT][PT];
        mmT<DT, DT, PT, false>(P, Kg, PK, d);                   // P K
        mmT<DT, PT, PT, true>(Kg, PK, X, d);                    // K^T P K, upper tiles
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = aa; b < PT; ++b) X[aa][b] += Qi[aa][b];
        mirror<IL, PT, false>(X, img, g, c);
        if (KF > 0) {
          // T2 = (P K) F^T (all tiles), then P -= T2 + T2^T on the upper tiles
          to_image<IL, DT, PT>(PK, img, g, c);
          wave_sync();                                          // (also publishes tv)
          d4 T2[DT][DT];
          image_times_ft<IL, DT, DT, KFA>(T2, img, frix, frvl, g);
          wave_sync();
          to_image<IL, DT, DT>(T2, img, g, c);
          wave_sync();
#pragma unroll
          for (int aa = 0; aa < DT; ++aa)
#pragma unroll
            for (int b = aa; b < DT; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) P[aa][b][r] -= T2[aa][b][r] + img[(16 * b + c) * IL + 16 * aa + 4 * r + g];
          wave_sync();
        } else {
          transpose<IL, DT, PT>(PK, NPKT, img, g, c);               // (also publishes tv)
        }
      }
      double ftv[DT];
      if (KF > 0) {
        gather_vec<DT, KFA>(tv, frix, frvl, ftv);               // F (u - K^T q)
#pragma unroll
        for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) rv[16 * b + c] = qv[16 * b + c] + ftv[b];
        // T1 = F X F^T: (X F^T) by columns, then F (X F^T) read as its own transpose (symmetric), upper tiles
        to_image<IL, PT, PT>(X, img, g, c);
        wave_sync();
        d4 FXT[PT][DT];
        image_times_ft<IL, PT, DT, KFA>(FXT, img, frix, frvl, g);  // X F^T
        wave_sync();
        to_image<IL, PT, DT>(FXT, img, g, c);
        wave_sync();
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = aa; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g;
              double s_ = 0.0;
#pragma unroll
              for (int s2 = 0; s2 < KFA; ++s2) s_ = fma(img[frix[b][s2] * IL + i], frvl[b][s2], s_);
              P[aa][b][r] += s_;
            }
        wave_sync();
      } else {
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b) NPKT[aa][b] = -NPKT[aa][b];
        d4 FT[PT][DT];
        ft_tiles<DT, PT>(Fl, FT, g, c);
        matTvec<PT, DT>(FT, tv, g, ftv);                        // F (u - K^T q)
#pragma unroll
        for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) rv[16 * b + c] = qv[16 * b + c] + ftv[b];
        d4 FXT[PT][DT];
        mmT<PT, PT, DT, false>(X, FT, FXT, p);                  // X F^T = (F X)^T
        mmT_acc_up<PT, DT>(FXT, FT, P, p);                      // + (F X) F^T
        mmT_acc_up<PT, DT>(FT, NPKT, P, p);                     // - F (P K)^T
        mmT_acc_up<PT, DT>(NPKT, FT, P, p);                     // - (P K) F^T
      }
      // P's lower tiles are stale now: the congruence below reads the image written from the mirrored matrix
      mirror<IL, DT, false>(P, img, g, c);
    } else {
      if (lane < d) rv[lane] = qv[lane];
    }
    wave_sync();
    double qn[DT];
    gather_vec<DT, K>(rv, tix, tvl, qn);                        // q = G^T r
    wave_sync();
#pragma unroll
    for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) qv[16 * b + c] = qn[b];
    congruence<IL, DT, K, false>(P, P, 0.0, tix, tvl, img, g, c);   // P = G^T M G
    if (pcheck) {   // has P stopped moving?
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      double dmax = 0.0, pmax = 0.0;
      const int so = (t - 1) * recb;
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g;
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rout, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, 1);   // glc
            const double old = __hiloint2double((int)v[1], (int)v[0]);
            dmax = fmax(dmax, settle_absdiff(P[aa][b][r], old)); pmax = fmax(pmax, fabs(P[aa][b][r]));
          }
      for (int o_ = 32; o_ > 0; o_ >>= 1) { dmax = fmax(dmax, __shfl_xor(dmax, o_)); pmax = fmax(pmax, __shfl_xor(pmax, o_)); }
      psteady = settle_eval(settle, dmax, pmax, sreset);   // the geometric tail of the changes within DLM_SETTLE_TOL max|P|
      sreset = false;
    }
    if (can_steady && psteady && inherit && allobs && mk) {   // the next step can be a steady one: S_t (stored above) into the image
      wave_sync();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int so = t * recb;
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g;
            const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rout, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, 1);
            img[i * IL + 16 * b + c] = __hiloint2double((int)v[1], (int)v[0]);
          }
      wave_sync();
      smode = true;
    }
    }
  }
  bool bad = false;
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) bad |= !isfinite(P[aa][b][r]);
  if (lane < d) bad |= !isfinite(qv[lane]);
  if (a.counters && lane == 0 && nsteady) atomicAdd(&a.counters[1], (unsigned long long)nsteady);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}


// ---------------------------------------------------------------------------------------
// second half of the simulation smoother: mean-only backward pass on y*, theta = s* + x+, Gibbs statistics
// (Gibbs.scala:23-78, GibbsWishart.scala:16-35) on the fly.  No covariance recursion:
//   theta_t = m*_t + C_t q_t + x+_t
//   q_{t-1} = G^T [ q + F (Qm^-1 e - K^T q) ],  K = C F Vm^-1,  Qm^-1 = Vm^-1 - Vm^-1 F^T K
// innov: the innovations of the forward pass on y* (NaN = missing), record t at innov[(t-1) p ..].
// ---------------------------------------------------------------------------------------
template <int DT, int PT, int K, int KF>
__global__ __launch_bounds__(64, (DT <= 2 && PT == 1 && KF <= 1) ? 2 : 1) void k_simsmooth_w48(KArgs a, const double* __restrict__ xplus, const double* __restrict__ innov) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int IL = il_of(DT, PT), IMG = img_of(DT, PT), FIMG = fimg_of(DT, PT), VL = vl_of(DT, PT);
  double* img = sm;        double* Fl = sm + IMG;  double* vec0 = Fl + FIMG + inv_extra(DT, PT);   double* inv = inv_in_img(DT, PT) ? img + inv_of(PT) : Fl + FIMG;
  double* qv = vec0;  double* rv = qv + VL;   double* ev = rv + VL;   double* ob = ev + VL;
  double* tv = ob + VL;    double* thc = tv + VL;  double* thn = thc + VL; double* dfv = thn + VL;
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0, stats = a.stats != nullptr;
  int st = 0;
  for (int i = lane; i < 10 * VL; i += 64) qv[i] = 0.0;

  bool jd[DT], jp[PT];
  int cpart[DT];
#pragma unroll
  for (int b = 0; b < DT; ++b) { jd[b] = 16 * b + c < d; cpart[b] = (d + (16 * b + c) * d) * 8; }
#pragma unroll
  for (int b = 0; b < PT; ++b) jp[b] = 16 * b + c < p;
  const double* V = a.V + (size_t)n * a.v_stride;
  constexpr int KFA = KF > 0 ? KF : 1;
  int* fci = (int*)Fl;            // structured F: its column tables live where the dense copy of F would
  double* fcv = Fl + 64;
  int frix[DT][KFA], fix[PT][KFA];
  double frvl[DT][KFA], fvl[PT][KFA];
  if (KF > 0) {
    load_f_tables<DT, KFA>(a.spf, fci, fcv, frix, frvl, lane, c);
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int s = 0; s < KFA; ++s) { fix[b][s] = a.spf->cidx[16 * b + c][s]; fvl[b][s] = a.spf->cval[16 * b + c][s]; }
  } else load_f_lds<DT, PT>(Fl, a.F, d, p, lane);

  int tix[DT][K], rix[DT][K];       // columns of G (q = G^T r) / rows of G (the system residual of the statistics)
  double tvl[DT][K], rvl[DT][K];
  int gcur = -1, rcur = -1;
  auto load_cols = [&](int gi) {
    const SparseBig* tab = a.spb + 2 * gi + 1;
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) { tix[b][s] = tab->idx[16 * b + c][s]; tvl[b][s] = tab->val[16 * b + c][s]; }
    gcur = gi;
  };
  auto load_rows = [&](int gi) {
    const SparseBig* tab = a.spb + 2 * gi;
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) { rix[b][s] = tab->idx[16 * b + c][s]; rvl[b][s] = tab->val[16 * b + c][s]; }
    rcur = gi;
  };
#pragma unroll
  for (int b = 0; b < DT; ++b)
#pragma unroll
    for (int s = 0; s < K; ++s) { rix[b][s] = 0; rvl[b][s] = 0.0; }

  const __amdgpu_buffer_rsrc_t rin = mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const __amdgpu_buffer_rsrc_t rinn = mk_rsrc(innov + (size_t)n * T * p, (size_t)T * p * 8);
  const __amdgpu_buffer_rsrc_t rxp = mk_rsrc(xplus + (size_t)n * (T + 1) * d, (size_t)(T + 1) * d * 8);
  const __amdgpu_buffer_rsrc_t ryo = mk_rsrc(a.y + (size_t)n * T * p, (size_t)T * p * 8);
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  const __amdgpu_buffer_rsrc_t rth = mk_rsrc(thout, thout ? (size_t)(T + 1) * d * 8 : 0);
  int eoff[PT], moff[DT];
#pragma unroll
  for (int b = 0; b < PT; ++b) eoff[b] = jp[b] ? (16 * b + c) * 8 : OOB;
#pragma unroll
  for (int b = 0; b < DT; ++b) moff[b] = jd[b] ? (16 * b + c) * 8 : OOB;

  d4 C[DT][DT], Vi[PT][PT], OUT[DT][DT];
  double mcol[DT], xcol[DT], ecol[PT], ycol[PT], obsP[PT], ssd[DT], ssy[PT], nob[PT];
  auto request = [&](int t, int g, int c) {   // record t, x+_t, innovation and original observation of record t
    const int so = t * recb;
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g;
          C[aa][b][r] = bld(rin, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so);
        }
#pragma unroll
    for (int b = 0; b < DT; ++b) { mcol[b] = bld(rin, moff[b], so); xcol[b] = bld(rxp, moff[b], t * d * 8); }
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      ecol[b] = bld(rinn, t > 0 ? eoff[b] : OOB, (t > 0 ? t - 1 : 0) * p * 8);
      ycol[b] = bld(ryo, t > 0 ? eoff[b] : OOB, (t > 0 ? t - 1 : 0) * p * 8);
    }
  };
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b) OUT[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int aa = 0; aa < PT; ++aa)
#pragma unroll
    for (int b = 0; b < PT; ++b) Vi[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int b = 0; b < PT; ++b) { obsP[b] = -1.0; ssy[b] = 0.0; nob[b] = 0.0; }
#pragma unroll
  for (int b = 0; b < DT; ++b) ssd[b] = 0.0;
  request(T, g, c);
  wave_sync();

  for (int t = T; t >= 0; --t) {
    int g_ = g, c_ = c;   // opaque copies, see k_filter_w48
    asm volatile("" : "+v"(g_), "+v"(c_));
    {
    const int g = g_, c = c_;
    if (a.f_stride && t > 0) { wave_sync(); load_f_lds<DT, PT>(Fl, a.F + (size_t)(t - 1) * a.f_stride, d, p, lane); }
    double obs[PT];
    bool anyobs = false, changed = false;
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const bool o = t > 0 && jp[b] && (ecol[b] == ecol[b]);
      obs[b] = o ? 1.0 : 0.0; anyobs |= o; changed |= jp[b] && obs[b] != obsP[b];
      if (g == 0) { ev[16 * b + c] = o ? ecol[b] : 0.0; ob[16 * b + c] = obs[b]; }
    }
    const bool any = __ballot(anyobs) != 0ull;
    // theta_t = m*_t + C_t q_t + x+_t
    double cq[DT], th[DT];
    matTvec<DT, DT>(C, qv, g, cq);
#pragma unroll
    for (int b = 0; b < DT; ++b) {
      th[b] = mcol[b] + cq[b] + xcol[b];
      bst(rth, g == 0 ? moff[b] : OOB, t * d * 8, th[b]);
      if (g == 0 && jd[b]) thc[16 * b + c] = th[b];
    }
    wave_sync();
    d4 Ft[DT][PT];
    if (KF == 0 && (any || stats)) f_tiles<DT, PT>(Fl, Ft, g, c);
    if (stats) {
      if (t < T) {   // system residual theta_{t+1} - G_{t+1} theta_t, scaled by 1 / sqrt(dt)
        const int gn = a.g_index ? a.g_index[t] : 0;
        const double dtn = a.dt ? a.dt[t] : 1.0;
        if (gn != rcur) load_rows(gn);
        double gth[DT];
        gather_vec<DT, K>(thc, rix, rvl, gth);
        const double dts = (dtn == 0.0) ? 1.0 : dtn, isd = 1.0 / sqrt(dts);
#pragma unroll
        for (int b = 0; b < DT; ++b) {
          const double s_ = jd[b] ? thn[16 * b + c] - (dtn == 0.0 ? th[b] : gth[b]) : 0.0;
          ssd[b] = fma(s_, s_ / dts, ssd[b]);
          if (g == 0) dfv[16 * b + c] = s_ * isd;
        }
        if (outer) {
          wave_sync();
#pragma unroll
          for (int aa = 0; aa < DT; ++aa)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double di = dfv[16 * aa + 4 * r + g];
#pragma unroll
              for (int b = 0; b < DT; ++b) OUT[aa][b][r] = fma(di, dfv[16 * b + c], OUT[aa][b][r]);
            }
        }
      }
      if (t > 0) {   // observation residual of theta_t against the ORIGINAL y_t
        double fth[PT];
        if (KF > 0) gather_vec<PT, KFA>(thc, fix, fvl, fth);
        else matTvec<DT, PT>(Ft, thc, g, fth);
#pragma unroll
        for (int b = 0; b < PT; ++b)
          if (jp[b] && ycol[b] == ycol[b]) { const double r_ = ycol[b] - fth[b]; ssy[b] = fma(r_, r_, ssy[b]); nob[b] += 1.0; }
      }
    }
    wave_sync();
    if (lane < d) thn[lane] = thc[lane];
    if (t == 0) break;

    const int gi = a.g_index ? a.g_index[t - 1] : 0;   // G of the step INTO record t
    const double dtt = a.dt ? a.dt[t - 1] : 1.0;
    if (gi != gcur) load_cols(gi);
    if (any) {
      if (__ballot(changed) != 0ull) {   // Vm^-1 for this missingness pattern
        d4 Vm[PT][PT];
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
              Vm[aa][b][r] = (i < p && jp[b]) ? ((__shfl(obs[aa], 4 * r + g) != 0.0 && obs[b] != 0.0) ? V[i + j * p] : (i == j ? 1.0 : 0.0)) : 0.0;
            }
        bool offd = false;   // a diagonal V_m inverts entry by entry (see k_smoother_w48)
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) offd |= (16 * aa + 4 * r + g != 16 * b + c) && Vm[aa][b][r] != 0.0;
        if (__ballot(offd) == 0ull) {
#pragma unroll
          for (int aa = 0; aa < PT; ++aa)
#pragma unroll
            for (int b = 0; b < PT; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const bool dg = (16 * aa + 4 * r + g == 16 * b + c) && jp[b];
                if (dg && !(Vm[aa][b][r] > 0.0)) st |= DLM_ST_NOT_PD;
                Vi[aa][b][r] = dg ? 1.0 / Vm[aa][b][r] : 0.0;
              }
        } else if (direct_inverse<PT>(Vm, Vi, p, img, inv, lane, g, c)) st |= DLM_ST_NOT_PD;
#pragma unroll
        for (int aa = 0; aa < PT; ++aa)
#pragma unroll
          for (int b = 0; b < PT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (!(__shfl(obs[aa], 4 * r + g) != 0.0 && obs[b] != 0.0)) Vi[aa][b][r] = 0.0;
#pragma unroll
        for (int b = 0; b < PT; ++b) obsP[b] = obs[b];
      }
      d4 Kg[DT][PT];
      {
        d4 CFT[PT][DT];
        if (KF > 0) {
          to_image<IL, DT, DT>(C, img, g, c);
          wave_sync();
          request(t - 1, g, c);                                 // C is free
          ft_times_image<IL, PT, DT, KFA>(CFT, img, fci, fcv, g, c);   // F^T C
          wave_sync();
        } else {
          mmT<DT, PT, DT, false>(Ft, C, CFT, d);                // F^T C
          request(t - 1, g, c);                                 // C is free
        }
        mmT<PT, DT, PT, false>(CFT, Vi, Kg, p);                 // K = C F Vm^-1
      }
      d4 Qi[PT][PT];
      {
        d4 X0[PT][PT];
        if (KF > 0) {
          to_image<IL, DT, PT>(Kg, img, g, c);
          wave_sync();
          ft_times_image<IL, PT, PT, KFA>(X0, img, fci, fcv, g, c);    // F^T K
          wave_sync();
        } else {
          mmT<DT, PT, PT, false>(Ft, Kg, X0, d);                // F^T K
        }
        mmT<PT, PT, PT, true>(Vi, X0, Qi, p);
      }
#pragma unroll
      for (int aa = 0; aa < PT; ++aa)
#pragma unroll
        for (int b = aa; b < PT; ++b) Qi[aa][b] = Vi[aa][b] - Qi[aa][b];
      mirror<IL, PT, false>(Qi, img, g, c);                         // Qm^-1
      double ucol[PT], ktq[PT];
      matTvec<PT, PT>(Qi, ev, g, ucol);                         // Qm^-1 e
      matTvec<DT, PT>(Kg, qv, g, ktq);                          // K^T q
#pragma unroll
      for (int b = 0; b < PT; ++b) if (g == 0) tv[16 * b + c] = jp[b] ? ucol[b] - ktq[b] : 0.0;
      wave_sync();
      double ftv[DT];
      if (KF > 0) gather_vec<DT, KFA>(tv, frix, frvl, ftv);      // F (Qm^-1 e - K^T q)
      else {
        d4 FT[PT][DT];
        ft_tiles<DT, PT>(Fl, FT, g, c);
        matTvec<PT, DT>(FT, tv, g, ftv);
      }
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) rv[16 * b + c] = qv[16 * b + c] + ftv[b];
    } else {
      request(t - 1, g, c);
      if (lane < d) rv[lane] = qv[lane];
    }
    wave_sync();
    if (dtt != 0.0) {
      double qn[DT];
      gather_vec<DT, K>(rv, tix, tvl, qn);                      // q = G^T r
      wave_sync();
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) qv[16 * b + c] = qn[b];
    } else {
      if (lane < d) qv[lane] = rv[lane];
    }
    wave_sync();
    }
  }
  bool bad = (lane < d) && !isfinite(thn[lane]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (stats) {
    const int L = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * L;
#pragma unroll
    for (int b = 0; b < PT; ++b) if (g == 0 && jp[b]) { so[16 * b + c] = ssy[b]; so[p + 16 * b + c] = nob[b]; }
    if (outer) {
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
            if (i < d && jd[b]) so[2 * p + i + j * d] = OUT[aa][b][r];
          }
    } else {
#pragma unroll
      for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) so[2 * p + 16 * b + c] = ssd[b];
    }
    if (lane == 0) so[L - 1] = (double)T;
  }
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}


// ---------------------------------------------------------------------------------------
// prologue of the simulation smoother: simulate (x+, y+) from the model and leave y* = y - y+ (Dlm.simulateRegular,
// Dlm.scala:245-292, on the filter's time grid: a zero increment is an identity advance without noise).  One wave per
// series; chol(W), chol(V) row-major in LDS with odd leading dimensions, G x by the row tables of G, F^T x by the column
// tables of a structured F (KF > 0) or from the LDS copy of F.  Normals: (seed, series, record t, i), i < d state noise
// (record 0: the initial state), d <= i < d + p observation noise, or a.z [N][T+1][d+p] when given.
// ---------------------------------------------------------------------------------------
template <int K, int KF>
__global__ __launch_bounds__(64) void k_sim_prologue_w48(KArgs a, double* __restrict__ xplus, double* __restrict__ ystar) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T;
  const int ldw = d | 1, ldv = p | 1;
  double* Lw = sm;               double* Lv = Lw + 48 * 49;     double* Fl = Lv + 32 * 33;
  double* x = Fl + (KF > 0 ? 0 : FIMG);   double* xn = x + 48;   double* z = xn + 48;   // z: d + p <= 80 normals
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * (d + p) : nullptr;
  const double* yin = a.y + (size_t)n * T * p;
  double* xo = xplus + (size_t)n * (T + 1) * d;
  double* yo = ystar + (size_t)n * T * p;
  int st = 0;
  // x+_0 = m0 + chol(C0) z_0 (chol(C0) built where chol(W) goes afterwards)
  for (int idx = lane; idx < d * d; idx += 64) { const int i = idx % d, j = idx / d; Lw[i * ldw + j] = C0[idx]; }
  for (int i = lane; i < d; i += 64) z[i] = zin ? zin[i] : philox_normal(a.seed, series, 0u, (unsigned)i);
  wave_sync();
  if (chol_rows(Lw, d, ldw, lane)) st |= DLM_ST_NOT_PD;
  if (lane < d) {
    double acc = m0[lane];
    for (int k = 0; k <= lane; ++k) acc = fma(Lw[lane * ldw + k], z[k], acc);
    x[lane] = acc; xo[lane] = acc;
  }
  wave_sync();
  for (int idx = lane; idx < d * d; idx += 64) { const int i = idx % d, j = idx / d; Lw[i * ldw + j] = W[idx]; }
  for (int idx = lane; idx < p * p; idx += 64) { const int i = idx % p, j = idx / p; Lv[i * ldv + j] = V[idx]; }
  if (KF == 0) load_f_lds<3, 2>(Fl, a.F, d, p, lane);
  wave_sync();
  if (chol_rows(Lw, d, ldw, lane)) st |= DLM_ST_NOT_PD;
  if (chol_rows(Lv, p, ldv, lane)) st |= DLM_ST_NOT_PD;
  // tables of this lane's state row / observation column
  int gix[K], fix[KF > 0 ? KF : 1];
  double gvl[K], fvl[KF > 0 ? KF : 1];
  int gcur = -1;
  const int li = lane < 48 ? lane : 0, lj = lane < 32 ? lane : 0;
#pragma unroll
  for (int s = 0; s < (KF > 0 ? KF : 1); ++s) { fix[s] = KF > 0 ? a.spf->cidx[lj][s] : 0; fvl[s] = KF > 0 ? a.spf->cval[lj][s] : 0.0; }
  for (int t = 0; t < T; ++t) {
    const double dt = a.dt ? a.dt[t] : 1.0, sdt = sqrt(dt);
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) {
      const SparseBig* tab = a.spb + 2 * gi;
#pragma unroll
      for (int s = 0; s < K; ++s) { gix[s] = tab->idx[li][s]; gvl[s] = tab->val[li][s]; }
      gcur = gi;
    }
    if (KF == 0 && a.f_stride) { wave_sync(); load_f_lds<3, 2>(Fl, a.F + (size_t)t * a.f_stride, d, p, lane); }
    if (a.w_tstride) {   // W_t of the transition into record t + 1 (DlmFsvSystem.scala:137-208 feeds one W per step): its factor, d pivots by one wave
      wave_sync();
      const double* Wt = W + (size_t)t * a.w_tstride;
      for (int idx = lane; idx < d * d; idx += 64) { const int i = idx % d, j = idx / d; Lw[i * ldw + j] = Wt[idx]; }
      wave_sync();
      if (chol_rows(Lw, d, ldw, lane)) st |= DLM_ST_NOT_PD;
    }
    for (int i = lane; i < d + p; i += 64)
      z[i] = zin ? zin[(size_t)(t + 1) * (d + p) + i] : philox_normal(a.seed, series, (unsigned)(t + 1), (unsigned)i);
    wave_sync();
    if (lane < d) {
      double acc = 0.0;
#pragma unroll
      for (int s = 0; s < K; ++s) acc = fma(x[gix[s]], gvl[s], acc);                    // G x
      double nz = 0.0;
      for (int k = 0; k <= lane; ++k) nz = fma(Lw[lane * ldw + k], z[k], nz);          // chol(W) z
      const double v = (dt == 0.0) ? x[lane] : fma(nz, sdt, acc);
      xn[lane] = v;
      xo[(size_t)(t + 1) * d + lane] = v;
    }
    wave_sync();
    if (lane < p) {
      double acc = 0.0;
      if (KF > 0) {
#pragma unroll
        for (int s = 0; s < (KF > 0 ? KF : 1); ++s) acc = fma(xn[fix[s]], fvl[s], acc);   // F^T x
      } else {
        for (int k = 0; k < d; ++k) acc = fma(Fl[k * FLD + lane], xn[k], acc);
      }
      for (int k = 0; k <= lane; ++k) acc = fma(Lv[lane * ldv + k], z[d + k], acc);     // + chol(V) z
      yo[(size_t)t * p + lane] = yin[(size_t)t * p + lane] - acc;                        // NaN stays NaN
    }
    if (lane < d) x[lane] = xn[lane];
  }
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// The steady steps of k_filter_w48 as a kernel of their own (records-free pooled FFBS): once the covariance
// recursion has settled a step is a = G m, e = y - F^T a, m = a + K e with the gain of the step before -- a few dozen registers,
// where the filter holds its matrices in 476.  The series without a gap leave k_filter_w48 at the step it settles (the same
// step for all of them: the recursion does not see the data) and continue here, several waves per SIMD, with K^T as the series
// of zeros left it (KArgs::ktab) and the operations of the filter's steady branch one for one.  Only the means are stored.
// ---------------------------------------------------------------------------------------
template <int DT, int PT, int K, int KF>
__global__ __launch_bounds__(64, 4) void k_steady_filter_w48(KArgs a, const double* __restrict__ ktab, const int* __restrict__ settle) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int IL = il_of(DT, PT), FIMG = fimg_of(DT, PT), VL = vl_of(DT, PT), KIMG = 16 * PT * IL;
  double* img = sm;  double* Fl = sm + KIMG;  double* mv = Fl + (KF > 0 ? 0 : FIMG);  double* av = mv + VL;  double* ev = av + VL;
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  if (!a.keep_cov || a.keep_cov[n] != 0) return;            // the series with a gap ran the whole filter
  const int d = a.d, p = a.p, T = a.T, rec = d + d * d, recb = rec * 8;
  // the record this series left k_filter_w48 at (its own convergence test).  It is the zero series' step (settle[0]) -- the recursion does
  // not see the data and both ran the same kernel -- but nothing here depends on that: the means continue from the series' own record,
  // with the zero series' steady gain (the two agree within DLM_SETTLE_TOL even if the steps should ever differ)
  const int t0 = a.leave_step ? a.leave_step[n] : settle[0];
  if (t0 >= T) return;                                       // this series did not settle: k_filter_w48 went all the way
  if (settle[0] >= T) {                                      // (cannot happen: the series left, but the zero series has no steady gain to give)
    if (a.status && lane == 0) atomicOr(&a.status[n], DLM_ST_NOCONV | DLM_ST_NONFINITE);
    return;
  }
  for (int i = lane; i < 3 * VL; i += 64) mv[i] = 0.0;
  bool jd[DT], jp[PT];
#pragma unroll
  for (int b = 0; b < DT; ++b) jd[b] = 16 * b + c < d;
#pragma unroll
  for (int b = 0; b < PT; ++b) jp[b] = 16 * b + c < p;
  if (KF == 0) load_f_lds<DT, PT>(Fl, a.F, d, p, lane);
  for (int i = lane; i < KIMG; i += 64) img[i] = ktab[i];
  int tix[DT][K];
  double tvl[DT][K];
  {
    const SparseBig* tab = a.spb;     // the ROWS of G (time-invariant here)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) { tix[b][s] = tab->idx[16 * b + c][s]; tvl[b][s] = tab->val[16 * b + c][s]; }
  }
  constexpr int KFA = KF > 0 ? KF : 1;
  int fix[PT][KFA];
  double fvl[PT][KFA];
#pragma unroll
  for (int b = 0; b < PT; ++b)
#pragma unroll
    for (int s = 0; s < KFA; ++s) {
      fix[b][s] = KF > 0 ? a.spf->cidx[16 * b + c][s] : 0;
      fvl[b][s] = KF > 0 ? a.spf->cval[16 * b + c][s] : 0.0;
    }
  double* out = a.filt + (size_t)n * (T + 1) * rec;
  const __amdgpu_buffer_rsrc_t rfo = mk_rsrc(out, (size_t)(T + 1) * recb);
  const double* y = a.y + (size_t)n * T * p;
  const __amdgpu_buffer_rsrc_t ry = mk_rsrc(y, (size_t)T * p * 8);
  int yoff[PT], moff[DT];
#pragma unroll
  for (int b = 0; b < PT; ++b) yoff[b] = jp[b] ? (16 * b + c) * 8 : OOB;
#pragma unroll
  for (int b = 0; b < DT; ++b) moff[b] = (jd[b] && g == 0) ? (16 * b + c) * 8 : OOB;
  if (lane < d) mv[lane] = out[(size_t)t0 * rec + lane];     // m_{t0}, as k_filter_w48 left it
  double ynext[PT];
#pragma unroll
  for (int b = 0; b < PT; ++b) ynext[b] = bld(ry, yoff[b], t0 * p * 8);
  wave_sync();
  for (int t = t0; t < T; ++t) {
    double ycur[PT];
#pragma unroll
    for (int b = 0; b < PT; ++b) { ycur[b] = ynext[b]; ynext[b] = bld(ry, t + 1 < T ? yoff[b] : OOB, (t + 1 < T ? t + 1 : 0) * p * 8); }
    double an[DT];
    gather_vec<DT, K>(mv, tix, tvl, an);                          // a = G m
#pragma unroll
    for (int b = 0; b < DT; ++b) if (g == 0 && jd[b]) av[16 * b + c] = an[b];
    wave_sync();
    double fcol[PT];
    if (KF > 0) {
#pragma unroll
      for (int b = 0; b < PT; ++b) {
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < KFA; ++s) s_ = fma(av[fix[b][s]], fvl[b][s], s_);
        fcol[b] = s_;
      }
    } else {
      d4 Fm[DT][PT];
      f_tiles<DT, PT>(Fl, Fm, g, c);
      matTvec<DT, PT>(Fm, av, g, fcol);
    }
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const double e = ycur[b] - fcol[b];
      if (g == 0) ev[16 * b + c] = jp[b] ? e : 0.0;
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < DT; ++b) {                                // K e, as the filter's steady branch sums it
      double s_ = 0.0;
#pragma unroll
      for (int jj = 0; jj < 4 * PT; ++jj) s_ = fma(img[(4 * jj + g) * IL + 16 * b + c], ev[4 * jj + g], s_);
      s_ = sum_g(s_);
      if (g == 0 && jd[b]) mv[16 * b + c] = av[16 * b + c] + s_;
    }
    wave_sync();
    const int so = (t + 1) * recb;
#pragma unroll
    for (int b = 0; b < DT; ++b) bst(rfo, moff[b], so, mv[16 * b + c]);
  }
  if (a.counters && lane == 0) atomicAdd(&a.counters[0], (unsigned long long)(T - t0));
  if (__ballot(lane < d && !isfinite(mv[lane < VL ? lane : 0])) != 0ull && a.status && lane == 0) atomicOr(&a.status[n], DLM_ST_NONFINITE);
}

// ---------------------------------------------------------------------------------------
// The reference's backward sampler (Smoothing.sampleDlm / Smoothing.step, Smoothing.scala:74-122) for 16 <= d <= 48 on
// register tiles: dlm_sampler16.hip's kernel with DT x DT tiles per matrix.  Per step t = T-1 .. 0, given theta_{t+1}:
//   a+ = G m, R+ = G C G^T + W dt (dt == 0: a+ = m, R+ = C);  J = C G^T R+^-1 (always the table entry g(dt));
//   h = m + J (theta_{t+1} - a+);  H = (I - J G) C (I - J G)^T + dt J W J^T, symmetrised;  theta_t = h + chol(H) z_t
// with the conditional-moment records and the Gibbs statistics (diagonal or outer product) of k_sampler_generic, whose
// draws it reproduces.  Products are MFMA chains on the tiles (X^T Y, as everywhere in this file), the three products
// with G gathers through the image, R+^-1 the warm-started Newton-Schulz refinement, the factor of H is taken with lane i
// holding row i in registers (pivots and multipliers by v_readlane: no LDS round trip on the d dependent pivots).
// Once the filtered covariance has stopped moving (|C_t - C| <= DLM_SETTLE_TOL max|C| against the last step computed in full), J, H
// and the factor are reused and a step is a gather, two matrix-vector products and the draw.
// LDS per wave: two images (scratch; the second also keeps the factor L in its lower triangle and the comparison copy of
// C in its strict upper triangle) and six vectors -- 39 KB at DT = 3: four series per CU.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Shared factors (dlm_sampler16.hip has the d <= 15 form).  EXP: the series is the one series of zeros, workgroup b
// makes the table rows of stretch b -- per step computed in full, per lane [ J^T tiles (4 DT^2) | row `lane` of L (16 DT, zero
// beyond the diagonal) ] and SampTabs::need[t] = 1; a step taken in the steady form leaves need[t] = 0.
// Stretches (every series, see dlm_sampler16.hip): 8 steps long below step 64, where the covariances are still moving and every
// step is computed in full (~120 us at d = 40: the longest stretch is what a table waits for), 32 steps long above.
constexpr int WS_STRETCH = 32, WS_SHORT = 8, WS_SHORT_END = 64;
__host__ __device__ constexpr bool ws_stretch_top(int t) { return (t & (WS_STRETCH - 1)) == WS_STRETCH - 1 || (t < WS_SHORT_END && (t & (WS_SHORT - 1)) == WS_SHORT - 1); }
__host__ __device__ constexpr int ws_stretches(int T) { return T <= WS_SHORT_END ? (T + WS_SHORT - 1) / WS_SHORT : WS_SHORT_END / WS_SHORT + (T - WS_SHORT_END + WS_STRETCH - 1) / WS_STRETCH; }
__host__ __device__ constexpr int ws_stretch_lo(int b) { return b < WS_SHORT_END / WS_SHORT ? b * WS_SHORT : WS_SHORT_END + (b - WS_SHORT_END / WS_SHORT) * WS_STRETCH; }
constexpr int ws_row_doubles(int DT) { return 64 * (4 * DT * DT + 16 * DT); }

template <int DT, int K, bool OUTER>
__global__ __launch_bounds__(64) void k_sampler_w48(KArgs a, SampTabs tb) {
  // The table of a shared-factor call is made by THIS kernel (tb.rows set), not by an instantiation of its own: the series that
  // compute their own factors and the table then run the same machine code, and what they compute agrees bit for bit by construction.
  const bool EXP = tb.rows != nullptr;
  if (EXP) __builtin_amdgcn_s_setprio(3);   // the table's few waves run beside the kernel that filters the batch and are what the draw kernel waits for
  if (!EXP && a.route && (a.route[blockIdx.x] != 0) != (a.route_take != 0)) return;   // shared-factor call: only the series routed here
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int IL = 16 * DT + 1, IMG = 16 * DT * IL, VL = 16 * DT, ND = 16 * DT;
  double* img = sm;       double* keep = sm + IMG;      // keep: direct-inverse scratch; afterwards L (lower + diagonal) and C (strict upper)
  double* mv = keep + IMG; double* thv = mv + VL; double* uv = thv + VL; double* hv = uv + VL; double* zv = hv + VL; double* cdv = zv + VL;
  const int n = EXP ? 0 : blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8;
  const int t_lo = EXP ? ws_stretch_lo((int)blockIdx.x) : 0;
  const int t_nx = EXP ? ws_stretch_lo((int)blockIdx.x + 1) : T;
  const int t_hi = (t_nx < T ? t_nx : T) - 1;
  for (int i = lane; i < 6 * VL; i += 64) mv[i] = 0.0;
  // EXP: row t of the table from the J^T tiles (nullptr: none, the step of theta_T) and the factor in `keep`
  auto export_row = [&](int t, const d4 (*JT)[DT]) {
    double* row = tb.rows + (size_t)t * ws_row_doubles(DT) + lane * (4 * DT * DT + 16 * DT);
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) row[(aa * DT + b) * 4 + r] = JT ? JT[aa][b][r] : 0.0;
    for (int k = 0; k < ND; ++k) row[4 * DT * DT + k] = (lane < d && k <= lane) ? keep[lane * IL + k] : 0.0;
    if (lane == 0) tb.need[t] = 1;
  };

  bool jd[DT];
  int cpart[DT], moff[DT];
#pragma unroll
  for (int b = 0; b < DT; ++b) { jd[b] = 16 * b + c < d; cpart[b] = (d + (16 * b + c) * d) * 8; moff[b] = (jd[b] && g == 0) ? (16 * b + c) * 8 : OOB; }
  int rix[DT][K], cix[DT][K];       // nonzeros of the lanes' ROWS / COLUMNS of G
  double rvl[DT][K], cvl[DT][K];
  int gcur = -1;
  auto load_tables = [&](int gi) {
    const SparseBig* tr = a.spb + 2 * gi;
    const SparseBig* tc = a.spb + 2 * gi + 1;
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) {
        rix[b][s] = tr->idx[16 * b + c][s]; rvl[b][s] = jd[b] ? tr->val[16 * b + c][s] : 0.0;
        cix[b][s] = tc->idx[16 * b + c][s]; cvl[b][s] = jd[b] ? tc->val[16 * b + c][s] : 0.0;
      }
    gcur = gi;
  };
  const double* W0 = a.W + (size_t)n * a.w_stride;
  const __amdgpu_buffer_rsrc_t rin = mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  double* cond = a.cond ? a.cond + (size_t)n * (T + 1) * rec : nullptr;
  const __amdgpu_buffer_rsrc_t rco = mk_rsrc(cond, cond ? (size_t)(T + 1) * recb : 0);
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  const double* y = a.y ? a.y + (size_t)n * T * p : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * d : nullptr;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  int st = 0;
  // column `lane` of a structured, time-invariant F (SparseF: at most four nonzeros, ascending row index, unused slots 0 * theta[0])
  const bool fsp = a.spf != nullptr && !a.f_stride && a.stats != nullptr && y != nullptr;
  int fci[4];
  double fcv[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) { fci[s] = (fsp && lane < p) ? a.spf->cidx[lane][s] : 0; fcv[s] = (fsp && lane < p) ? a.spf->cval[lane][s] : 0.0; }

  const int t_last = EXP ? tb.settle[0] : T;      // EXP: the records above this one were not written -- they repeat its covariance
  auto load_record = [&](d4 (&C)[DT][DT], int t, int g, int c) {
    const int so = (EXP && t > t_last ? t_last : t) * recb;
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g;
          C[aa][b][r] = bld(rin, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so);
        }
    if (lane < ND) mv[lane] = bld(rin, lane < d ? lane * 8 : OOB, so);
  };
  auto load_W = [&](d4 (&Wt)[DT][DT], const double* Wp, int g, int c) {
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
          Wt[aa][b][r] = (i < d && jd[b]) ? Wp[i + j * d] : 0.0;
        }
  };
  // lower Cholesky factor of the symmetric matrix in `img` (rows by lanes, in registers), written to the lower triangle of `keep`
  auto factor_to_keep = [&]() {
    double row[ND];
    const double* src = img + (lane < ND ? lane : 0) * IL;
#pragma unroll
    for (int j = 0; j < ND; ++j) row[j] = src[j];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < ND; ++k)
      if (k < d) {
        const double akk = readlane_d(row[k], k);
        const bool np = !(akk > 0.0);
        bad |= np;
        double inv = __builtin_amdgcn_rsq(np ? 1.0 : akk);
        inv = inv * fma(-0.5 * akk * inv, inv, 1.5);
        inv = inv * fma(-0.5 * akk * inv, inv, 1.5);
        inv = np ? 0.0 : inv;
        const double lik = row[k] * inv;
        row[k] = lik;
#pragma unroll
        for (int j = k + 1; j < ND; ++j)
          if (j < d) row[j] = fma(-lik, readlane_d(lik, j), row[j]);
      }
    if (bad) st |= DLM_ST_NOT_PD;
    wave_sync();
    if (lane < d) {
#pragma unroll
      for (int j = 0; j < ND; ++j) if (j <= lane) keep[lane * IL + j] = row[j];
    }
    wave_sync();
  };
  // theta = h + L z from the factor in `keep`; hv holds h, zv the normals; returns this lane's component (lane < d)
  auto draw = [&]() {
    double th = 0.0;
    if (lane < d) {
      th = hv[lane];
      const double* Lr = keep + lane * IL;
      for (int k = 0; k <= lane; ++k) th = fma(Lr[k], zv[k], th);
    }
    return th;
  };
  auto store_cond = [&](const d4 (&H)[DT][DT], int t, int g, int c) {
    const int so = t * recb;
#pragma unroll
    for (int aa = 0; aa < DT; ++aa)
#pragma unroll
      for (int b = 0; b < DT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * aa + 4 * r + g;
          bst(rco, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so, H[aa][b][r]);
        }
    if (lane < ND) bst(rco, lane < d ? lane * 8 : OOB, so, hv[lane]);
  };

  // Persistent tiles: the Newton-Schulz warm start and (GibbsWishart) the outer-product statistics.  J^T of the last full
  // step lives in `img` (the steady-state path uses the image for nothing else) -- a step that recomputes it may use the
  // image as scratch again.
  d4 Rinv[DT][DT], OUT[OUTER ? DT : 1][OUTER ? DT : 1];
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b) Rinv[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
    for (int b = 0; b < (OUTER ? DT : 1); ++b) OUT[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
  double ssd[DT];
#pragma unroll
  for (int b = 0; b < DT; ++b) ssd[b] = 0.0;
  double ssy = 0.0, nob = 0.0;
  bool warm = false, have = false;
  int gprev = -1;
  double dtprev = 0.0, cmaxp = 0.0;
#ifdef DLM_STAMP
  int n_reuse = 0;      // diagnostic build: steps that took the steady-state path, reported in status[n] >> 8
#endif

  if (!EXP || t_hi == T - 1) {   // theta_T ~ N(m_T, C_T)
    d4 C[DT][DT];
    load_record(C, T, g, c);
    to_image<IL, DT, DT>(C, keep, g, c);
    if (lane < ND) { zv[lane] = (lane < d && !EXP) ? (zin ? zin[(size_t)T * d + lane] : philox_normal(a.seed, series, (unsigned)T, (unsigned)lane)) : 0.0; }
    wave_sync();
    if (lane < ND) hv[lane] = mv[lane];
    if (chol_rows(keep, d, IL, lane)) st |= DLM_ST_NOT_PD;   // once per series: the in-LDS factorisation (psd pivots: see below)
    wave_sync();
    if (EXP) { export_row(T, nullptr); wave_sync(); }
    if (cond) store_cond(C, T, g, c);
    const double th = draw();
    wave_sync();
    if (lane < ND) thv[lane] = lane < d ? th : 0.0;
    if (thout && lane < d) thout[(size_t)T * d + lane] = th;
    wave_sync();
  }

  for (int t = t_hi; t >= t_lo; --t) {
    int g_ = g, c_ = c;
    asm volatile("" : "+v"(g_), "+v"(c_));
    {
    const int g = g_, c = c_;
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) load_tables(gi);
    if ((EXP || a.stretches) && ws_stretch_top(t)) { have = false; warm = false; }   // a stretch starts from scratch: nothing a step computes depends on the steps above its stretch
    if (a.stats && y && lane < p) {   // observation residual of theta_{t+1}, component `lane` (Gibbs.scala:29-39)
      const double yv = y[(size_t)t * p + lane];
      if (yv == yv) {
        double f = 0.0;
        if (fsp) {   // structured F: the few nonzeros of column `lane`, in the order of the dense sum (whose other terms add exact zeros)
#pragma unroll
          for (int s = 0; s < 4; ++s) f = fma(fcv[s], thv[fci[s]], f);
        } else {     // d dependent loads per step: 20 of the 23 us of a steady step at d = 40 before the tables were used here
          const double* Fj = a.F + (size_t)t * a.f_stride + (size_t)lane * d;
          for (int k = 0; k < d; ++k) f = fma(Fj[k], thv[k], f);
        }
        ssy += (yv - f) * (yv - f); nob += 1.0;
      }
    }
    if (lane < ND) zv[lane] = (lane < d && !EXP) ? (zin ? zin[(size_t)t * d + lane] : philox_normal(a.seed, series, (unsigned)t, (unsigned)lane)) : 0.0;
    // steady state: C against the copy kept from the last full step (strict upper triangle of `keep`, diagonal in cdv)
    bool reuse = false;
    {
      d4 C[DT][DT];
      load_record(C, t, g, c);
      wave_sync();
    if (have && gi == gprev && dt == dtprev && !a.w_tstride) {
      bool moved = false;
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
            if (i < d && j < d) {
              const double old = (i == j) ? cdv[i] : (i < j ? keep[i * IL + j] : keep[j * IL + i]);
              moved |= !(fabs(C[aa][b][r] - old) <= DLM_SETTLE_TOL * cmaxp);   // this step's record against the covariance of the last full step: a bound on the distance itself
            }
          }
      reuse = __ballot(moved) == 0ull;
    }
    }
    double a1[DT];
    if (dt != 0.0) gather_vec<DT, K>(mv, rix, rvl, a1);
    else {
#pragma unroll
      for (int b = 0; b < DT; ++b) a1[b] = mv[16 * b + c];
    }
    if (g == 0) {
#pragma unroll
      for (int b = 0; b < DT; ++b) uv[16 * b + c] = jd[b] ? thv[16 * b + c] - a1[b] : 0.0;
    }
    wave_sync();
    double hc[DT];
    if (__builtin_expect(reuse, 1)) {   // (the likely path: the register allocator then spills in the full step, not here)
#ifdef DLM_STAMP
      ++n_reuse;
#endif
      d4 JTl[DT][DT];
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) JTl[aa][b][r] = img[(16 * aa + 4 * r + g) * IL + 16 * b + c];
      matTvec<DT, DT>(JTl, uv, g, hc);
      if (EXP) { if (lane == 0) tb.need[t] = 0; }
    } else {
      // R+ = G C G^T + W dt on a copy of C (the congruence works in place), exactly symmetric.  C itself is fetched again
      // when it is next needed (the record is in L2): holding it across the inverse costs 24 registers per tile.
      d4 R[DT][DT];
      load_record(R, t, g, c);
      if (dt != 0.0) {
        d4 Wt[DT][DT];
        load_W(Wt, W0 + (size_t)t * a.w_tstride, g, c);
        congruence<IL, DT, K, true>(R, Wt, dt, rix, rvl, img, g, c);
      }
      if (spd_inverse_warm<IL, DT>(R, Rinv, d, warm, img, keep, lane, g, c)) st |= DLM_ST_NOT_PD;
      warm = true;
      // G C = (C G^T)^T: pass 1 of the congruence on the image of C, read back transposed
      d4 C[DT][DT], GC[DT][DT];
      load_record(C, t, g, c);
      {   // remember C (strict upper triangle of `keep` -- the inverse is done with it --, diagonal in cdv) and its scale
        double mx = 0.0;
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
              if (i < d && j < d) {
                mx = fmax(mx, fabs(C[aa][b][r]));
                if (i < j) keep[i * IL + j] = C[aa][b][r];
                if (i == j) cdv[i] = C[aa][b][r];
              }
            }
        for (int o_ = 32; o_ > 0; o_ >>= 1) mx = fmax(mx, __shfl_xor(mx, o_));
        cmaxp = mx;
      }
      to_image<IL, DT, DT>(C, img, g, c);
      wave_sync();
      {
        d4 T1[DT][DT];
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const double* row = img + (16 * aa + 4 * r + g) * IL;
              double s_ = 0.0;
#pragma unroll
              for (int s = 0; s < K; ++s) s_ = fma(row[rix[b][s]], rvl[b][s], s_);
              T1[aa][b][r] = s_;
            }
        wave_sync();
        transpose<IL, DT, DT>(T1, GC, img, g, c);
      }
      d4 JT[DT][DT];
      mmT<DT, DT, DT, false>(Rinv, GC, JT, d);            // J^T = R+^-1 G C
      matTvec<DT, DT>(JT, uv, g, hc);
      // Dm = I - J G: J (= J^T transposed into the image), gathered with the COLUMN tables of G
      d4 DmT[DT][DT];
      {
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[(16 * b + c) * IL + 16 * aa + 4 * r + g] = JT[aa][b][r];
        wave_sync();
        d4 Dm[DT][DT];
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
              const double* row = img + i * IL;
              double s_ = 0.0;
#pragma unroll
              for (int s = 0; s < K; ++s) s_ = fma(row[cix[b][s]], cvl[b][s], s_);
              Dm[aa][b][r] = ((i == j && j < d) ? 1.0 : 0.0) - s_;
            }
        wave_sync();
        transpose<IL, DT, DT>(Dm, DmT, img, g, c);
      }
      d4 H[DT][DT];
      {
        d4 CD[DT][DT];
        mmT<DT, DT, DT, false>(C, DmT, CD, d);            // C Dm^T
        mmT<DT, DT, DT, false>(DmT, CD, H, d);            // Dm C Dm^T
      }
      if (dt != 0.0) {
        d4 Wt[DT][DT], WJ[DT][DT], H2[DT][DT];
        load_W(Wt, W0 + (size_t)t * a.w_tstride, g, c);
        mmT<DT, DT, DT, false>(Wt, JT, WJ, d);            // W J^T
        mmT<DT, DT, DT, false>(JT, WJ, H2, d);            // J W J^T
#pragma unroll
        for (int aa = 0; aa < DT; ++aa)
#pragma unroll
          for (int b = 0; b < DT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) H[aa][b][r] = fma(H2[aa][b][r], dt, H[aa][b][r]);
      }
      mirror<IL, DT, true, false>(H, img, g, c);           // (H + H^T) / 2 (Smoothing.scala:95)
      if (cond) {
        if (g == 0) {
#pragma unroll
          for (int b = 0; b < DT; ++b) hv[16 * b + c] = jd[b] ? mv[16 * b + c] + hc[b] : 0.0;
        }
        wave_sync();
        store_cond(H, t, g, c);
      }
      to_image<IL, DT, DT>(H, img, g, c);
      wave_sync();
      factor_to_keep();
      if (EXP) export_row(t, JT);
      have = true; gprev = gi; dtprev = dt;
      wave_sync();
      to_image<IL, DT, DT>(JT, img, g, c);                // J^T stays in the image for the steady-state steps that follow
      wave_sync();
    }
    if (g == 0) {
#pragma unroll
      for (int b = 0; b < DT; ++b) hv[16 * b + c] = jd[b] ? mv[16 * b + c] + hc[b] : 0.0;      // h = m + J (theta+ - a+)
    }
    wave_sync();
    if (cond && reuse) {   // same conditional covariance as the step before: copy it from that record
      d4 Hc[DT][DT];
      const int so = (t + 1) * recb;
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g;
            Hc[aa][b][r] = bld(rco, (i < d && jd[b]) ? cpart[b] + i * 8 : OOB, so);
          }
      store_cond(Hc, t, g, c);
    }
    const double th = draw();
    wave_sync();
    if (lane < ND) hv[lane] = lane < d ? th : 0.0;        // hv <- theta_t
    wave_sync();
    if (a.stats) {   // system residual theta_{t+1} - G theta_t (always the table entry)
      double gth[DT];
      gather_vec<DT, K>(hv, rix, rvl, gth);
      const double dts = (dt == 0.0) ? 1.0 : dt;
      const bool unit = __builtin_amdgcn_readfirstlane((int)(dts == 1.0)) != 0;   // a regular grid: x / 1.0 is x, and 36 fp64 divisions a step are ~1000 instructions
      double df[DT];
#pragma unroll
      for (int b = 0; b < DT; ++b) { df[b] = jd[b] ? thv[16 * b + c] - gth[b] : 0.0; ssd[b] += unit ? df[b] * df[b] : df[b] * df[b] / dts; }
      if (OUTER) {
        if (g == 0) {
#pragma unroll
          for (int b = 0; b < DT; ++b) uv[16 * b + c] = df[b];
        }
        wave_sync();
        if (unit) {
#pragma unroll
          for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
            for (int b = 0; b < (OUTER ? DT : 1); ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) OUT[aa][b][r] += uv[16 * aa + 4 * r + g] * df[b];
        } else {
#pragma unroll
          for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
            for (int b = 0; b < (OUTER ? DT : 1); ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) OUT[aa][b][r] += uv[16 * aa + 4 * r + g] * df[b] / dts;
        }
      }
    }
    wave_sync();
    if (lane < ND) thv[lane] = hv[lane];
    if (thout && lane < d) thout[(size_t)t * d + lane] = th;
    wave_sync();
    }
  }
  if (!EXP && __ballot(lane < d && !isfinite(thv[lane < ND ? lane : 0])) != 0ull) st |= DLM_ST_NONFINITE;
  if (!EXP && a.route && a.counters && lane == 0) atomicAdd(&a.counters[3], 1ull);   // a series of a shared-factor call that computed its own
  if (a.stats) {
    const int L = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * L;
    if (lane < p) { so[lane] = ssy; so[p + lane] = nob; }
    if (lane == 0) so[L - 1] = (double)T;
    if (OUTER) {
#pragma unroll
      for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
        for (int b = 0; b < (OUTER ? DT : 1); ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
            if (i < d && j < d) so[2 * p + i + j * d] = OUT[aa][b][r];
          }
    } else if (g == 0) {
#pragma unroll
      for (int b = 0; b < DT; ++b) if (jd[b]) so[2 * p + 16 * b + c] = ssd[b];
    }
  }
#ifdef DLM_STAMP
  st |= n_reuse << 8;
#endif
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}


// ---------------------------------------------------------------------------------------
// The draw against the shared factors, 16 <= d <= 48: one wave per series, k_sampler_w48's steady-state step with nothing of the
// covariances left in it -- no record of C_t is read (the filtered mean alone, 8 d of the 8 (d + d^2) bytes), J^T lives in registers
// (the per-series kernel re-reads it from its image every step), the factor's row `lane` too, F in LDS (the per-series kernel
// reads it from memory in a dependent chain), two steps' normals per Philox / Box-Muller pass.  The operations on the data are
// k_sampler_w48's, one for one: gather_vec, matTvec, the draw's chain over the pivots (terms beyond the diagonal are zeros of the
// table: they add nothing), the residual sums in its order.  LDS per wave: F and seven vectors -- no images.
// ---------------------------------------------------------------------------------------
template <int DT, int K, bool OUTER>
__global__ __launch_bounds__(64, 2) void k_mean_sampler_w48(KArgs a, SampTabs tb) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int VL = 16 * DT, ND = 16 * DT, RW = 4 * DT * DT + 16 * DT;
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  if (a.route[n] != 0) return;                      // a series with a missing observation: k_sampler_w48 serves it
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8, FLD = d + 1;
  double* mv = sm; double* thv = mv + VL; double* uv = thv + VL; double* hv = uv + VL; double* zv = hv + VL;   // zv: two steps' normals, [t & 1][VL]
  double* Ll = zv + 2 * VL;                         // the factor, lower triangle by rows: L[i][k] at i (i + 1) / 2 + k
  double* Fl = Ll + (ND * (ND + 1)) / 2;            // F by observation component: Fl[j * FLD + k] = F[k][j]
  for (int i = lane; i < 6 * VL; i += 64) mv[i] = 0.0;
  const bool stats = a.stats != nullptr, obs = stats && a.y;
  if (obs) for (int i = lane; i < p * d; i += 64) Fl[(i / d) * FLD + i % d] = a.F[i];
  bool jd[DT];
#pragma unroll
  for (int b = 0; b < DT; ++b) jd[b] = 16 * b + c < d;
  int rix[DT][K];
  double rvl[DT][K];
  {
    const SparseBig* tr = a.spb;                    // rows of G (time-invariant: the only table)
#pragma unroll
    for (int b = 0; b < DT; ++b)
#pragma unroll
      for (int s = 0; s < K; ++s) { rix[b][s] = tr->idx[16 * b + c][s]; rvl[b][s] = jd[b] ? tr->val[16 * b + c][s] : 0.0; }
  }
  const __amdgpu_buffer_rsrc_t rin = mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const int moff = lane < d ? lane * 8 : OOB;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  const double* y = obs ? a.y + (size_t)n * T * p : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * d : nullptr;
  const unsigned long long series = a.series_offset + (unsigned long long)n;

  d4 JT[DT][DT];
#pragma unroll
  for (int aa = 0; aa < DT; ++aa)
#pragma unroll
    for (int b = 0; b < DT; ++b) JT[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
  const int loff = (lane * (lane + 1)) / 2;
  auto load_row = [&](int t, bool withJ) {
    const double* row = tb.rows + (size_t)t * ws_row_doubles(DT) + lane * RW;
    if (withJ) {
#pragma unroll
      for (int aa = 0; aa < DT; ++aa)
#pragma unroll
        for (int b = 0; b < DT; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) JT[aa][b][r] = row[(aa * DT + b) * 4 + r];
    }
    if (lane < d) for (int k = 0; k <= lane; ++k) Ll[loff + k] = row[4 * DT * DT + k];
  };
  // the normals of steps tb_ (lanes 0..31) and tb_ - 1 (lanes 32..63): lane q of a half makes components 2 q and 2 q + 1
  auto normals2 = [&](int tb_) {
    const int te = tb_ - (lane >> 5), q = lane & 31;
    if (te >= 0 && 2 * q < ND) {
      double ze = 0.0, zo = 0.0;
      if (2 * q < d) philox_normal2(a.seed, series, (unsigned)te, (unsigned)q, ze, zo);
      double* zr = zv + (te & 1) * VL + 2 * q;
      zr[0] = ze; zr[1] = (2 * q + 1 < d) ? zo : 0.0;
    }
  };
  auto draw = [&](const double* zk) {                // theta = h + L z: hv holds h
    double th = 0.0;
    if (lane < d) {
      th = hv[lane];
      const double* Lr = Ll + loff;
      for (int k = 0; k <= lane; ++k) th = fma(Lr[k], zk[k], th);
    }
    return th;
  };
  unsigned long long nmask = 0;
  auto need_of = [&](int s_) -> bool {
    if (s_ < 0) return false;
    if (s_ == T || (s_ & 63) == 63) { const int b = (s_ & ~63) + lane; nmask = __ballot(b <= T && tb.need[b] != 0); }
    return ((nmask >> (s_ & 63)) & 1ull) != 0;
  };
  (void)need_of(T);
  double mnext;
  {   // theta_T = m_T + chol(C_T) z_T
    const double mT = bld(rin, moff, T * recb);
    mnext = bld(rin, moff, (T > 0 ? T - 1 : 0) * recb);
    load_row(T, false);
    if (zin) { if (lane < ND) zv[(T & 1) * VL + lane] = lane < d ? zin[(size_t)T * d + lane] : 0.0; }
    else { if (lane < ND) zv[(T & 1) * VL + lane] = lane < d ? philox_normal(a.seed, series, (unsigned)T, (unsigned)lane) : 0.0; }
    if (lane < ND) hv[lane] = lane < d ? mT : 0.0;
    wave_sync();
    const double th = draw(zv + (T & 1) * VL);
    wave_sync();
    if (lane < ND) thv[lane] = th;
    if (thout && lane < d) thout[(size_t)T * d + lane] = th;
    if (!zin) normals2(T - 1);                       // steps T - 1 and T - 2
    wave_sync();
  }
  d4 OUT[OUTER ? DT : 1][OUTER ? DT : 1];
#pragma unroll
  for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
    for (int b = 0; b < (OUTER ? DT : 1); ++b) OUT[aa][b] = d4{0.0, 0.0, 0.0, 0.0};
  double ssd[DT];
#pragma unroll
  for (int b = 0; b < DT; ++b) ssd[b] = 0.0;
  double ssy = 0.0, nob = 0.0;
  double ynext = (y && lane < p && T > 0) ? y[(size_t)(T - 1) * p + lane] : 0.0;
  double znext = (zin && lane < d && T > 0) ? zin[(size_t)(T - 1) * d + lane] : 0.0;   // given normals (injected, or k_normals_rows'): one step ahead
  for (int t = T - 1; t >= 0; --t) {
    const bool nd = need_of(t);
    const double yv = ynext, mc = mnext, zc = znext;
    if (t > 0) {
      mnext = bld(rin, moff, (t - 1) * recb);
      if (y && lane < p) ynext = y[(size_t)(t - 1) * p + lane];
      if (zin && lane < d) znext = zin[(size_t)(t - 1) * d + lane];
    }
    if (obs && lane < p && yv == yv) {   // observation residual of theta_{t+1}, component `lane` (Gibbs.scala:29-39)
      const double* Fj = Fl + lane * FLD;
      double f = 0.0;
      for (int k = 0; k < d; ++k) f = fma(Fj[k], thv[k], f);
      ssy += (yv - f) * (yv - f); nob += 1.0;
    }
    if (zin) { if (lane < ND) zv[(t & 1) * VL + lane] = lane < d ? zc : 0.0; }
    if (nd) load_row(t, true);
    if (lane < ND) mv[lane] = mc;
    wave_sync();
    double a1[DT];
    gather_vec<DT, K>(mv, rix, rvl, a1);
    if (g == 0) {
#pragma unroll
      for (int b = 0; b < DT; ++b) uv[16 * b + c] = jd[b] ? thv[16 * b + c] - a1[b] : 0.0;
    }
    wave_sync();
    double hc[DT];
    matTvec<DT, DT>(JT, uv, g, hc);
    if (g == 0) {
#pragma unroll
      for (int b = 0; b < DT; ++b) hv[16 * b + c] = jd[b] ? mv[16 * b + c] + hc[b] : 0.0;      // h = m + J (theta+ - a+)
    }
    wave_sync();
    const double th = draw(zv + (t & 1) * VL);
    wave_sync();
    if (lane < ND) hv[lane] = th;                     // hv <- theta_t
    wave_sync();
    if (stats) {   // system residual theta_{t+1} - G theta_t
      double gth[DT];
      gather_vec<DT, K>(hv, rix, rvl, gth);
      double df[DT];
#pragma unroll
      for (int b = 0; b < DT; ++b) { df[b] = jd[b] ? thv[16 * b + c] - gth[b] : 0.0; ssd[b] += df[b] * df[b]; }
      if (OUTER) {
        if (g == 0) {
#pragma unroll
          for (int b = 0; b < DT; ++b) uv[16 * b + c] = df[b];
        }
        wave_sync();
#pragma unroll
        for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
          for (int b = 0; b < (OUTER ? DT : 1); ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) OUT[aa][b][r] += uv[16 * aa + 4 * r + g] * df[b];
      }
    }
    wave_sync();
    if (lane < ND) thv[lane] = hv[lane];
    if (thout && lane < d) thout[(size_t)t * d + lane] = th;
    if (!zin && ((T - 1 - t) & 1) == 1) normals2(t - 1);   // the normals of steps t - 1 and t - 2
    wave_sync();
  }
  int st = tb.status[0];
  if (__ballot(lane < d && !isfinite(thv[lane < ND ? lane : 0])) != 0ull) st |= DLM_ST_NONFINITE;
  if (stats) {
    const int L = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * L;
    if (lane < p) { so[lane] = ssy; so[p + lane] = nob; }
    if (lane == 0) so[L - 1] = (double)T;
    if (OUTER) {
#pragma unroll
      for (int aa = 0; aa < (OUTER ? DT : 1); ++aa)
#pragma unroll
        for (int b = 0; b < (OUTER ? DT : 1); ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * aa + 4 * r + g, j = 16 * b + c;
            if (i < d && j < d) so[2 * p + i + j * d] = OUT[aa][b][r];
          }
    } else if (g == 0) {
#pragma unroll
      for (int b = 0; b < DT; ++b) if (jd[b]) so[2 * p + 16 * b + c] = ssd[b];
    }
  }
  if (a.counters && lane == 0) atomicAdd(&a.counters[2], 1ull);
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// The normals of a shared-factor call as rows [N][T+1][d] (the layout of injected normals), made while the batch is filtered: one
// thread per Box-Muller pair, philox_normal2 = philox_normal's values.  They are 60 % of the draw kernel's arithmetic otherwise.
__global__ __launch_bounds__(256) void k_normals_rows(KArgs a, double* __restrict__ z) {
  const int d = a.d, np = (d + 1) / 2;
  const long long total = (long long)a.N * (a.T + 1) * np;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(i % np);
    const long long nt = i / np;
    const int t = (int)(nt % (a.T + 1));
    const long long n = nt / (a.T + 1);
    double ze, zo;
    philox_normal2(a.seed, a.series_offset + (unsigned long long)n, (unsigned)t, (unsigned)q, ze, zo);
    double* o = z + (size_t)nt * d + 2 * q;
    o[0] = ze;
    if (2 * q + 1 < d) o[1] = zo;
  }
}

// route[n] = 1: series n has a missing observation component.  One workgroup per series.
__global__ __launch_bounds__(256) void k_mark_gaps_w48(const double* __restrict__ y, int N, int Tp, unsigned char* __restrict__ route) {
  const int n = blockIdx.x;
  if (n >= N) return;
  __shared__ int any;
  if (threadIdx.x == 0) any = 0;
  __syncthreads();
  const double* yn = y + (size_t)n * Tp;
  bool gap = false;
  for (int t = threadIdx.x; t < Tp; t += 256) { const double v = yn[t]; gap |= !(v == v); }
  if (__ballot(gap) != 0ull && (threadIdx.x & 63) == 0) any = 1;   // (every writer writes the same value)
  __syncthreads();
  if (threadIdx.x == 0) route[n] = any ? 1 : 0;
}

}  // namespace w48


// The per-wave kernels take the structured-G models; a dense G stays on dlm_tiled.hip.
// One wave per series needs more series than the chip has SIMDs to pay off: up to one series per CU (N <= 256) the
// workgroup-per-series kernels of dlm_tiled.hip finish a step sooner (10.8 against 13 us at d = 40, p = 20).
// dlm_options.flags: DLM_OPT_NO_WAVE sends everything to dlm_tiled.hip, DLM_OPT_FORCE_WAVE lifts the batch-size rule (A/B
// measurements and the parity tests of the two paths; the library reads no environment variables).
// Multivariate models below the tiled range (d <= 15, 2 <= p <= 32) run on the same kernels with one tile per dimension:
// their only alternative is the generic LDS kernel, so the batch-size rule does not apply to them.
bool wave48_small_shape(const KArgs& a) {
  return a.d >= 1 && a.d <= 15 && a.p >= 2 && a.p <= 32 && ((size_t)a.T + 1) * (size_t)(a.d + a.d * a.d) * 8 < ((size_t)1 << 31);
}
// A time-invariant model on a regular grid takes the per-wave kernels at every batch size: their steady-state steps (a few
// microseconds once the covariance recursion has settled: C4 at 250 series 6.8 ms) beat the workgroup kernels' 10.8 us per step
// (24.4 ms) by more than the latter win when nothing settles (58.5 against 63.6 ms with 5 % of the components missing).
static bool steady_capable(const KArgs& a) {
  return !a.g_index && !a.dt && !a.f_stride && !a.v_tstride && !a.w_tstride && !(a.flags & DLM_OPT_NO_STEADY);
}
static bool wave48_wanted(const KArgs& a) {
  if (a.flags & DLM_OPT_NO_WAVE) return false;
  return wave48_small_shape(a) || a.N > 256 || (a.flags & DLM_OPT_FORCE_WAVE) || steady_capable(a);
}
static bool shape_ok(const KArgs& a) { return tiled_supported(a) || wave48_small_shape(a); }
bool wave48_small_ok(const KArgs& a) { return wave48_small_shape(a) && a.spb && !(a.flags & DLM_OPT_NO_WAVE); }
bool wave48_filter_supported(const KArgs& a) {
  return shape_ok(a) && a.spb && wave48_wanted(a);
}

template <int DT, int PT>
static hipError_t launch_w48_filter_k(const KArgs& a, int K, double* innov, int zero_m0, hipStream_t s) {
  const size_t lds = sizeof(double) * w48::lds_doubles(DT, PT);
  const int kf = (a.spf && !(a.flags & DLM_OPT_NO_SPARSE_F)) ? a.spf_k : 0;   // 0: dense (or time-varying) F
  if (K <= 2) {
    if (kf == 1) hipLaunchKernelGGL((w48::k_filter_w48<DT, PT, 2, 1>), dim3(a.N), dim3(64), lds, s, a, innov, zero_m0);
    else if (kf > 1) hipLaunchKernelGGL((w48::k_filter_w48<DT, PT, 2, 4>), dim3(a.N), dim3(64), lds, s, a, innov, zero_m0);
    else hipLaunchKernelGGL((w48::k_filter_w48<DT, PT, 2, 0>), dim3(a.N), dim3(64), lds, s, a, innov, zero_m0);
  } else {
    if (kf >= 1) hipLaunchKernelGGL((w48::k_filter_w48<DT, PT, 4, 4>), dim3(a.N), dim3(64), lds, s, a, innov, zero_m0);
    else hipLaunchKernelGGL((w48::k_filter_w48<DT, PT, 4, 0>), dim3(a.N), dim3(64), lds, s, a, innov, zero_m0);
  }
  return hipGetLastError();
}

bool wave48_smoother_supported(const KArgs& a) { return shape_ok(a) && a.spb && wave48_wanted(a); }

template <int DT, int PT>
static hipError_t launch_w48_smoother_k(const KArgs& a, int K, const double* innov, hipStream_t s) {
  const size_t lds = sizeof(double) * w48::lds_doubles(DT, PT);
  const int kf = (a.spf && !a.f_stride && !(a.flags & DLM_OPT_NO_SPARSE_F)) ? a.spf_k : 0;   // 0: dense (or time-varying) F
  if (K <= 2) {
    if (kf == 1) hipLaunchKernelGGL((w48::k_smoother_w48<DT, PT, 2, 1>), dim3(a.N), dim3(64), lds, s, a, innov);
    else if (kf > 1) hipLaunchKernelGGL((w48::k_smoother_w48<DT, PT, 2, 4>), dim3(a.N), dim3(64), lds, s, a, innov);
    else hipLaunchKernelGGL((w48::k_smoother_w48<DT, PT, 2, 0>), dim3(a.N), dim3(64), lds, s, a, innov);
  } else {
    if (kf >= 1) hipLaunchKernelGGL((w48::k_smoother_w48<DT, PT, 4, 4>), dim3(a.N), dim3(64), lds, s, a, innov);
    else hipLaunchKernelGGL((w48::k_smoother_w48<DT, PT, 4, 0>), dim3(a.N), dim3(64), lds, s, a, innov);
  }
  return hipGetLastError();
}

hipError_t launch_wave48_smoother(const KArgs& a, int K, const double* innov, hipStream_t s) {
  const bool d2 = a.d <= 32, p1 = a.p <= 16;
  if (a.d <= 16 && p1) return launch_w48_smoother_k<1, 1>(a, K, innov, s);
  if (a.d <= 16) return launch_w48_smoother_k<1, 2>(a, K, innov, s);
  if (d2 && p1) return launch_w48_smoother_k<2, 1>(a, K, innov, s);
  if (d2) return launch_w48_smoother_k<2, 2>(a, K, innov, s);
  if (p1) return launch_w48_smoother_k<3, 1>(a, K, innov, s);
  return launch_w48_smoother_k<3, 2>(a, K, innov, s);
}

static hipError_t launch_wave48_filter_z(const KArgs& a, int K, double* innov, int zero_m0, hipStream_t s) {
  const bool d2 = a.d <= 32, p1 = a.p <= 16;
  if (a.d <= 16 && p1) return launch_w48_filter_k<1, 1>(a, K, innov, zero_m0, s);
  if (a.d <= 16) return launch_w48_filter_k<1, 2>(a, K, innov, zero_m0, s);
  if (d2 && p1) return launch_w48_filter_k<2, 1>(a, K, innov, zero_m0, s);
  if (d2) return launch_w48_filter_k<2, 2>(a, K, innov, zero_m0, s);
  if (p1) return launch_w48_filter_k<3, 1>(a, K, innov, zero_m0, s);
  return launch_w48_filter_k<3, 2>(a, K, innov, zero_m0, s);
}
hipError_t launch_wave48_filter(const KArgs& a, int K, double* innov, hipStream_t s) { return launch_wave48_filter_z(a, K, innov, 0, s); }

// Durbin-Koopman simulation smoother (FFBS draw + Gibbs statistics): the prologue simulates (x+, y+) and leaves
// y* = y - y+ in `ystar`; the forward pass filters y* from a zero prior mean and overwrites y*_t by its innovation (it has
// read y*_{t+1} by then); the mean-only backward pass adds x+.
bool wave48_simsmooth_supported(const KArgs& a) {
  return shape_ok(a) && a.spb && !a.v_tstride && !a.cond && wave48_wanted(a);   // (a W_t stream only enters the simulation: k_sim_prologue_w48 factors it per step; V_t would enter the backward gain)
}

template <int DT, int PT>
static hipError_t launch_w48_sims_k(const KArgs& a, int K, const double* xplus, const double* innov, hipStream_t s) {
  const size_t lds = sizeof(double) * w48::lds_doubles(DT, PT);
  const int kf = (a.spf && !a.f_stride && !(a.flags & DLM_OPT_NO_SPARSE_F)) ? a.spf_k : 0;   // 0: dense (or time-varying) F
  if (K <= 2) {
    if (kf == 1) hipLaunchKernelGGL((w48::k_simsmooth_w48<DT, PT, 2, 1>), dim3(a.N), dim3(64), lds, s, a, xplus, innov);
    else if (kf > 1) hipLaunchKernelGGL((w48::k_simsmooth_w48<DT, PT, 2, 4>), dim3(a.N), dim3(64), lds, s, a, xplus, innov);
    else hipLaunchKernelGGL((w48::k_simsmooth_w48<DT, PT, 2, 0>), dim3(a.N), dim3(64), lds, s, a, xplus, innov);
  } else {
    if (kf >= 1) hipLaunchKernelGGL((w48::k_simsmooth_w48<DT, PT, 4, 4>), dim3(a.N), dim3(64), lds, s, a, xplus, innov);
    else hipLaunchKernelGGL((w48::k_simsmooth_w48<DT, PT, 4, 0>), dim3(a.N), dim3(64), lds, s, a, xplus, innov);
  }
  return hipGetLastError();
}

hipError_t launch_wave48_simsmooth(const KArgs& a, int K, double* xplus, double* ystar, hipStream_t s) {
  {
    const int kf = (a.spf && !a.f_stride && !(a.flags & DLM_OPT_NO_SPARSE_F)) ? a.spf_k : 0;
    const size_t lds = sizeof(double) * (48 * 49 + 32 * 33 + (kf ? 0 : w48::FIMG) + 48 + 48 + 96);
    if (K <= 2) {
      if (kf == 1) hipLaunchKernelGGL((w48::k_sim_prologue_w48<2, 1>), dim3(a.N), dim3(64), lds, s, a, xplus, ystar);
      else if (kf > 1) hipLaunchKernelGGL((w48::k_sim_prologue_w48<2, 4>), dim3(a.N), dim3(64), lds, s, a, xplus, ystar);
      else hipLaunchKernelGGL((w48::k_sim_prologue_w48<2, 0>), dim3(a.N), dim3(64), lds, s, a, xplus, ystar);
    } else {
      if (kf >= 1) hipLaunchKernelGGL((w48::k_sim_prologue_w48<4, 4>), dim3(a.N), dim3(64), lds, s, a, xplus, ystar);
      else hipLaunchKernelGGL((w48::k_sim_prologue_w48<4, 0>), dim3(a.N), dim3(64), lds, s, a, xplus, ystar);
    }
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  KArgs f = a;
  f.y = ystar; f.z = nullptr; f.theta = nullptr; f.stats = nullptr; f.fq = nullptr; f.prior = nullptr; f.loglik = nullptr;
  e = launch_wave48_filter_z(f, K, ystar, 1, s);
  if (e != hipSuccess) return e;
  KArgs b = a;
  b.filt_in = a.filt;
  const bool d2 = a.d <= 32, p1 = a.p <= 16;
  if (a.d <= 16 && p1) return launch_w48_sims_k<1, 1>(b, K, xplus, ystar, s);
  if (a.d <= 16) return launch_w48_sims_k<1, 2>(b, K, xplus, ystar, s);
  if (d2 && p1) return launch_w48_sims_k<2, 1>(b, K, xplus, ystar, s);
  if (d2) return launch_w48_sims_k<2, 2>(b, K, xplus, ystar, s);
  if (p1) return launch_w48_sims_k<3, 1>(b, K, xplus, ystar, s);
  return launch_w48_sims_k<3, 2>(b, K, xplus, ystar, s);
}

// reference-form backward sampler on register tiles, 16 <= d <= 48 with a structured G (a.filt_in -> theta / cond / stats)
bool wave48_sampler_supported(const KArgs& a) { return tiled_supported(a) && a.spb && wave48_wanted(a); }

template <int DT>
static hipError_t launch_w48_sampler_k(const KArgs& a, int K, hipStream_t s) {
  const size_t lds = sizeof(double) * (size_t)(2 * 16 * DT * (16 * DT + 1) + 6 * 16 * DT);
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0 && a.stats;
  if (K <= 2) {
    if (outer) hipLaunchKernelGGL((w48::k_sampler_w48<DT, 2, true>), dim3(a.N), dim3(64), lds, s, a, SampTabs{});
    else hipLaunchKernelGGL((w48::k_sampler_w48<DT, 2, false>), dim3(a.N), dim3(64), lds, s, a, SampTabs{});
  } else {
    if (outer) hipLaunchKernelGGL((w48::k_sampler_w48<DT, 4, true>), dim3(a.N), dim3(64), lds, s, a, SampTabs{});
    else hipLaunchKernelGGL((w48::k_sampler_w48<DT, 4, false>), dim3(a.N), dim3(64), lds, s, a, SampTabs{});
  }
  return hipGetLastError();
}
hipError_t launch_wave48_sampler(const KArgs& a, hipStream_t s) {
  if (a.d <= 32) return launch_w48_sampler_k<2>(a, a.spb_k, s);
  return launch_w48_sampler_k<3>(a, a.spb_k, s);
}

// ---- shared factors: regular grid, time-invariant F / V / W, V, W, C0 shared by the batch -------------------------
bool wave48_sampler_shared_model_ok(const KArgs& a) {
  const size_t rec = (size_t)a.d + (size_t)a.d * a.d;
  return tiled_supported(a) && a.d >= 16 && !a.g_index && !a.dt && !a.f_stride && !a.v_tstride && !a.w_tstride && !a.v_stride && !a.w_stride && !a.c0_stride &&
         !a.packed && a.T >= 1 && a.T <= 100000 && ((size_t)a.T + 1) * rec * 8 < ((size_t)1 << 31);
}
bool wave48_sampler_shared_eligible(const KArgs& a) {
  return wave48_sampler_shared_model_ok(a) && wave48_sampler_supported(a) && wave48_filter_supported(a) && !a.cond && (!a.stats || a.y) &&
         !(a.flags & (DLM_OPT_FORCE_GENERIC | DLM_OPT_NO_SAMPLER16 | DLM_OPT_SAMPLER_PER_SERIES));
}
static size_t up64w(size_t x) { return (x + 63) & ~(size_t)63; }
static int ws_dt(const KArgs& a) { return a.d <= 32 ? 2 : 3; }
size_t wave48_sampler_shared_ws_bytes(const KArgs& a) {
  const size_t n1 = (size_t)a.T + 1, rec = (size_t)a.d + (size_t)a.d * a.d, zn = (size_t)a.T * a.p > 64 ? (size_t)a.T * a.p : 64;
  return up64w(n1 * w48::ws_row_doubles(ws_dt(a)) * 8) + up64w(n1 * rec * 8) + up64w(zn * 8) + up64w(n1) + 64 + up64w(wave48_ktab_doubles(a) * 8);
}
void wave48_sampler_shared_carve(void* ws, const KArgs& a, SampTabs& tb) {
  const size_t n1 = (size_t)a.T + 1, rec = (size_t)a.d + (size_t)a.d * a.d, zn = (size_t)a.T * a.p > 64 ? (size_t)a.T * a.p : 64;
  char* p = (char*)ws;
  tb.rows = (double*)p;  p += up64w(n1 * w48::ws_row_doubles(ws_dt(a)) * 8);
  tb.zrec = (double*)p;  p += up64w(n1 * rec * 8);
  tb.zeros = (double*)p; p += up64w(zn * 8);
  tb.need = (unsigned char*)p; p += up64w(n1);
  tb.status = (int*)p; tb.settle = tb.status + 1; p += 64;
  tb.ktab = (double*)p;
}
template <int DT>
static void launch_w48_tables_k(const KArgs& kp, const SampTabs& tb, bool outer, hipStream_t s) {
  const size_t lds = sizeof(double) * (size_t)(2 * 16 * DT * (16 * DT + 1) + 6 * 16 * DT);
  const dim3 grid(w48::ws_stretches(kp.T));   // one wave per stretch
  // (the instantiation launch_w48_sampler_k picks for the series of this call that compute their own factors)
  if (kp.spb_k <= 2) {
    if (outer) hipLaunchKernelGGL((w48::k_sampler_w48<DT, 2, true>), grid, dim3(64), lds, s, kp, tb);
    else hipLaunchKernelGGL((w48::k_sampler_w48<DT, 2, false>), grid, dim3(64), lds, s, kp, tb);
  } else {
    if (outer) hipLaunchKernelGGL((w48::k_sampler_w48<DT, 4, true>), grid, dim3(64), lds, s, kp, tb);
    else hipLaunchKernelGGL((w48::k_sampler_w48<DT, 4, false>), grid, dim3(64), lds, s, kp, tb);
  }
}
hipError_t launch_wave48_sampler_shared_tables(const KArgs& a, const SampTabs& tb, hipStream_t s, hipEvent_t after_filter) {
  const size_t zn = (size_t)a.T * a.p > 64 ? (size_t)a.T * a.p : 64;
  hipError_t err = hipMemsetAsync(tb.zeros, 0, zn * 8, s);
  if (err != hipSuccess) return err;
  if ((err = hipMemsetAsync(tb.status, 0, sizeof(int), s)) != hipSuccess) return err;
  KArgs kf = a;   // the filter on a series of zeros: the covariances of every series without a missing observation, bit for bit
  kf.N = 1; kf.y = tb.zeros; kf.m0 = tb.zeros; kf.m0_stride = 0; kf.filt = tb.zrec; kf.status = tb.status; kf.stats = nullptr; kf.loglik = nullptr;
  kf.prior = nullptr; kf.fq = nullptr; kf.route = nullptr; kf.counters = nullptr; kf.theta = nullptr; kf.z = nullptr; kf.series_offset = 0; kf.keep_cov = nullptr;
  kf.flags |= DLM_OPT_FORCE_WAVE;   // the kernel family that filters the batch, whatever the batch size
  if ((err = hipMemsetD32Async((hipDeviceptr_t)tb.settle, a.T, 1, s)) != hipSuccess) return err;
  kf.settle_step = tb.settle;       // (stops where its covariance recursion has settled: within 30 steps for the C4 model)
  kf.ktab = tb.ktab;                // ... and leaves the steady gain there
  if ((err = launch_wave48_filter(kf, kf.spb_k, nullptr, s)) != hipSuccess) return err;
  if (after_filter && (err = hipEventRecord(after_filter, s)) != hipSuccess) return err;
  KArgs kp = kf;
  kp.y = nullptr; kp.filt_in = tb.zrec; kp.settle_step = nullptr; kp.ktab = nullptr;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0 && a.stats;
  if (a.d <= 32) launch_w48_tables_k<2>(kp, tb, outer, s); else launch_w48_tables_k<3>(kp, tb, outer, s);
  return hipGetLastError();
}
template <int DT>
static void launch_w48_mean_k(const KArgs& a, const SampTabs& tb, hipStream_t s) {
  const size_t lds = sizeof(double) * (size_t)(6 * 16 * DT + (16 * DT * (16 * DT + 1)) / 2 + a.p * (a.d + 1));
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0 && a.stats;
  if (a.spb_k <= 2) {
    if (outer) hipLaunchKernelGGL((w48::k_mean_sampler_w48<DT, 2, true>), dim3(a.N), dim3(64), lds, s, a, tb);
    else hipLaunchKernelGGL((w48::k_mean_sampler_w48<DT, 2, false>), dim3(a.N), dim3(64), lds, s, a, tb);
  } else {
    if (outer) hipLaunchKernelGGL((w48::k_mean_sampler_w48<DT, 4, true>), dim3(a.N), dim3(64), lds, s, a, tb);
    else hipLaunchKernelGGL((w48::k_mean_sampler_w48<DT, 4, false>), dim3(a.N), dim3(64), lds, s, a, tb);
  }
}
size_t wave48_sampler_shared_normals_bytes(const KArgs& a) { return (size_t)a.N * ((size_t)a.T + 1) * a.d * 8; }
hipError_t launch_wave48_sampler_shared_normals(const KArgs& a, double* z, hipStream_t s) {
  const long long total = (long long)a.N * (a.T + 1) * ((a.d + 1) / 2);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(w48::k_normals_rows, dim3((unsigned)(blocks < 65536 * 4 ? blocks : 65536 * 4)), dim3(256), 0, s, a, z);
  return hipGetLastError();
}
size_t wave48_ktab_doubles(const KArgs& a) { const int PT = a.p <= 16 ? 1 : 2, DT = a.d <= 16 ? 1 : (a.d <= 32 ? 2 : 3); return (size_t)16 * PT * w48::il_of(DT, PT); }
template <int DT, int PT>
static void launch_w48_steady_k(const KArgs& a, const double* ktab, const int* settle, hipStream_t s) {
  const int kf = (a.spf && !(a.flags & DLM_OPT_NO_SPARSE_F)) ? a.spf_k : 0;
  const size_t lds = sizeof(double) * (size_t)(16 * PT * w48::il_of(DT, PT) + (kf ? 0 : w48::fimg_of(DT, PT)) + 3 * w48::vl_of(DT, PT));
  const int K = a.spb_k;
  if (K <= 2) {
    if (kf == 1) hipLaunchKernelGGL((w48::k_steady_filter_w48<DT, PT, 2, 1>), dim3(a.N), dim3(64), lds, s, a, ktab, settle);
    else if (kf > 1) hipLaunchKernelGGL((w48::k_steady_filter_w48<DT, PT, 2, 4>), dim3(a.N), dim3(64), lds, s, a, ktab, settle);
    else hipLaunchKernelGGL((w48::k_steady_filter_w48<DT, PT, 2, 0>), dim3(a.N), dim3(64), lds, s, a, ktab, settle);
  } else {
    if (kf >= 1) hipLaunchKernelGGL((w48::k_steady_filter_w48<DT, PT, 4, 4>), dim3(a.N), dim3(64), lds, s, a, ktab, settle);
    else hipLaunchKernelGGL((w48::k_steady_filter_w48<DT, PT, 4, 0>), dim3(a.N), dim3(64), lds, s, a, ktab, settle);
  }
}
hipError_t launch_wave48_steady_filter(const KArgs& a, const double* ktab, const int* settle, hipStream_t s) {
  const bool d2 = a.d <= 32, p1 = a.p <= 16;
  if (a.d <= 16 && p1) launch_w48_steady_k<1, 1>(a, ktab, settle, s);
  else if (a.d <= 16) launch_w48_steady_k<1, 2>(a, ktab, settle, s);
  else if (d2 && p1) launch_w48_steady_k<2, 1>(a, ktab, settle, s);
  else if (d2) launch_w48_steady_k<2, 2>(a, ktab, settle, s);
  else if (p1) launch_w48_steady_k<3, 1>(a, ktab, settle, s);
  else launch_w48_steady_k<3, 2>(a, ktab, settle, s);
  return hipGetLastError();
}
hipError_t launch_wave48_mark_gaps(const KArgs& a, unsigned char* route, hipStream_t s) {
  hipLaunchKernelGGL(w48::k_mark_gaps_w48, dim3(a.N), dim3(256), 0, s, a.y, a.N, a.T * a.p, route);
  return hipGetLastError();
}
hipError_t launch_wave48_sampler_shared_draw(const KArgs& a, const SampTabs& tb, hipStream_t s) {
  if (!a.route) return hipErrorInvalidValue;
  hipError_t err;
  if (tb.marked) { /* the engine marked the gaps before the forward pass */ }
  else if (a.y) hipLaunchKernelGGL(w48::k_mark_gaps_w48, dim3(a.N), dim3(256), 0, s, a.y, a.N, a.T * a.p, a.route);
  else if ((err = hipMemsetAsync(a.route, 0, (size_t)a.N, s)) != hipSuccess) return err;
  if ((err = hipGetLastError()) != hipSuccess) return err;
  KArgs km = a;
  km.route_take = 0;
  if (!km.z) km.z = tb.z4;          // the call's normals made beside the filter (k_normals_rows); nullptr: the draw kernel makes them itself
  if (a.d <= 32) launch_w48_mean_k<2>(km, tb, s); else launch_w48_mean_k<3>(km, tb, s);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  KArgs kg = a;   // the series with a missing observation: their own factors
  kg.route_take = 1;
  return launch_wave48_sampler(kg, s);
}

}  // namespace dlm

