// The reference's backward sampler, literally, for d <= 15 with a structured G -- one wavefront per series, every d x d
// matrix one 16 x 16 tile in the fp64 MFMA accumulator layout (the single-tile case of dlm_wave48.hip).
//
// Smoothing.sampleDlm / backSampleStep (Smoothing.scala:74-122), per step t = T-1 .. 0, given theta_{t+1}:
//   a+ = G m, R+ = G C G^T + W dt                      (dt == 0: a+ = m, R+ = C; KalmanFilter.scala:279-280)
//   J  = C G^T R+^-1                                   (the table entry g(dt) also for dt == 0, Smoothing.scala:85)
//   h  = m + J (theta_{t+1} - a+)
//   H  = (I - J G) C (I - J G)^T + dt J W J^T          (:88-93), symmetrised (:95)
//   theta_t = h + chol(H) z_t                          (the engine's canonical factor)
// with the conditional-moment records and the Gibbs statistics of k_sampler_generic (dlm_generic.hip), whose draws it
// reproduces.  What makes it fast: G C G^T, C G^T and J G are gathers through a wave-private LDS image (G has at most
// four nonzeros per row and column), the five dense products are MFMA chains on registers, R+^-1 is a Newton-Schulz
// refinement of the previous step's inverse (R+ changes slowly along a series; direct Cholesky inverse when the start
// is too far off), and the factor of H -- thirteen dependent pivots -- is taken in registers, lane i holding row i, with
// pivots and multipliers broadcast by v_readlane (chol_draw).
//
// Also here (round 3): the shared factors of a pooled batch -- k_sampler_sp16 with its table export (one wave per
// stretch of 32 steps on a series of zeros), k_mean_sampler_sp16 (the draw against the table, four series per wave, means / table rows
// / normals by LDS DMA), k_normals4 (the call's normals made beside the forward pass), k_mark_gaps, and their launchers.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {
namespace s16 {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr int IL = 17;
constexpr int IMG = 16 * IL;
constexpr int OOB = 0x7ffffff0;

__device__ __forceinline__ void wave_sync() {
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
// X^T Y over the first `rows` rows (lane 16 g + c, register r <-> element (4 r + g, c))
__device__ __forceinline__ d4 mm(const d4& x, const d4& y, int rows) {
  d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (4 * r < rows) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[r], y[r], acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ void to_img(const d4& x, double* img, int g, int c) {
#pragma unroll
  for (int r = 0; r < 4; ++r) img[(4 * r + g) * IL + c] = x[r];
}
__device__ __forceinline__ d4 transposed(const d4& x, double* img, int g, int c) {
  to_img(x, img, g, c);
  wave_sync();
  d4 y;
#pragma unroll
  for (int r = 0; r < 4; ++r) y[r] = img[c * IL + 4 * r + g];
  wave_sync();
  return y;
}
// y[c] = sum_i M[i][c] x[i]  (x: LDS vector, zero beyond its length)
__device__ __forceinline__ double matTvec(const d4& M, const double* x, int g) {
  double acc = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) acc = fma(M[r], x[4 * r + g], acc);
  return sum_g(acc);
}
// in-place lower Cholesky of the n x n row-major LDS matrix (leading dimension IL).  psd: a non-positive pivot gives a zero
// column (a direction without variance, as the oracle's factor) instead of failing the solve that follows.
__device__ __forceinline__ bool chol_rows(double* L, int n, int lane, bool psd) {
  bool bad = false;
  for (int k = 0; k < n; ++k) {
    double akk = L[k * IL + k];
    const bool np = !(akk > 0.0);
    if (np) { bad = true; akk = 1e-300; }
    const double lkk = (np && psd) ? 0.0 : sqrt(akk), inv = (np && psd) ? 0.0 : 1.0 / lkk;
    wave_sync();
    if (lane >= k && lane < n) L[lane * IL + k] = (lane == k) ? lkk : L[lane * IL + k] * inv;
    wave_sync();
    {   // trailing update, a lane's four elements (4 r + g, c) of the lower triangle
      const int g_ = lane >> 4, j = lane & 15;
      if (j > k && j < n) {
        const double ljk = L[j * IL + k];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * r + g_;
          if (i >= j && i < n) L[i * IL + j] = fma(-L[i * IL + k], ljk, L[i * IL + j]);
        }
      }
    }
    wave_sync();
  }
  return bad;
}
// X = (SPD Q)^-1 by Cholesky and n triangular solves in LDS (img: factor, B: the inverse)
__device__ __forceinline__ bool direct_inverse(const d4& Q, d4& X, int n, double* img, double* B, int lane, int g, int c) {
  to_img(Q, img, g, c);
  wave_sync();
  const bool bad = chol_rows(img, n, lane, false);
  if (lane < n) {
    double* x = B + lane;
    for (int i = 0; i < n; ++i) {
      double s = (i == lane) ? 1.0 : 0.0;
      for (int l = 0; l < i; ++l) s = fma(-img[i * IL + l], x[l * IL], s);
      x[i * IL] = s / img[i * IL + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = x[i * IL];
      for (int l = i + 1; l < n; ++l) s = fma(-img[l * IL + i], x[l * IL], s);
      x[i * IL] = s / img[i * IL + i];
    }
  }
  wave_sync();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    X[r] = (i < n && c < n) ? 0.5 * (B[i * IL + c] + B[c * IL + i]) : 0.0;
  }
  wave_sync();
  return bad;
}

// theta = h + chol(H) z with the factorisation in REGISTERS: lane i < d takes row i of H from the image, every pivot and
// every multiplier travels by v_readlane (the indices are compile-time), so the thirteen dependent pivots cost no LDS
// round trip and no barrier.  A non-positive pivot gives a zero column (a direction without variance, as the oracle's
// factor) and is reported.  hcol / zc: h[lane], z[lane] in the lanes 0..d-1; returns theta[lane] there.
__device__ __forceinline__ double readlane_d(double v, int src) {
  const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void chol_factor(const double* img, int d, int lane, double (&row)[15], bool& bad) {
  const double* src = img + (lane < 16 ? lane : 0) * IL;
#pragma unroll
  for (int j = 0; j < 15; ++j) row[j] = src[j];
#pragma unroll
  for (int k = 0; k < 15; ++k)
    if (k < d) {
      const double akk = readlane_d(row[k], k);
      const bool np = !(akk > 0.0);
      bad |= np;
      double inv = __builtin_amdgcn_rsq(np ? 1.0 : akk);         // 1/sqrt by the hardware estimate and two Newton steps (about 1 ulp)
      inv = inv * fma(-0.5 * akk * inv, inv, 1.5);
      inv = inv * fma(-0.5 * akk * inv, inv, 1.5);
      inv = np ? 0.0 : inv;
      const double lik = row[k] * inv;                            // L[lane][k] for lane >= k (the pivot itself: akk / sqrt(akk))
      row[k] = lik;
#pragma unroll
      for (int j = k + 1; j < 15; ++j)
        if (j < d) row[j] = fma(-lik, readlane_d(lik, j), row[j]);
    }
}
// theta[lane] = h[lane] + sum_{k <= lane} L[lane][k] z[k]   (lanes 0..d-1)
__device__ __forceinline__ double chol_draw(const double (&row)[15], int d, int lane, double hcol, double zc) {
  double th = hcol;
#pragma unroll
  for (int k = 0; k < 15; ++k)
    if (k < d) th = fma((lane >= k) ? row[k] : 0.0, readlane_d(zc, k), th);
  return th;
}

// Shared factors.  J_t, H_t and the factor of H_t depend on the filtered COVARIANCES alone: when the batch shares
// V, W, C0 on a regular grid they are those of every series without a missing observation.  One wave then runs k_sampler_sp16
// on the records of a series of zeros with EXP set and leaves, per step that computed them, a table row; every series draws
// with k_mean_sampler_sp16 below -- four series per wave, the mean recursion and the draw only.
//   row t (SF_ROW doubles): per component c < d  [ J_t^T[0..15][c] | L_t[c][0..14], 0 | 2 pad ]      need[t] = 1
//   a step the producer took in its steady form leaves no row (need[t] = 0): the factors are those of the last row above it.
constexpr int SF_ENT = 34;                 // 272 bytes per component: sixteen lanes' 16-byte LDS reads fall into distinct banks
constexpr int SF_ROW = 16 * SF_ENT;
constexpr int SF_SLOT = 4096;              // LDS bytes of a ring slot (15 * 272 = 4080 at most travel)
constexpr int SF_AHEAD = 8;                // the filtered means are requested this many steps ahead
// shared factors of the RTS smoother (k_smoother_rts16<EXP>, k_mean_rts16): a row of RtsTabs::jrows
constexpr int RJ_ENT = 18;                 // 144 bytes per component: sixteen lanes' 16-byte LDS reads fall into distinct banks
constexpr int RJ_ROW = 16 * RJ_ENT;
constexpr int RS_SLOT = 1936;              // LDS bytes of a record slot: a record of d = 15 is 1920 bytes, + 16
// Stretches: steps t at the top of one start from scratch (k_sampler_sp16, every series).  32 steps long -- and 16 below step SF_SHORT_END, where
// the filtered covariance of a typical model is still moving and every step of a stretch is computed in full: the longest stretch is what the
// draw kernel of a shared-factor call waits for (C3: the table 1.24 -> 0.7 ms, no longer behind the forward mean kernel).
constexpr int SF_STRETCH = 32, SF_SHORT = 16, SF_SHORT_END = 384;
__host__ __device__ constexpr bool sf_stretch_top(int t) { return (t & (SF_STRETCH - 1)) == SF_STRETCH - 1 || (t < SF_SHORT_END && (t & (SF_SHORT - 1)) == SF_SHORT - 1); }
__host__ __device__ constexpr int sf_stretches(int T) { return T <= SF_SHORT_END ? (T + SF_SHORT - 1) / SF_SHORT : SF_SHORT_END / SF_SHORT + (T - SF_SHORT_END + SF_STRETCH - 1) / SF_STRETCH; }
__host__ __device__ constexpr int sf_stretch_lo(int b) { return b < SF_SHORT_END / SF_SHORT ? b * SF_SHORT : SF_SHORT_END + (b - SF_SHORT_END / SF_SHORT) * SF_STRETCH; }

// Tab: SparseT (the d <= 15, p = 1 tables) or SparseBig (the tables of the multivariate paths); any p <= 64 -- the
// observations only enter the statistics.
template <int K, class Tab, bool EXP = false>
__global__ __launch_bounds__(64, 2) void k_sampler_sp16(KArgs a, const Tab* __restrict__ sp, SampTabs tb) {   // two waves per SIMD: the step is a long dependent chain
  if (!EXP && a.route && (a.route[blockIdx.x] != 0) != (a.route_take != 0)) return;   // shared-factor call: only the series routed here
  if constexpr (EXP) __builtin_amdgcn_s_setprio(3);   // the table's few waves run beside the kernel that filters the batch and are what the draw kernel waits for
  __shared__ __attribute__((aligned(16))) double lds[2 * IMG + 8 * 16 + 64];
  double* img = lds;       double* inv = lds + IMG;
  double* mv = inv + IMG;  double* thv = mv + 16;   double* uv = thv + 16;   double* zv = uv + 16;
  double* hv = zv + 16;    double* dfv = hv + 16;   double* zq = mv + 8 * 16;   // zq: 64 normals, four records x 16 components
  // EXP: the series is the one series of zeros and workgroup b makes the rows of the steps of stretch b (a stretch starts from
  // scratch, see SF_STRETCH)
  const int n = EXP ? 0 : blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8;
  const int t_lo = EXP ? sf_stretch_lo((int)blockIdx.x) : 0;
  const int t_nx = EXP ? sf_stretch_lo((int)blockIdx.x + 1) : T;
  const int t_hi = (t_nx < T ? t_nx : T) - 1;
  const bool vc = c < d;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;
  for (int i = lane; i < 8 * 16; i += 64) mv[i] = 0.0;

  int ridx[K], cidx[K];          // nonzeros of row c / column c of G
  double rval[K], cval[K];
  int gcur = -1;
  auto load_tables = [&](int gi) {
#pragma unroll
    for (int s = 0; s < K; ++s) {
      ridx[s] = sp[2 * gi].idx[c][s];      rval[s] = vc ? sp[2 * gi].val[c][s] : 0.0;
      cidx[s] = sp[2 * gi + 1].idx[c][s];  cval[s] = vc ? sp[2 * gi + 1].val[c][s] : 0.0;
    }
    gcur = gi;
  };
  const double* W0 = a.W + (size_t)n * a.w_stride;
  d4 Wt;
  bool va[4];
  int offC[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    va[r] = i < d && vc;
    offC[r] = va[r] ? (d + i + c * d) * 8 : OOB;      // C[i][c] of a record, column-major
    Wt[r] = va[r] ? W0[i + c * d] : 0.0;
  }
  const int offM = (vc && g == 0) ? c * 8 : OOB;
  const int rstr = (EXP && tb.zstride) ? tb.zstride : recb;   // bytes between the input records (the table kernel may read the rows of a covariance table)
  const __amdgpu_buffer_rsrc_t rin = mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * rstr);
  double* cond = a.cond ? a.cond + (size_t)n * (T + 1) * rec : nullptr;
  const __amdgpu_buffer_rsrc_t rco = mk_rsrc(cond, cond ? (size_t)(T + 1) * recb : 0);
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  const double* y = a.y ? a.y + (size_t)n * T * p : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * d : nullptr;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  int st = 0;
  d4 OUT = {0.0, 0.0, 0.0, 0.0}, Rinv = {0.0, 0.0, 0.0, 0.0};
  double ssd = 0.0, ssy = 0.0, nob = 0.0;
  bool warm = false;

  // theta = h + chol(H) z, with the conditional record; leaves theta in thv and returns this lane's component
  // conditional record, then theta = h + L z with the factor rows L (lanes 0..15); leaves nothing in LDS
  auto emit = [&](int t, double hcol, const d4& H, double zc, const double (&rows)[15]) {
    if (cond) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bst(rco, offC[r], t * recb, H[r]);
      bst(rco, offM, t * recb, hcol);
    }
    const double v = chol_draw(rows, d, lane, hcol, vc ? zc : 0.0);
    return (vc && g == 0) ? v : 0.0;
  };
  auto export_row = [&](int t, const d4* JT, const double (&rows)[15]) {
    double* row = tb.rows + (size_t)t * SF_ROW + c * SF_ENT;
    if (vc) {
#pragma unroll
      for (int r = 0; r < 4; ++r) row[4 * r + g] = JT ? (*JT)[r] : 0.0;
      if (g == 0) {
#pragma unroll
        for (int k = 0; k < 15; ++k) row[16 + k] = (lane >= k && k < d) ? rows[k] : 0.0;   // as chol_draw takes them
        row[31] = 0.0;
      }
    }
    if (lane == 0) tb.need[t] = 1;
  };
  auto factor = [&](const d4& H, double (&rows)[15]) {
    to_img(H, img, g, c);
    wave_sync();
    bool bad = false;
    chol_factor(img, d, lane, rows, bad);
    if (bad) st |= DLM_ST_NOT_PD;
    wave_sync();
  };

  if (!EXP || t_hi == T - 1) {   // theta_T ~ N(m_T, C_T)
    d4 C;
#pragma unroll
    for (int r = 0; r < 4; ++r) C[r] = bld(rin, offC[r], T * rstr);
    const double mc = bld(rin, vc ? c * 8 : OOB, T * rstr);
    const double zc = (vc && !EXP) ? (zin ? zin[(size_t)T * d + c] : philox_normal(a.seed, series, (unsigned)T, (unsigned)c)) : 0.0;
    double rows[15];
    factor(C, rows);
    if constexpr (EXP) export_row(T, nullptr, rows);
    const double th = emit(T, mc, C, zc, rows);
    if (g == 0) thv[c] = th;
    if (thout && g == 0 && vc) thout[(size_t)T * d + c] = th;
    wave_sync();
  }
  d4 Cp = {0.0, 0.0, 0.0, 0.0}, JTs = Cp, Hs = Cp;   // steady-state reuse: the covariance, J^T, H and the factor of the last full step
  double Ls[15];
#pragma unroll
  for (int j = 0; j < 15; ++j) Ls[j] = 0.0;
  bool have = false;
  int gprev = -1;
  double dtprev = 0.0, cmaxp = 0.0;
  // the record of the next step travels while this one is computed
  d4 nC;
  double nm;
  {
    const int tp = t_hi > 0 ? t_hi : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) nC[r] = bld(rin, offC[r], tp * rstr);
    nm = bld(rin, vc ? c * 8 : OOB, tp * rstr);
  }
  const bool f1 = p == 1 && !a.f_stride;
  const double Fl1 = (f1 && lane < d) ? a.F[lane] : 0.0;
  double ych = 0.0;
  for (int t = t_hi; t >= t_lo; --t) {
    // Every SF_STRETCH steps the recursion starts from scratch -- a full step, its inverse not refined from the one before: what a
    // step computes then depends on nothing above its stretch, and the stretches of a shared-factor table are made side by side.
    if ((EXP || a.stretches) && sf_stretch_top(t)) { have = false; warm = false; }
    // the normals of records t, t-1, t-2, t-3 (16 components each) are drawn together, one per lane, every fourth step:
    // the generator is the same few hundred instructions whether 13 lanes or 64 need a value
    if (!EXP && !zin && ((T - 1 - t) & 3) == 0) {
      const int tr = t - g;
      zq[lane] = (vc && tr >= 0) ? philox_normal(a.seed, series, (unsigned)tr, (unsigned)c) : 0.0;
      wave_sync();
    }
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) load_tables(gi);
    if (a.stats && y && f1) {
      // one observation component, time-invariant F (C2 / C3): F.theta by a wave reduction on registers.  The loop below issues d
      // dependent global loads -- ~4000 cycles of every step for a wave with one neighbour on its SIMD (measured: the whole
      // steady step took 3.6 us)
      const int kk = (T - 1 - t) & 63;                       // the observations 64 at a time: a load per step would be waited for at once,
      if (kk == 0) ych = (t - lane >= 0) ? y[t - lane] : 0.0;   // together with the record requested for the next step
      const double yv = readlane_d(ych, kk);
      double part = lane < d ? Fl1 * thv[lane] : 0.0;
      for (int o_ = 8; o_ > 0; o_ >>= 1) part += __shfl_xor(part, o_);
      if (lane == 0 && yv == yv) { ssy += (yv - part) * (yv - part); nob += 1.0; }
    } else if (a.stats && y && lane < p) {   // observation residual of theta_{t+1}, component `lane` (Gibbs.scala:29-39)
      const double yv = y[(size_t)t * p + lane];
      if (yv == yv) {
        const double* Fj = a.F + (size_t)t * a.f_stride + (size_t)lane * d;
        double f = 0.0;
        for (int k = 0; k < d; ++k) f = fma(Fj[k], thv[k], f);
        ssy += (yv - f) * (yv - f); nob += 1.0;
      }
    }
    const d4 C = nC;
    const double mc = nm;
    {
      const int tp = t > 0 ? t - 1 : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) nC[r] = bld(rin, offC[r], tp * rstr);
      nm = bld(rin, vc ? c * 8 : OOB, tp * rstr);
    }
    const double zc = (vc && !EXP) ? (zin ? zin[(size_t)t * d + c] : zq[16 * ((T - 1 - t) & 3) + c]) : 0.0;
    if (a.w_tstride) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = va[r] ? (W0 + (size_t)t * a.w_tstride)[4 * r + g + c * d] : 0.0;
    }
    if (g == 0) mv[c] = mc;
    // Steady state: once the filtered covariance has stopped moving (a regular stretch without missing observations), J, H
    // and its factor are those of the step before -- only a+, h and the draw are left to do.
    bool reuse = false;
    if (have && gi == gprev && dt == dtprev && !a.w_tstride) {
      bool moved = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) moved |= !(fabs(C[r] - Cp[r]) <= DLM_SETTLE_TOL * cmaxp);   // THIS step's record against the covariance J, H and the factor were computed from: a bound on the distance itself (no drift can build up), relative to the largest entry of C
      reuse = __ballot(moved) == 0ull;
    }
    d4 JT, H;
    double hcol;
    if (reuse) {
      wave_sync();
      double a1 = mc;
      if (dt != 0.0) {
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < K; ++s) s_ = fma(mv[ridx[s]], rval[s], s_);
        a1 = s_;
      }
      JT = JTs; H = Hs;
      if (g == 0) uv[c] = vc ? thv[c] - a1 : 0.0;
      wave_sync();
      hcol = mc + matTvec(JT, uv, g);
      if constexpr (EXP) { if (lane == 0) tb.need[t] = 0; }
    } else {
    to_img(C, img, g, c);
    wave_sync();
    // C G^T (pass 1), a+ and R+
    d4 T1, R;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double* row = img + (4 * r + g) * IL;
      double s_ = 0.0;
#pragma unroll
      for (int s = 0; s < K; ++s) s_ = fma(row[ridx[s]], rval[s], s_);
      T1[r] = s_;
    }
    double a1 = mc;
    if (dt != 0.0) {
      double s_ = 0.0;
#pragma unroll
      for (int s = 0; s < K; ++s) s_ = fma(mv[ridx[s]], rval[s], s_);
      a1 = s_;
    }
    wave_sync();
    const d4 GC = transposed(T1, img, g, c);         // (C G^T)^T = G C
    if (dt != 0.0) {
      // pass 2 on the image of C G^T:  R[i][j] = sum_s (C G^T)[idx_s(j)][i] val_s(j) + W dt
      to_img(T1, img, g, c);
      wave_sync();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        double s_ = Wt[r] * dt;
#pragma unroll
        for (int s = 0; s < K; ++s) s_ = fma(img[ridx[s] * IL + i], rval[s], s_);
        R[r] = s_;
      }
      wave_sync();
      const d4 Rt = transposed(R, img, g, c);
#pragma unroll
      for (int r = 0; r < 4; ++r) R[r] = 0.5 * (R[r] + Rt[r]);
    } else R = C;
    // R+^-1: Newton-Schulz from the previous step's inverse, direct otherwise
    {
      bool done = false;
      if (warm) {
        for (int it = 0; it < 6 && !done; ++it) {
          d4 E = mm(R, Rinv, d);
          bool big = false, far = false;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 4 * r + g;
            const double e = ((i == c) ? 1.0 : 0.0) - E[r];
            E[r] = e;
            if (i < d && vc) { big |= !(fabs(e) <= 2e-10); far |= !(fabs(e) * d < 0.5); }
          }
          const bool any_big = __ballot(big) != 0ull, any_far = __ballot(far) != 0ull;
          if (any_far && any_big) break;
          if (!any_big) done = true;
          const d4 D = mm(Rinv, E, d);
#pragma unroll
          for (int r = 0; r < 4; ++r) Rinv[r] += D[r];
          const d4 Xt = transposed(Rinv, img, g, c);
#pragma unroll
          for (int r = 0; r < 4; ++r) Rinv[r] = 0.5 * (Rinv[r] + Xt[r]);
        }
      }
      if (!done && direct_inverse(R, Rinv, d, img, inv, lane, g, c)) st |= DLM_ST_NOT_PD;
      warm = true;
    }
    JT = mm(Rinv, GC, d);                             // J^T = R+^-1 G C
    if (g == 0) uv[c] = vc ? thv[c] - a1 : 0.0;
    wave_sync();
    hcol = mc + matTvec(JT, uv, g);                   // h = m + J (theta+ - a+)
    // J G by columns of G on the image of J (= J^T written transposed), Dm = I - J G
#pragma unroll
    for (int r = 0; r < 4; ++r) img[c * IL + 4 * r + g] = JT[r];
    wave_sync();
    d4 Dm;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double* row = img + (4 * r + g) * IL;
      double s_ = 0.0;
#pragma unroll
      for (int s = 0; s < K; ++s) s_ = fma(row[cidx[s]], cval[s], s_);
      Dm[r] = ((4 * r + g == c && vc) ? 1.0 : 0.0) - s_;
    }
    wave_sync();
    const d4 DmT = transposed(Dm, img, g, c);
    const d4 CD = mm(C, DmT, d);                      // C Dm^T
    H = mm(DmT, CD, d);                               // Dm C Dm^T
    const d4 WJ = mm(Wt, JT, d);                      // W J^T
    const d4 H2 = mm(JT, WJ, d);                      // J W J^T
#pragma unroll
    for (int r = 0; r < 4; ++r) H[r] = fma(H2[r], dt, H[r]);
    {
      const d4 Ht = transposed(H, img, g, c);
#pragma unroll
      for (int r = 0; r < 4; ++r) H[r] = (4 * r + g == c) ? H[r] : (H[r] + Ht[r]) / 2.0;
    }
    factor(H, Ls);
    if constexpr (EXP) export_row(t, &JT, Ls);
    Cp = C; JTs = JT; Hs = H; have = true; gprev = gi; dtprev = dt;
    {
      double mx = fmax(fmax(fabs(C[0]), fabs(C[1])), fmax(fabs(C[2]), fabs(C[3])));
      for (int o_ = 32; o_ > 0; o_ >>= 1) mx = fmax(mx, __shfl_xor(mx, o_));
      cmaxp = mx;
    }
    }
    const double th = emit(t, hcol, H, zc, Ls);
    if (a.stats) {   // system residual theta_{t+1} - G theta_t (always the table entry)
      if (g == 0) hv[c] = th;
      wave_sync();
      double gth = 0.0;
#pragma unroll
      for (int s = 0; s < K; ++s) gth = fma(hv[ridx[s]], rval[s], gth);
      const double dts = (dt == 0.0) ? 1.0 : dt;
      const double df = vc ? thv[c] - gth : 0.0;
      ssd += df * df / dts;
      if (outer) {
        if (g == 0) dfv[c] = df;
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) OUT[r] += dfv[4 * r + g] * df / dts;
      }
    }
    wave_sync();
    if (g == 0) thv[c] = th;
    if (thout && g == 0 && vc) thout[(size_t)t * d + c] = th;
    wave_sync();
  }
  if (!EXP && __ballot(vc && !isfinite(thv[c])) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.stats) {
    const int L = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * L;
    if (lane < p) { so[lane] = ssy; so[p + lane] = nob; }
    if (lane == 0) so[L - 1] = (double)T;
    if (outer) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (va[r]) so[2 * p + 4 * r + g + c * d] = OUT[r];
    } else if (g == 0 && vc) so[2 * p + c] = ssd;
  }
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
  if (!EXP && a.route && a.counters && lane == 0) atomicAdd(&a.counters[3], 1ull);   // a series of a shared-factor call that computed its own
}

// ---------------------------------------------------------------------------------------------------------------------
// The draw against the shared factors: four series per wave (lane 16 j + c = component c of series j of the wave's four
// consecutive series).  Per step only what depends on the data is left --
//     a+ = G m_t,  h = m_t + J_t (theta_{t+1} - a+),  theta_t = h + L_t z_t,  the Gibbs sums
// -- in the operations of k_sampler_sp16, one for one: the same FMA chains (matTvec's four partial chains and their
// (0 + 1) + (2 + 3) sum, chol_draw's chain over the pivots, the gathers of G in table order), the same normals.  The factors
// live in registers (column c of J^T, row c of L) and are replaced when the table has a row for the step (SampTabs::need);
// rows travel two steps ahead by LDS DMA from L2, the four series' filtered means SF_AHEAD steps ahead from their records.
// All vector-memory traffic of the loop is issued by hand and waited for by count (vm_wait), see dlm_sparse16.hip.
// NR: DMA instructions per table row = ceil(17 d / 64).
// ---------------------------------------------------------------------------------------------------------------------
typedef int i4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ i4 rsrc_words(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i4 r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
          __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  return r;
}
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
template <int OFF>
__device__ __forceinline__ double lds_read64(unsigned addr) {
  double v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int OFF>
__device__ __forceinline__ d2 lds_read128(unsigned addr) {
  d2 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// 16 bytes per lane from byte offset voff + soff of the buffer to LDS address lds_addr + 16 lane (+ 1024 per further instruction)
template <int NR>
__device__ __forceinline__ void dma_row(const i4& rs, unsigned lds_addr, int soff, int lane, int n16) {
  const int voff = lane * 16;
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  // NR = ceil(n16 / 64): every instruction has lanes to serve, so the count of operations in flight is NR whatever the exec mask
  if (lane < n16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if constexpr (NR > 1) if (lane + 64 < n16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if constexpr (NR > 2) if (lane + 128 < n16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if constexpr (NR > 3) if (lane + 192 < n16) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:3072 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ double mul_rn(double x, double y) {   // a product that is rounded before it is used (never contracted into the addition behind it)
#pragma clang fp contract(off)
  return x * y;
}
__device__ __forceinline__ double row_pick(double v, int lane, int src) {   // the value of lane src (0..15, wave-uniform) of this lane's 16-lane row
  const int a_ = ((lane & 48) + src) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(a_, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(a_, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ZD: the normals come from SampTabs::z4 (k_normals4) by LDS DMA, four steps ahead; otherwise they are the injected ones (KArgs::z)
template <int K, int NR, bool ZD>
__global__ __launch_bounds__(64, 3) void k_mean_sampler_sp16(KArgs a, const SparseT* __restrict__ sp, SampTabs tb) {
  __shared__ __attribute__((aligned(16))) double lds[2 * 64 + 4 * 64 + 2 * (SF_SLOT / 8) + SF_AHEAD * 64];
  const int lane = threadIdx.x, j = lane >> 4, c = lane & 15;
  const int n0 = 4 * blockIdx.x;
  if (n0 >= a.N) return;
  const int nser = a.N - n0 < 4 ? a.N - n0 : 4;
  const bool have = j < nser;
  const int n = have ? n0 + j : n0;                 // (a row without a series shadows the first: loads stay in bounds, nothing is stored)
  const bool dead = !have || a.route[n] != 0;       // a series with a missing observation: k_sampler_sp16 serves it
  if (__ballot(!dead) == 0ull) return;
  double* vU = lds;            // theta_{t+1} - a+   [4][16]
  double* vT = vU + 64;        // theta_t
  double* vZ = vT + 64;        // the normals of four steps, [t & 3][4][16]
  char* ring = (char*)(vZ + 4 * 64);               // two table slots, then the slots of the means
  const int d = a.d, T = a.T, rec = d + d * d, recb = rec * 8;
  const bool vc = c < d;
  int idx[K];
  double val[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = (16 * j + sp[0].idx[c][s]) * 8; val[s] = vc ? sp[0].val[c][s] : 0.0; }
  const size_t sbytes = (size_t)(T + 1) * recb;
  // the filtered means: the head of each series' records -- or, when the call keeps no records, the mean-only forward kernel's compact stream
  const bool cm = tb.mc4 != nullptr;
  const int mstep = cm ? 512 : recb;                 // bytes between two steps of the means
  const i4 rmean = cm ? rsrc_words(tb.mc4 + (size_t)(n0 / 4) * (T + 1) * 64, (unsigned)((size_t)(T + 1) * 512))
                      : rsrc_words((const char*)a.filt_in + (size_t)n0 * sbytes, (unsigned)((size_t)nser * sbytes));
  const i4 rtab = rsrc_words(tb.rows, (unsigned)((size_t)(T + 1) * SF_ROW * 8));
  const int mvoff = cm ? lane * 16 : ((lane < 32 && (lane >> 3) < nser) ? (int)((size_t)(lane >> 3) * sbytes) + (lane & 7) * 16 : OOB);   // 16 doubles from the head of each record
  const unsigned ring_lds = lds_addr_of(ring), mring_lds = ring_lds + 2 * SF_SLOT;
  const unsigned vT_lds = lds_addr_of(vT);
  const int n16 = d * (SF_ENT / 2);                  // 16-byte pieces of a row that travel
  auto dma_means = [&](unsigned lds_addr, int soff) {
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    if (lane < 32)   // 4 x 8 pieces; the other lanes' LDS destinations lie beyond the slot
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(mvoff), "s"(rmean), "s"(soff) : "memory");
  };
  const i4 rz = rsrc_words(ZD ? tb.z4 + (size_t)(n0 / 4) * (T + 1) * 64 : nullptr, ZD ? (unsigned)((size_t)(T + 1) * 512) : 0u);
  const unsigned zring_lds = lds_addr_of(vZ);
  const int zvoff = lane * 16;
  auto dma_z = [&](int tz) {   // the wave's 512 bytes of normals of step tz into slot tz & 3
    const unsigned la = (unsigned)__builtin_amdgcn_readfirstlane((int)(zring_lds + (unsigned)(tz & 3) * 512u));
    const int soff = __builtin_amdgcn_readfirstlane((tz > 0 ? tz : 0) * 512);
    if (lane < 32)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(la), "v"(zvoff), "s"(rz), "s"(soff) : "memory");
  };
  double* thout = a.theta ? a.theta + (size_t)n0 * (T + 1) * d : nullptr;
  const __amdgpu_buffer_rsrc_t rth = mk_rsrc(thout, thout ? (size_t)nser * (T + 1) * d * 8 : 0);
  const int offth = (!dead && vc && thout) ? (int)((size_t)j * (T + 1) * d * 8) + c * 8 : OOB;
  const double* y = (a.stats && a.y) ? a.y + (size_t)n * T : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * d : nullptr;
  const double Fc = vc ? a.F[c] : 0.0;
  const unsigned char* need = tb.need;

  double Jc[16], Lr[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { Jc[k] = 0.0; Lr[k] = 0.0; }
  auto read8 = [&](unsigned base, double (&dst)[16]) {
    d2 q[8];
    q[0] = lds_read128<0>(base); q[1] = lds_read128<16>(base); q[2] = lds_read128<32>(base); q[3] = lds_read128<48>(base);
    q[4] = lds_read128<64>(base); q[5] = lds_read128<80>(base); q[6] = lds_read128<96>(base); q[7] = lds_read128<112>(base);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])::"memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) { dst[2 * k] = q[k][0]; dst[2 * k + 1] = q[k][1]; }   // (a lane beyond d holds component 0's: finite, and what it computes is never used)
  };
  auto read_row = [&](unsigned slot, bool withJ) {
    const unsigned base = slot + (vc ? c : 0) * (SF_ENT * 8);
    if (withJ) read8(base, Jc);
    read8(base + 128, Lr);
  };
  // theta = h + L z over the pivots, as chol_draw
  auto draw = [&](double hcol, const double* zk) {
    double th = hcol;
#pragma unroll
    for (int k = 0; k < 15; ++k) th = fma(Lr[k], zk[16 * j + k], th);   // L[c][k] = 0 for k > c and for k >= d: those terms add nothing
    return vc ? th : 0.0;
  };

  // requests: the means of steps T .. T - SF_AHEAD + 1 (the oldest), then the table rows T and T - 1
  for (int k = 0; k < SF_AHEAD; ++k) { const int tk = T - k > 0 ? T - k : 0; dma_means(mring_lds + ((T - k) & (SF_AHEAD - 1)) * 512, tk * mstep); }
  if constexpr (ZD) for (int k = 0; k < 4; ++k) dma_z(T - k);   // the normals of steps T .. T - 3
  dma_row<NR>(rtab, ring_lds + (T & 1) * SF_SLOT, T * (SF_ROW * 8), lane, n16);
  dma_row<NR>(rtab, ring_lds + ((T - 1) & 1) * SF_SLOT, (T > 0 ? T - 1 : 0) * (SF_ROW * 8), lane, n16);
  // need[t], need[t - 1], need[t - 2] as the loop goes down: bit (s & 63) of the mask of s's block of 64 steps
  unsigned long long nmask = 0;
  auto need_of = [&](int s_) -> bool {
    if (s_ < 0) return false;
    if (s_ == T || (s_ & 63) == 63) { const int b = (s_ & ~63) + lane; nmask = __ballot(b <= T && need[b] != 0); }
    return ((nmask >> (s_ & 63)) & 1ull) != 0;
  };
  bool nd0 = need_of(T), nd1 = need_of(T - 1), nd2 = need_of(T - 2);
  (void)nd0;
  double thc;
  {   // theta_T = m_T + chol(C_T) z_T
    if constexpr (!ZD) vZ[(T & 3) * 64 + lane] = vc ? zin[(size_t)T * d + c] : 0.0;
    vm_wait<0>();
    wave_sync();
    const unsigned slot = ring_lds + (T & 1) * SF_SLOT, mslot = mring_lds + (T & (SF_AHEAD - 1)) * 512;
    double mr = lds_read64<0>(mslot + lane * 8);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr)::"memory");
    read_row(slot, false);
    thc = draw(vc ? mr : 0.0, vZ + (T & 3) * 64);
    wave_sync();
    dma_means(mslot, (T > SF_AHEAD ? T - SF_AHEAD : 0) * mstep);
    if (nd2) dma_row<NR>(rtab, slot, (T - 2) * (SF_ROW * 8), lane, n16);
    if constexpr (ZD) dma_z(T - 4);
    bst(rth, offth, T * d * 8, thc);
  }
  double ssd = 0.0, ssy = 0.0;
  int nob = 0;
  double yk[2] = {0.0, 0.0};             // the observations of this row's series, 32 steps at a time
  for (int t = T - 1; t >= 0; --t) {
    nd0 = nd1; nd1 = nd2; nd2 = need_of(t - 2);
    if constexpr (!ZD) { vZ[(t & 3) * 64 + lane] = vc ? zin[(size_t)t * d + c] : 0.0; wave_sync(); }
    if (y) {   // observation residual of theta_{t+1} (Gibbs.scala:29-39), as k_sampler_sp16
      if (t == T - 1 || (t & 31) == 31) {
        const int base = t & ~31;
#pragma unroll
        for (int k = 0; k < 2; ++k) yk[k] = (base + 16 * k + c < T) ? y[base + 16 * k + c] : 0.0;
        asm volatile("" ::"v"(yk[0]), "v"(yk[1]));
      }
      const double yv = row_pick(((t >> 4) & 1) ? yk[1] : yk[0], lane, t & 15);
      double part = vc ? mul_rn(Fc, thc) : 0.0;
      for (int o_ = 8; o_ > 0; o_ >>= 1) part += __shfl_xor(part, o_);
      if (c == 0 && yv == yv) { ssy += (yv - part) * (yv - part); nob += 1; }
    }
    // A step issues, in this order: the request for the means (SF_AHEAD steps ahead), for a table row (two steps ahead, when it
    // exists), for the normals (ZD: four steps ahead, once the draw has read their slot), the store of the draw.  Operations younger
    // than the request for row t: that step's normals and store, then the step above this one: means, row t - 1 (when it exists),
    // (normals,) store.  Without a row for this step the normals of step t are what is waited for (the means are older): four steps
    // up, behind them that step's store and three steps of at least three operations -- or, with injected normals, the means:
    // 2 SF_AHEAD - 1 operations ago at least.
    if constexpr (ZD) { if (nd0) { if (nd1) vm_wait<5 + NR>(); else vm_wait<5>(); } else vm_wait<10>(); }
    else { if (nd0) { if (nd1) vm_wait<3 + NR>(); else vm_wait<3>(); } else vm_wait<2 * SF_AHEAD - 1>(); }
    const unsigned slot = ring_lds + (t & 1) * SF_SLOT, mslot = mring_lds + (t & (SF_AHEAD - 1)) * 512;
    double mr = lds_read64<0>(mslot + lane * 8);
    double mg[K];
#pragma unroll
    for (int s = 0; s < K; ++s) mg[s] = lds_read64<0>(mslot + idx[s]);
    if constexpr (K == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(mg[0])::"memory");
    else if constexpr (K == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(mg[0]), "+v"(mg[1])::"memory");
    else if constexpr (K == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(mg[0]), "+v"(mg[1]), "+v"(mg[2])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(mg[0]), "+v"(mg[1]), "+v"(mg[2]), "+v"(mg[3])::"memory");
    if (nd0) read_row(slot, true);
    dma_means(mslot, (t > SF_AHEAD ? t - SF_AHEAD : 0) * mstep);
    if (nd2) dma_row<NR>(rtab, slot, (t - 2) * (SF_ROW * 8), lane, n16);
    const double mc = vc ? mr : 0.0;
    double a1 = 0.0;
#pragma unroll
    for (int s = 0; s < K; ++s) a1 = fma(vc ? mg[s] : 0.0, val[s], a1);
    vU[lane] = vc ? thc - a1 : 0.0;
    wave_sync();
    double ch[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) ch[g] = fma(Jc[4 * r + g], vU[16 * j + 4 * r + g], ch[g]);
    const double hcol = mc + ((ch[0] + ch[1]) + (ch[2] + ch[3]));
    const double th = draw(hcol, vZ + (t & 3) * 64);
    if (a.stats) {   // system residual theta_{t+1} - G theta_t
      vT[lane] = th;
      wave_sync();
      double gth = 0.0;
#pragma unroll
      for (int s = 0; s < K; ++s) gth = fma(vT[idx[s] >> 3], val[s], gth);
      const double df = vc ? thc - gth : 0.0;
      ssd += mul_rn(df, df);
    }
    wave_sync();
    thc = th;
    if constexpr (ZD) dma_z(t - 4);                   // (the draw has read this slot)
    bst(rth, offth, t * d * 8, th);
  }
  vm_wait<0>();   // no DMA may still be writing this block's LDS when the wave ends
  (void)vT_lds;
  const int stz = tb.status[0];
  const unsigned long long badl = __ballot(!dead && vc && !isfinite(thc));
  if (!dead) {
    if (a.stats) {
      const int L = stats_len(d, 1, a.flags);
      double* so = a.stats + (size_t)n * L;
      if (c == 0) { so[0] = ssy; so[1] = (double)nob; so[L - 1] = (double)T; }
      if (vc) so[2 + c] = ssd;
    }
    if (c == 0) {
      const int sj = stz | (((badl >> (16 * j)) & 0xffffull) ? DLM_ST_NONFINITE : 0);
      if (a.status && sj) atomicOr(&a.status[n], sj);
    }
  }
  const unsigned long long live = __ballot(!dead && c == 0);
  if (a.counters && lane == 0) atomicAdd(&a.counters[2], (unsigned long long)__builtin_popcountll(live));
}

// route[n] = 1: the series has a missing observation
__global__ __launch_bounds__(256) void k_mark_gaps(const double* __restrict__ y, int N, int T, unsigned char* __restrict__ route, int* __restrict__ count = nullptr) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const bool live = n < N;   // (no early return: the workgroup meets at the barriers below)
  const double* yn = y + (size_t)(live ? n : 0) * T;
  bool gap = false;
  if (live) for (int t = lane; t < T; t += 64) { const double v = yn[t]; gap |= !(v == v); }
  const bool any = __ballot(gap) != 0ull;
  if (live && lane == 0) route[n] = any ? 1 : 0;
  if (count) {   // one atomic per workgroup of four series
    __shared__ int found;
    if (threadIdx.x == 0) found = 0;
    __syncthreads();
    if (lane == 0 && any) atomicAdd(&found, 1);
    __syncthreads();
    if (threadIdx.x == 0 && found) atomicAdd(count, found);
  }
}
// More than half of the series have a gap: the tables would serve too few of them to be worth one wave's 2-3 ms beside the forward pass (and
// the per-series kernel's launch for the rest): every series is routed to the per-series kernel and the table kernels return at once.
__global__ __launch_bounds__(256) void k_rts_decide(unsigned char* __restrict__ route, int N, const int* __restrict__ gaps, int* __restrict__ skip) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const bool sk = 2 * (long long)gaps[0] > (long long)N;
  if (n == 0) skip[0] = sk ? 1 : 0;
  if (sk && n < N) route[n] = 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// The reference's RTS smoother on the same register tiles: Smoothing.smoothStep / backwardsSmoother (Smoothing.scala:31-64),
// from filter records alone (no innovations from a forward pass needed: serves dlm_smooth_batch and the literal mode).
//   a+ = G m, R+ = G C G^T + W dt (dt == 0: a+ = m, R+ = C);  J = C G^T R+^-1 (always the table entry g(dt), Smoothing.scala:41)
//   s_t = m_t + J (s_{t+1} - a+)
//   S_t = C_t - J (R+ - S_{t+1}) J^T          textbook
//   S_t = C_t - J (R+ - S_{t+1}) J            DLM_OPT_SMOOTHER_COMPAT_Q1: Smoothing.scala:44 as written (no transpose)
// With mm(A, B) = A^T B on the tiles:  J^T = mm(Rinv, G C),  Y = mm(X, J^T) = (J X)^T,  J X J^T = mm(J^T, Y),  J X J = mm(Y, J)
// with J = mm(G C, Rinv).  Rinv is the warm-started Newton-Schulz inverse of the sampler above; once the filtered
// covariance has stopped moving, J and R+ are those of the step before and a step is two products.
// ---------------------------------------------------------------------------------------------------------------------
// EXP (shared factors, below): the series is the one series of zeros; the J_t^T of every full step goes to RtsTabs::jrows, the smoothed
// records -- a.smooth = RtsTabs::srec -- are the S_t table.
template <int K, class Tab, bool EXP = false>
__global__ __launch_bounds__(64) void k_smoother_rts16(KArgs a, const Tab* __restrict__ sp, RtsTabs tb) {
  if (!EXP && a.route && (a.route[blockIdx.x] != 0) != (a.route_take != 0)) return;   // shared-factor call: only the series routed here
  if constexpr (EXP) { if (tb.skip && *tb.skip) return; }   // (the call found that it does not want the tables)
  if constexpr (EXP) __builtin_amdgcn_s_setprio(3);   // the table's one wave runs beside the kernel that filters the batch and is what the mean kernel waits for
  __shared__ __attribute__((aligned(16))) double lds[2 * IMG + 4 * 16];
  double* img = lds;       double* inv = lds + IMG;
  double* mv = inv + IMG;  double* sv = mv + 16;   double* uv = sv + 16;
  const int n = blockIdx.x, lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  const int d = a.d, T = a.T, dd = d * d, rec = d + dd, recb = rec * 8;
  const bool vc = c < d;
  const bool compat = (a.flags & DLM_OPT_SMOOTHER_COMPAT_Q1) != 0;
  for (int i = lane; i < 4 * 16; i += 64) mv[i] = 0.0;

  int ridx[K];                   // nonzeros of row c of G
  double rval[K];
  int gcur = -1;
  auto load_tables = [&](int gi) {
#pragma unroll
    for (int s = 0; s < K; ++s) { ridx[s] = sp[2 * gi].idx[c][s]; rval[s] = vc ? sp[2 * gi].val[c][s] : 0.0; }
    gcur = gi;
  };
  const double* W0 = a.W + (size_t)n * a.w_stride;
  d4 Wt;
  bool va[4];
  int offC[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    va[r] = i < d && vc;
    offC[r] = va[r] ? (d + i + c * d) * 8 : OOB;      // element (i, c) of a record's covariance, column-major
    Wt[r] = va[r] ? W0[i + c * d] : 0.0;
  }
  const int offM = (vc && g == 0) ? c * 8 : OOB;
  const int rstr = (EXP && tb.crec) ? tb.crec_stride : recb;   // bytes between the input records (the table run may read the rows of a covariance table)
  const __amdgpu_buffer_rsrc_t rin = (EXP && tb.crec) ? mk_rsrc(tb.crec, (size_t)(T + 1) * rstr) : mk_rsrc(a.filt_in + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  const __amdgpu_buffer_rsrc_t rout = mk_rsrc(a.smooth + (size_t)n * (T + 1) * rec, (size_t)(T + 1) * recb);
  int st = 0;
  d4 S, Rinv = {0.0, 0.0, 0.0, 0.0};
  bool warm = false;
  {   // init = last filter state (Smoothing.scala:59-61)
#pragma unroll
    for (int r = 0; r < 4; ++r) { S[r] = bld(rin, offC[r], T * rstr); bst(rout, offC[r], T * recb, S[r]); }
    const double mc = bld(rin, vc ? c * 8 : OOB, T * rstr);
    bst(rout, offM, T * recb, mc);
    if (g == 0) sv[c] = mc;
    wave_sync();
    if constexpr (EXP) { if (lane == 0) tb.need[T] = 0; }
  }
  d4 Cp = {0.0, 0.0, 0.0, 0.0}, JTs = Cp, Js = Cp, Rs = Cp;   // steady-state reuse (see k_sampler_sp16)
  bool have = false;
  int gprev = -1;
  double dtprev = 0.0, cmaxp = 0.0;
  d4 nC;
  double nm;
  {
    const int tp = T > 0 ? T - 1 : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) nC[r] = bld(rin, offC[r], tp * rstr);
    nm = bld(rin, vc ? c * 8 : OOB, tp * rstr);
  }
  double scol = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    const double dt = a.dt ? a.dt[t] : 1.0;
    const int gi = a.g_index ? a.g_index[t] : 0;
    if (gi != gcur) load_tables(gi);
    const d4 C = nC;
    const double mc = nm;
    {
      const int tp = t > 0 ? t - 1 : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) nC[r] = bld(rin, offC[r], tp * rstr);
      nm = bld(rin, vc ? c * 8 : OOB, tp * rstr);
    }
    if (a.w_tstride) {
#pragma unroll
      for (int r = 0; r < 4; ++r) Wt[r] = va[r] ? (W0 + (size_t)t * a.w_tstride)[4 * r + g + c * d] : 0.0;
    }
    if (!EXP && g == 0) mv[c] = mc;   // (EXP: nobody reads the means of the series of zeros -- the table run leaves the mean recursion out of its chain)
    bool reuse = false;
    if (have && gi == gprev && dt == dtprev && !a.w_tstride) {
      bool moved = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) moved |= !(fabs(C[r] - Cp[r]) <= DLM_SETTLE_TOL * cmaxp);   // this step's record against the covariance of the last full step: a bound on the distance itself
      reuse = __ballot(moved) == 0ull;
    }
    d4 JT, J, R;
    double a1 = mc;
    if (reuse) {
      if constexpr (!EXP) {
        wave_sync();
        if (dt != 0.0) {
          double s_ = 0.0;
#pragma unroll
          for (int s = 0; s < K; ++s) s_ = fma(mv[ridx[s]], rval[s], s_);
          a1 = s_;
        }
      }
      JT = JTs; J = Js; R = Rs;
      if constexpr (EXP) { if (lane == 0) tb.need[t] = 0; }
    } else {
      to_img(C, img, g, c);
      wave_sync();
      d4 T1;                                           // C G^T (pass 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double* row = img + (4 * r + g) * IL;
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < K; ++s) s_ = fma(row[ridx[s]], rval[s], s_);
        T1[r] = s_;
      }
      if (!EXP && dt != 0.0) {
        double s_ = 0.0;
#pragma unroll
        for (int s = 0; s < K; ++s) s_ = fma(mv[ridx[s]], rval[s], s_);
        a1 = s_;
      }
      wave_sync();
      const d4 GC = transposed(T1, img, g, c);         // (C G^T)^T = G C
      if (dt != 0.0) {
        to_img(T1, img, g, c);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * r + g;
          double s_ = Wt[r] * dt;
#pragma unroll
          for (int s = 0; s < K; ++s) s_ = fma(img[ridx[s] * IL + i], rval[s], s_);
          R[r] = s_;
        }
        wave_sync();
        const d4 Rt = transposed(R, img, g, c);
#pragma unroll
        for (int r = 0; r < 4; ++r) R[r] = 0.5 * (R[r] + Rt[r]);
      } else R = C;
      {
        bool done = false;
        if (warm) {
          for (int it = 0; it < 6 && !done; ++it) {
            d4 E = mm(R, Rinv, d);
            bool big = false, far = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = 4 * r + g;
              const double e = ((i == c) ? 1.0 : 0.0) - E[r];
              E[r] = e;
              if (i < d && vc) { big |= !(fabs(e) <= 2e-10); far |= !(fabs(e) * d < 0.5); }
            }
            const bool any_big = __ballot(big) != 0ull, any_far = __ballot(far) != 0ull;
            if (any_far && any_big) break;
            if (!any_big) done = true;
            const d4 D = mm(Rinv, E, d);
#pragma unroll
            for (int r = 0; r < 4; ++r) Rinv[r] += D[r];
            const d4 Xt = transposed(Rinv, img, g, c);
#pragma unroll
            for (int r = 0; r < 4; ++r) Rinv[r] = 0.5 * (Rinv[r] + Xt[r]);
          }
        }
        if (!done && direct_inverse(R, Rinv, d, img, inv, lane, g, c)) st |= DLM_ST_NOT_PD;
        warm = true;
      }
      JT = mm(Rinv, GC, d);                             // J^T = R+^-1 G C
      J = mm(GC, Rinv, d);                              // J   = C G^T R+^-1
      if constexpr (EXP) {                              // column c of J^T, as k_mean_rts16's lane c takes it
        double* row = tb.jrows + (size_t)t * RJ_ROW + c * RJ_ENT;
        if (vc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) row[4 * r + g] = JT[r];
        }
        if (lane == 0) tb.need[t] = 1;
      }
      Cp = C; JTs = JT; Js = J; Rs = R; have = true; gprev = gi; dtprev = dt;
      {
        double mx = fmax(fmax(fabs(C[0]), fabs(C[1])), fmax(fabs(C[2]), fabs(C[3])));
        for (int o_ = 32; o_ > 0; o_ >>= 1) mx = fmax(mx, __shfl_xor(mx, o_));
        cmaxp = mx;
      }
    }
    if constexpr (!EXP) {
      if (g == 0) uv[c] = vc ? sv[c] - a1 : 0.0;
      wave_sync();
      scol = mc + matTvec(JT, uv, g);                   // s = m + J (s+ - a+)
    }
    d4 X;
#pragma unroll
    for (int r = 0; r < 4; ++r) X[r] = R[r] - S[r];     // R+ - S+
    const d4 Y = mm(X, JT, d);                          // X^T J^T = (J X)^T
    const d4 JXJ = compat ? mm(Y, J, d) : mm(JT, Y, d); // J X J (Smoothing.scala:44)  |  J X J^T
#pragma unroll
    for (int r = 0; r < 4; ++r) { S[r] = va[r] ? C[r] - JXJ[r] : 0.0; bst(rout, offC[r], t * recb, S[r]); }
    bst(rout, offM, t * recb, scol);
    if constexpr (!EXP) {
      wave_sync();
      if (g == 0) sv[c] = vc ? scol : 0.0;
      wave_sync();
    }
  }
  bool bad = vc && g == 0 && !isfinite(scol);
#pragma unroll
  for (int r = 0; r < 4; ++r) bad |= va[r] && !isfinite(S[r]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------------------------------------
// Shared factors of the RTS smoother.  J_t and S_t (Smoothing.scala:38-47) depend on V, W, C0 and the time grid, not on the data:
// where the batch shares them and a series has no missing observation, k_smoother_rts16 above computes the same J_t, S_t -- d^3
// work and a chain of eight matrix products per step -- in every series.  Here it runs ONCE, with EXP set, on the filter records of
// a series of zeros (made beside the batch's forward pass), and every series runs k_mean_rts16: four series per wave (lane 16 j + c =
// component c of series j), per step
//     a+ = G m_t,   s_t = m_t + J_t (s_{t+1} - a+),   record t = [ s_t | S_t of the table ]
// in the operations of k_smoother_rts16, one for one (the gather of G in table order, matTvec's four chains and their (0 + 1) +
// (2 + 3) sum): the records are bit for bit those of the per-series kernel.  Column c of J^T lives in registers and is replaced
// where the table has a row for the step (RtsTabs::need: a full step of the table run; the others reuse J as the per-series kernel
// does); J rows and S_t records travel two steps ahead by LDS DMA from L2, the four series' filtered means SF_AHEAD steps ahead
// from the heads of their filter records; a record leaves as 16-byte pieces: the table's S_t with s_t patched into the first d doubles.
//   row t of RtsTabs::jrows (RJ_ROW doubles): per component c < d  [ J_t^T[0..15][c] | 2 pad ]
// NP: store instructions per step (64-lane groups of 16-byte pieces covering four records, rounded up to 2, 4, 6, 8);
// NRS / NRJ: DMA instructions per S_t record / per J row -- exact, the waits below count them.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef DLM_RTS_STORE_AUX
#define DLM_RTS_STORE_AUX 2   // nt: the records are written once and read by nobody in this call; non-temporal stores leave the table rows and the means in L2
#endif
__device__ __forceinline__ void bst128(__amdgpu_buffer_rsrc_t r, int voff, int soff, d2 x) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = {(unsigned)__double2loint(x[0]), (unsigned)__double2hiint(x[0]), (unsigned)__double2loint(x[1]), (unsigned)__double2hiint(x[1])};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, DLM_RTS_STORE_AUX);
}
__device__ __forceinline__ d2 lds_read128v(unsigned addr) {
  d2 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
template <int NP>
__device__ __forceinline__ void lds_wait_np(d2 (&pc)[NP]) {
  static_assert(NP == 2 || NP == 4 || NP == 6 || NP == 8, "pieces");
  if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1])::"memory");
  else if constexpr (NP == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3])::"memory");
  else if constexpr (NP == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]), "+v"(pc[4]), "+v"(pc[5])::"memory");
  else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc[0]), "+v"(pc[1]), "+v"(pc[2]), "+v"(pc[3]), "+v"(pc[4]), "+v"(pc[5]), "+v"(pc[6]), "+v"(pc[7])::"memory");
}

template <int K, int NP, int NRS, int NRJ>
__global__ __launch_bounds__(64, 3) void k_mean_rts16(KArgs a, const SparseT* __restrict__ sp, RtsTabs tb) {
  __shared__ __attribute__((aligned(16))) double lds[2 * 64 + 2 * (RS_SLOT / 8) + 2 * RJ_ROW + SF_AHEAD * 64];
  const int lane = threadIdx.x, j = lane >> 4, c = lane & 15;
  const int n0 = 4 * blockIdx.x;
  if (n0 >= a.N) return;
  const int nser = a.N - n0 < 4 ? a.N - n0 : 4;
  const bool have = j < nser;
  const int n = have ? n0 + j : n0;                 // (a row without a series shadows the first: loads stay in bounds, nothing is stored)
  const bool dead = !have || a.route[n] != 0;       // a series with a missing observation: k_smoother_rts16 serves it
  unsigned deadmask = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) deadmask |= (__builtin_amdgcn_readlane((int)dead, 16 * q) & 1) << q;
  if (a.counters && lane == 0) {   // the wave's series that are served here, and those that are left to the per-series kernel
    const int routed = __builtin_popcount(deadmask & ((1u << nser) - 1u));
    if (nser - routed) atomicAdd(&a.counters[2], (unsigned long long)(nser - routed));
    if (routed) atomicAdd(&a.counters[3], (unsigned long long)routed);
  }
  if (deadmask == 0xfu) return;
  double* vU = lds;            // s_{t+1} - a+   [4][16]
  double* vH = vU + 64;        // the first 16 doubles of each series' smoothed record
  char* ring = (char*)(vH + 64);                    // two S_t slots, two J slots, then the slots of the means
  const int d = a.d, T = a.T, rec = d + d * d, recb = rec * 8;
  const bool vc = c < d;
  int idx[K];
  double val[K];
#pragma unroll
  for (int s = 0; s < K; ++s) { idx[s] = (16 * j + sp[0].idx[c][s]) * 8; val[s] = vc ? sp[0].val[c][s] : 0.0; }
  const size_t sbytes = (size_t)(T + 1) * recb;
  const i4 rmean = rsrc_words((const char*)a.filt_in + (size_t)n0 * sbytes, (unsigned)((size_t)nser * sbytes));
  const i4 rjt = rsrc_words(tb.jrows, (unsigned)((size_t)(T + 1) * RJ_ROW * 8));
  const i4 rst = rsrc_words(tb.srec, (unsigned)sbytes);
  const __amdgpu_buffer_rsrc_t rout = mk_rsrc((char*)a.smooth + (size_t)n0 * sbytes, (size_t)nser * sbytes);
  const int mvoff = (lane < 32 && (lane >> 3) < nser) ? (int)((size_t)(lane >> 3) * sbytes) + (lane & 7) * 16 : OOB;   // 16 doubles from the head of each record
  const unsigned sring_lds = lds_addr_of(ring), jring_lds = sring_lds + 2 * RS_SLOT, mring_lds = jring_lds + 2 * (RJ_ROW * 8);
  const unsigned vH_lds = lds_addr_of(vH);
  const int npc = rec / 2;                           // 16-byte pieces of a record
  const int nj16 = d * (RJ_ENT / 2);                 // ... of a J row that travel
  auto dma_means = [&](unsigned lds_addr, int soff) {
    lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    if (lane < 32)   // 4 x 8 pieces; the other lanes' LDS destinations lie beyond the slot
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(mvoff), "s"(rmean), "s"(soff) : "memory");
  };
  unsigned psrc[NP];
  int pdst[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int q = 64 * k + lane, sj = q / npc, pp = q - sj * npc;
    pdst[k] = (sj < nser && !((deadmask >> (sj & 3)) & 1u)) ? (int)((size_t)sj * sbytes) + pp * 16 : OOB;
    psrc[k] = pp < 8 ? (0x80000000u | (unsigned)((16 * (sj & 3) + 2 * pp) * 8)) : (unsigned)(pp * 16);
  }
  const unsigned ptd = (unsigned)(c * 8);            // double c of the S_t record: what the head of a record holds beyond the mean
  const unsigned char* need = tb.need;

  double Jc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Jc[k] = 0.0;
  auto read_J = [&](unsigned slot) {
    const unsigned base = slot + (vc ? c : 0) * (RJ_ENT * 8);
    d2 q[8];
    q[0] = lds_read128<0>(base); q[1] = lds_read128<16>(base); q[2] = lds_read128<32>(base); q[3] = lds_read128<48>(base);
    q[4] = lds_read128<64>(base); q[5] = lds_read128<80>(base); q[6] = lds_read128<96>(base); q[7] = lds_read128<112>(base);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])::"memory");
#pragma unroll
    for (int k = 0; k < 8; ++k) { Jc[2 * k] = q[k][0]; Jc[2 * k + 1] = q[k][1]; }   // (a lane beyond d holds component 0's: finite, and what it computes is never used)
  };
  // the record of step t: the head from vH, the rest from the S_t slot; then the slots of step t are free for step t - 2
  auto emit = [&](int t, unsigned sslot, unsigned jslot, unsigned mslot, bool nd2) {
    d2 pc[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) pc[k] = lds_read128v((psrc[k] & 0x80000000u) ? vH_lds + (psrc[k] & 0x7fffffffu) : sslot + psrc[k]);
    lds_wait_np<NP>(pc);
    dma_means(mslot, (t > SF_AHEAD ? t - SF_AHEAD : 0) * recb);
    if (nd2) dma_row<NRJ>(rjt, jslot, (t - 2) * (RJ_ROW * 8), lane, nj16);
    dma_row<NRS>(rst, sslot, (t > 1 ? t - 2 : 0) * recb, lane, npc);
    const int so = t * recb;
#pragma unroll
    for (int k = 0; k < NP; ++k) bst128(rout, pdst[k], so, pc[k]);
  };

  // requests: the means of steps T .. T - SF_AHEAD + 1, the S_t records T and T - 1, the J row of step T - 1 (its first step is a full one)
  for (int k = 0; k < SF_AHEAD; ++k) { const int tk = T - k > 0 ? T - k : 0; dma_means(mring_lds + ((T - k) & (SF_AHEAD - 1)) * 512, tk * recb); }
  dma_row<NRS>(rst, sring_lds + (T & 1) * RS_SLOT, T * recb, lane, npc);
  dma_row<NRS>(rst, sring_lds + ((T - 1) & 1) * RS_SLOT, (T - 1) * recb, lane, npc);
  // need[t], need[t - 1], need[t - 2] as the loop goes down: bit (s & 63) of the mask of s's block of 64 steps
  unsigned long long nmask = 0;
  auto need_of = [&](int s_) -> bool {
    if (s_ < 0) return false;
    if (s_ == T || (s_ & 63) == 63) { const int b = (s_ & ~63) + lane; nmask = __ballot(b <= T && need[b] != 0); }
    return ((nmask >> (s_ & 63)) & 1ull) != 0;
  };
  bool nd0 = need_of(T), nd1 = need_of(T - 1), nd2 = need_of(T - 2);
  (void)nd0;
  if (nd1) dma_row<NRJ>(rjt, jring_lds + ((T - 1) & 1) * (RJ_ROW * 8), (T - 1) * (RJ_ROW * 8), lane, nj16);
  double sc;
  {   // s_T = m_T, S_T = C_T (Smoothing.scala:59-61): the table's record T
    vm_wait<0>();
    wave_sync();
    const unsigned sslot = sring_lds + (T & 1) * RS_SLOT, mslot = mring_lds + (T & (SF_AHEAD - 1)) * 512;
    double mr = lds_read64<0>(mslot + lane * 8), td = lds_read64<0>(sslot + ptd);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(td)::"memory");
    sc = vc ? mr : 0.0;
    vH[lane] = vc ? mr : td;
    wave_sync();
    emit(T, sslot, jring_lds + (T & 1) * (RJ_ROW * 8), mslot, nd2);
  }
  for (int t = T - 1; t >= 0; --t) {
    nd0 = nd1; nd1 = nd2; nd2 = need_of(t - 2);
    // A step issues, in this order: the request for the means (SF_AHEAD steps ahead), for a J row (two steps ahead, when it exists), for
    // an S_t record (two steps ahead), its NP stores.  Operations younger than the request for record t (issued by step t + 2): that step's
    // stores, then step t + 1: means, J row t - 1 (when it exists), record t - 1, stores.  The J row of step t is older than its record, the
    // means older still.  (Step T - 1: everything it reads was waited for before step T.)
    if (nd1) vm_wait<2 * NP + 1 + NRS + NRJ>(); else vm_wait<2 * NP + 1 + NRS>();
    const unsigned sslot = sring_lds + (t & 1) * RS_SLOT, jslot = jring_lds + (t & 1) * (RJ_ROW * 8), mslot = mring_lds + (t & (SF_AHEAD - 1)) * 512;
    double mr = lds_read64<0>(mslot + lane * 8), td = lds_read64<0>(sslot + ptd);
    double mg[K];
#pragma unroll
    for (int s = 0; s < K; ++s) mg[s] = lds_read64<0>(mslot + idx[s]);
    if constexpr (K == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(td), "+v"(mg[0])::"memory");
    else if constexpr (K == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(td), "+v"(mg[0]), "+v"(mg[1])::"memory");
    else if constexpr (K == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(td), "+v"(mg[0]), "+v"(mg[1]), "+v"(mg[2])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mr), "+v"(td), "+v"(mg[0]), "+v"(mg[1]), "+v"(mg[2]), "+v"(mg[3])::"memory");
    if (nd0) read_J(jslot);
    const double mc = vc ? mr : 0.0;
    double a1 = 0.0;
#pragma unroll
    for (int s = 0; s < K; ++s) a1 = fma(vc ? mg[s] : 0.0, val[s], a1);
    vU[lane] = vc ? sc - a1 : 0.0;
    wave_sync();
    double ch[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) ch[g] = fma(Jc[4 * r + g], vU[16 * j + 4 * r + g], ch[g]);
    const double scol = mc + ((ch[0] + ch[1]) + (ch[2] + ch[3]));
    vH[lane] = vc ? scol : td;
    wave_sync();
    emit(t, sslot, jslot, mslot, nd2);
    sc = vc ? scol : 0.0;
  }
  vm_wait<0>();   // no DMA may still be writing this block's LDS when the wave ends
  const int stz = tb.status[0];
  const unsigned long long badl = __ballot(!dead && vc && !isfinite(sc));
  if (!dead && c == 0) {
    const int sj = stz | (((badl >> (16 * j)) & 0xffffull) ? DLM_ST_NONFINITE : 0);
    if (a.status && sj) atomicOr(&a.status[n], sj);
  }
}

}  // namespace s16

template <class Tab>
static hipError_t launch_sampler_t(const KArgs& a, int K, const Tab* tabs_dev, hipStream_t s) {
  switch (K) {
    case 1: hipLaunchKernelGGL((s16::k_sampler_sp16<1, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, SampTabs{}); break;
    case 2: hipLaunchKernelGGL((s16::k_sampler_sp16<2, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, SampTabs{}); break;
    case 3: hipLaunchKernelGGL((s16::k_sampler_sp16<3, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, SampTabs{}); break;
    case 4: hipLaunchKernelGGL((s16::k_sampler_sp16<4, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, SampTabs{}); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_sparse16_sampler(const KArgs& a, int K, const SparseT* tabs_dev, hipStream_t s) { return launch_sampler_t(a, K, tabs_dev, s); }

template <class Tab>
static hipError_t launch_rts_t(const KArgs& a, int K, const Tab* tabs_dev, hipStream_t s) {
  switch (K) {
    case 1: hipLaunchKernelGGL((s16::k_smoother_rts16<1, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, RtsTabs{}); break;
    case 2: hipLaunchKernelGGL((s16::k_smoother_rts16<2, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, RtsTabs{}); break;
    case 3: hipLaunchKernelGGL((s16::k_smoother_rts16<3, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, RtsTabs{}); break;
    case 4: hipLaunchKernelGGL((s16::k_smoother_rts16<4, Tab>), dim3(a.N), dim3(64), 0, s, a, tabs_dev, RtsTabs{}); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
// ---- shared factors ------------------------------------------------------------------------------------------------------------
#ifndef DLM_SAMPLER_SHARED_MIN
#define DLM_SAMPLER_SHARED_MIN 1   // (measured, tools/sweep_c3.sh: the table pays at every batch size -- its stretches are made side by side, a series' own wave makes them one after the other)
#endif
bool sampler_shared_model_ok(const KArgs& a) {
  const size_t rec = (size_t)a.d + (size_t)a.d * a.d;
  return a.d <= 15 && a.p == 1 && !a.g_index && !a.dt && !a.f_stride && !a.v_tstride && !a.w_tstride && !a.v_stride && !a.w_stride && !a.c0_stride &&
         !a.packed && a.T >= 1 && a.T <= 400000 && 4 * ((size_t)a.T + 1) * rec * 8 < ((size_t)1 << 31);   // four series' records under one buffer resource
}
bool sampler_shared_eligible(const KArgs& a) {
  return sampler_shared_model_ok(a) && !a.cond && (!a.stats || a.y) &&
         !(a.flags & (DLM_OPT_STATS_OUTER | DLM_OPT_FORCE_GENERIC | DLM_OPT_NO_SAMPLER16 | DLM_OPT_SAMPLER_PER_SERIES)) &&
         (a.N >= DLM_SAMPLER_SHARED_MIN || (a.flags & DLM_OPT_NO_SMALL_BATCH));
}
static size_t up64(size_t x) { return (x + 63) & ~(size_t)63; }
size_t sampler_shared_ws_bytes(const KArgs& a) {
  const size_t n1 = (size_t)a.T + 1, rec = (size_t)a.d + (size_t)a.d * a.d;
  return up64(n1 * s16::SF_ROW * 8) + up64(n1 * rec * 8) + up64((size_t)(a.T > 16 ? a.T : 16) * 8) + up64(n1) + 64;
}
void sampler_shared_carve(void* ws, const KArgs& a, SampTabs& tb) {
  const size_t n1 = (size_t)a.T + 1, rec = (size_t)a.d + (size_t)a.d * a.d;
  char* p = (char*)ws;
  tb.rows = (double*)p;  p += up64(n1 * s16::SF_ROW * 8);
  tb.zrec = (double*)p;  p += up64(n1 * rec * 8);
  tb.zeros = (double*)p; p += up64((size_t)(a.T > 16 ? a.T : 16) * 8);
  tb.need = (unsigned char*)p; p += up64(n1);
  tb.status = (int*)p; tb.settle = tb.status + 1;
}
hipError_t launch_sampler_shared_tables(const KArgs& a, int K, const SparseT* tabs_dev, const SampTabs& tb, hipStream_t s) {
  hipError_t err = hipMemsetAsync(tb.zeros, 0, (size_t)(a.T > 16 ? a.T : 16) * 8, s);
  if (err != hipSuccess) return err;
  if ((err = hipMemsetAsync(tb.status, 0, sizeof(int), s)) != hipSuccess) return err;
  KArgs kf = a;   // the filter on a series of zeros: the covariances of every series without a missing observation, bit for bit
  kf.N = 1; kf.y = tb.zeros; kf.m0 = tb.zeros; kf.m0_stride = 0; kf.filt = tb.zrec; kf.status = tb.status; kf.stats = nullptr; kf.loglik = nullptr;
  kf.prior = nullptr; kf.fq = nullptr; kf.route = nullptr; kf.counters = nullptr; kf.theta = nullptr; kf.z = nullptr; kf.series_offset = 0;
  if ((err = launch_sparse16_filter(kf, K, tabs_dev, nullptr, nullptr, s)) != hipSuccess) return err;
  KArgs kp = kf;
  kp.y = nullptr; kp.filt_in = tb.zrec;
  const dim3 grid(s16::sf_stretches(a.T));   // one wave per stretch
  switch (K) {
    case 1: hipLaunchKernelGGL((s16::k_sampler_sp16<1, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    case 2: hipLaunchKernelGGL((s16::k_sampler_sp16<2, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    case 3: hipLaunchKernelGGL((s16::k_sampler_sp16<3, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    case 4: hipLaunchKernelGGL((s16::k_sampler_sp16<4, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
template <int K, bool ZD>
static hipError_t launch_mean_sampler(const KArgs& a, const SparseT* sp, const SampTabs& tb, hipStream_t s) {
  const dim3 grid((a.N + 3) / 4), blk(64);
  const int nr = (17 * a.d + 63) / 64;
  switch (nr) {
    case 1: hipLaunchKernelGGL((s16::k_mean_sampler_sp16<K, 1, ZD>), grid, blk, 0, s, a, sp, tb); break;
    case 2: hipLaunchKernelGGL((s16::k_mean_sampler_sp16<K, 2, ZD>), grid, blk, 0, s, a, sp, tb); break;
    case 3: hipLaunchKernelGGL((s16::k_mean_sampler_sp16<K, 3, ZD>), grid, blk, 0, s, a, sp, tb); break;
    default: hipLaunchKernelGGL((s16::k_mean_sampler_sp16<K, 4, ZD>), grid, blk, 0, s, a, sp, tb); break;
  }
  return hipGetLastError();
}
// The normals of the whole call in the draw kernel's layout, made while the batch is filtered: they are a third of the draw
// kernel's instructions otherwise, in its dependent chain.  One wave per group of four series and 16 steps; lane 16 j + c makes the
// pair (2 q, 2 q + 1), q = c >> 1, of series j at the steps te = t0 + 2 i + (c & 1): philox_normal2, i.e. philox_normal's values.
__global__ __launch_bounds__(256) void k_normals4(KArgs a, double* __restrict__ z4) {
  const int lane = threadIdx.x & 63, j = lane >> 4, c = lane & 15;
  const int nch = (a.T + 16) / 16;                           // chunks of 16 steps covering 0 .. T
  const long long w = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long g4 = w / nch;
  if (g4 >= (a.N + 3) / 4) return;
  const int t0 = (int)(w - g4 * nch) * 16, n = (int)g4 * 4 + j;
  const bool live = n < a.N && (c & ~1) < a.d;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  double* out = z4 + (size_t)g4 * (a.T + 1) * 64 + 16 * j + (c & ~1);
  for (int i = 0; i < 8; ++i) {
    const int te = t0 + 2 * i + (c & 1);
    if (te > a.T) break;
    double ze = 0.0, zo = 0.0;
    if (live) philox_normal2(a.seed, series, (unsigned)te, (unsigned)(c >> 1), ze, zo);
    if ((c | 1) >= a.d) zo = 0.0;
    double* o = out + (size_t)te * 64;
    o[0] = ze; o[1] = zo;
  }
}
size_t sampler_shared_normals_bytes(const KArgs& a) { return (size_t)((a.N + 3) / 4) * ((size_t)a.T + 1) * 512; }
hipError_t launch_sampler_shared_normals(const KArgs& a, double* z4, hipStream_t s) {
  const long long waves = (long long)((a.N + 3) / 4) * ((a.T + 16) / 16);
  hipLaunchKernelGGL(k_normals4, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a, z4);
  return hipGetLastError();
}
hipError_t launch_sampler_shared_tables_from(const KArgs& a, int K, const SparseT* tabs_dev, SampTabs tb, const double* crec, int stride, hipStream_t s) {
  hipError_t err = hipMemsetAsync(tb.status, 0, sizeof(int), s);
  if (err != hipSuccess) return err;
  KArgs kp = a;
  kp.N = 1; kp.y = nullptr; kp.filt_in = crec; kp.filt = nullptr; kp.status = tb.status; kp.stats = nullptr; kp.loglik = nullptr; kp.prior = nullptr; kp.fq = nullptr;
  kp.route = nullptr; kp.counters = nullptr; kp.theta = nullptr; kp.z = nullptr; kp.series_offset = 0; kp.m0_stride = 0;
  tb.zstride = stride;
  const dim3 grid(s16::sf_stretches(a.T));
  switch (K) {
    case 1: hipLaunchKernelGGL((s16::k_sampler_sp16<1, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    case 2: hipLaunchKernelGGL((s16::k_sampler_sp16<2, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    case 3: hipLaunchKernelGGL((s16::k_sampler_sp16<3, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    case 4: hipLaunchKernelGGL((s16::k_sampler_sp16<4, SparseT, true>), grid, dim3(64), 0, s, kp, tabs_dev, tb); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
hipError_t launch_sampler_shared_draw(const KArgs& a, int K, const SparseT* tabs_dev, const SampTabs& tb, hipStream_t s) {
  if (!a.route) return hipErrorInvalidValue;
  hipError_t err;
  if (tb.mc4) { /* the mean-only forward kernel has marked the series it left */ }
  else if (a.y) hipLaunchKernelGGL(s16::k_mark_gaps, dim3((a.N + 3) / 4), dim3(256), 0, s, a.y, a.N, a.T, a.route, (int*)nullptr);
  else if ((err = hipMemsetAsync(a.route, 0, (size_t)a.N, s)) != hipSuccess) return err;
  if ((err = hipGetLastError()) != hipSuccess) return err;
  KArgs km = a;
  km.route_take = 0;
  if (!a.z && !tb.z4) return hipErrorInvalidValue;   // the normals: injected, or made by launch_sampler_shared_normals
#define DLM_MS(KK) err = a.z ? launch_mean_sampler<KK, false>(km, tabs_dev, tb, s) : launch_mean_sampler<KK, true>(km, tabs_dev, tb, s)
  switch (K) {
    case 1: DLM_MS(1); break;
    case 2: DLM_MS(2); break;
    case 3: DLM_MS(3); break;
    case 4: DLM_MS(4); break;
    default: return hipErrorInvalidValue;
  }
#undef DLM_MS
  if (err != hipSuccess) return err;
  KArgs kg = a;   // the series with a missing observation: their own factors
  kg.route_take = 1;
  return launch_sampler_t(kg, K, tabs_dev, s);
}

// ---- shared factors of the RTS smoother ------------------------------------------------------------------------------------------
#ifndef DLM_RTS_SHARED_MIN
#define DLM_RTS_SHARED_MIN 2048            // literal Q1: against k_smoother_rts16 per series
#endif
#ifndef DLM_RTS_SHARED_MIN_TEXTBOOK
#define DLM_RTS_SHARED_MIN_TEXTBOOK 6144   // textbook covariance: against k_smoother_sp16 per series
#endif
// The table run is one wave's 2.3 ms (C2, T = 1000) whatever the batch: what the batch must be worth.  Literal Q1
// (tools/sweep_rts_threshold_q1.sh, profiles/r04_rts_threshold_q1.json): the per-series kernel takes 2.53 / 2.72 / 3.74 / 6.98 ms at 512 / 1024 / 2048 / 4096
// series against 2.90 / 3.47 / 3.77 / 4.44 ms through the tables -- up to two waves per SIMD it takes what the table run takes, and the mean kernel comes on top.  Textbook
// (tools/sweep_rts_threshold.sh, profiles/r04_rts_threshold.json): the per-series information-form kernels take 4.79 / 6.56 / 7.46 / 9.27 / 17.6 ms at
// 5000 / 6500 / 7500 / 10 000 / 20 000 series against 5.48 / 6.25 / 6.71 / 7.96 / 15.3 ms through the tables: from six waves per SIMD.
// DLM_OPT_NO_STEADY asks for every series' own recursion at every step: never through the tables.
bool rts_shared_eligible(const KArgs& a, bool textbook) {
  return sampler_shared_model_ok(a) && a.y && !(a.flags & (DLM_OPT_FORCE_GENERIC | DLM_OPT_SMOOTHER_PER_SERIES | DLM_OPT_NO_STEADY)) &&
         (a.N >= (textbook ? DLM_RTS_SHARED_MIN_TEXTBOOK : DLM_RTS_SHARED_MIN) || (a.flags & DLM_OPT_NO_SMALL_BATCH));
}
size_t rts_shared_ws_bytes(const KArgs& a) {
  const size_t n1 = (size_t)a.T + 1, rec = (size_t)a.d + (size_t)a.d * a.d;
  return up64(n1 * s16::RJ_ROW * 8) + up64(n1 * rec * 8) + up64(n1) + 64;   // (64: status block of 16 ints)
}
void rts_shared_carve(void* ws, const KArgs& a, RtsTabs& tb) {
  const size_t n1 = (size_t)a.T + 1, rec = (size_t)a.d + (size_t)a.d * a.d;
  char* p = (char*)ws;
  tb.jrows = (double*)p; p += up64(n1 * s16::RJ_ROW * 8);
  tb.srec = (double*)p;  p += up64(n1 * rec * 8);
  tb.need = (unsigned char*)p; p += up64(n1);
  tb.status = (int*)p; tb.gaps = tb.status + 8; tb.skip = tb.status + 9;
  tb.crec = nullptr; tb.crec_stride = 0;
}
// the tables: the covariance-only filter (one wave; it stops where the recursion settles and a copy kernel fills the rows above) -- IN FRONT of
// the batch's forward pass: beside it, with the memory system saturated by the batch's record stores, the one wave's dependent round trips
// take ten times as long (0.22 -> 2.3 ms measured) -- then, beside the forward pass, the smoother with its export on, reading that table's rows
// as its filter records
hipError_t launch_rts_shared_cov(const KArgs& a, int K, const SparseT* tabs_dev, RtsTabs& tb, const CovTabs& ctb, hipStream_t s) {
  KArgs kc = a;
  kc.smooth = nullptr; kc.filt = nullptr; kc.stats = nullptr; kc.theta = nullptr; kc.z = nullptr;
  tb.crec = ctb.ftab; tb.crec_stride = ctb.frow;
  CovTabs cs = ctb;
  cs.skip = tb.skip;
  return launch_sparse16_cov_filter(kc, K, tabs_dev, cs, s);
}
// Dynamic LDS that, with the kernel's static LDS, fills a CU's 160 KB: no other workgroup that uses LDS -- every batch kernel of this library --
// becomes resident beside it.  A table run is ONE wave whose dependent chain is what the call waits for; on a CU it shares with eight waves
// per SIMD of the batch's forward pass it queues behind their MFMAs (64 cycles of the pipe each) and LDS traffic at every link of the chain
// (measured: 1.8 ms alone, 4.5 ms beside k_filter_sp16).  One CU of 256 is what the isolation costs the batch.
template <int TAG, class F>   // TAG: one static per kernel (instantiations of one template share their function type)
static size_t whole_cu_lds(F kernel) {
  auto ask = [&]() -> size_t {
    hipFuncAttributes at;
    if (hipFuncGetAttributes(&at, (const void*)kernel) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const size_t whole = 160 * 1024;
    if (at.sharedSizeBytes >= whole) return 0;
    const size_t dyn = whole - at.sharedSizeBytes;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return dyn;
  };
  static const size_t dyn = ask();
  return dyn;
}
hipError_t launch_rts_shared_tables(const KArgs& a, int K, const SparseT* tabs_dev, const RtsTabs& tb, hipStream_t s) {
  KArgs kp = a;   // the covariances of every series without a missing observation, bit for bit
  kp.N = 1; kp.y = nullptr; kp.m0_stride = 0; kp.filt_in = nullptr; kp.filt = nullptr; kp.smooth = tb.srec; kp.status = tb.status; kp.stats = nullptr; kp.loglik = nullptr;
  kp.prior = nullptr; kp.fq = nullptr; kp.route = nullptr; kp.counters = nullptr; kp.theta = nullptr; kp.z = nullptr; kp.series_offset = 0; kp.plain = nullptr;
  switch (K) {
    case 1: hipLaunchKernelGGL((s16::k_smoother_rts16<1, SparseT, true>), dim3(1), dim3(64), whole_cu_lds<1>(s16::k_smoother_rts16<1, SparseT, true>), s, kp, tabs_dev, tb); break;
    case 2: hipLaunchKernelGGL((s16::k_smoother_rts16<2, SparseT, true>), dim3(1), dim3(64), whole_cu_lds<2>(s16::k_smoother_rts16<2, SparseT, true>), s, kp, tabs_dev, tb); break;
    case 3: hipLaunchKernelGGL((s16::k_smoother_rts16<3, SparseT, true>), dim3(1), dim3(64), whole_cu_lds<3>(s16::k_smoother_rts16<3, SparseT, true>), s, kp, tabs_dev, tb); break;
    case 4: hipLaunchKernelGGL((s16::k_smoother_rts16<4, SparseT, true>), dim3(1), dim3(64), whole_cu_lds<4>(s16::k_smoother_rts16<4, SparseT, true>), s, kp, tabs_dev, tb); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
template <int K>
static hipError_t launch_mean_rts(const KArgs& a, const SparseT* sp, const RtsTabs& tb, hipStream_t s) {
  const dim3 grid((a.N + 3) / 4), blk(64);
  const int d = a.d;   // store / DMA instructions per step: see k_mean_rts16
  if (d <= 7) hipLaunchKernelGGL((s16::k_mean_rts16<K, 2, 1, 1>), grid, blk, 0, s, a, sp, tb);
  else if (d <= 10) hipLaunchKernelGGL((s16::k_mean_rts16<K, 4, 1, 2>), grid, blk, 0, s, a, sp, tb);
  else if (d <= 13) hipLaunchKernelGGL((s16::k_mean_rts16<K, 6, 2, 2>), grid, blk, 0, s, a, sp, tb);
  else if (d == 14) hipLaunchKernelGGL((s16::k_mean_rts16<K, 8, 2, 2>), grid, blk, 0, s, a, sp, tb);
  else hipLaunchKernelGGL((s16::k_mean_rts16<K, 8, 2, 3>), grid, blk, 0, s, a, sp, tb);
  return hipGetLastError();
}
// route [N]: the series with a missing observation -- or, where more than half of them have one, every series (tb.skip: no tables)
hipError_t launch_rts_shared_mark(const KArgs& a, unsigned char* route, const RtsTabs& tb, hipStream_t s) {
  if (!route || !a.y) return hipErrorInvalidValue;
  hipError_t err = hipMemsetAsync(tb.status, 0, 16 * sizeof(int), s);   // the zero series' status, the gap count, the decision
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(s16::k_mark_gaps, dim3((a.N + 3) / 4), dim3(256), 0, s, a.y, a.N, a.T, route, tb.gaps);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  hipLaunchKernelGGL(s16::k_rts_decide, dim3((a.N + 255) / 256), dim3(256), 0, s, route, a.N, (const int*)tb.gaps, tb.skip);
  return hipGetLastError();
}
// a.route [N] as launch_rts_shared_mark left it; the mean-only kernel smooths the series without a gap, k_smoother_rts16 the others
// (own_rts; otherwise the caller launches its per-series kernel for them)
hipError_t launch_rts_shared_means(const KArgs& a, int K, const SparseT* tabs_dev, const RtsTabs& tb, bool own_rts, hipStream_t s) {
  if (!a.route || !a.y) return hipErrorInvalidValue;
  hipError_t err;
  KArgs km = a;
  km.route_take = 0;
  switch (K) {
    case 1: err = launch_mean_rts<1>(km, tabs_dev, tb, s); break;
    case 2: err = launch_mean_rts<2>(km, tabs_dev, tb, s); break;
    case 3: err = launch_mean_rts<3>(km, tabs_dev, tb, s); break;
    case 4: err = launch_mean_rts<4>(km, tabs_dev, tb, s); break;
    default: return hipErrorInvalidValue;
  }
  if (err != hipSuccess || !own_rts) return err;
  KArgs kg = a;   // the series with a missing observation: their own J_t, S_t
  kg.route_take = 1;
  return launch_rts_t(kg, K, tabs_dev, s);
}

// RTS smoother from filter records alone (a.filt_in -> a.smooth), textbook or literal Q1: structured d <= 15
hipError_t launch_sparse16_rts(const KArgs& a, int K, const SparseT* tabs_dev, hipStream_t s) { return launch_rts_t(a, K, tabs_dev, s); }
hipError_t launch_small_mv_rts(const KArgs& a, hipStream_t s) { return launch_rts_t(a, a.spb_k, a.spb, s); }
// d <= 15 with several observation components: the tables of the multivariate paths (a.spb, [2 gi] rows / [2 gi + 1] columns)
hipError_t launch_small_mv_sampler(const KArgs& a, hipStream_t s) { return launch_sampler_t(a, a.spb_k, a.spb, s); }

}  // namespace dlm
