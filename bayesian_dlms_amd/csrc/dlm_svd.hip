// SVD (square-root) Kalman filter and backward sampler: d <= 48, p <= 32 (two instantiations: NM = 16 -- d, p <= 16, fourteen
// one-wave workgroups per CU -- and NM = 48 -- the multivariate models of BASELINE configs[3], one workgroup per CU).
//
// Restates SvdFilter.scala:38-95, :183-236 and SvdSampler.scala:15-60, :94-102 with the two
// LAPACK dgesdd calls per filter step (one per sampler step) replaced by a per-wavefront
// one-sided (Hestenes) Jacobi SVD held entirely in LDS: the stacked matrix ((2d) x d or
// (p+d) x d) and the accumulated right vectors live in the wave's LDS slice, the n/2 disjoint
// column pairs of a round-robin round are rotated concurrently by 8-lane groups, and the three
// inner products of a pair are reduced with wave shuffles.  One wavefront per series.
//
// Only U D^2 U^T and the means are defined by the algorithm (singular-vector order and sign are
// LAPACK-defined in the reference); the filter records therefore hold the factors in Jacobi
// order.  The sampler, whose draw h + U diag(d) z does depend on order and sign, canonicalises
// its factor: singular values descending, largest-|component| of each column of the FINAL
// factor positive -- the same convention as oracle/dlm_oracle.c.
//
// Quirk switches: DLM_OPT_SVD_RAW_W_Q2 (time update receives the raw W, SvdFilter.scala:158-161)
// and DLM_OPT_SVD_SAMPLER_Q9 (backward step uses sqrt(W) where sqrt(W)^-1 is needed,
// SvdSampler.scala:71-73).  Defaults are the mathematically consistent forms.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

// SL = NM + 1 (defined in every template below): leading dimension of the d x d LDS matrices; `stl` (odd, sized by the shape): that of the stacked matrix
// NM: the largest state / observation dimension of the instantiation (16 or 48).  PK: the row of the stacked matrix where the
// measurement update parks vm fm^T (at or beyond the largest p of the instantiation).
template <int NM> struct SvdDim { static constexpr int SL = NM + 1, PK = NM == 16 ? 16 : 32, PMAX = NM == 16 ? 16 : 32; };
#define M17(buf, i, j) (buf)[(i) + (j) * SL]
#define STK(i, j) stack[(i) + (j) * stl]

// A block is ONE wavefront: LDS hand-offs need only keep the compiler (and the in-order LDS queue) in order --
// no s_barrier, and none of the vmcnt(0) a __syncthreads() fence would add after the record stores.
__device__ __forceinline__ void ssync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// sum over the 8 lanes of a half-row, result in all of them: row_half_mirror (i <-> 7 - i), then the quad
// permutations xor 1 and xor 2 -- three DPP steps, no LDS
__device__ __forceinline__ double sum8(double v) {
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  return v;
}
// 1/sqrt(x) and 1/x from the hardware estimates + Newton steps (about 1 ulp): the Jacobi rotation needs a division
// and two square roots per pair, computed redundantly by every lane of the pair's group
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// One-sided Jacobi SVD of the m x n (m <= 32, n <= 16) LDS matrix A (leading dim lda).
// On return the columns of A are U * Sigma, V (n x n, ld SL) holds the right singular vectors,
// sig[j] = ||A[:, j]||.  Returns 1 if not converged after 40 sweeps.
// The n/2 disjoint column pairs of a round-robin round are rotated concurrently, one pair per 8-lane group; a lane
// keeps its (up to 4) rows of the two columns in registers from the inner products to the rotation.
// warm: V already holds an orthogonal matrix close to the answer (the right vectors of the same decomposition one
// time step earlier): A is first multiplied by it, after which one or two sweeps suffice instead of six to ten.
#ifndef JACOBI_ENDGAME
#define JACOBI_ENDGAME 1e-13
#endif
template <int NM>
__device__ int jacobi_svd(int lane, int m, int n, double* A, int lda, double* V, double* sig, bool warm = false) {
  constexpr int SL = NM + 1;
  constexpr int RK = NM / 4;    // rows of a column per lane of its 8-lane group (m <= 2 NM)
  constexpr int VK = NM / 8;    // rows of V per lane
  constexpr int NCH = NM / 16;  // an 8-lane group takes NCH of the up to NM / 2 pairs of a round, one after the other
  if (!warm) {
    for (int k = lane; k < n * n; k += 64) M17(V, k % n, k / n) = (k % n == k / n) ? 1.0 : 0.0;
  } else {
    ssync();
    // row i of A, replaced by row^T V: lane i and, beyond 64 rows (NM = 48: m <= 96), lane i - 64 in a second pass
    for (int i0 = 0; i0 < m; i0 += 64) {
      double row[NM];
      const int i = i0 + lane;
      if (i < m) {
#pragma unroll
        for (int j = 0; j < NM; ++j) row[j] = j < n ? A[i + j * lda] : 0.0;
        for (int j = 0; j < n; ++j) {
          double acc = 0.0;
#pragma unroll
          for (int l = 0; l < NM; ++l) acc = fma(row[l], l < n ? M17(V, l, j) : 0.0, acc);
          A[i + j * lda] = acc;
        }
      }
    }
  }
  const int np = (n + 1) & ~1, half = np >> 1;
  const int grp = lane >> 3, sub = lane & 7;
  bool conv = (n < 2);
#ifdef DLM_STAMP
  int nsweeps = 0;
#endif
  for (int sweep = 0; sweep < 40 && !conv; ++sweep) {
#ifdef DLM_STAMP
    ++nsweeps;
#endif
    bool rot_any = false, big_any = false;
    // round-robin schedule: group 0 pairs the fixed column np-1 with column r; group g pairs (r + g) and (r - g)
    // modulo np-1 -- both advance by one per round, so no integer division in the loop
    int ras[NCH], rbs[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { const int gg = grp + 8 * ch; ras[ch] = gg % (np - 1); rbs[ch] = (np - 1 - gg % (np - 1)) % (np - 1); }
    for (int r = 0; r < np - 1; ++r) {
      ssync();
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
      const int gg = grp + 8 * ch;
      int ca, cb;
      if (gg == 0) { ca = np - 1; cb = r; }
      else { ca = ras[ch]; cb = rbs[ch]; }
      ras[ch] = (ras[ch] + 1 == np - 1) ? 0 : ras[ch] + 1;
      rbs[ch] = (rbs[ch] + 1 == np - 1) ? 0 : rbs[ch] + 1;
      const bool act = gg < half && ca < n && cb < n;
      const int p = ca < cb ? ca : cb, q = ca < cb ? cb : ca;
      double* Ap = A + p * lda;
      double* Aq = A + q * lda;
      double x[RK], y[RK];
      double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
      for (int k = 0; k < RK; ++k) {
        const int i = sub + 8 * k;
        const bool ok = act && i < m;
        x[k] = ok ? Ap[i] : 0.0;
        y[k] = ok ? Aq[i] : 0.0;
        al = fma(x[k], x[k], al); be = fma(y[k], y[k], be); ga = fma(x[k], y[k], ga);
      }
      al = sum8(al); be = sum8(be); ga = sum8(ga);
      // columns count as orthogonal once |a.b| <= 1e-14 |a||b| (fp64 dot products of <= 32 terms)
      const double g2 = ga * ga, ab = al * be;
      const bool rot = act && ga != 0.0 && g2 > 1e-28 * ab;
      big_any |= act && g2 > JACOBI_ENDGAME * ab;   // a rotation above 3e-7: not yet in the quadratic endgame
      if (rot) {
        // tan of the rotation angle from the raw hardware estimates (~1e-7 relative: any t gives an exact rotation,
        // a slightly wrong one only costs a little convergence); c and s from it at full accuracy, c^2 + s^2 = 1
        const double zeta = (be - al) * 0.5 * __builtin_amdgcn_rcp(ga);
        const double h = fma(zeta, zeta, 1.0);
        const double t = copysign(1.0, zeta) * __builtin_amdgcn_rcp(fabs(zeta) + h * __builtin_amdgcn_rsq(h));
        const double c = fast_rsqrt(fma(t, t, 1.0)), s = t * c;
#pragma unroll
        for (int k = 0; k < RK; ++k) {
          const int i = sub + 8 * k;
          if (i < m) { Ap[i] = fma(c, x[k], -(s * y[k])); Aq[i] = fma(s, x[k], c * y[k]); }
        }
#pragma unroll
        for (int k = 0; k < VK; ++k) {
          const int i = sub + 8 * k;
          if (i < n) {
            const double vx = M17(V, i, p), vy = M17(V, i, q);
            M17(V, i, p) = fma(c, vx, -(s * vy)); M17(V, i, q) = fma(s, vx, c * vy);
          }
        }
      }
      rot_any |= rot;
      }
    }
    // Converged when a sweep rotated nothing -- or when all its rotations were below 3e-7: the off-diagonal mass
    // then drops quadratically to ~1e-13 within this very sweep (two decades below the 1e-11 at which factors count as
    // settled), and the verification sweep (a third of the work with a warm start) can be skipped.
    conv = (__ballot(rot_any) == 0ull) || (__ballot(big_any) == 0ull);
  }
  ssync();
  if (lane < n) {
    double s = 0.0;
    for (int i = 0; i < m; ++i) s = fma(A[i + lane * lda], A[i + lane * lda], s);
    sig[lane] = sqrt(s);
  }
  ssync();
#ifdef DLM_STAMP
  return (conv ? 0 : 1) | (nsweeps << 8);
#else
  return conv ? 0 : 1;
#endif
}


// Eigen-decomposition of diag(delta) + w w^T (delta > 0) by the secular equation -- the measurement update of the SVD filter
// when ONE observation component is present:  the stacked matrix [w^T ; diag(1 / dr)] of SvdFilter.updateState
// (SvdFilter.scala:56) has A^T A = diag(1 / dr^2) + w w^T, so its right singular vectors and squared singular values are
// the eigen-pairs of a diagonal plus a rank-one matrix.  No sweeps: lane i finds eigenvalue i as the root of
//   1 + sum_j w_j^2 / (delta_j - lambda) = 0   in (delta_i, delta_{i+1})        (poles sorted ascending)
// by a safeguarded Newton iteration on F(tau) = tau (1 + r(tau)) - w_o^2 with the origin o at the nearer pole (lambda =
// delta_o + tau, so that delta_j - lambda is computed without cancellation), then the Gu-Eisenstat weights
//   zhat_j^2 = prod_i (lambda_i - delta_j) / prod_{i != j} (delta_i - delta_j)
// make the vectors  v_i ~ zhat_j / (delta_j - lambda_i)  the EXACT eigenvectors of a matrix within rounding of the given
// one -- orthogonal to working precision without any deflation logic.  Equal poles are separated by 16 eps scale and
// vanishing weights floored at 1e-13 sqrt(scale) first (a 1e-13 relative perturbation of the matrix, below the
// steady-state tolerance of this file).  Prototype and stress test: 3000 adversarial cases, |V^T V - I| <= 8e-16.
// In: lane j < d holds delta_j and w_j.  Out: V (d x d, ld SL, columns = eigenvectors in the caller's coordinates),
// sig[i] = sqrt(lambda_i).  scr: 6 x NM doubles of LDS scratch.
template <int NM>
__device__ void secular_update(int lane, int d, double delta, double w, double* V, double* sig, double* scr) {
  constexpr int SL = NM + 1;
  // LDS scratch (6 x NM doubles): sorted poles, squared weights, roots' offsets and origins, Gu-Eisenstat weights, permutation.
  // All loops over j read sds[j] / sw2[j] at a wave-uniform address (LDS broadcast): no per-lane register arrays.
  double* sds = scr; double* sw2 = scr + NM; double* stau = scr + 2 * NM; double* sdso = scr + 3 * NM; double* szh = scr + 4 * NM;
  int* sperm = (int*)(scr + 5 * NM);
  const double EPS = 2.220446049250313e-16;
  const bool act = lane < d;
  if (act) stau[lane] = delta;
  ssync();
  int rank = 0;
  if (act)
    for (int k = 0; k < d; ++k) { const double dk = stau[k]; rank += (dk < delta) || (dk == delta && k < lane); }
  ssync();
  if (act) { sds[rank] = delta; sw2[rank] = w; sperm[rank] = lane; }
  ssync();
  // regularise (every lane the same scalar recurrences, values broadcast from LDS): strictly increasing poles, no vanishing weights
  double wn2 = 0.0, scale = 0.0;
  for (int j = 0; j < d; ++j) { const double wj = sw2[j]; wn2 = fma(wj, wj, wn2); scale = fmax(scale, sds[j]); }
  scale = fmax(scale, wn2);
  const double sep = 16.0 * EPS * scale, floorw = 1e-13 * sqrt(scale);
  double mine = 0.0, prev = 0.0, wmine = 0.0;
  wn2 = 0.0;
  for (int j = 0; j < d; ++j) {
    double dj = sds[j];
    if (j > 0) dj = fmax(dj, prev + sep);
    prev = dj;
    double wj = sw2[j];
    if (fabs(wj) < floorw) wj = wj < 0.0 ? -floorw : floorw;
    wn2 = fma(wj, wj, wn2);
    if (j == lane) { mine = dj; wmine = wj; }
  }
  ssync();
  if (act) { sds[lane] = mine; sw2[lane] = wmine * wmine; }
  const double wsgn = wmine < 0.0 ? -1.0 : 1.0;
  ssync();
  // ---- roots: lane i, eigenvalue i in (delta_i, delta_{i+1}) ----------------------------------------------------------
  const int i = act ? lane : 0;
  const bool last = i == d - 1;
  const double di = sds[i], dip = last ? di : sds[i + 1];
  double lo, hi, dso, w2o, t;
  int o;
  {
    const double mid = 0.5 * (dip - di);
    double hm = 1.0;
    for (int j = 0; j < d; ++j) hm = fma(sw2[j], fast_rcp((sds[j] - di) - mid), hm);
    const bool lower = last || hm >= 0.0;
    o = lower ? i : i + 1;
    dso = lower ? di : dip;
    lo = lower ? 0.0 : -mid;
    hi = last ? wn2 * (1.0 + 1e-12) : (lower ? mid : 0.0);
    w2o = sw2[o];
    double r0 = 0.0;
    for (int j = 0; j < d; ++j) { const double dj = sds[j] - dso; r0 += (j == o) ? 0.0 : sw2[j] * fast_rcp(dj); }
    t = w2o * fast_rcp(1.0 + r0);
    if (!(lo < t && t < hi)) t = 0.5 * (lo + hi);
  }
  bool done = !act;
  for (int it = 0; it < 48; ++it) {
    if (__ballot(!done) == 0ull) break;
    double r = 0.0, rp = 0.0;
    for (int j = 0; j < d; ++j) {
      const double inv = fast_rcp((sds[j] - dso) - t);
      const double tt = (j == o) ? 0.0 : sw2[j] * inv;
      r += tt; rp = fma(tt, inv, rp);
    }
    const double f = fma(t, 1.0 + r, -w2o), fp = (1.0 + r) + t * rp;
    const double step = fp != 0.0 ? f * fast_rcp(fp) : 0.0;
    if (!done) {
      if (fp != 0.0 && fabs(step) <= 4.0 * EPS * fabs(t)) done = true;
      else {
        if ((f > 0.0) == (t > 0.0)) hi = t; else lo = t;
        double tn = fp != 0.0 ? t - step : 0.5 * (lo + hi);
        if (!(lo < tn && tn < hi)) {
          const double a_ = lo >= 0.0 ? lo : -hi, b_ = lo >= 0.0 ? hi : -lo;
          const double m_ = a_ == 0.0 ? 0.125 * fabs(t) : (b_ > 4.0 * a_ ? sqrt(a_ * b_) : 0.5 * (a_ + b_));
          tn = lo >= 0.0 ? m_ : -m_;
        }
        t = tn;
      }
    }
  }
  if (act) { stau[lane] = t; sdso[lane] = dso; }
  ssync();
  // ---- Gu-Eisenstat weights (lane j = i) -------------------------------------------------------------------------------
  {
    double prod = 1.0;
    for (int k = 0; k < d; ++k) {
      const double num = (sdso[k] - di) + stau[k];                          // lambda_k - delta_j
      const double den = (k < i) ? sds[k] - di : (k < d - 1 ? sds[k + 1] - di : 1.0);
      prod *= num * fast_rcp(den);
    }
    if (act) szh[lane] = sqrt(fabs(prod)) * wsgn;
  }
  ssync();
  // ---- eigenvectors: lane i builds column i, rows back in the caller's order ----------------------------------------------
  {
    double nrm = 0.0;
    for (int j = 0; j < d; ++j) { const double x = szh[j] * fast_rcp((sds[j] - dso) - t); nrm = fma(x, x, nrm); }
    const double inv = fast_rsqrt(nrm);
    if (act) {
      for (int j = 0; j < d; ++j) M17(V, sperm[j], lane) = szh[j] * fast_rcp((sds[j] - dso) - t) * inv;
      sig[lane] = sqrt(dso + t);
    }
  }
  ssync();
}

// sqrtSvd / sqrtInvSvd (SvdFilter.scala:210-227): out = diag(sig^{+-1/2}) V^T for the SPD n x n Mx.
template <int NM>
__device__ int sqrt_svd(int lane, int n, const double* Mx /* global, col-major n x n */, bool inverse,
                        double* out, double* stack, int stl, double* Vacc, double* sig) {
  constexpr int SL = NM + 1;
  for (int k = lane; k < n * n; k += 64) STK(k % n, k / n) = Mx[k];
  ssync();
  const int rc = jacobi_svd<NM>(lane, n, n, stack, stl, Vacc, sig);
  for (int k = lane; k < n * n; k += 64) {
    const int i = k % n, j = k / n;
    const double s = inverse ? 1.0 / sqrt(sig[i]) : sqrt(sig[i]);
    M17(out, i, j) = s * M17(Vacc, j, i);
  }
  ssync();
  return rc;
}

// LDS carve shared by both kernels
struct SvdLds {
  double *m, *a, *dc, *dr, *sig, *e, *yv, *tv, *uc, *ur, *V, *Wadv, *tmp, *stack, *sVinv, *gs, *sWb;
  int* idx;
};
// The filter needs only the first part (through the first 16 doubles of gs): its vm fm^T scratch lives in rows 16..31
// of the stacked matrix, so that ten of its one-wave workgroups fit a CU instead of seven (the Jacobi rounds are
// latency-bound: more waves per SIMD is throughput).
// (the sampler: eight vectors, six NM x NM matrices, the 2 NM-row stack, two vectors of statistics -- 18 KB at NM = 16, 151 KB at NM = 48)
template <int NM>
__device__ __forceinline__ SvdLds carve(double* sm) {
  constexpr int SL = NM + 1, STLN = 2 * NM + 1;
  SvdLds L;
  L.m = sm; L.a = L.m + NM; L.dc = L.a + NM; L.dr = L.dc + NM; L.sig = L.dr + NM; L.e = L.sig + NM;
  L.yv = L.e + NM; L.tv = L.yv + NM;
  L.uc = L.tv + NM; L.ur = L.uc + NM * SL; L.V = L.ur + NM * SL; L.Wadv = L.V + NM * SL;
  L.sVinv = nullptr;
  L.stack = L.Wadv + NM * SL;
  L.idx = nullptr;
  L.gs = L.stack + NM * STLN;
  L.tmp = L.gs + 2 * NM; L.sWb = L.tmp + NM * SL;
  return L;
}
template <int NM> constexpr size_t svd_sampler_lds_doubles() { return 8 * NM + 6 * NM * (NM + 1) + NM * (2 * NM + 1) + 2 * NM; }
// The filter's LDS, sized by the shape: the decompositions are latency-bound (a Jacobi round is a dependent chain of ~2400
// cycles), so waves per SIMD is throughput -- 11 KB per one-wave workgroup at d = 13, p = 1 (14 per CU) instead of 18.6.
__device__ __host__ inline int svd_nm(int d, int p) { return (d <= 16 && p <= 16) ? 16 : 48; }
__device__ __host__ inline int svd_filter_stl(int d, int p) {
  const int pk = svd_nm(d, p) == 16 ? 16 : 32;
  int r = 2 * d; if (pk + p > r) r = pk + p; if (p + d > r) r = p + d; if (r < 6) r = 6; return r | 1;
}
// doubles of the stacked matrix's region: n columns of stl rows -- and never less than the 6 NM doubles of scratch the secular
// update takes there (tiny models used to reach past it, into vectors that happened to be dead)
__device__ __host__ inline int svd_stack_doubles(int d, int p) {
  const int n = d > p ? d : p, a = n * svd_filter_stl(d, p), b = 6 * svd_nm(d, p);
  return a > b ? a : b;
}
template <int NM>
__device__ __forceinline__ SvdLds carve_filter(double* sm, int d, int p) {
  constexpr int SL = NM + 1;
  const int n = d > p ? d : p;
  SvdLds L;
  L.m = sm; L.a = L.m + NM; L.dc = L.a + NM; L.dr = L.dc + NM; L.sig = L.dr + NM; L.e = L.sig + NM;
  L.yv = L.e + NM; L.tv = L.yv + NM;
  L.uc = L.tv + NM; L.ur = L.uc + n * SL; L.V = L.ur + n * SL; L.Wadv = L.V + n * SL;
  L.sVinv = L.Wadv + d * SL;
  L.stack = L.sVinv + p * SL;
  L.idx = (int*)(L.stack + svd_stack_doubles(d, p));
  L.gs = L.stack + svd_stack_doubles(d, p) + NM / 2;
  L.tmp = nullptr; L.sWb = nullptr;
  return L;
}
__device__ __host__ inline size_t svd_filter_lds_doubles(int d, int p) {
  const int n = d > p ? d : p, nm = svd_nm(d, p), sl = nm + 1;
  return (size_t)(8 * nm + 3 * n * sl + d * sl + p * sl + svd_stack_doubles(d, p) + nm / 2 + nm + 2);   // (+ 2: the state of the steady-state test behind gs)
}
size_t svd_filter_lds_bytes(int d, int p) { return sizeof(double) * svd_filter_lds_doubles(d, p) + 16; }
// the table run's instantiation (LONE) also keeps G and F in LDS, behind everything else
size_t svd_filter_lone_lds_bytes(int d, int p) { return sizeof(double) * (((svd_filter_lds_doubles(d, p) + 1) & ~(size_t)1) + (size_t)d * d + (size_t)d * p) + 16; }
size_t svd_sampler_lds_bytes(int d, int p) { return sizeof(double) * (svd_nm(d, p) == 16 ? svd_sampler_lds_doubles<16>() : svd_sampler_lds_doubles<48>()) + 16; }

// ---------------------------------------------------------------------------------------
// SVD filter.  Record t: [m_t (d) | dc_t (d) | uc_t (d x d, column-major)].
// ---------------------------------------------------------------------------------------
// rec_stride (doubles, 0: 2 d + d^2) / aux: the covariance-only run of the shared-factor path (k_svd_mean_filter below) writes its
// records as padded table rows and leaves sqrt(V)^-1[0][0] in aux[0].
// LONE: the instantiation of the shared-factor table run -- ONE wave whose dependent chain the whole call waits for: no occupancy bound, so that
// none of its values is spilled (the batch instantiation trades 15 spilled values for a fourth wave per SIMD).
template <int NM, bool LONE = false>
__global__ __launch_bounds__(64, (NM == 16 && !LONE) ? 4 : 1) void k_svd_filter(KArgs a, double* __restrict__ rec_out, int rec_stride, double* __restrict__ aux) {   // 128 VGPRs (15 spilled): 14 one-wave workgroups per CU instead of 12 -- the rotation rounds are latency-bound (C5 84.8 -> 81.9 ms)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, srec = rec_stride ? rec_stride : 2 * d + dd;
  if (a.route && (a.route[n] != 0) != (a.route_take != 0)) return;   // shared-factor call: only the series routed here
  constexpr int SL = NM + 1, PK = SvdDim<NM>::PK;
  SvdLds L = carve_filter<NM>(sm, d, p);
  double* stack = L.stack;
  const int stl = svd_filter_stl(d, p);
  const double* V = a.V + (size_t)n * a.v_stride;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const double* y = a.y + (size_t)n * T * p;
  double* out = rec_out + (size_t)n * (T + 1) * srec;
  int st = 0;

  // transformParams (SvdFilter.scala:232-236)
  if (a.flags & DLM_OPT_SVD_RAW_W_Q2) {
    for (int k = lane; k < dd; k += 64) M17(L.Wadv, k % d, k / d) = W[k];
    ssync();
  } else if (sqrt_svd<NM>(lane, d, W, false, L.Wadv, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
  if (sqrt_svd<NM>(lane, p, V, true, L.sVinv, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
  if (aux && lane == 0) aux[0] = M17(L.sVinv, 0, 0);
  // initialiseState (SvdFilter.scala:83-95): svd(C0) -> dc0 = sqrt(sigma), uc0 = V
  for (int k = lane; k < dd; k += 64) STK(k % d, k / d) = C0[k];
  ssync();
  if (jacobi_svd<NM>(lane, d, d, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
  for (int i = lane; i < d; i += 64) { L.m[i] = m0[i]; L.dc[i] = sqrt(L.sig[i]); out[i] = m0[i]; out[d + i] = sqrt(L.sig[i]); }
  for (int k = lane; k < dd; k += 64) { M17(L.uc, k % d, k / d) = M17(L.V, k % d, k / d); out[2 * d + k] = M17(L.V, k % d, k / d); }
  ssync();

  // Warm starts: L.ur / L.V still hold the right vectors of the previous step's two decompositions.  They are
  // dropped every 64th step (a rotation product drifts from orthogonality by ~1e-16 per step) and whenever the
  // buffers were used for something else.
  bool warm_r = false, warm_c = false;
  // Steady state.  On a regular stretch without missing observations the Riccati recursion converges: the posterior
  // factors a measurement update writes approach the ones it overwrites.  `settled` records for the last update actually
  // computed that the factors are within DLM_SETTLE_TOL (relative to their largest entry, for uc and for dc) of their LIMIT:
  // settle_test (dlm_internal.h) bounds the geometric tail of the step-to-step changes, tested at every computed step -- a
  // small one-step change alone says nothing when the recursion contracts slowly.  While it holds and the transition is
  // the same, (ur, dr) ARE the answer, and with the same observation pattern so are (uc, dc): both decompositions are skipped
  // and only the mean moves.  Anything that disturbs the covariance (a missing observation, another dt, a variance
  // stream) clears it, the test starts over and the full path resumes.
  bool have_r = false, have_c = false, reuse_r = false, settled = false;
  float* settle = (float*)(L.gs + NM);
  settle_reset(settle);
  unsigned nsteady = 0;   // steps that skipped both decompositions (KArgs::counters[0])
  bool chain = false;     // the last step computed a measurement update (with transition chain_g / chain_dt, pattern chain_mask)
  int chain_g = -1;
  double chain_dt = 0.0;
  unsigned long long chain_mask = 0ull;
  int gprev = -1;
  double dtprev = 0.0;
  unsigned long long mask_prev = 0ull;
  int dbg_sw1 = 0, dbg_sw2 = 0;   // sweep counts (reported by the diagnostic build only)
#ifdef DLM_STAMP
  unsigned long long tj = 0, t0_ = 0, tstart;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tstart)::"memory");
#define SVD_T0 asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory");
#define SVD_T1 { unsigned long long t1_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory"); tj += t1_ - t0_; }
#else
#define SVD_T0
#define SVD_T1
#endif
  // LONE (the shared-factor table run: a time-invariant model): G and F from LDS -- in the one wave's dependent chain a global load, even one that
  // hits the cache, is several hundred cycles per matrix product
  double* Gs = sm + ((svd_filter_lds_doubles(d, p) + 1) & ~(size_t)1);
  double* Fs = Gs + dd;
  // ... and the (row, column) of this lane's elements of a d x d matrix come from registers instead of an integer division per element and loop
  int ki[4] = {0, 0, 0, 0}, kj[4] = {0, 0, 0, 0};
  if constexpr (LONE) {
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) { const int k = lane + 64 * s_; ki[s_] = k % d; kj[s_] = k / d; }
  }
  auto each_dd = [&](auto&& f) {   // f(k, row, column) over this lane's elements k = lane, lane + 64, ... of a d x d matrix
    if constexpr (LONE) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) { const int k = lane + 64 * s_; if (k < dd) f(k, ki[s_], kj[s_]); }
    } else {
      for (int k = lane; k < dd; k += 64) f(k, k % d, k / d);
    }
  };
  if constexpr (LONE) {
    for (int k = lane; k < dd; k += 64) Gs[k] = a.G[k];
    for (int k = lane; k < d * p; k += 64) Fs[k] = a.F[k];
    ssync();
  }
  for (int t = 0; t < T; ++t) {
    const double* Gt = LONE ? Gs : a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
    const double* Ft = LONE ? Fs : a.F + (size_t)t * a.f_stride;
    const double dt = a.dt ? a.dt[t] : 1.0;
    if ((t & 63) == 0) { warm_r = false; warm_c = false; }
    // V_t / W_t streams: step t runs with transformParams(p.copy(v = V_t)) / transformParams(p.copy(w = W_t))
    // (DlmFsv.ffbsSvd, DlmFsv.scala:208-228; DlmFsvSystem.ffbsSvd, DlmFsvSystem.scala:176-208).  The square roots use
    // the stack and L.V as scratch: the update step's warm start is gone.
    if (a.w_tstride) {
      const double* Wt = W + (size_t)t * a.w_tstride;
      if (a.flags & DLM_OPT_SVD_RAW_W_Q2) {
        ssync();
        for (int k = lane; k < dd; k += 64) M17(L.Wadv, k % d, k / d) = Wt[k];
        ssync();
      } else {
        ssync();
        if (sqrt_svd<NM>(lane, d, Wt, false, L.Wadv, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
        warm_c = false;
      }
    }
    if (a.v_tstride) {
      ssync();
      if (sqrt_svd<NM>(lane, p, V + (size_t)t * a.v_tstride, true, L.sVinv, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
      warm_c = false;
    }
    // advState (SvdFilter.scala:183-202)
    if (dt == 0.0) {
      for (int i = lane; i < d; i += 64) { L.a[i] = L.m[i]; L.dr[i] = L.dc[i]; }
      for (int k = lane; k < dd; k += 64) M17(L.ur, k % d, k / d) = M17(L.uc, k % d, k / d);
      warm_r = false;
      ssync();
      have_r = false;
      reuse_r = false;
    } else {
      const double sdt = sqrt(dt);
      const int gi = a.g_index ? a.g_index[t] : 0;
      for (int i = lane; i < d; i += 64) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s = fma(Gt[i + k * d], L.m[k], s);
        L.a[i] = s;
      }
      reuse_r = settled && have_r && gi == gprev && dt == dtprev && !a.w_tstride && !(a.flags & DLM_OPT_FORCE_GENERIC);
      if (reuse_r) { ssync(); } else {
      have_r = true; have_c = false; gprev = gi; dtprev = dt;
      each_dd([&](int, int i, int j) {   // stack = [diag(dc) uc^T G^T ; Wadv sqrt(dt)]
        double s = 0.0;
        for (int l = 0; l < d; ++l) s = fma(M17(L.uc, l, i), Gt[j + l * d], s);
        STK(i, j) = L.dc[i] * s;
        STK(d + i, j) = M17(L.Wadv, i, j) * sdt;
      });
      ssync();
      SVD_T0
      { const int rc = jacobi_svd<NM>(lane, 2 * d, d, stack, stl, L.ur, L.dr, warm_r); if (rc & 1) st |= DLM_ST_NOCONV; dbg_sw1 += rc >> 8; }   // dr = sigma, ur = V
      SVD_T1
      warm_r = true;
      }
    }
    // updateState (SvdFilter.scala:38-68)
    const double yl = (lane < p) ? y[(size_t)t * p + lane] : __builtin_nan("");
    const unsigned long long mask = __ballot(yl == yl);
    const int pm = __popcll(mask);
    if (yl == yl) { const int pos = __popcll(mask & ((1ull << lane) - 1ull)); L.idx[pos] = lane; L.yv[pos] = yl; }
    ssync();
    // same prior factors, same observation pattern, same F and V: (uc, dc) of the step before are the posterior factors
    const bool reuse_c = reuse_r && have_c && mask == mask_prev && pm > 0 && !a.f_stride && !a.v_tstride;
    if (pm == 0) {
      for (int i = lane; i < d; i += 64) { L.m[i] = L.a[i]; L.dc[i] = L.dr[i]; }
      for (int k = lane; k < dd; k += 64) M17(L.uc, k % d, k / d) = M17(L.ur, k % d, k / d);
      have_c = false; settled = false; chain = false;
    } else {
      // e = y - fm^T a ; tmp(pm x d) = vm fm^T   (vm = sqrtVinv[idx, idx], fm = F[:, idx])
      for (int j = lane; j < pm; j += 64) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) s = fma(Ft[k + L.idx[j] * d], L.a[k], s);
        L.e[j] = L.yv[j] - s;
      }
      for (int k = lane; k < pm * d; k += 64) {
        const int i = LONE ? 0 : k % pm, j = LONE ? k : k / pm;   // (LONE: p = 1, the one component observed)
        double s = 0.0;
        for (int l = 0; l < pm; ++l) s = fma(M17(L.sVinv, L.idx[i], L.idx[l]), Ft[j + L.idx[l] * d], s);
        STK(PK + i, j) = s;                    // vm fm^T, parked in rows PK.. of the stack
      }
      ssync();
      if (reuse_c) ++nsteady;
      if (!reuse_c) {
      // stack ((pm + d) x d) = [vm fm^T ur ; diag(1/dr)]
      for (int k = lane; k < pm * d; k += 64) {
        const int i = LONE ? 0 : k % pm, j = LONE ? k : k / pm;
        double s = 0.0;
        for (int l = 0; l < d; ++l) s = fma(STK(PK + i, l), M17(L.ur, l, j), s);
        STK(i, j) = s;
      }
      ssync();                                 // rows PK.. are overwritten next
      if (pm == 1 && !(a.flags & DLM_OPT_FORCE_GENERIC)) {
        // one observed component: [w^T ; diag(1 / dr)] has A^T A = diag(1 / dr^2) + w w^T -- no Jacobi sweeps (secular_update)
        const double wj = lane < d ? STK(0, lane) : 0.0;
        const double rj = lane < d ? 1.0 / L.dr[lane] : 0.0;
        ssync();
        SVD_T0
        secular_update<NM>(lane, d, rj * rj, wj, L.V, L.sig, stack);
        SVD_T1
        warm_c = false;
      } else {
      for (int k = lane; k < dd; k += 64) { const int i = k % d, j = k / d; STK(pm + i, j) = (i == j) ? 1.0 / L.dr[i] : 0.0; }
      ssync();
      SVD_T0
      { const int rc = jacobi_svd<NM>(lane, pm + d, d, stack, stl, L.V, L.sig, warm_c); if (rc & 1) st |= DLM_ST_NOCONV; dbg_sw2 += rc >> 8; }
      SVD_T1
      warm_c = true;
      }
      // uc = ur V ; dc = 1 / sigma -- and how far they moved (steady-state test)
      double mxu = 0.0, dfu = 0.0, mxd = 0.0, dfd = 0.0;
      each_dd([&](int, int i, int j) {
        double s = 0.0;
        for (int l = 0; l < d; ++l) s = fma(M17(L.ur, i, l), M17(L.V, l, j), s);
        mxu = fmax(mxu, fabs(s)); dfu = fmax(dfu, fabs(s - M17(L.uc, i, j)));
        M17(L.uc, i, j) = s;
      });
      for (int i = lane; i < d; i += 64) { const double v = 1.0 / L.sig[i]; mxd = fmax(mxd, fabs(v)); dfd = fmax(dfd, fabs(v - L.dc[i])); L.dc[i] = v; }
      for (int o_ = 32; o_ > 0; o_ >>= 1) { mxu = fmax(mxu, __shfl_xor(mxu, o_)); dfu = fmax(dfu, __shfl_xor(dfu, o_)); mxd = fmax(mxd, __shfl_xor(mxd, o_)); dfd = fmax(dfd, __shfl_xor(dfd, o_)); }
      {   // the larger of the two relative changes, in single precision (ratios of maxima)
        const float xu = (float)dfu * __builtin_amdgcn_rcpf((float)mxu), xd = (float)dfd * __builtin_amdgcn_rcpf((float)mxd);
        // the update before this one ran one step earlier with the same transition and observation pattern: else start over
        const int giu = a.g_index ? a.g_index[t] : 0;
        const bool contiguous = chain && chain_g == giu && chain_dt == dt && chain_mask == mask && dt != 0.0;
        if (!contiguous) settle_reset(settle);
        settled = settle_test(settle, fmaxf(xu, xd), 1.f, 1);
        chain = true; chain_g = giu; chain_dt = dt; chain_mask = mask;
      }
      have_c = true; mask_prev = mask;
      }
      // tv (pm) = vm^T vm e ; gs (d) = fm tv ; gain e = uc dc^2 uc^T gs
      for (int j = lane; j < pm; j += 64) {
        double s = 0.0;
        for (int l = 0; l < pm; ++l) {
          double vv = 0.0;   // (vm^T vm)[j][l]
          for (int r = 0; r < pm; ++r) vv = fma(M17(L.sVinv, L.idx[r], L.idx[j]), M17(L.sVinv, L.idx[r], L.idx[l]), vv);
          s = fma(vv, L.e[l], s);
        }
        L.tv[j] = s;
      }
      ssync();
      for (int i = lane; i < d; i += 64) {
        double s = 0.0;
        for (int j = 0; j < pm; ++j) s = fma(Ft[i + L.idx[j] * d], L.tv[j], s);
        L.gs[i] = s;
      }
      ssync();
      for (int i = lane; i < d; i += 64) {     // yv <- dc^2 * (uc^T gs)
        double s = 0.0;
        for (int l = 0; l < d; ++l) s = fma(M17(L.uc, l, i), L.gs[l], s);
        L.yv[i] = L.dc[i] * L.dc[i] * s;
      }
      ssync();
      for (int i = lane; i < d; i += 64) {
        double s = L.a[i];
        for (int l = 0; l < d; ++l) s = fma(M17(L.uc, i, l), L.yv[l], s);
        L.m[i] = s;
      }
    }
    ssync();
    double* o = out + (size_t)(t + 1) * srec;
    for (int i = lane; i < d; i += 64) { o[i] = L.m[i]; o[d + i] = L.dc[i]; }
    each_dd([&](int k, int i, int j) { o[2 * d + k] = M17(L.uc, i, j); });
  }
#ifdef DLM_STAMP
  if (n == 0 && lane == 0 && a.status) { unsigned long long tend; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tend)::"memory"); a.status[1] = dbg_sw1; a.status[2] = dbg_sw2; a.status[3] = (int)(tj / T); a.status[4] = (int)((tend - tstart) / T); }
#else
  (void)dbg_sw1; (void)dbg_sw2;
#endif
  bool bad = false;
  for (int i = lane; i < d; i += 64) bad |= !isfinite(L.m[i]) || !isfinite(L.dc[i]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
  if (a.counters && lane == 0) {
    if (nsteady) atomicAdd(&a.counters[0], (unsigned long long)nsteady);
    if (a.route) atomicAdd(&a.counters[3], 1ull);          // a series of a shared-factor call that ran its own decompositions
  }
}

// ---------------------------------------------------------------------------------------
// Shared factors.  With V, W, C0 shared by the batch and no missing observation the factors (uc_t, dc_t) of the SVD filter do
// not depend on the data: the two decompositions per step -- all of this filter's cost -- are done ONCE per call, by k_svd_filter
// itself on a series of zeros (one wave: its records are the table), and every series runs only
//   a = G m,  e = y - F^T a,  m = a + uc dc^2 uc^T F (sqrt(V)^-1)^2 e          (SvdFilter.scala:58-66)
// with the operations and their order exactly those of k_svd_filter: the records are bit for bit its own.  A series with a
// missing observation is marked in KArgs::route and served by k_svd_filter (p = 1, regular grid, time-invariant model).
// One wave per series; table row t + 1 ([0 | dc | uc], padded to a multiple of 16 bytes) travels two steps ahead into an LDS ring.
// ---------------------------------------------------------------------------------------
typedef int i4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void svd_dma_row(const i4s& rs, unsigned lds_addr, int soff, int lane, int n16) {
  const int voff = lane * 16;
  lds_addr = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  if (lane < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if (lane + 64 < n16)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
  if (lane + 128 < n16)   // (d = 16: 144 pieces)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__global__ __launch_bounds__(256) void k_svd_mean_filter(KArgs a, const double* __restrict__ tab, int tstride, const double* __restrict__ aux,
                                                         const int* __restrict__ cov_status, double* __restrict__ rec_out) {
  extern __shared__ __attribute__((aligned(16))) char ring_all[];
  __shared__ __attribute__((aligned(16))) double lds[4 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = blockIdx.x * (int)(blockDim.x >> 6) + wave;
  if (n >= a.N) return;
  const int d = a.d, T = a.T, dd = d * d, srec = 2 * d + dd, rowb = tstride * 8;
  double* vM = lds + wave * 64;   // m
  double* vA = vM + 16;           // a
  double* vG = vA + 16;           // gs, then yv
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double* y = a.y + (size_t)n * T;
  double* out = rec_out + (size_t)n * (T + 1) * srec;
  const int OOBo = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)(T + 1) * srec * 8), 0x00020000);
  // G and F in registers: lane i holds row i of G and F[i] (d <= 16)
  double Gi[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) Gi[k] = (lane < d && k < d) ? a.G[lane + k * d] : 0.0;
  const double s00 = aux[0];
  const double vv = fma(s00, s00, 0.0);               // (vm^T vm)[0][0]
  const double Fl = lane < d ? a.F[lane] : 0.0;
  double* vF = vG + 16;                               // F
  if (lane < 16) vF[lane] = Fl;
  char* ring = ring_all + wave * 2 * (rowb + 16);
  const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const unsigned long long ta = (unsigned long long)tab;
  const i4s rtab = {__builtin_amdgcn_readfirstlane((int)(unsigned)ta), __builtin_amdgcn_readfirstlane((int)(unsigned)((ta >> 32) & 0xffffu)),
                    __builtin_amdgcn_readfirstlane((int)((size_t)(T + 1) * rowb)), 0x00020000};
  const int n16 = rowb / 16;
  int offp[5];          // this lane's doubles of a record: lane, lane + 64, ... (2 d + d^2 <= 288)
#pragma unroll
  for (int k = 0; k < 5; ++k) offp[k] = (lane + 64 * k < srec) ? (lane + 64 * k) * 8 : OOBo;
  svd_dma_row(rtab, ring_lds, 0, lane, n16);
  svd_dma_row(rtab, ring_lds + rowb + 16, rowb, lane, n16);
  double mi = lane < d ? m0[lane] : 0.0;
  if (lane < 16) vM[lane] = mi;
  double ychunk = (lane < T) ? y[lane] : 0.0;
  asm volatile("" ::"v"(ychunk));
  ssync();
  asm volatile("s_waitcnt vmcnt(1)" ::: "memory");   // row 0
  {   // record 0: [m0 | dc0 | uc0]
    const double* row = (const double*)ring;
    double v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (lane + 64 * k < srec) ? row[lane + 64 * k] : 0.0;
    v[0] = lane < d ? mi : v[0];
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");
    svd_dma_row(rtab, ring_lds, (T >= 2 ? 2 : T) * rowb, lane, n16);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const unsigned lo = (unsigned)__double2loint(v[k]), hi = (unsigned)__double2hiint(v[k]);
      const unsigned __attribute__((ext_vector_type(2))) w = {lo, hi};
      __builtin_amdgcn_raw_buffer_store_b64(w, rout, offp[k], 0, 0);
    }
  }
  for (int t = 0; t < T; ++t) {
    if (t > 0 && (t & 63) == 0) {
      ychunk = (t + lane < T) ? y[t + lane] : 0.0;
      asm volatile("" ::"v"(ychunk));
    }
    const double yt = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ychunk), t & 63), __builtin_amdgcn_readlane(__double2loint(ychunk), t & 63));
    if (!(yt == yt)) {   // a missing observation: the factors of this series are its own -- k_svd_filter takes it
      if (lane == 0) a.route[n] = 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      return;
    }
    // operations issued after the request for row t + 1: the 5 stores of record t - 1, the request for row t + 2 (>= 1), the 5 stores of record t
    if (t == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    const double* row = (const double*)(ring + ((t + 1) & 1) * (rowb + 16));
    // a = G m
    double ai = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < d) ai = fma(Gi[k], vM[k], ai);
    if (lane < 16) vA[lane] = lane < d ? ai : 0.0;
    ssync();
    double fs = 0.0;
    for (int k = 0; k < d; ++k) fs = fma(vF[k], vA[k], fs);
    const double e = yt - fs;
    const double tv = fma(vv, e, 0.0);
    if (lane < 16) vG[lane] = lane < d ? fma(Fl, tv, 0.0) : 0.0;              // gs = F tv
    ssync();
    double yv = 0.0;
    if (lane < d) {
      double s_ = 0.0;
      for (int l = 0; l < d; ++l) s_ = fma(row[2 * d + l + lane * d], vG[l], s_);   // uc[l][i]
      const double dci = row[d + lane];
      yv = dci * dci * s_;
    }
    ssync();
    if (lane < 16) vG[lane] = lane < d ? yv : 0.0;
    ssync();
    if (lane < d) {
      double s_ = ai;
      for (int l = 0; l < d; ++l) s_ = fma(row[2 * d + lane + l * d], vG[l], s_);   // uc[i][l]
      mi = s_;
    }
    ssync();
    if (lane < 16) vM[lane] = lane < d ? mi : 0.0;
    ssync();
    double v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (lane + 64 * k < srec) ? row[lane + 64 * k] : 0.0;
    v[0] = lane < d ? mi : v[0];
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4])::"memory");
    { const int tn = t + 3 <= T ? t + 3 : T; svd_dma_row(rtab, ring_lds + ((t + 1) & 1) * (rowb + 16), tn * rowb, lane, n16); }
    const int so = (t + 1) * srec * 8;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const unsigned lo = (unsigned)__double2loint(v[k]), hi = (unsigned)__double2hiint(v[k]);
      const unsigned __attribute__((ext_vector_type(2))) w = {lo, hi};
      __builtin_amdgcn_raw_buffer_store_b64(w, rout, offp[k], so, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) a.route[n] = 0;
  if (a.counters && lane == 0) atomicAdd(&a.counters[2], 1ull);
  int st = cov_status[0];
  if (__ballot(lane < d && !isfinite(mi)) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// The same mean recursion, four series per wave (round 4): lane 16 j + c = component c of series j of the wave's four consecutive series.
// k_svd_mean_filter above gives a whole wave to one series and 13 of its 64 lanes to the arithmetic: 10 000 waves of ~200 instructions per
// step are issue-bound (6.7 ms for 15.7 GB of records: 2.3 TB/s).  Here the table row is shared by four series (one LDS DMA, broadcast
// reads), every operation of a series' step is the one of k_svd_mean_filter in the same order -- a = G m, f = F^T a, e, tv = vv e,
// gs = F tv, yv_i = dc_i^2 sum_l uc[l][i] gs_l, m_i = a_i + sum_l uc[i][l] yv_l, all sums ascending from their first term -- so the records
// are the same bits, and the four records leave as one stream of 8-byte stores (records are 2 d + d^2 doubles: 8-byte aligned only).
// NS: store instructions per step = ceil(4 (2 d + d^2) / 64) rounded up to 4, 8, 13 or 18 (an instruction whose lanes all lie beyond the
// records is still issued: the waits count it).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double svd_row_pick(double v, int lane, int src) {   // the value of lane src (0..15, wave-uniform) of this lane's 16-lane row
  const int a_ = ((lane & 48) + src) << 2;
  const int lo = __builtin_amdgcn_ds_bpermute(a_, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(a_, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int NS>
__global__ __launch_bounds__(64, 3) void k_svd_mean_filter4(KArgs a, const double* __restrict__ tab, int tstride, const double* __restrict__ aux,
                                                            const int* __restrict__ cov_status, double* __restrict__ rec_out) {
  extern __shared__ __attribute__((aligned(16))) char ring[];      // two table slots of tstride doubles (+ 16 bytes each)
  __shared__ __attribute__((aligned(16))) double lds[5 * 64];
  const int lane = threadIdx.x, j = lane >> 4, c = lane & 15;
  const int n0 = 4 * blockIdx.x;
  if (n0 >= a.N) return;
  const int nser = a.N - n0 < 4 ? a.N - n0 : 4;
  const bool have = j < nser;
  const int n = have ? n0 + j : n0;                 // (a row without a series shadows the first: loads stay in bounds, nothing is stored)
  const int d = a.d, T = a.T, dd = d * d, srec = 2 * d + dd, rowb = tstride * 8;
  const bool vc = c < d;
  double* vM = lds;            // m   [4][16]
  double* vA = vM + 64;        // a
  double* vG = vA + 64;        // gs
  double* vY = vG + 64;        // yv
  double* vF = vY + 64;        // F (16), then unused
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double* y = a.y + (size_t)n * T;
  const size_t sbytes = (size_t)(T + 1) * srec * 8;
  const int OOBo = 0x7ffffff0;
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((char*)rec_out + (size_t)n0 * sbytes, 0, (int)((size_t)nser * sbytes), 0x00020000);
  double Gi[16];               // row c of G
#pragma unroll
  for (int k = 0; k < 16; ++k) Gi[k] = (vc && k < d) ? a.G[c + k * d] : 0.0;
  const double s00 = aux[0];
  const double vv = fma(s00, s00, 0.0);               // (vm^T vm)[0][0]
  const double Fl = vc ? a.F[c] : 0.0;
  if (lane < 16) vF[lane] = Fl;
  const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const unsigned vM_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)vM;
  const unsigned long long ta = (unsigned long long)tab;
  const i4s rtab = {__builtin_amdgcn_readfirstlane((int)(unsigned)ta), __builtin_amdgcn_readfirstlane((int)(unsigned)((ta >> 32) & 0xffffu)),
                    __builtin_amdgcn_readfirstlane((int)((size_t)(T + 1) * rowb)), 0x00020000};
  const int n16 = rowb / 16;
  bool dead = !have;
  // the four records as one stream of doubles: double q = 64 k + lane is double q % srec of series q / srec; the first d of a record are its mean (vM)
  unsigned psrc[NS];
  int pdst[NS], pser[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int q = 64 * k + lane, sj = q / srec, pp = q - sj * srec;
    pser[k] = sj < 4 ? sj : 4;
    pdst[k] = sj < nser ? (int)((size_t)sj * sbytes) + pp * 8 : OOBo;
    psrc[k] = pp < d ? (0x80000000u | (unsigned)((16 * (sj & 3) + pp) * 8)) : (unsigned)(pp * 8);
  }
  unsigned deadmask = 0;
  auto read_doubles = [&](unsigned slot, double (&v)[NS]) {   // this lane's doubles of the four records: the means from vM, the rest from the table row
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const unsigned ad = (psrc[k] & 0x80000000u) ? vM_lds + (psrc[k] & 0x7fffffffu) : slot + psrc[k];
      asm volatile("ds_read_b64 %0, %1" : "=v"(v[k]) : "v"(ad) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NS; ++k) asm volatile("" : "+v"(v[k]));
  };
  auto store_doubles = [&](const double (&v)[NS], int so, unsigned dm) {
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      const unsigned lo = (unsigned)__double2loint(v[k]), hi = (unsigned)__double2hiint(v[k]);
      const unsigned __attribute__((ext_vector_type(2))) w = {lo, hi};
      __builtin_amdgcn_raw_buffer_store_b64(w, rout, ((dm >> pser[k]) & 1u) ? OOBo : pdst[k], so, 0);
    }
  };
  svd_dma_row(rtab, ring_lds, 0, lane, n16);
  svd_dma_row(rtab, ring_lds + rowb + 16, rowb, lane, n16);
  double mi = vc ? m0[c] : 0.0;
  vM[lane] = mi;
  double yk[4];                                      // the observations of this row's series, 64 steps at a time
#pragma unroll
  for (int k = 0; k < 4; ++k) yk[k] = (16 * k + c < T) ? y[16 * k + c] : 0.0;
  asm volatile("" ::"v"(yk[0]), "v"(yk[1]), "v"(yk[2]), "v"(yk[3]));
  ssync();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // rows 0 and 1
  {   // record 0: [m0 | dc0 | uc0]
    deadmask = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) deadmask |= (__builtin_amdgcn_readlane((int)dead, 16 * q) & 1) << q;
    deadmask |= 16u;
    double v[NS];
    read_doubles(ring_lds, v);
    svd_dma_row(rtab, ring_lds, (T >= 2 ? 2 : T) * rowb, lane, n16);
    store_doubles(v, 0, deadmask);
  }
  for (int t = 0; t < T; ++t) {
    if (t > 0 && (t & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) yk[k] = (t + 16 * k + c < T) ? y[t + 16 * k + c] : 0.0;
      asm volatile("" ::"v"(yk[0]), "v"(yk[1]), "v"(yk[2]), "v"(yk[3]));
    }
    const int kk = (t >> 4) & 3;
    const double yt = svd_row_pick(kk == 0 ? yk[0] : kk == 1 ? yk[1] : kk == 2 ? yk[2] : yk[3], lane, t & 15);
    if (!dead && !(yt == yt)) {   // a missing observation: the factors of this series are its own -- k_svd_filter takes it (all of it)
      dead = true;
      if (c == 0) a.route[n] = 1;
    }
    deadmask = 16u;
#pragma unroll
    for (int q = 0; q < 4; ++q) deadmask |= (__builtin_amdgcn_readlane((int)dead, 16 * q) & 1) << q;
    if ((deadmask & 15u) == 15u) break;
    // operations issued after the request for row t + 1: the NS stores of record t - 1, the request for row t + 2 (>= 1), the NS stores of record t
    if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + 1) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NS + 1) : "memory");
    const unsigned slot = ring_lds + ((t + 1) & 1) * (rowb + 16);
    const double* row = (const double*)(ring + ((t + 1) & 1) * (rowb + 16));
    // a = G m
    double ai = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) if (k < d) ai = fma(Gi[k], vM[16 * j + k], ai);
    vA[lane] = vc ? ai : 0.0;
    ssync();
    double fs = 0.0;
    for (int k = 0; k < d; ++k) fs = fma(vF[k], vA[16 * j + k], fs);
    const double e = yt - fs;
    const double tv = fma(vv, e, 0.0);
    vG[lane] = vc ? fma(Fl, tv, 0.0) : 0.0;              // gs = F tv
    ssync();
    double yv = 0.0;
    if (vc) {
      double s_ = 0.0;
      for (int l = 0; l < d; ++l) s_ = fma(row[2 * d + l + c * d], vG[16 * j + l], s_);   // uc[l][i]
      const double dci = row[d + c];
      yv = dci * dci * s_;
    }
    vY[lane] = vc ? yv : 0.0;
    ssync();
    if (vc) {
      double s_ = ai;
      for (int l = 0; l < d; ++l) s_ = fma(row[2 * d + c + l * d], vY[16 * j + l], s_);   // uc[i][l]
      mi = s_;
    }
    ssync();                                             // (the reads of vM by this step's a = G m are done)
    vM[lane] = vc ? mi : 0.0;
    ssync();
    double v[NS];
    read_doubles(slot, v);
    { const int tn = t + 3 <= T ? t + 3 : T; svd_dma_row(rtab, slot, tn * rowb, lane, n16); }
    store_doubles(v, (t + 1) * srec * 8, deadmask);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be writing this block's LDS when the wave ends
  int st = cov_status[0];
  const unsigned long long badl = __ballot(!dead && vc && !isfinite(mi));
  if (have && c == 0) {
    if (!dead) {
      a.route[n] = 0;
      const int sj = st | (((badl >> (16 * j)) & 0xffffull) ? DLM_ST_NONFINITE : 0);
      if (a.status && sj) atomicOr(&a.status[n], sj);
    }
  }
  const unsigned long long live = __ballot(have && !dead && c == 0);
  if (a.counters && lane == 0 && live) atomicAdd(&a.counters[2], (unsigned long long)__builtin_popcountll(live));
}

bool svd_shared_eligible(const KArgs& a) {
  return a.d <= 16 && a.p == 1 && !a.g_index && !a.dt && !a.f_stride && !a.v_tstride && !a.w_tstride && !a.v_stride && !a.w_stride && !a.c0_stride &&
         !(a.flags & (DLM_OPT_SVD_PER_SERIES | DLM_OPT_FORCE_GENERIC));
}
size_t svd_shared_ws_doubles(const KArgs& a) { return (size_t)(a.T + 1) * (size_t)((2 * a.d + a.d * a.d + 1) & ~1) + (size_t)a.T + 64; }
// ws: svd_shared_ws_doubles(a) doubles; route: [N] bytes
hipError_t launch_svd_filter_shared(const KArgs& a, double* svd_rec, double* ws, unsigned char* route, hipStream_t s) {
  const int d = a.d, srec = 2 * d + d * d, tstride = (srec + 1) & ~1;
  double* tab = ws;
  double* zeros = tab + (size_t)(a.T + 1) * tstride;      // [T] zero observations, then: aux, the covariance-only run's status
  double* aux = zeros + a.T;
  int* cst = (int*)(aux + 8);
  hipError_t err = hipMemsetAsync(zeros, 0, sizeof(double) * ((size_t)a.T + 16), s);
  if (err != hipSuccess) return err;
  KArgs kc = a;
  kc.N = 1; kc.y = zeros; kc.m0 = zeros; kc.m0_stride = 0; kc.status = cst; kc.counters = nullptr; kc.route = nullptr;
  hipLaunchKernelGGL((k_svd_filter<16, true>), dim3(1), dim3(64), svd_filter_lone_lds_bytes(a.d, a.p), s, kc, tab, tstride, aux);
  if ((err = hipGetLastError()) != hipSuccess) return err;
  KArgs km = a;
  km.route = route; km.route_take = 0;
#ifdef DLM_SVD_MEAN_ONE_PER_WAVE
  const int wpb = 4;
  hipLaunchKernelGGL(k_svd_mean_filter, dim3((a.N + wpb - 1) / wpb), dim3(64 * wpb), (size_t)wpb * 2 * (tstride * 8 + 16), s, km, (const double*)tab, tstride,
                     (const double*)aux, (const int*)cst, svd_rec);
#else
  {
    const dim3 grid((a.N + 3) / 4), blk(64);
    const size_t ring = 2 * ((size_t)tstride * 8 + 16);
    const int ns = (4 * srec + 63) / 64;
    if (ns <= 4) hipLaunchKernelGGL(k_svd_mean_filter4<4>, grid, blk, ring, s, km, (const double*)tab, tstride, (const double*)aux, (const int*)cst, svd_rec);
    else if (ns <= 8) hipLaunchKernelGGL(k_svd_mean_filter4<8>, grid, blk, ring, s, km, (const double*)tab, tstride, (const double*)aux, (const int*)cst, svd_rec);
    else if (ns <= 13) hipLaunchKernelGGL(k_svd_mean_filter4<13>, grid, blk, ring, s, km, (const double*)tab, tstride, (const double*)aux, (const int*)cst, svd_rec);
    else hipLaunchKernelGGL(k_svd_mean_filter4<18>, grid, blk, ring, s, km, (const double*)tab, tstride, (const double*)aux, (const int*)cst, svd_rec);
  }
#endif
  if ((err = hipGetLastError()) != hipSuccess) return err;
  KArgs kg = a;
  kg.route = route; kg.route_take = 1;
  hipLaunchKernelGGL(k_svd_filter<16>, dim3(a.N), dim3(64), svd_filter_lds_bytes(a.d, a.p), s, kg, svd_rec, 0, (double*)nullptr);
  return hipGetLastError();
}

// canonical factor: columns of U (d x d, LDS ld SL) and entries of s reordered so that the
// ordering key is descending, then each column's largest-|.| entry made positive.
template <int NM>
__device__ void canon_factor(int lane, int d, double* U, double* s, const double* key, double* Utmp, double* stmp) {
  constexpr int SL = NM + 1;
  if (lane < d) {
    int rank = 0;
    for (int k = 0; k < d; ++k) rank += (key[k] > key[lane]) || (key[k] == key[lane] && k < lane);
    stmp[rank] = s[lane];
    // sign: first index attaining the maximum |.| (with the oracle's 1e-12 relative tie margin)
    int arg = 0; double best = -1.0;
    for (int i = 0; i < d; ++i) { const double v = fabs(M17(U, i, lane)); if (v > best * (1.0 + 1e-12)) { best = v; arg = i; } }
    const double sg = M17(U, arg, lane) < 0.0 ? -1.0 : 1.0;
    for (int i = 0; i < d; ++i) M17(Utmp, i, rank) = sg * M17(U, i, lane);
  }
  ssync();
  for (int k = lane; k < d * d; k += 64) M17(U, k % d, k / d) = M17(Utmp, k % d, k / d);
  if (lane < d) s[lane] = stmp[lane];
  ssync();
}

// ---------------------------------------------------------------------------------------
// SVD backward sampler + Gibbs statistics (SvdSampler.scala:15-60)
// ---------------------------------------------------------------------------------------
template <int NM>
__global__ __launch_bounds__(64) void k_svd_sampler(KArgs a, const double* __restrict__ rec_in) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int n = blockIdx.x, lane = threadIdx.x;
  const int d = a.d, p = a.p, T = a.T, dd = d * d, srec = 2 * d + dd;
  constexpr int SL = NM + 1;
  SvdLds L = carve<NM>(sm);
  double* stack = L.stack;
  constexpr int stl = 2 * NM + 1;
  double* th = L.e;      // theta_{t+1}
  double* zv = L.yv;     // normals
  double* ssv = L.dr;    // per-state sum of squares (diag statistics)
  double* outer = L.ur;  // outer-product statistics
  double* ssy = L.gs; double* nob = L.gs + NM;
  const double* W = a.W + (size_t)n * a.w_stride;
  const double* rin = rec_in + (size_t)n * (T + 1) * srec;
  const double* y = a.y ? a.y + (size_t)n * T * p : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * d : nullptr;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * d : nullptr;
  const bool want_outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  int st = 0;

  // ps.w of SvdSampler.ffbs: sqrtSvd(W) literally (Q9), sqrtInvSvd(W) for the consistent form
  if (sqrt_svd<NM>(lane, d, W, !(a.flags & DLM_OPT_SVD_SAMPLER_Q9), L.sWb, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
  for (int i = lane; i < d; i += 64) ssv[i] = 0.0;
  for (int k = lane; k < dd; k += 64) M17(outer, k % d, k / d) = 0.0;
  for (int i = lane; i < 2 * NM; i += 64) ssy[i] = 0.0;
  ssync();

  // initialise (SvdSampler.scala:38-45): theta_T = m_T + uc_T diag(dc_T) z, canonical factor order
  {
    const double* r = rin + (size_t)T * srec;
    for (int i = lane; i < d; i += 64) {
      L.m[i] = r[i]; L.dc[i] = r[d + i]; L.sig[i] = 1.0 / r[d + i];   // key: sigma = 1/dc, descending
      zv[i] = zin ? zin[(size_t)T * d + i] : philox_normal(a.seed, series, (unsigned)T, (unsigned)i);
    }
    for (int k = lane; k < dd; k += 64) M17(L.uc, k % d, k / d) = r[2 * d + k];
    ssync();
    canon_factor<NM>(lane, d, L.uc, L.dc, L.sig, L.tmp, L.tv);
    for (int i = lane; i < d; i += 64) {
      double s = L.m[i];
      for (int k = 0; k < d; ++k) s = fma(M17(L.uc, i, k) * L.dc[k], zv[k], s);
      th[i] = s;
      if (thout) thout[(size_t)T * d + i] = s;
    }
    ssync();
  }
  for (int t = T - 1; t >= 0; --t) {
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * dd;
    const double* Ft = a.F + (size_t)t * a.f_stride;
    const double dt = a.dt ? a.dt[t] : 1.0;
    const double* r = rin + (size_t)t * srec;
    if (a.stats && y) {   // observation residuals of theta_{t+1} (Gibbs.scala:29-39)
      for (int j = lane; j < p; j += 64) {
        const double yj = y[(size_t)t * p + j];
        if (yj == yj) {
          double f = 0.0;
          for (int k = 0; k < d; ++k) f = fma(Ft[k + j * d], th[k], f);
          ssy[j] += (yj - f) * (yj - f); nob[j] += 1.0;
        }
      }
    }
    for (int i = lane; i < d; i += 64) {
      L.m[i] = r[i]; L.dc[i] = r[d + i];
      zv[i] = zin ? zin[(size_t)t * d + i] : philox_normal(a.seed, series, (unsigned)t, (unsigned)i);
    }
    for (int k = lane; k < dd; k += 64) M17(L.uc, k % d, k / d) = r[2 * d + k];
    ssync();
    if (a.w_tstride) {   // the step from record t uses W_t, the transition into observation t (DlmFsvSystem.scala:196-205)
      if (sqrt_svd<NM>(lane, d, W + (size_t)t * a.w_tstride, !(a.flags & DLM_OPT_SVD_SAMPLER_Q9), L.sWb, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
    }
    // a_{t+1} = G m_t ; tmp = sqrtWb G
    for (int i = lane; i < d; i += 64) {
      double s = 0.0;
      for (int k = 0; k < d; ++k) s = fma(Gt[i + k * d], L.m[k], s);
      L.a[i] = s;
    }
    for (int k = lane; k < dd; k += 64) {
      const int i = k % d, j = k / d;
      double s = 0.0;
      for (int l = 0; l < d; ++l) s = fma(M17(L.sWb, i, l), Gt[l + j * d], s);
      M17(L.tmp, i, j) = s;
    }
    ssync();
    // stack (2d x d) = [sqrtWb G uc ; diag(1/dc)]
    for (int k = lane; k < dd; k += 64) {
      const int i = k % d, j = k / d;
      double s = 0.0;
      for (int l = 0; l < d; ++l) s = fma(M17(L.tmp, i, l), M17(L.uc, l, j), s);
      STK(i, j) = s;
      STK(d + i, j) = (i == j) ? 1.0 / L.dc[i] : 0.0;
    }
    ssync();
    if (jacobi_svd<NM>(lane, 2 * d, d, stack, stl, L.V, L.sig)) st |= DLM_ST_NOCONV;
    // uh = uc V -> Wadv buffer ; dh = 1/sigma -> tv
    for (int k = lane; k < dd; k += 64) {
      const int i = k % d, j = k / d;
      double s = 0.0;
      for (int l = 0; l < d; ++l) s = fma(M17(L.uc, i, l), M17(L.V, l, j), s);
      M17(L.Wadv, i, j) = s;
    }
    for (int i = lane; i < d; i += 64) L.tv[i] = 1.0 / L.sig[i];
    ssync();
    canon_factor<NM>(lane, d, L.Wadv, L.tv, L.sig, L.V, L.dc);   // V, dc are free scratch now
    double* uh = L.Wadv; double* dh = L.tv;
    // h = m + uh dh^2 uh^T G^T sqrtWb^T sqrtWb (theta_{t+1} - a_{t+1})
    double* u = L.sig;   // d-vectors: reuse sig, dc as scratch
    double* v1 = L.dc;
    for (int i = lane; i < d; i += 64) u[i] = th[i] - L.a[i];
    ssync();
    for (int i = lane; i < d; i += 64) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(M17(L.sWb, i, k), u[k], s); v1[i] = s; }
    ssync();
    for (int i = lane; i < d; i += 64) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(M17(L.sWb, k, i), v1[k], s); u[i] = s; }
    ssync();
    for (int i = lane; i < d; i += 64) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(Gt[k + i * d], u[k], s); v1[i] = s; }
    ssync();
    for (int i = lane; i < d; i += 64) { double s = 0.0; for (int k = 0; k < d; ++k) s = fma(M17(uh, k, i), v1[k], s); u[i] = dh[i] * dh[i] * s; }
    ssync();
    for (int i = lane; i < d; i += 64) {
      double h = L.m[i];
      for (int k = 0; k < d; ++k) h = fma(M17(uh, i, k), u[k], h);
      double s = h;
      for (int k = 0; k < d; ++k) s = fma(M17(uh, i, k) * dh[k], zv[k], s);
      v1[i] = s;   // theta_t
    }
    ssync();
    if (a.stats) {
      for (int i = lane; i < d; i += 64) {
        double s = th[i];
        for (int k = 0; k < d; ++k) s = fma(-Gt[i + k * d], v1[k], s);
        u[i] = s;   // theta_{t+1} - G theta_t
      }
      ssync();
      const double dts = (dt == 0.0) ? 1.0 : dt;
      if (want_outer) for (int k = lane; k < dd; k += 64) M17(outer, k % d, k / d) += u[k % d] * u[k / d] / dts;
      for (int i = lane; i < d; i += 64) ssv[i] += u[i] * u[i] / dts;
    }
    ssync();
    for (int i = lane; i < d; i += 64) { th[i] = v1[i]; if (thout) thout[(size_t)t * d + i] = v1[i]; }
    ssync();
  }
  bool bad = false;
  for (int i = lane; i < d; i += 64) bad |= !isfinite(th[i]);
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.stats) {
    const int Ls = stats_len(d, p, a.flags);
    double* so = a.stats + (size_t)n * Ls;
    for (int j = lane; j < p; j += 64) { so[j] = ssy[j]; so[p + j] = nob[j]; }
    if (want_outer) for (int k = lane; k < dd; k += 64) so[2 * p + k] = M17(outer, k % d, k / d);
    else for (int i = lane; i < d; i += 64) so[2 * p + i] = ssv[i];
    if (lane == 0) so[Ls - 1] = (double)T;
  }
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

bool svd_supported(const KArgs& a) { return a.d >= 1 && a.d <= 48 && a.p >= 1 && a.p <= 32; }
// the NM = 48 instantiations take up to 151 KB of the CU's 160 KB of LDS: above the 64 KB a kernel gets without asking
static hipError_t svd_big_lds_once() {
  static hipError_t done = []() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_svd_filter<48>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_svd_sampler<48>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }();
  return done;
}
hipError_t launch_svd_filter(const KArgs& a, double* svd_rec, hipStream_t s) {
  if (!svd_supported(a)) return hipErrorNotSupported;
  if (svd_nm(a.d, a.p) == 16) hipLaunchKernelGGL(k_svd_filter<16>, dim3(a.N), dim3(64), svd_filter_lds_bytes(a.d, a.p), s, a, svd_rec, 0, (double*)nullptr);
  else {
    const hipError_t e = svd_big_lds_once();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_svd_filter<48>, dim3(a.N), dim3(64), svd_filter_lds_bytes(a.d, a.p), s, a, svd_rec, 0, (double*)nullptr);
  }
  return hipGetLastError();
}

hipError_t launch_svd_sampler(const KArgs& a, const double* svd_rec, hipStream_t s) {
  if (!svd_supported(a)) return hipErrorNotSupported;
  if (svd_nm(a.d, a.p) == 16) hipLaunchKernelGGL(k_svd_sampler<16>, dim3(a.N), dim3(64), svd_sampler_lds_bytes(a.d, a.p), s, a, svd_rec);
  else {
    const hipError_t e = svd_big_lds_once();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_svd_sampler<48>, dim3(a.N), dim3(64), svd_sampler_lds_bytes(a.d, a.p), s, a, svd_rec);
  }
  return hipGetLastError();
}

}  // namespace dlm
