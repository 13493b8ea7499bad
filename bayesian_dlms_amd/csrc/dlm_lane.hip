// One LANE per series for the smallest models: d <= 5, p = 1 (local level, linear growth, quadratic trend, a trend with
// one or two harmonics, ...; Dlm.scala:139-243) -- the shapes of the reference's first examples (FirstOrderDlm.scala)
// run over very many series.
//
// With d <= 5 a wavefront per series (dlm_sparse16.hip) computes on a 16 x 16 tile that is 1-10 % full and moves
// 16-240 bytes per step; the recursion of such a model is a few dozen to a few hundred scalar operations.  Here a lane owns a series:
// the state (m, C) lives in a handful of registers, the model matrices are wave-uniform (scalar loads), and the only
// traffic is the record stream -- the pass is bound by how well a lane's private, contiguous stream uses HBM, like
// the AR(1) kernel (dlm_ar1.hip): observations / records are requested a TILE of steps ahead as 16-byte loads and
// every step's record leaves as whole 16-byte stores (a record of d + d^2 = d (d + 1) doubles is an even number of doubles).
//
// Filter: KalmanFilter.scala:64-118, 262-294 (advance, forecast, masked update; dt == 0: identity advance);
// log-likelihood :138-153.  Smoother: Smoothing.scala:31-64 in its textbook form
//   J = C G^T R^-1,  s = m + J (s+ - a+),  S = C - J (R+ - S+) J^T
// with (a+, R+) recomputed from (m, C) exactly as the filter formed them -- no side buffer, so the same kernel serves
// the fused call and dlm_smooth_batch.  FFBS: the simulation smoother (k_simfilter_lane / k_simsmooth_lane, further down).
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {
namespace lane {

typedef double dbl2 __attribute__((ext_vector_type(2), aligned(8)));
constexpr int YT = 16;   // observations requested per tile in the forward pass (one 128-byte line per lane)

template <int D>
struct Adv { double a[D], R[D][D]; };

// a = G m, R = G C G^T + W dt (lower triangle computed, mirrored); dt == 0: identity advance
template <int D>
__device__ __forceinline__ void advance(const double* __restrict__ Gt, double dt, const double (&W)[D][D],
                                        const double (&m)[D], const double (&C)[D][D], double (&a)[D], double (&R)[D][D]) {
  if (dt == 0.0) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      a[i] = m[i];
#pragma unroll
      for (int j = 0; j < D; ++j) R[i][j] = C[i][j];
    }
    return;
  }
  double G[D][D], T[D][D];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int k = 0; k < D; ++k) G[i][k] = Gt[i + k * D];   // column-major table, wave-uniform
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) s = fma(G[i][k], m[k], s);
    a[i] = s;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double t_ = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t_ = fma(G[i][k], C[k][j], t_);
      T[i][j] = t_;
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      double s = W[i][j] * dt;
#pragma unroll
      for (int k = 0; k < D; ++k) s = fma(T[i][k], G[j][k], s);
      R[i][j] = s; R[j][i] = s;
    }
}

template <int D>
__device__ __forceinline__ void load_w(const double* __restrict__ Wp, double (&W)[D][D]) {
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = 0; j < D; ++j) W[i][j] = Wp[i + j * D];
}

// ---------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------
// IRR = false: one G, one F, unit time increments -- the model is read once, before the time loop (a step of a single
// series is then a chain of a few dozen dependent operations with no load in it)
template <int D, bool IRR>
__global__ __launch_bounds__(64) void k_filter_lane(KArgs a) {
  constexpr int REC = D + D * D;
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= a.N) return;
  const int T = a.T;
  const double* V0 = a.V + (size_t)n * a.v_stride;
  const double* W0 = a.W + (size_t)n * a.w_stride;
  double V = V0[0], W[D][D], m[D], C[D][D];
  load_w<D>(W0, W);
  {
    const double* m0 = a.m0 + (size_t)n * a.m0_stride;
    const double* C0 = a.C0 + (size_t)n * a.c0_stride;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      m[i] = m0[i];
#pragma unroll
      for (int j = 0; j < D; ++j) C[i][j] = C0[i + j * D];
    }
  }
  const double* y = a.y + (size_t)n * T;
  double* out = a.filt ? a.filt + (size_t)n * (T + 1) * REC : nullptr;
  double* pri = a.prior ? a.prior + (size_t)n * (T + 1) * REC : nullptr;
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * 2 : nullptr;
  int st = 0;
  double ll = 0.0;
  auto store = [&](double* base, int t, const double (&mm)[D], const double (&CC)[D][D]) {
    double r[REC];
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = mm[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) r[D + i + j * D] = CC[i][j];
    dbl2* o = (dbl2*)(base + (size_t)t * REC);
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) o[q] = dbl2{r[2 * q], r[2 * q + 1]};
  };
  if (out) store(out, 0, m, C);
  if (pri) store(pri, 0, m, C);
  if (fq) { fq[0] = __builtin_nan(""); fq[1] = __builtin_nan(""); }

  double Gc[D * D], Fc[D];     // the time-invariant model (IRR = false)
#pragma unroll
  for (int i = 0; i < D * D; ++i) Gc[i] = a.G[i];
#pragma unroll
  for (int i = 0; i < D; ++i) Fc[i] = a.F[i];
  const int Tl = T - 1;
  double yb[YT];
  auto request = [&](int t0) {
    if (t0 + YT <= T && (T & 1) == 0) {   // whole tile, 16-byte aligned stream
#pragma unroll
      for (int j = 0; j < YT / 2; ++j) { const dbl2 v = *(const dbl2*)(y + t0 + 2 * j); yb[2 * j] = v.x; yb[2 * j + 1] = v.y; }
    } else {
#pragma unroll
      for (int j = 0; j < YT; ++j) { const int t = t0 + j < Tl ? t0 + j : Tl; yb[j] = y[t]; }
    }
  };
  // D <= 2: a record is 16 / 48 bytes; eight of them leave together as whole 128-byte lines (a line written piecemeal
  // over eight steps is evicted from L2 in between, and HBM then sees 16-byte masked writes)
  constexpr int OT = D <= 2 ? 8 : 1;
  dbl2 ob[OT][REC / 2];
  if (T > 0) request(0);
  for (int t0 = 0; t0 < T; t0 += YT) {
    double yc[YT];
#pragma unroll
    for (int j = 0; j < YT; ++j) yc[j] = yb[j];
    if (t0 + YT < T) request(t0 + YT);
    const bool whole = t0 + YT <= T;
#pragma unroll
    for (int j = 0; j < YT; ++j) {
      const int t = t0 + j;
      if (t < T) {
        const double dt = (IRR && a.dt) ? a.dt[t] : 1.0;
        const double* Gt = IRR ? a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * (D * D) : Gc;
        const double* Ft = IRR ? a.F + (size_t)t * a.f_stride : Fc;
        if (IRR && a.v_tstride) V = V0[(size_t)t * a.v_tstride];
        if (IRR && a.w_tstride) load_w<D>(W0 + (size_t)t * a.w_tstride, W);
        double av[D], R[D][D];
        advance<D>(Gt, dt, W, m, C, av, R);
        if (pri) store(pri, t + 1, av, R);
        double F[D], RF[D], f = 0.0, Q = V;
#pragma unroll
        for (int i = 0; i < D; ++i) { F[i] = Ft[i]; f = fma(F[i], av[i], f); }
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) s = fma(R[i][k], F[k], s);
          RF[i] = s;
        }
#pragma unroll
        for (int i = 0; i < D; ++i) Q = fma(F[i], RF[i], Q);
        if (fq) { fq[2 * (t + 1)] = f; fq[2 * (t + 1) + 1] = Q; }
        const double yt = yc[j];
        if (yt == yt) {
          if (!(Q > 0.0)) st |= DLM_ST_NOT_PD;
          const double e = yt - f, iq = 1.0 / Q;
#pragma unroll
          for (int i = 0; i < D; ++i) {
            m[i] = fma(RF[i], e * iq, av[i]);
#pragma unroll
            for (int k = 0; k <= i; ++k) { const double c_ = fma(-RF[i] * iq, RF[k], R[i][k]); C[i][k] = c_; C[k][i] = c_; }
          }
          if (a.loglik) ll -= 0.5 * (1.8378770664093453 + log(Q) + e * e * iq);
        } else {
#pragma unroll
          for (int i = 0; i < D; ++i) {
            m[i] = av[i];
#pragma unroll
            for (int k = 0; k < D; ++k) C[i][k] = R[i][k];
          }
        }
        if (out) {
          if (OT == 1 || !whole) store(out, t + 1, m, C);
          else {
            double r[REC];
#pragma unroll
            for (int i = 0; i < D; ++i) r[i] = m[i];
#pragma unroll
            for (int jj = 0; jj < D; ++jj)
#pragma unroll
              for (int i = 0; i < D; ++i) r[D + i + jj * D] = C[i][jj];
#pragma unroll
            for (int q = 0; q < REC / 2; ++q) ob[j % OT][q] = dbl2{r[2 * q], r[2 * q + 1]};
            if (j % OT == OT - 1) {
              dbl2* o = (dbl2*)(out + (size_t)(t + 2 - OT) * REC);   // records t + 2 - OT .. t + 1
#pragma unroll
              for (int k = 0; k < OT; ++k)
#pragma unroll
                for (int q = 0; q < REC / 2; ++q) o[k * (REC / 2) + q] = ob[k][q];
            }
          }
        }
      }
    }
  }
  if (a.loglik) a.loglik[n] = ll;
  bool bad = false;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    bad |= !isfinite(m[i]);
#pragma unroll
    for (int k = 0; k < D; ++k) bad |= !isfinite(C[i][k]);
  }
  if (bad) st |= DLM_ST_NONFINITE;
  if (a.status && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// backward pass
// ---------------------------------------------------------------------------------------
template <int D, bool IRR>
__global__ __launch_bounds__(64) void k_smoother_lane(KArgs a) {
  constexpr int REC = D + D * D;
  constexpr int RT = D <= 2 ? 8 : (D == 3 ? 4 : (D == 4 ? 2 : 1));     // records requested per tile (128 - 384 bytes per lane)
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= a.N) return;
  const int T = a.T;
  const double* W0 = a.W + (size_t)n * a.w_stride;
  double W[D][D];
  load_w<D>(W0, W);
  const dbl2* in = (const dbl2*)(a.filt_in + (size_t)n * (T + 1) * REC);
  double* outp = a.smooth + (size_t)n * (T + 1) * REC;
  int st = 0;
  double s[D], S[D][D];
  double Gc[D * D];
#pragma unroll
  for (int i = 0; i < D * D; ++i) Gc[i] = a.G[i];
  auto unpack = [&](const dbl2* r, double (&mm)[D], double (&CC)[D][D]) {
    double v[REC];
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) { v[2 * q] = r[q].x; v[2 * q + 1] = r[q].y; }
#pragma unroll
    for (int i = 0; i < D; ++i) mm[i] = v[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) CC[i][j] = v[D + i + j * D];
  };
  auto store = [&](int t, const double (&mm)[D], const double (&CC)[D][D]) {
    double r[REC];
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = mm[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) r[D + i + j * D] = CC[i][j];
    dbl2* o = (dbl2*)(outp + (size_t)t * REC);
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) o[q] = dbl2{r[2 * q], r[2 * q + 1]};
  };
  {   // record T: the smoothed moments are the filtered ones
    dbl2 r[REC / 2];
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) r[q] = in[(size_t)T * (REC / 2) + q];
    unpack(r, s, S);
    store(T, s, S);
  }
  // tiles of records t0 .. t0 + RT - 1 walked downwards from T - 1; the top tile may be ragged
  dbl2 buf[RT][REC / 2];
  int t0 = T > 0 ? ((T - 1) / RT) * RT : -1;
  auto request = [&](int t0) {
#pragma unroll
    for (int j = 0; j < RT; ++j) {
      const int t = t0 + j < T ? t0 + j : T - 1;
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) buf[j][q] = in[(size_t)t * (REC / 2) + q];
    }
  };
  if (T > 0) request(t0);
  for (; t0 >= 0; t0 -= RT) {
    dbl2 cur[RT][REC / 2];
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) cur[j][q] = buf[j][q];
    if (t0 > 0) request(t0 - RT);
#pragma unroll
    for (int j = RT - 1; j >= 0; --j) {
      const int t = t0 + j;
      if (t < T) {
        double m[D], C[D][D];
        unpack(cur[j], m, C);
        const double dt = (IRR && a.dt) ? a.dt[t] : 1.0;
        const double* Gt = IRR ? a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * (D * D) : Gc;
        if (IRR && a.w_tstride) load_w<D>(W0 + (size_t)t * a.w_tstride, W);
        double a1[D], R1[D][D];
        advance<D>(Gt, dt, W, m, C, a1, R1);
        // Cholesky of R+ (lower), B = C G^T, J = B R+^-1 row by row
        double L[D][D];
        bool bad = false;
#pragma unroll
        for (int jj = 0; jj < D; ++jj) {
          double sd = R1[jj][jj];
#pragma unroll
          for (int k = 0; k < jj; ++k) sd = fma(-L[jj][k], L[jj][k], sd);
          if (!(sd > 0.0)) { bad = true; sd = 1e-300; }
          const double ljj = sqrt(sd), inv = 1.0 / ljj;
          L[jj][jj] = inv;   // the RECIPROCAL of the diagonal
#pragma unroll
          for (int i = jj + 1; i < D; ++i) {
            double v = R1[i][jj];
#pragma unroll
            for (int k = 0; k < jj; ++k) v = fma(-L[i][k], L[jj][k], v);
            L[i][jj] = v * inv;
          }
        }
        if (bad) st |= DLM_ST_NOT_PD;
        double J[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double x[D];
#pragma unroll
          for (int jj = 0; jj < D; ++jj) {   // B[i][jj] = sum_k C[i][k] G[jj][k]  (the table entry g(dt), also for dt == 0: Smoothing.scala:41)
            double b = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) b = fma(C[i][k], Gt[jj + k * D], b);
            x[jj] = b;
          }
#pragma unroll
          for (int r = 0; r < D; ++r) {      // L z = b
            double v = x[r];
#pragma unroll
            for (int k = 0; k < r; ++k) v = fma(-L[r][k], x[k], v);
            x[r] = v * L[r][r];
          }
#pragma unroll
          for (int r = D - 1; r >= 0; --r) { // L^T x = z
            double v = x[r];
#pragma unroll
            for (int k = r + 1; k < D; ++k) v = fma(-L[k][r], x[k], v);
            x[r] = v * L[r][r];
          }
#pragma unroll
          for (int jj = 0; jj < D; ++jj) J[i][jj] = x[jj];
        }
        double sn[D], Sn[D][D], X[D][D], JX[D][D];
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int k = 0; k < D; ++k) X[i][k] = R1[i][k] - S[i][k];
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double v = m[i];
#pragma unroll
          for (int k = 0; k < D; ++k) v = fma(J[i][k], s[k] - a1[k], v);
          sn[i] = v;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            double w = 0.0;
#pragma unroll
            for (int q = 0; q < D; ++q) w = fma(J[i][q], X[q][k], w);
            JX[i][k] = w;
          }
        }
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int k = 0; k <= i; ++k) {
            double w = C[i][k];
#pragma unroll
            for (int q = 0; q < D; ++q) w = fma(-JX[i][q], J[k][q], w);
            Sn[i][k] = w; Sn[k][i] = w;
          }
#pragma unroll
        for (int i = 0; i < D; ++i) {
          s[i] = sn[i];
#pragma unroll
          for (int k = 0; k < D; ++k) S[i][k] = Sn[i][k];
        }
        if (D > 2 || t0 + RT > T) store(t, s, S);
        else {
          double r[REC];
#pragma unroll
          for (int i = 0; i < D; ++i) r[i] = s[i];
#pragma unroll
          for (int jj = 0; jj < D; ++jj)
#pragma unroll
            for (int i = 0; i < D; ++i) r[D + i + jj * D] = S[i][jj];
#pragma unroll
          for (int q = 0; q < REC / 2; ++q) cur[j][q] = dbl2{r[2 * q], r[2 * q + 1]};   // cur[j] has been consumed: reuse it as the output tile
          if (j == 0) {
            dbl2* o = (dbl2*)(outp + (size_t)t0 * REC);
#pragma unroll
            for (int k = 0; k < RT; ++k)
#pragma unroll
              for (int q = 0; q < REC / 2; ++q) o[k * (REC / 2) + q] = cur[k][q];
          }
        }
      }
    }
  }
  bool nf = false;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    nf |= !isfinite(s[i]);
#pragma unroll
    for (int k = 0; k < D; ++k) nf |= !isfinite(S[i][k]);
  }
  if (nf) st |= DLM_ST_NONFINITE;
  if (a.status && st) atomicOr(&a.status[n], st);
}


// ---------------------------------------------------------------------------------------
// FFBS by the Durbin-Koopman simulation smoother, one lane per series (DLM_OPT_FFBS_SIMSMOOTH; the d <= 5 counterpart
// of k_filter_sp16<SIM> / k_simsmooth_sp16): the forward kernel simulates (x+, y+) from the model, filters
// y* = y - y+ from a zero prior mean and stores the filtered records and x+; the backward kernel runs the MEAN
// recursion of the smoother on y*,  s*_t = m*_t + J_t (s*_{t+1} - a*_{t+1}),  theta_t = s*_t + x+_t, and accumulates
// the Gibbs statistics (Gibbs.scala:23-78, GibbsWishart.scala:16-35).  Normals: (seed, series, record t, i), i < d
// state noise (record 0: the initial state), i = d observation noise, or a.z [N][T+1][d+1].
// ---------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ bool chol_small(const double (&A)[D][D], double (&L)[D][D]) {   // lower factor, true: not positive definite
  bool bad = false;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double sd = A[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) sd = fma(-L[j][k], L[j][k], sd);
    if (!(sd > 0.0)) { bad = true; sd = 0.0; }
    const double ljj = sqrt(sd), inv = ljj > 0.0 ? 1.0 / ljj : 0.0;
    L[j][j] = ljj;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double v = A[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) v = fma(-L[i][k], L[j][k], v);
      L[i][j] = v * inv;
    }
#pragma unroll
    for (int i = 0; i < j; ++i) L[i][j] = 0.0;
  }
  return bad;
}

template <int D>
__global__ __launch_bounds__(64) void k_simfilter_lane(KArgs a, double* __restrict__ xplus) {
  constexpr int REC = D + D * D;
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= a.N) return;
  const int T = a.T;
  const double V = a.V[(size_t)n * a.v_stride], sqV = sqrt(V);
  double W[D][D], Lw[D][D], m[D], C[D][D], x[D];
  load_w<D>(a.W + (size_t)n * a.w_stride, W);
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;
  if (chol_small<D>(W, Lw)) st |= DLM_ST_NOT_PD;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * (D + 1) : nullptr;
  {
    const double* m0 = a.m0 + (size_t)n * a.m0_stride;
    const double* C0 = a.C0 + (size_t)n * a.c0_stride;
    double L0[D][D], z0[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      z0[i] = zin ? zin[i] : philox_normal(a.seed, series, 0u, (unsigned)i);
#pragma unroll
      for (int j = 0; j < D; ++j) C[i][j] = C0[i + j * D];
    }
    if (chol_small<D>(C, L0)) st |= DLM_ST_NOT_PD;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double v = m0[i];
#pragma unroll
      for (int k = 0; k <= i; ++k) v = fma(L0[i][k], z0[k], v);
      x[i] = v;
      m[i] = 0.0;                 // y* is filtered from a zero prior mean
    }
  }
  const double* y = a.y + (size_t)n * T;
  double* out = a.filt + (size_t)n * (T + 1) * REC;
  double* xo = xplus + (size_t)n * (T + 1) * D;
  auto store = [&](int t) {
    double r[REC];
#pragma unroll
    for (int i = 0; i < D; ++i) r[i] = m[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) r[D + i + j * D] = C[i][j];
    dbl2* o = (dbl2*)(out + (size_t)t * REC);
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) o[q] = dbl2{r[2 * q], r[2 * q + 1]};
#pragma unroll
    for (int i = 0; i < D; ++i) xo[(size_t)t * D + i] = x[i];
  };
  store(0);
  // tiles of ST steps: the tile's observations are requested one tile ahead, its records and x+ leave together (whole
  // lines for d <= 2; the step body with its normal generator is unrolled ST times, hence the small tiles)
  constexpr int ST = D == 1 ? 8 : (D == 2 ? 4 : 1);
  const int Tl = T - 1;
  double yb[ST];
  auto request = [&](int t0) {
#pragma unroll
    for (int j = 0; j < ST; ++j) { const int t = t0 + j < Tl ? t0 + j : Tl; yb[j] = y[t]; }
  };
  if (T > 0) request(0);
  for (int t0 = 0; t0 < T; t0 += ST) {
  double yc[ST];
#pragma unroll
  for (int j = 0; j < ST; ++j) yc[j] = yb[j];
  if (t0 + ST < T) request(t0 + ST);
  const bool whole = ST > 1 && t0 + ST <= T;
  dbl2 ob[ST][REC / 2];
  double xb[ST][D];
#pragma unroll
  for (int j = 0; j < ST; ++j) {
    const int t = t0 + j;
    if (t >= T) break;
    const double ycur = yc[j];
    const double dt = a.dt ? a.dt[t] : 1.0;
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * (D * D);
    const double* Ft = a.F + (size_t)t * a.f_stride;
    double z[D + 2];   // D + 1 normals, drawn as Box-Muller pairs
    if (zin) {
#pragma unroll
      for (int i = 0; i <= D; ++i) z[i] = zin[(size_t)(t + 1) * (D + 1) + i];
    } else {
#pragma unroll
      for (int q = 0; q < (D + 2) / 2; ++q) philox_normal2(a.seed, series, (unsigned)(t + 1), (unsigned)q, z[2 * q], z[2 * q + 1]);
    }
    double av[D], R[D][D];
    advance<D>(Gt, dt, W, m, C, av, R);
    if (dt != 0.0) {              // x+ = G x+ + sqrt(dt) chol(W) z ; a zero increment leaves x+ as it is
      const double sdt = sqrt(dt);
      double xn[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double v = 0.0, w_ = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(Gt[i + k * D], x[k], v);
#pragma unroll
        for (int k = 0; k <= i; ++k) w_ = fma(Lw[i][k], z[k], w_);
        xn[i] = fma(w_, sdt, v);
      }
#pragma unroll
      for (int i = 0; i < D; ++i) x[i] = xn[i];
    }
    double F[D], RF[D], f = 0.0, Q = V, yp = sqV * z[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { F[i] = Ft[i]; f = fma(F[i], av[i], f); yp = fma(F[i], x[i], yp); }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double s_ = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) s_ = fma(R[i][k], F[k], s_);
      RF[i] = s_;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) Q = fma(F[i], RF[i], Q);
    const double ys = ycur - yp;              // NaN (missing) stays NaN
    if (ys == ys) {
      const double e = ys - f, iq = 1.0 / Q;
#pragma unroll
      for (int i = 0; i < D; ++i) {
        m[i] = fma(RF[i], e * iq, av[i]);
#pragma unroll
        for (int k = 0; k <= i; ++k) { const double c_ = fma(-RF[i] * iq, RF[k], R[i][k]); C[i][k] = c_; C[k][i] = c_; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        m[i] = av[i];
#pragma unroll
        for (int k = 0; k < D; ++k) C[i][k] = R[i][k];
      }
    }
    if (!whole) store(t + 1);
    else {
      double r[REC];
#pragma unroll
      for (int i = 0; i < D; ++i) { r[i] = m[i]; xb[j][i] = x[i]; }
#pragma unroll
      for (int jj = 0; jj < D; ++jj)
#pragma unroll
        for (int i = 0; i < D; ++i) r[D + i + jj * D] = C[i][jj];
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) ob[j][q] = dbl2{r[2 * q], r[2 * q + 1]};
    }
  }
  if (whole) {
    dbl2* o = (dbl2*)(out + (size_t)(t0 + 1) * REC);
#pragma unroll
    for (int j = 0; j < ST; ++j)
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) o[j * (REC / 2) + q] = ob[j][q];
#pragma unroll
    for (int j = 0; j < ST; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) xo[(size_t)(t0 + 1 + j) * D + i] = xb[j][i];
  }
  }
  if (a.status && st) atomicOr(&a.status[n], st);
}

template <int D>
__global__ __launch_bounds__(64) void k_simsmooth_lane(KArgs a, const double* __restrict__ xplus) {
  constexpr int REC = D + D * D;
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= a.N) return;
  const int T = a.T;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;
  double W[D][D];
  load_w<D>(a.W + (size_t)n * a.w_stride, W);
  const dbl2* in = (const dbl2*)(a.filt_in + (size_t)n * (T + 1) * REC);
  const double* xp = xplus + (size_t)n * (T + 1) * D;
  const double* y = a.y ? a.y + (size_t)n * T : nullptr;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * D : nullptr;
  int st = 0;
  double s[D], thn[D], ssd[D], OUT[D][D], ssy = 0.0, nob = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    ssd[i] = 0.0; thn[i] = 0.0; s[i] = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) OUT[i][k] = 0.0;
  }
  // tiles of BT records walked downwards from record T: the tile below is requested before this one is computed, the
  // tile's draws leave together (whole lines for d <= 2)
  constexpr int BT = D <= 2 ? 8 : (D == 3 ? 4 : 2);
  dbl2 nrec[BT][REC / 2];
  double nx[BT][D], ny[BT];
  auto request = [&](int t0) {
#pragma unroll
    for (int j = 0; j < BT; ++j) {
      const int t = t0 + j < T ? t0 + j : T;
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) nrec[j][q] = in[(size_t)t * (REC / 2) + q];
#pragma unroll
      for (int i = 0; i < D; ++i) nx[j][i] = xp[(size_t)t * D + i];
      ny[j] = (y && t > 0) ? y[t - 1] : 0.0;
    }
  };
  int t0 = (T / BT) * BT;   // the tile that holds record T
  request(t0);
  for (; t0 >= 0; t0 -= BT) {
  dbl2 crec[BT][REC / 2];
  double cx[BT][D], cy[BT], thb[BT][D];
#pragma unroll
  for (int j = 0; j < BT; ++j) {
    cy[j] = ny[j];
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) crec[j][q] = nrec[j][q];
#pragma unroll
    for (int i = 0; i < D; ++i) { cx[j][i] = nx[j][i]; thb[j][i] = 0.0; }
  }
  if (t0 > 0) request(t0 - BT);
#pragma unroll
  for (int j = BT - 1; j >= 0; --j) {
    const int t = t0 + j;
    if (t > T) continue;
    double m[D], C[D][D], xcur[D];
    const double ycur = cy[j];
    {
      double v[REC];
#pragma unroll
      for (int q = 0; q < REC / 2; ++q) { v[2 * q] = crec[j][q].x; v[2 * q + 1] = crec[j][q].y; }
#pragma unroll
      for (int i = 0; i < D; ++i) { m[i] = v[i]; xcur[i] = cx[j][i]; }
#pragma unroll
      for (int jj = 0; jj < D; ++jj)
#pragma unroll
        for (int i = 0; i < D; ++i) C[i][jj] = v[D + i + jj * D];
    }
    const double dt = (t < T && a.dt) ? a.dt[t] : 1.0;                                     // the step t -> t + 1
    const double* Gt = a.G + (size_t)((t < T && a.g_index) ? a.g_index[t] : 0) * (D * D);
    if (t == T) {
#pragma unroll
      for (int i = 0; i < D; ++i) s[i] = m[i];
    } else if (dt != 0.0) {       // a zero increment was an identity advance for x+ and for the filter: s*_t = s*_{t+1}
      double a1[D], R1[D][D], L[D][D];
      advance<D>(Gt, dt, W, m, C, a1, R1);
      if (chol_small<D>(R1, L)) st |= DLM_ST_NOT_PD;
      // s = m + C G^T R+^-1 (s+ - a+): solve R+ u = s+ - a+, then s = m + C G^T u
      double u[D];
#pragma unroll
      for (int r = 0; r < D; ++r) {
        double v = s[r] - a1[r];
#pragma unroll
        for (int k = 0; k < r; ++k) v = fma(-L[r][k], u[k], v);
        u[r] = v / L[r][r];
      }
#pragma unroll
      for (int r = D - 1; r >= 0; --r) {
        double v = u[r];
#pragma unroll
        for (int k = r + 1; k < D; ++k) v = fma(-L[k][r], u[k], v);
        u[r] = v / L[r][r];
      }
      double gu[D];
#pragma unroll
      for (int k = 0; k < D; ++k) {     // (G^T u)[k]
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < D; ++j) v = fma(Gt[j + k * D], u[j], v);
        gu[k] = v;
      }
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double v = m[i];
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(C[i][k], gu[k], v);
        s[i] = v;
      }
    }
    double th[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { th[i] = s[i] + xcur[i]; thb[j][i] = th[i]; }
    if (a.stats) {
      if (t < T) {   // system residual (theta_{t+1} - G_{t+1} theta_t) / sqrt(dt)
        const double dts = (dt == 0.0) ? 1.0 : dt, isd = 1.0 / sqrt(dts);
        double df[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
          double g_ = th[i];
          if (dt != 0.0) {
            g_ = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) g_ = fma(Gt[i + k * D], th[k], g_);
          }
          const double r_ = thn[i] - g_;
          ssd[i] = fma(r_, r_ / dts, ssd[i]);
          df[i] = r_ * isd;
        }
        if (outer) {
#pragma unroll
          for (int i = 0; i < D; ++i)
#pragma unroll
            for (int k = 0; k < D; ++k) OUT[i][k] = fma(df[i], df[k], OUT[i][k]);
        }
      }
      if (t > 0 && y) {   // observation residual of theta_t against the original y_t
        const double yv = ycur;
        if (yv == yv) {
          const double* Ft = a.F + (size_t)(t - 1) * a.f_stride;
          double f = 0.0;
#pragma unroll
          for (int i = 0; i < D; ++i) f = fma(Ft[i], th[i], f);
          ssy = fma(yv - f, yv - f, ssy); nob += 1.0;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) thn[i] = th[i];
  }
  if (thout) {
#pragma unroll
    for (int j = 0; j < BT; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) if (t0 + j <= T) thout[(size_t)(t0 + j) * D + i] = thb[j][i];
  }
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < D; ++i) bad |= !isfinite(thn[i]);
  if (bad) st |= DLM_ST_NONFINITE;
  if (a.stats) {
    const int L_ = stats_len(D, 1, a.flags);
    double* so = a.stats + (size_t)n * L_;
    so[0] = ssy; so[1] = nob; so[L_ - 1] = (double)T;
    if (outer) {
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int k = 0; k < D; ++k) so[2 + i + k * D] = OUT[i][k];
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) so[2 + i] = ssd[i];
    }
  }
  if (a.status && st) atomicOr(&a.status[n], st);
}


// ---------------------------------------------------------------------------------------
// The reference's backward sampler, literally (Smoothing.sampleDlm / backSampleStep, Smoothing.scala:74-122), one lane per
// series: theta_T ~ N(m_T, C_T);  J = C G^T R+^-1,  h = m + J (theta+ - a+),  H = (I - J G) C (I - J G)^T + dt J W J^T
// (symmetrised, :95),  theta_t = h + chol(H) z_t  (the engine's canonical factor), with the optional
// conditional-moment records and the Gibbs statistics of k_sampler_generic.  Normals: a.z [N][T+1][d] or Philox
// (seed, series, record t, i).
// ---------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void k_sampler_lane(KArgs a) {
  constexpr int REC = D + D * D;
  const int n = blockIdx.x * 64 + threadIdx.x;
  if (n >= a.N) return;
  const int T = a.T;
  const bool outer = (a.flags & DLM_OPT_STATS_OUTER) != 0;
  const double* W0 = a.W + (size_t)n * a.w_stride;
  double W[D][D];
  load_w<D>(W0, W);
  const dbl2* in = (const dbl2*)(a.filt_in + (size_t)n * (T + 1) * REC);
  const double* y = a.y ? a.y + (size_t)n * T : nullptr;
  const double* zin = a.z ? a.z + (size_t)n * (T + 1) * D : nullptr;
  double* thout = a.theta ? a.theta + (size_t)n * (T + 1) * D : nullptr;
  double* cond = a.cond ? a.cond + (size_t)n * (T + 1) * REC : nullptr;
  const unsigned long long series = a.series_offset + (unsigned long long)n;
  int st = 0;
  double th[D], ssd[D], OUT[D][D], ssy = 0.0, nob = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    ssd[i] = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) OUT[i][k] = 0.0;
  }
  auto load_rec = [&](int t, double (&m)[D], double (&C)[D][D]) {
    double v[REC];
#pragma unroll
    for (int q = 0; q < REC / 2; ++q) { const dbl2 r = in[(size_t)t * (REC / 2) + q]; v[2 * q] = r.x; v[2 * q + 1] = r.y; }
#pragma unroll
    for (int i = 0; i < D; ++i) m[i] = v[i];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int i = 0; i < D; ++i) C[i][j] = v[D + i + j * D];
  };
  auto normals = [&](int t, double (&z)[D + 1]) {
    if (zin) {
#pragma unroll
      for (int i = 0; i < D; ++i) z[i] = zin[(size_t)t * D + i];
    } else {
#pragma unroll
      for (int q = 0; q < (D + 1) / 2; ++q) philox_normal2(a.seed, series, (unsigned)t, (unsigned)q, z[2 * q], z[2 * q + 1]);
    }
  };
  auto emit = [&](int t, const double (&h)[D], const double (&H)[D][D], const double (&z)[D + 1], double (&out)[D]) {
    if (cond) {
      double* c = cond + (size_t)t * REC;
#pragma unroll
      for (int i = 0; i < D; ++i) c[i] = h[i];
#pragma unroll
      for (int j = 0; j < D; ++j)
#pragma unroll
        for (int i = 0; i < D; ++i) c[D + i + j * D] = H[i][j];
    }
    double L[D][D];
    if (chol_small<D>(H, L)) st |= DLM_ST_NOT_PD;
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double v = h[i];
#pragma unroll
      for (int k = 0; k <= i; ++k) v = fma(L[i][k], z[k], v);
      out[i] = v;
      if (thout) thout[(size_t)t * D + i] = v;
    }
  };
  {
    double m[D], C[D][D], z[D + 1];
    load_rec(T, m, C);
    normals(T, z);
    emit(T, m, C, z, th);
  }
  for (int t = T - 1; t >= 0; --t) {
    const double dt = a.dt ? a.dt[t] : 1.0;
    const double* Gt = a.G + (size_t)(a.g_index ? a.g_index[t] : 0) * (D * D);
    const double* Ft = a.F + (size_t)t * a.f_stride;
    if (a.stats && y) {   // observation residual of theta_{t+1} (Gibbs.scala:29-39)
      const double yv = y[t];
      if (yv == yv) {
        double f = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) f = fma(Ft[i], th[i], f);
        ssy += (yv - f) * (yv - f); nob += 1.0;
      }
    }
    double m[D], C[D][D], z[D + 1];
    load_rec(t, m, C);
    normals(t, z);
    if (a.w_tstride) load_w<D>(W0 + (size_t)t * a.w_tstride, W);
    double G[D][D], a1[D], R1[D][D], Lr[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int k = 0; k < D; ++k) G[i][k] = Gt[i + k * D];
    advance<D>(Gt, dt, W, m, C, a1, R1);
    if (chol_small<D>(R1, Lr)) st |= DLM_ST_NOT_PD;
    // J^T = R+^-1 (G C): column i of J^T = row i of J solves R+ x = (G C)[:, i]
    double J[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double x[D];
#pragma unroll
      for (int r = 0; r < D; ++r) {
        double b = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) b = fma(G[r][k], C[k][i], b);
        x[r] = b;
      }
#pragma unroll
      for (int r = 0; r < D; ++r) {
        double v = x[r];
#pragma unroll
        for (int k = 0; k < r; ++k) v = fma(-Lr[r][k], x[k], v);
        x[r] = v / Lr[r][r];
      }
#pragma unroll
      for (int r = D - 1; r >= 0; --r) {
        double v = x[r];
#pragma unroll
        for (int k = r + 1; k < D; ++k) v = fma(-Lr[k][r], x[k], v);
        x[r] = v / Lr[r][r];
      }
#pragma unroll
      for (int r = 0; r < D; ++r) J[i][r] = x[r];
    }
    double h[D], Dm[D][D], DC[D][D], JW[D][D], H[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double v = m[i];
#pragma unroll
      for (int k = 0; k < D; ++k) v = fma(J[i][k], th[k] - a1[k], v);
      h[i] = v;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
        for (int l = 0; l < D; ++l) acc = fma(-J[i][l], G[l][j], acc);
        Dm[i][j] = acc;
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double v = 0.0, w_ = 0.0;
#pragma unroll
        for (int l = 0; l < D; ++l) { v = fma(Dm[i][l], C[l][j], v); w_ = fma(J[i][l], W[l][j], w_); }
        DC[i][j] = v; JW[i][j] = w_;
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        double v = 0.0, w_ = 0.0;
#pragma unroll
        for (int l = 0; l < D; ++l) { v = fma(DC[i][l], Dm[j][l], v); w_ = fma(JW[i][l], J[j][l], w_); }
        H[i][j] = fma(w_, dt, v);
      }
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = 0; j < i; ++j) { const double v = (H[i][j] + H[j][i]) / 2.0; H[i][j] = v; H[j][i] = v; }
    double tn[D];
    emit(t, h, H, z, tn);
    if (a.stats) {   // system residual theta_{t+1} - G theta_t (always the table entry)
      const double dts = (dt == 0.0) ? 1.0 : dt;
      double df[D];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        double v = th[i];
#pragma unroll
        for (int k = 0; k < D; ++k) v = fma(-G[i][k], tn[k], v);
        df[i] = v;
        ssd[i] += v * v / dts;
      }
      if (outer) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int k = 0; k < D; ++k) OUT[i][k] += df[i] * df[k] / dts;
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) th[i] = tn[i];
  }
  bool bad = false;
#pragma unroll
  for (int i = 0; i < D; ++i) bad |= !isfinite(th[i]);
  if (bad) st |= DLM_ST_NONFINITE;
  if (a.stats) {
    const int L_ = stats_len(D, 1, a.flags);
    double* so = a.stats + (size_t)n * L_;
    so[0] = ssy; so[1] = nob; so[L_ - 1] = (double)T;
    if (outer) {
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int k = 0; k < D; ++k) so[2 + i + k * D] = OUT[i][k];
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) so[2 + i] = ssd[i];
    }
  }
  if (a.status && st) atomicOr(&a.status[n], st);
}

}  // namespace lane

constexpr int LANE_MIN_N = 1;   // a lane's step is a few hundred cycles, a wavefront-per-series step a few thousand: faster at every batch size (tools/lane_small_n.py)
constexpr int LANE_MAX_D = 5;   // at d = 6 the backward pass no longer fits the register file (78 spilled registers)

// d <= 5, p = 1; per-step V_t / W_t streams are read per lane, any time grid, time-varying F.
bool lane_supported(const KArgs& a) {
  return a.d >= 1 && a.d <= LANE_MAX_D && a.p == 1 && a.N >= LANE_MIN_N;
}

hipError_t launch_lane_filter(const KArgs& a, hipStream_t s) {
  const dim3 grid((a.N + 63) / 64), block(64);
  const bool irr = a.g_index || a.dt || a.f_stride || a.v_tstride || a.w_tstride;
  switch (a.d) {
    case 1: if (irr) hipLaunchKernelGGL((lane::k_filter_lane<1, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_filter_lane<1, false>), grid, block, 0, s, a); break;
    case 2: if (irr) hipLaunchKernelGGL((lane::k_filter_lane<2, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_filter_lane<2, false>), grid, block, 0, s, a); break;
    case 3: if (irr) hipLaunchKernelGGL((lane::k_filter_lane<3, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_filter_lane<3, false>), grid, block, 0, s, a); break;
    case 4: if (irr) hipLaunchKernelGGL((lane::k_filter_lane<4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_filter_lane<4, false>), grid, block, 0, s, a); break;
    case 5: if (irr) hipLaunchKernelGGL((lane::k_filter_lane<5, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_filter_lane<5, false>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_lane_smoother(const KArgs& a, hipStream_t s) {
  const dim3 grid((a.N + 63) / 64), block(64);
  const bool irr = a.g_index || a.dt || a.f_stride || a.v_tstride || a.w_tstride;
  switch (a.d) {
    case 1: if (irr) hipLaunchKernelGGL((lane::k_smoother_lane<1, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_smoother_lane<1, false>), grid, block, 0, s, a); break;
    case 2: if (irr) hipLaunchKernelGGL((lane::k_smoother_lane<2, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_smoother_lane<2, false>), grid, block, 0, s, a); break;
    case 3: if (irr) hipLaunchKernelGGL((lane::k_smoother_lane<3, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_smoother_lane<3, false>), grid, block, 0, s, a); break;
    case 4: if (irr) hipLaunchKernelGGL((lane::k_smoother_lane<4, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_smoother_lane<4, false>), grid, block, 0, s, a); break;
    case 5: if (irr) hipLaunchKernelGGL((lane::k_smoother_lane<5, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((lane::k_smoother_lane<5, false>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}


hipError_t launch_lane_sampler(const KArgs& a, hipStream_t s) {
  const dim3 grid((a.N + 63) / 64), block(64);
  switch (a.d) {
    case 1: hipLaunchKernelGGL(lane::k_sampler_lane<1>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(lane::k_sampler_lane<2>, grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL(lane::k_sampler_lane<3>, grid, block, 0, s, a); break;
    case 4: hipLaunchKernelGGL(lane::k_sampler_lane<4>, grid, block, 0, s, a); break;
    case 5: hipLaunchKernelGGL(lane::k_sampler_lane<5>, grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// forward SIM pass (filt records + x+) followed by the mean-only backward pass; xplus [N][T+1][d] is engine workspace
hipError_t launch_lane_simsmooth(const KArgs& a, double* xplus, hipStream_t s) {
  const dim3 grid((a.N + 63) / 64), block(64);
  KArgs b = a;
  b.filt_in = a.filt;
  switch (a.d) {
    case 1: hipLaunchKernelGGL(lane::k_simfilter_lane<1>, grid, block, 0, s, a, xplus); hipLaunchKernelGGL(lane::k_simsmooth_lane<1>, grid, block, 0, s, b, (const double*)xplus); break;
    case 2: hipLaunchKernelGGL(lane::k_simfilter_lane<2>, grid, block, 0, s, a, xplus); hipLaunchKernelGGL(lane::k_simsmooth_lane<2>, grid, block, 0, s, b, (const double*)xplus); break;
    case 3: hipLaunchKernelGGL(lane::k_simfilter_lane<3>, grid, block, 0, s, a, xplus); hipLaunchKernelGGL(lane::k_simsmooth_lane<3>, grid, block, 0, s, b, (const double*)xplus); break;
    case 4: hipLaunchKernelGGL(lane::k_simfilter_lane<4>, grid, block, 0, s, a, xplus); hipLaunchKernelGGL(lane::k_simsmooth_lane<4>, grid, block, 0, s, b, (const double*)xplus); break;
    case 5: hipLaunchKernelGGL(lane::k_simfilter_lane<5>, grid, block, 0, s, a, xplus); hipLaunchKernelGGL(lane::k_simsmooth_lane<5>, grid, block, 0, s, b, (const double*)xplus); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace dlm
