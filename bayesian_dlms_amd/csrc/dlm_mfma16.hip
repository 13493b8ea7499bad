// Fast path: d <= 15, p == 1, one model matrix G, regular time grid (dt == 1).
//
// One wavefront per series.  Every d x d matrix lives in registers in the accumulator
// layout of v_mfma_f64_16x16x4_f64 ("std layout", d padded to 16 with zeros):
//     lane l = 16*g + c (g = 0..3, c = 0..15), register r = 0..3  <->  X[row 4r+g][col c]
// That layout is at once the C/D layout and the B-operand layout of the instruction, and
// read as the A operand it supplies X^T.  So with x = std(X), y = std(Y)
//     mmT(x, y) = X^T * Y          (4 MFMAs, no data movement, no LDS)
// and chains such as G (C G^T) or C (P C) run register-to-register.  Vectors ride along as
// column 15 of a B operand (free, since d <= 15): G [C G^T | m] = [G C G^T | G m].
//
// Forward pass  = Kalman filter (KalmanFilter.scala:64-107, :273-286, :311-321) with the
//   p = 1 Joseph form expanded algebraically:
//   (I-KF^T) R (I-KF^T)^T + K V K^T  =  R - K (RF)^T - (RF) K^T + Q K K^T.
// Backward pass = the smoothing distribution of Smoothing.backwardsSmoother
//   (Smoothing.scala:31-64) computed in information form (de Jong 1989 / Durbin-Koopman):
//   s_t = m_t + C_t q_t,  S_t = C_t - C_t P_t C_t,  with
//   q_{t-1} = G^T [ q_t + F (e_t/Q_t - K_t^T q_t) ],
//   P_{t-1} = G^T [ P_t + F F^T (1/Q_t + K^T P K) - F (P K)^T - (P K) F^T ] G,   K_t = C_t F / V.
//   This is algebraically identical to the RTS recursion s = m + J (s+ - a+),
//   S = C - J (R+ - S+) J^T with J = C G^T R+^-1 (substitute R+^-1 (R+ - S+) R+^-1 = M),
//   needs no d x d solve, and never re-reads or recomputes R+.  It requires V > 0.
//   The innovations (e_t/Q_t, 1/Q_t) come from the forward pass through a side buffer.
#include "dlm_internal.h"
#include "../../include/dlm_engine.h"

namespace dlm {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 mmT(const d4& x, const d4& y) {  // X^T * Y
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[0], y[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[1], y[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[2], y[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[3], y[3], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ double sum_over_g(double v) {  // sum lanes c, c+16, c+32, c+48
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ double uniform_from_lane(double v, int src) {
  const int lo = __builtin_amdgcn_readlane((int)__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane((int)__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// The fast kernels address a series' records through 32-bit buffer offsets: one series' record stream must stay
// below 2 GiB (T < 1.4 million steps at d = 13), longer series take the generic kernels.
bool fast_shape(const KArgs& a) {
  return a.d <= 15 && a.p == 1 && ((size_t)a.T + 1) * (size_t)(a.d + a.d * a.d) * 8 < ((size_t)1 << 31);
}
bool mfma16_supported(const KArgs& a) {
  return fast_shape(a) && a.f_stride == 0 && a.g_index == nullptr && a.dt == nullptr && a.v_tstride == 0 && a.w_tstride == 0;
}

// ---------------------------------------------------------------------------------------
// forward pass
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_filter_mfma16(KArgs a, double* __restrict__ side) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
  if (n >= a.N) return;
  const int d = a.d, T = a.T, rec = d + d * d;
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d, col15 = (c == 15);

  const double* W = a.W + (size_t)n * a.w_stride;
  const double* C0 = a.C0 + (size_t)n * a.c0_stride;
  const double* m0 = a.m0 + (size_t)n * a.m0_stride;
  const double V = a.V[(size_t)n * a.v_stride];
  const double* y = a.y + (size_t)n * T;
  double* out = a.filt ? a.filt + (size_t)n * (T + 1) * rec : nullptr;   // null: likelihood only, nothing stored
  double* fq = a.fq ? a.fq + (size_t)n * (T + 1) * 2 : nullptr;
  double* sd = side ? side + (size_t)n * (T + 1) * 2 : nullptr;
  double ll = 0.0;   // sum_t log N(y_t; f_t, Q_t) (KalmanFilter.scala:138-153)

  d4 gt, w, cc, mrow;
  double Fr[4];
  bool vr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    vr[r] = i < d;
    const bool ok = vr[r] && vc;
    gt[r] = ok ? a.G[c + i * d] : 0.0;   // G^T[i][c] = G[c][i]
    w[r] = ok ? W[i * d + c] : 0.0;
    cc[r] = ok ? C0[i * d + c] : 0.0;
    Fr[r] = vr[r] ? a.F[i] : 0.0;
    mrow[r] = vr[r] ? m0[i] : 0.0;
  }
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;

  // record 0: the initial state at t0 - 1 (KalmanFilter.scala:112-118)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    if (out && vr[r] && vc) out[d + i * d + c] = cc[r];
    if (out && vr[r] && col15) out[i] = mrow[r];
  }
  if (lane == 0) {
    if (fq) { fq[0] = __builtin_nan(""); fq[1] = __builtin_nan(""); }
    if (sd) { sd[0] = __builtin_nan(""); sd[1] = __builtin_nan(""); }
  }

  double ychunk = 0.0;
  for (int t = 0; t < T; ++t) {
    // observation stream: one coalesced 64-step chunk per 64 iterations
    if ((t & 63) == 0) ychunk = (t + lane < T) ? y[t + lane] : 0.0;
    const double yt = uniform_from_lane(ychunk, t & 63);

    // advState: a = G m, R = G C G^T + W
    const d4 cgt = mmT(cc, gt);                       // C G^T   (C symmetric)
    d4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = col15 ? mrow[r] : cgt[r];
    const d4 rp = mmT(gt, b);                         // G [C G^T | m] = [G C G^T | a]
    d4 R;
#pragma unroll
    for (int r = 0; r < 4; ++r) R[r] = col15 ? 0.0 : rp[r] + w[r];

    // one-step forecast f = F^T a (a sits in column 15), RF, Q = F^T R F + V
    double fp = 0.0, rfc = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { fp = fma(Fr[r], rp[r], fp); rfc = fma(R[r], Fr[r], rfc); }
    fp = col15 ? fp : 0.0;
    const double f = uniform_from_lane(sum_over_g(fp), 15);
    rfc = sum_over_g(rfc);                            // (R F)[c], all lanes
    double rfr[4], qp = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { rfr[r] = __shfl(rfc, 4 * r + g); qp = fma(Fr[r], rfr[r], qp); }
    const double Q = uniform_from_lane(sum_over_g(qp), 0) + V;

    double* o = out + (size_t)(t + 1) * rec;
    if (yt == yt) {
      // updateState, Joseph form expanded for p = 1
      // Joseph form for p = 1, factored: R - (RF_i/Q) RF_j (2 - Q (1/Q))
      const double e = yt - f, rq = 1.0 / Q;
      const double gam = rfc * (2.0 - Q * rq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double Kr = rfr[r] * rq;
        cc[r] = fma(-Kr, gam, R[r]);
        mrow[r] = fma(Kr, e, rp[r]);                  // meaningful in column-15 lanes
      }
      if (sd && lane == 0) { sd[2 * (t + 1)] = e * rq; sd[2 * (t + 1) + 1] = rq; }
      if (a.loglik) ll -= 0.5 * (1.8378770664093453 + log(Q) + e * e * rq);
    } else {
      cc = R;
      mrow = rp;
      if (sd && lane == 0) { sd[2 * (t + 1)] = __builtin_nan(""); sd[2 * (t + 1) + 1] = __builtin_nan(""); }
    }
    if (fq && lane == 0) { fq[2 * (t + 1)] = f; fq[2 * (t + 1) + 1] = Q; }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + g;
      if (out && vr[r] && vc) o[d + i * d + c] = cc[r];
      if (out && vr[r] && col15) o[i] = mrow[r];
    }
  }
  if (a.loglik && lane == 0) a.loglik[n] = ll;
  // a non-finite value, once present, propagates to every later state: test the last one
  bool bad = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) bad |= (vr[r] && vc && !isfinite(cc[r])) || (vr[r] && col15 && !isfinite(mrow[r]));
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

// ---------------------------------------------------------------------------------------
// backward pass
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_smoother_mfma16(KArgs a, const double* __restrict__ side) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
  if (n >= a.N) return;
  const int d = a.d, T = a.T, rec = d + d * d;
  const int g = lane >> 4, c = lane & 15;
  const bool vc = c < d, col15 = (c == 15);

  const double V = a.V[(size_t)n * a.v_stride];
  const double rV = 1.0 / V;
  const double* fin = a.filt_in + (size_t)n * (T + 1) * rec;
  double* out = a.smooth + (size_t)n * (T + 1) * rec;
  const double* sd = side + (size_t)n * (T + 1) * 2;
  __shared__ double tlds[4 * 16 * 17];
  double* timg = tlds + (threadIdx.x >> 6) * (16 * 17);   // wave-private transpose scratch

  d4 gm;             // std(G)
  double Fr[4];
  bool vr[4];
  const double Fc = vc ? a.F[c] : 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    vr[r] = i < d;
    gm[r] = (vr[r] && vc) ? a.G[i + c * d] : 0.0;
    Fr[r] = vr[r] ? a.F[i] : 0.0;
  }
  d4 P = {0.0, 0.0, 0.0, 0.0}, qrow = {0.0, 0.0, 0.0, 0.0};
  int st = (V > 0.0) ? 0 : DLM_ST_NOT_PD;
  bool bad = false;  // P and q carry any non-finite value down to record 0: test that one

  // software prefetch of the next record
  d4 ncc, nm;
  {
    const double* r0 = fin + (size_t)T * rec;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + g;
      ncc[r] = (vr[r] && vc) ? r0[d + i * d + c] : 0.0;
      nm[r] = vr[r] ? r0[i] : 0.0;
    }
  }
  double neq = sd[2 * T], niq = sd[2 * T + 1];

  for (int t = T; t >= 0; --t) {
    const d4 cc = ncc, mrow = nm;
    const double eq = neq, iq = niq;
    if (t > 0) {
      const double* r0 = fin + (size_t)(t - 1) * rec;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * r + g;
        ncc[r] = (vr[r] && vc) ? r0[d + i * d + c] : 0.0;
        nm[r] = vr[r] ? r0[i] : 0.0;
      }
      neq = sd[2 * (t - 1)]; niq = sd[2 * (t - 1) + 1];
    }
    const bool observed = (iq == iq) && t > 0;

    // K_t = C_t F / V (column-indexed, then row-indexed); zero when nothing was observed
    double kc = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) kc = fma(cc[r], Fr[r], kc);
    kc = observed ? sum_over_g(kc) * rV : 0.0;
    d4 krow, b1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { krow[r] = __shfl(kc, 4 * r + g); b1[r] = col15 ? krow[r] : cc[r]; }
    const d4 x1 = mmT(P, b1);                         // [P C | P K]
    d4 b2;
#pragma unroll
    for (int r = 0; r < 4; ++r) b2[r] = col15 ? qrow[r] : x1[r];
    const d4 x2 = mmT(cc, b2);                        // [C P C | C q]

    // s_t = m_t + C_t q_t ; S_t = C_t - C_t P_t C_t
    double* o = out + (size_t)t * rec;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * r + g;
      const double Sv = cc[r] - x2[r], sv = mrow[r] + x2[r];
      if (vr[r] && vc) o[d + i * d + c] = Sv;
      if (vr[r] && col15) o[i] = sv;
      if (t == 0) bad |= (vr[r] && vc && !isfinite(Sv)) || (vr[r] && col15 && !isfinite(sv));
    }
    if (t == 0) break;

    // (q_{t-1}, P_{t-1}) from (q_t, P_t)
    d4 M, rrow;
    if (observed) {
      d4 pkr;
      double kq = 0.0, kpk = 0.0, pkc = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pkr[r] = __shfl(x1[r], 16 * g + 15);          // (P K)[4r+g] in every lane of the row group
        kq = fma(krow[r], qrow[r], kq);
        kpk = fma(krow[r], pkr[r], kpk);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {                   // (P K)[c]: register c>>2 of a lane with g = c&3
        const double cand = __shfl(pkr[k], 16 * (c & 3));
        pkc = ((c >> 2) == k) ? cand : pkc;
      }
      kq = uniform_from_lane(sum_over_g(col15 ? kq : 0.0), 15);
      kpk = uniform_from_lane(sum_over_g(kpk), 0);
      const double sc = iq + kpk, sr = eq - kq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        M[r] = fma(-pkr[r], Fc, fma(-Fr[r], pkc, fma(Fr[r] * Fc, sc, P[r])));
        rrow[r] = fma(Fr[r], sr, qrow[r]);
      }
    } else {
      M = P;
      rrow = qrow;
    }
    // Symmetrise M through a wave-private LDS transpose: the rank-2 update above treats P as exactly
    // symmetric, and an antisymmetric rounding component would escape the contraction and grow
    // exponentially for unit-root models.
    if ((t & 7) == 0) {   // both cross terms use P^T K here: the asymmetry only rotates with G between clean-ups
#pragma unroll
      for (int r = 0; r < 4; ++r) timg[(4 * r + g) * 17 + c] = M[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int r = 0; r < 4; ++r) M[r] = 0.5 * (M[r] + timg[c * 17 + 4 * r + g]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    const d4 mg = mmT(M, gm);                         // M G   (M symmetric)
    d4 b3;
#pragma unroll
    for (int r = 0; r < 4; ++r) b3[r] = col15 ? rrow[r] : mg[r];
    const d4 pn = mmT(gm, b3);                        // G^T [M G | r] = [G^T M G | G^T r]
#pragma unroll
    for (int r = 0; r < 4; ++r) { P[r] = col15 ? 0.0 : pn[r]; qrow[r] = pn[r]; }
  }
  if (__ballot(bad) != 0ull) st |= DLM_ST_NONFINITE;
  if (a.status && lane == 0 && st) atomicOr(&a.status[n], st);
}

hipError_t launch_mfma16_filter(const KArgs& a, double* side, hipStream_t s) {
  hipLaunchKernelGGL(k_filter_mfma16, dim3((a.N + 3) / 4), dim3(256), 0, s, a, side);
  return hipGetLastError();
}

hipError_t launch_mfma16_smoother(const KArgs& a, const double* side, hipStream_t s) {
  hipLaunchKernelGGL(k_smoother_mfma16, dim3((a.N + 3) / 4), dim3(256), 0, s, a, side);
  return hipGetLastError();
}

}  // namespace dlm
