/*
 * dlm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, IEEE fp64) of the reference's Kalman filter /
 * RTS smoother / FFBS / SVD-filter / Gibbs-sufficient-statistic recursions for
 * ONE series.  It is the checker for the HIP engine: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link,
 * import or call anything in oracle/.  The product path never routes here.
 *
 * The reference (jonnylaw/bayesian_dlms) is Scala on Breeze 0.13.2 +
 * netlib LAPACK; neither a JVM nor Breeze exists in the build container, so
 * the reference itself cannot be run.  This restatement follows the reference
 * line by line (citations below are relative to
 * /root/reference/core/src/main/scala/dlm/model/) and is PINNED by the
 * reference's own known-answer tests and golden CSVs (tests/golden/):
 *   - core/src/test/scala/KalmanFilter.scala:78-189   (bivariate KF table)
 *   - core/src/test/scala/Smoothing.scala:10-76       (scalar RTS identities)
 *   - core/src/test/scala/SvdFilter.scala:102-158     (SVD filter == KF)
 *   - examples/data/first_order_dlm{,_filtered,_smoothed}.csv (T=1000, C1)
 * What those do not pin (d>1 smoother covariance, FFBS draws, Gibbs
 * statistics, SVD path with W != I) is "parity unpinned" against Breeze and is
 * cross-checked by algebraic identities instead (see tests/test_oracle.py).
 *
 * Layout conventions (Breeze DenseMatrix.data): all matrices column-major,
 * F is d x p and is used as F^T (KalmanFilter.scala:317), missing observation
 * components are NaN (the reference's Option None, Dlm.scala:94).
 *
 * Third-party arithmetic restated here (Breeze -> LAPACK, not vendored in
 * the reference): `\` = dgesv (LU, partial pivoting); `svd` = dgesdd
 * (restated as one-sided Jacobi, singular values descending); `eigSym` =
 * dsyevd (restated as cyclic Jacobi, eigenvalues ascending).  Factor signs
 * are implementation-defined in LAPACK; the canonical sign used here is
 * "largest-|component| of each vector is positive".
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IDX(i, j, ld) ((size_t)(i) + (size_t)(j) * (size_t)(ld))

/* ------------------------------------------------------------------ */
/* small dense helpers                                                 */
/* ------------------------------------------------------------------ */

/* C(m x n) = op(A) * op(B); op = transpose when ta/tb != 0.
 * A is (ta ? k x m : m x k) with leading dimension lda, likewise B. */
static void mm(int m, int n, int k, const double *restrict A, int lda, int ta,
               const double *restrict B, int ldb, int tb, double *restrict C, int ldc) {
  /* column-major: accumulate C[:, j] += op(A)[:, l] * op(B)[l, j] so that the inner loop runs
   * down a contiguous column whenever A is not transposed (the common case on this path) */
  for (int j = 0; j < n; ++j) {
    double *restrict c = C + (size_t)j * ldc;
    for (int i = 0; i < m; ++i) c[i] = 0.0;
    if (!ta) {
      for (int l = 0; l < k; ++l) {
        const double b = tb ? B[IDX(j, l, ldb)] : B[IDX(l, j, ldb)];
        const double *restrict a = A + (size_t)l * lda;
        for (int i = 0; i < m; ++i) c[i] += a[i] * b;
      }
    } else {
      for (int i = 0; i < m; ++i) {
        const double *restrict a = A + (size_t)i * lda;
        double s = 0.0;
        if (!tb) { const double *restrict b = B + (size_t)j * ldb; for (int l = 0; l < k; ++l) s += a[l] * b[l]; }
        else for (int l = 0; l < k; ++l) s += a[l] * B[IDX(j, l, ldb)];
        c[i] = s;
      }
    }
  }
}

/* Solve A X = B in place (B <- X); A n x n is destroyed.  LU with partial
 * pivoting, the algorithm behind Breeze's `\` (dgesv).  Returns 0, or k+1 if
 * the k-th pivot is exactly zero (Breeze throws MatrixSingularException). */
static int lu_solve(int n, double *A, int nrhs, double *B) {
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = fabs(A[IDX(k, k, n)]);
    for (int i = k + 1; i < n; ++i) {
      double v = fabs(A[IDX(i, k, n)]);
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0 || best != best) return k + 1;
    if (piv != k) {
      for (int j = 0; j < n; ++j) {
        double t = A[IDX(k, j, n)]; A[IDX(k, j, n)] = A[IDX(piv, j, n)]; A[IDX(piv, j, n)] = t;
      }
      for (int j = 0; j < nrhs; ++j) {
        double t = B[IDX(k, j, n)]; B[IDX(k, j, n)] = B[IDX(piv, j, n)]; B[IDX(piv, j, n)] = t;
      }
    }
    double pinv = 1.0 / A[IDX(k, k, n)];
    for (int i = k + 1; i < n; ++i) {
      double l = A[IDX(i, k, n)] * pinv;
      A[IDX(i, k, n)] = l;
      for (int j = k + 1; j < n; ++j) A[IDX(i, j, n)] -= l * A[IDX(k, j, n)];
      for (int j = 0; j < nrhs; ++j) B[IDX(i, j, n)] -= l * B[IDX(k, j, n)];
    }
  }
  for (int j = 0; j < nrhs; ++j)
    for (int i = n - 1; i >= 0; --i) {
      double s = B[IDX(i, j, n)];
      for (int l = i + 1; l < n; ++l) s -= A[IDX(i, l, n)] * B[IDX(l, j, n)];
      B[IDX(i, j, n)] = s / A[IDX(i, i, n)];
    }
  return 0;
}

/* canonical sign: make the largest-|.| entry of each column positive */
static void canon_cols(int rows, int cols, double *M, int ld) {
  for (int j = 0; j < cols; ++j) {
    int arg = 0; double best = -1.0;
    for (int i = 0; i < rows; ++i) {
      double v = fabs(M[IDX(i, j, ld)]);
      if (v > best * (1.0 + 1e-12)) { best = v; arg = i; }
    }
    if (M[IDX(arg, j, ld)] < 0.0)
      for (int i = 0; i < rows; ++i) M[IDX(i, j, ld)] = -M[IDX(i, j, ld)];
  }
}

/* Symmetric eigendecomposition (restates Breeze eigSym -> dsyevd): cyclic
 * Jacobi.  S (n x n) destroyed; lam ascending; E columns = eigenvectors. */
static void eig_sym(int n, double *S, double *lam, double *E) {
  for (int i = 0; i < n * n; ++i) E[i] = 0.0;
  for (int i = 0; i < n; ++i) E[IDX(i, i, n)] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < n; ++i) {
        double v = S[IDX(i, j, n)] * S[IDX(i, j, n)];
        if (i == j) dia += v; else off += v;
      }
    if (off <= 1e-60 + 1e-32 * dia) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double apq = S[IDX(p, q, n)];
        if (apq == 0.0) continue;
        double app = S[IDX(p, p, n)], aqq = S[IDX(q, q, n)];
        double tau = (aqq - app) / (2.0 * apq);
        double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) {
          double skp = S[IDX(k, p, n)], skq = S[IDX(k, q, n)];
          S[IDX(k, p, n)] = c * skp - s * skq;
          S[IDX(k, q, n)] = s * skp + c * skq;
        }
        for (int k = 0; k < n; ++k) {
          double spk = S[IDX(p, k, n)], sqk = S[IDX(q, k, n)];
          S[IDX(p, k, n)] = c * spk - s * sqk;
          S[IDX(q, k, n)] = s * spk + c * sqk;
        }
        for (int k = 0; k < n; ++k) {
          double ekp = E[IDX(k, p, n)], ekq = E[IDX(k, q, n)];
          E[IDX(k, p, n)] = c * ekp - s * ekq;
          E[IDX(k, q, n)] = s * ekp + c * ekq;
        }
      }
  }
  for (int i = 0; i < n; ++i) lam[i] = S[IDX(i, i, n)];
  /* ascending selection sort, permuting eigenvector columns */
  for (int i = 0; i < n - 1; ++i) {
    int m = i;
    for (int j = i + 1; j < n; ++j) if (lam[j] < lam[m]) m = j;
    if (m != i) {
      double t = lam[i]; lam[i] = lam[m]; lam[m] = t;
      for (int k = 0; k < n; ++k) {
        double e = E[IDX(k, i, n)]; E[IDX(k, i, n)] = E[IDX(k, m, n)]; E[IDX(k, m, n)] = e;
      }
    }
  }
  canon_cols(n, n, E, n);
}

/* Thin SVD (restates Breeze svd -> dgesdd for m >= n): one-sided (Hestenes)
 * Jacobi.  A (m x n) destroyed.  sig descending, Vt (n x n) = rightVectors
 * (rows are right singular vectors), canonical sign per row of Vt. */
static void svd_right(int m, int n, double *A, double *sig, double *Vt) {
  double *V = (double *)malloc(sizeof(double) * n * n);
  for (int i = 0; i < n * n; ++i) V[i] = 0.0;
  for (int i = 0; i < n; ++i) V[IDX(i, i, n)] = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    int rotated = 0;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int k = 0; k < m; ++k) {
          double ap = A[IDX(k, p, m)], aq = A[IDX(k, q, m)];
          alpha += ap * ap; beta += aq * aq; gamma += ap * aq;
        }
        if (gamma == 0.0 || fabs(gamma) <= 1e-16 * sqrt(alpha * beta)) continue;
        rotated = 1;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        double c = 1.0 / sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < m; ++k) {
          double ap = A[IDX(k, p, m)], aq = A[IDX(k, q, m)];
          A[IDX(k, p, m)] = c * ap - s * aq;
          A[IDX(k, q, m)] = s * ap + c * aq;
        }
        for (int k = 0; k < n; ++k) {
          double vp = V[IDX(k, p, n)], vq = V[IDX(k, q, n)];
          V[IDX(k, p, n)] = c * vp - s * vq;
          V[IDX(k, q, n)] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  for (int j = 0; j < n; ++j) {
    double s = 0;
    for (int k = 0; k < m; ++k) s += A[IDX(k, j, m)] * A[IDX(k, j, m)];
    sig[j] = sqrt(s);
  }
  for (int i = 0; i < n - 1; ++i) {
    int mx = i;
    for (int j = i + 1; j < n; ++j) if (sig[j] > sig[mx]) mx = j;
    if (mx != i) {
      double t = sig[i]; sig[i] = sig[mx]; sig[mx] = t;
      for (int k = 0; k < n; ++k) {
        double e = V[IDX(k, i, n)]; V[IDX(k, i, n)] = V[IDX(k, mx, n)]; V[IDX(k, mx, n)] = e;
      }
    }
  }
  canon_cols(n, n, V, n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Vt[IDX(i, j, n)] = V[IDX(j, i, n)];
  free(V);
}

/* ------------------------------------------------------------------ */
/* model description shared by every entry point                        */
/* ------------------------------------------------------------------ */
typedef struct {
  int d, p, T;
  const double *F;      /* [nF][d*p] column-major d x p                       */
  long f_stride;        /* 0: time-invariant, else F_t = F + t*f_stride       */
  const double *G;      /* [nG][d*d] column-major, one per distinct dt        */
  const int *g_index;   /* [T] index into G for step t (NULL: all 0)          */
  const double *dt;     /* [T] time increments (NULL: all 1.0); dt[0] is the  */
                        /* gap from the initial state at t0-1, i.e. 1.0       */
                        /* (KalmanFilter.scala:112-118)                       */
} model_t;

static const double *F_at(const model_t *M, int t) { return M->F + (size_t)t * M->f_stride; }
static const double *G_at(const model_t *M, int t) {
  return M->G + (size_t)(M->g_index ? M->g_index[t] : 0) * M->d * M->d;
}
static double dt_at(const model_t *M, int t) { return M->dt ? M->dt[t] : 1.0; }

/* ------------------------------------------------------------------ */
/* Kalman filter: KalmanFilter.scala:64-118, :262-286, :311-321;         */
/* driver loop Filter.scala:41-45 (`filter`, keeps the initial state).    */
/* Outputs have T+1 records; record 0 is the initial state at t0-1 with   */
/* m=a=m0, C=R=C0 and f,Q = NaN (the reference's None).                   */
/* Returns 0 or (t+1) of the first singular Q solve.                      */
/* ------------------------------------------------------------------ */
static int kf_filter_core(int d, int p, int T, const double *F, long f_stride,
                     const double *G, const int *g_index, const double *dts,
                     const double *V0, long v_tstride, const double *W0, long w_tstride, const double *m0,
                     const double *C0, const double *y, /* [T][p], NaN = missing */
                     double *m_out, double *C_out,       /* [T+1][d], [T+1][d*d] */
                     double *a_out, double *R_out,       /* same shapes, may be NULL */
                     double *f_out, double *Q_out)       /* [T+1][p], [T+1][p*p], may be NULL */
{
  model_t M = {d, p, T, F, f_stride, G, g_index, dts};
  const int dd = d * d;
  double *a = malloc(sizeof(double) * d), *R = malloc(sizeof(double) * dd);
  double *GC = malloc(sizeof(double) * dd);
  double *Fm = malloc(sizeof(double) * d * p), *Vm = malloc(sizeof(double) * p * p);
  double *Qm = malloc(sizeof(double) * p * p), *RF = malloc(sizeof(double) * d * p);
  double *X = malloc(sizeof(double) * p * d), *K = malloc(sizeof(double) * d * p);
  double *D = malloc(sizeof(double) * dd), *DR = malloc(sizeof(double) * dd);
  double *KV = malloc(sizeof(double) * d * p);
  double *yo = malloc(sizeof(double) * p), *res = malloc(sizeof(double) * p);
  int *idx = malloc(sizeof(int) * p);
  int rc = 0;

  /* initialiseState: KalmanFilter.scala:112-118 */
  memcpy(m_out, m0, sizeof(double) * d);
  memcpy(C_out, C0, sizeof(double) * dd);
  if (a_out) memcpy(a_out, m0, sizeof(double) * d);
  if (R_out) memcpy(R_out, C0, sizeof(double) * dd);
  if (f_out) for (int i = 0; i < p; ++i) f_out[i] = NAN;
  if (Q_out) for (int i = 0; i < p * p; ++i) Q_out[i] = NAN;

  for (int t = 0; t < T; ++t) {
    /* time-varying variances: V_t of observation t, W_t of the transition into it (StudentTGibbs.scala:100-136) */
    const double *V = V0 + (size_t)t * v_tstride, *W = W0 + (size_t)t * w_tstride;
    const double *m = m_out + (size_t)t * d, *C = C_out + (size_t)t * dd;
    double *mn = m_out + (size_t)(t + 1) * d, *Cn = C_out + (size_t)(t + 1) * dd;
    const double *Ft = F_at(&M, t), *Gt = G_at(&M, t);
    const double dt = dt_at(&M, t);
    /* advState: KalmanFilter.scala:273-286 */
    if (dt == 0.0) {
      memcpy(a, m, sizeof(double) * d);
      memcpy(R, C, sizeof(double) * dd);
    } else {
      mm(d, 1, d, Gt, d, 0, m, d, 0, a, d);
      mm(d, d, d, Gt, d, 0, C, d, 0, GC, d);
      mm(d, d, d, GC, d, 0, Gt, d, 1, R, d);
      for (int i = 0; i < dd; ++i) R[i] += W[i] * dt;
    }
    if (a_out) memcpy(a_out + (size_t)(t + 1) * d, a, sizeof(double) * d);
    if (R_out) memcpy(R_out + (size_t)(t + 1) * dd, R, sizeof(double) * dd);
    /* oneStepPrediction (unmasked): KalmanFilter.scala:311-321 */
    if (f_out) mm(p, 1, d, Ft, d, 1, a, d, 0, f_out + (size_t)(t + 1) * p, p);
    if (Q_out) {
      double *Qf = Q_out + (size_t)(t + 1) * p * p;
      mm(d, p, d, R, d, 0, Ft, d, 0, RF, d);
      mm(p, p, d, Ft, d, 1, RF, d, 0, Qf, p);
      for (int i = 0; i < p * p; ++i) Qf[i] += V[i];
    }
    /* updateState: KalmanFilter.scala:64-94 */
    int pm = 0;
    for (int i = 0; i < p; ++i) {
      double yi = y[(size_t)t * p + i];
      if (yi == yi) { idx[pm] = i; yo[pm] = yi; ++pm; }
    }
    if (pm == 0) {
      memcpy(mn, a, sizeof(double) * d);
      memcpy(Cn, R, sizeof(double) * dd);
      continue;
    }
    for (int j = 0; j < pm; ++j) {            /* missingF / missingV :202-218 */
      for (int i = 0; i < d; ++i) Fm[IDX(i, j, d)] = Ft[IDX(i, idx[j], d)];
      for (int i = 0; i < pm; ++i) Vm[IDX(i, j, pm)] = V[IDX(idx[i], idx[j], p)];
    }
    mm(pm, 1, d, Fm, d, 1, a, d, 0, res, pm);               /* predicted     */
    for (int i = 0; i < pm; ++i) res[i] = yo[i] - res[i];    /* residual      */
    mm(d, pm, d, R, d, 0, Fm, d, 0, RF, d);
    mm(pm, pm, d, Fm, d, 1, RF, d, 0, Qm, pm);
    for (int i = 0; i < pm * pm; ++i) Qm[i] += Vm[i];        /* predcov       */
    /* kalmanGain = (predcov.t \ (fm.t * rt.t)).t : KalmanFilter.scala:83 */
    {
      double *Qt = D; /* reuse as scratch pm x pm */
      for (int i = 0; i < pm; ++i)
        for (int j = 0; j < pm; ++j) Qt[IDX(i, j, pm)] = Qm[IDX(j, i, pm)];
      mm(pm, d, d, Fm, d, 1, R, d, 1, X, pm);
      int s = lu_solve(pm, Qt, d, X);
      if (s && !rc) rc = t + 1;
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < pm; ++j) K[IDX(i, j, d)] = X[IDX(j, i, pm)];
    }
    mm(d, 1, pm, K, d, 0, res, pm, 0, mn, d);
    for (int i = 0; i < d; ++i) mn[i] += a[i];
    /* Joseph form: KalmanFilter.scala:87-90 */
    mm(d, d, pm, K, d, 0, Fm, d, 1, D, d);
    for (int j = 0; j < d; ++j)
      for (int i = 0; i < d; ++i) D[IDX(i, j, d)] = (i == j ? 1.0 : 0.0) - D[IDX(i, j, d)];
    mm(d, d, d, D, d, 0, R, d, 0, DR, d);
    mm(d, d, d, DR, d, 0, D, d, 1, Cn, d);
    mm(d, pm, pm, K, d, 0, Vm, pm, 0, KV, d);
    mm(d, d, pm, KV, d, 0, K, d, 1, DR, d);
    for (int i = 0; i < dd; ++i) Cn[i] += DR[i];
  }
  free(a); free(R); free(GC); free(Fm); free(Vm); free(Qm); free(RF); free(X);
  free(K); free(D); free(DR); free(KV); free(yo); free(res); free(idx);
  return rc;
}

int oracle_kf_filter(int d, int p, int T, const double *F, long f_stride,
                     const double *G, const int *g_index, const double *dts,
                     const double *V, const double *W, const double *m0,
                     const double *C0, const double *y, double *m_out, double *C_out,
                     double *a_out, double *R_out, double *f_out, double *Q_out) {
  return kf_filter_core(d, p, T, F, f_stride, G, g_index, dts, V, 0, W, 0, m0, C0, y, m_out, C_out, a_out, R_out, f_out, Q_out);
}
/* the same with V_t [T][p*p] / W_t [T][d*d] streams (a stride of 0 keeps that matrix time-invariant) */
int oracle_kf_filter_tv(int d, int p, int T, const double *F, long f_stride,
                        const double *G, const int *g_index, const double *dts,
                        const double *V, long v_tstride, const double *W, long w_tstride, const double *m0,
                        const double *C0, const double *y, double *m_out, double *C_out,
                        double *a_out, double *R_out, double *f_out, double *Q_out) {
  return kf_filter_core(d, p, T, F, f_stride, G, g_index, dts, V, v_tstride, W, w_tstride, m0, C0, y, m_out, C_out, a_out, R_out, f_out, Q_out);
}

/* cgrinv = (rt1.t \ (g(dt) * ct.t)).t  -- Smoothing.scala:41 and :85.
 * J (d x d) out; R1 = R_{t+1}, C = C_t, Gn = G(dt_{t+1}). */
static int cgrinv(int d, const double *R1, const double *Gn, const double *C, double *J,
                  double *s1, double *s2) {
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) s1[IDX(i, j, d)] = R1[IDX(j, i, d)];   /* rt1.t */
  mm(d, d, d, Gn, d, 0, C, d, 1, s2, d);                                 /* g * ct.t */
  int rc = lu_solve(d, s1, d, s2);
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) J[IDX(i, j, d)] = s2[IDX(j, i, d)];
  return rc;
}

/* ------------------------------------------------------------------ */
/* RTS smoother: Smoothing.scala:31-64.  Inputs are the T+1 filter      */
/* records (m,C,a,R).  compat_q1 != 0 reproduces the reference literally  */
/* (`cgrinv * (rt1 - cov) * cgrinv`, Smoothing.scala:44, no transpose on   */
/* the trailing factor); compat_q1 == 0 uses the textbook J X J^T.         */
/* ------------------------------------------------------------------ */
int oracle_smoother(int d, int T, const double *G, const int *g_index,
                    const double *m, const double *C, const double *a, const double *R,
                    int compat_q1, double *s_out, double *S_out) {
  const int dd = d * d;
  double *J = malloc(sizeof(double) * dd), *s1 = malloc(sizeof(double) * dd);
  double *s2 = malloc(sizeof(double) * dd), *X = malloc(sizeof(double) * dd);
  double *u = malloc(sizeof(double) * d);
  int rc = 0;
  memcpy(s_out + (size_t)T * d, m + (size_t)T * d, sizeof(double) * d);
  memcpy(S_out + (size_t)T * dd, C + (size_t)T * dd, sizeof(double) * dd);
  for (int t = T - 1; t >= 0; --t) {
    /* state_{t+1} carries at1 = a_{t+1}, rt1 = R_{t+1}; dt = time_{t+1}-time_t
     * is the increment of filter step index t (0-based). */
    const double *Gn = G + (size_t)(g_index ? g_index[t] : 0) * dd;
    const double *R1 = R + (size_t)(t + 1) * dd, *a1 = a + (size_t)(t + 1) * d;
    const double *Ct = C + (size_t)t * dd, *mt = m + (size_t)t * d;
    const double *sn = s_out + (size_t)(t + 1) * d, *Sn = S_out + (size_t)(t + 1) * dd;
    double *st = s_out + (size_t)t * d, *St = S_out + (size_t)t * dd;
    int s = cgrinv(d, R1, Gn, Ct, J, s1, s2);
    if (s && !rc) rc = t + 1;
    for (int i = 0; i < d; ++i) u[i] = sn[i] - a1[i];
    mm(d, 1, d, J, d, 0, u, d, 0, st, d);
    for (int i = 0; i < d; ++i) st[i] += mt[i];
    for (int i = 0; i < dd; ++i) X[i] = R1[i] - Sn[i];
    mm(d, d, d, J, d, 0, X, d, 0, s1, d);
    mm(d, d, d, s1, d, 0, J, d, compat_q1 ? 0 : 1, s2, d);
    for (int i = 0; i < dd; ++i) St[i] = Ct[i] - s2[i];
  }
  free(J); free(s1); free(s2); free(X); free(u);
  return rc;
}

/* ------------------------------------------------------------------ */
/* Backward sampling: Smoothing.scala:74-122 (+ MultivariateGaussianSvd  */
/* .scala:13-22 for the draw).  z [T+1][d] are the standard normals the   */
/* caller injects (the reference's RNG is unseedable).         */
/* factor: 0 = eigSym (reference-literal: mu + E sqrt(Lambda) z),         */
/*         1 = lower Cholesky (engine-canonical: mu + L z).               */
/* h_out/H_out (optional) receive the conditional mean/covariance used    */
/* at each step (record T holds m_T, C_T).                                */
/* ------------------------------------------------------------------ */
static int chol_lower(int n, const double *A, double *L) {
  for (int i = 0; i < n * n; ++i) L[i] = 0.0;
  for (int j = 0; j < n; ++j) {
    double s = A[IDX(j, j, n)];
    for (int k = 0; k < j; ++k) s -= L[IDX(j, k, n)] * L[IDX(j, k, n)];
    if (!(s > 0.0)) { s = 0.0; }
    double ljj = sqrt(s);
    L[IDX(j, j, n)] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double v = A[IDX(i, j, n)];
      for (int k = 0; k < j; ++k) v -= L[IDX(i, k, n)] * L[IDX(j, k, n)];
      L[IDX(i, j, n)] = ljj > 0.0 ? v / ljj : 0.0;
    }
  }
  return 0;
}

/* Sum over a series of KalmanFilter.conditionalLikelihood (KalmanFilter.scala:138-153):
 *   log N(y_t^obs ; f_t^obs, Q_t^obs)  with the observed components selected as there
 * (indexNonMissing :190-197); a step with no observed component contributes 0.
 * f [T+1][p] and Q [T+1][p*p] (column-major) are the forecasts written by oracle_kf_filter
 * (record t+1 belongs to observation t); y [T][p], NaN = missing. */
double oracle_loglik(int p, int T, const double *f, const double *Q, const double *y) {
  double ll = 0.0;
  int *idx = (int *)malloc(sizeof(int) * (size_t)p);
  double *Qm = (double *)malloc(sizeof(double) * (size_t)p * p);
  double *L = (double *)malloc(sizeof(double) * (size_t)p * p);
  double *w = (double *)malloc(sizeof(double) * (size_t)p);
  for (int t = 0; t < T; ++t) {
    const double *ft = f + (size_t)(t + 1) * p, *Qt = Q + (size_t)(t + 1) * p * p, *yt = y + (size_t)t * p;
    int pm = 0;
    for (int j = 0; j < p; ++j) if (yt[j] == yt[j]) idx[pm++] = j;
    if (pm == 0) continue;
    for (int j = 0; j < pm; ++j) for (int i = 0; i < pm; ++i) Qm[i + j * pm] = Qt[idx[i] + idx[j] * p];
    chol_lower(pm, Qm, L);
    double logdet = 0.0, quad = 0.0;
    for (int i = 0; i < pm; ++i) {   /* w = L^-1 e */
      double v = yt[idx[i]] - ft[idx[i]];
      for (int l = 0; l < i; ++l) v -= L[i + l * pm] * w[l];
      w[i] = v / L[i + i * pm];
      quad += w[i] * w[i];
      logdet += 2.0 * log(L[i + i * pm]);
    }
    ll -= 0.5 * (pm * 1.8378770664093453 + logdet + quad);
  }
  free(idx); free(Qm); free(L); free(w);
  return ll;
}

/* KalmanFilter.logLikelihood (KalmanFilter.scala:175-183) as KalmanFilter.likelihood calls it (:299-306) -- the
 * "likelihood" MetropolisHastings.dlm evaluates (MetropolisHastings.scala:134, :205): over the T+1
 * filtered means m_0 .. m_T (m_0 = the initial state at t0 - 1, kf.initialiseState),
 *   sum_{t=1..T} MultivariateGaussian(mod.g(dt_t) * m_{t-1}, w * dt_t).logPdf(m_t).
 * Breeze's MultivariateGaussian(mean, cov).logPdf(x) = -1/2 |L^-1 (x - mean)|^2 - (n/2 log 2 pi + sum log diag L),
 * L = cholesky(cov) (lower).  Every consecutive pair counts, also across a missing observation (there m_t = a_t: the
 * quadratic form is 0).  dt = 0 makes the covariance the zero matrix: Breeze's cholesky throws; here the result is NaN and
 * the return value 1.  m [T+1][d]; G, g_index, dts as in oracle_kf_filter (mod.g(dt) is the table entry also for dt = 0);
 * W d x d column-major.  Returns 0, or 1 if some W dt was not positive definite. */
int oracle_loglik_q7(int d, int T, const double *G, const int *g_index, const double *dts, const double *W,
                     const double *m, double *out) {
  double ll = 0.0;
  int bad = 0;
  double *cov = (double *)malloc(sizeof(double) * (size_t)d * d);
  double *L = (double *)malloc(sizeof(double) * (size_t)d * d);
  double *c = (double *)malloc(sizeof(double) * (size_t)d);
  double *u = (double *)malloc(sizeof(double) * (size_t)d);
  for (int t = 0; t < T; ++t) {
    const double dt = dts ? dts[t] : 1.0;
    const double *Gt = G + (size_t)(g_index ? g_index[t] : 0) * d * d;
    const double *x0 = m + (size_t)t * d, *x1 = m + (size_t)(t + 1) * d;
    for (int i = 0; i < d; ++i) {   /* c = x1 - g(dt) x0 */
      double v = 0.0;
      for (int k = 0; k < d; ++k) v += Gt[IDX(i, k, d)] * x0[k];
      c[i] = x1[i] - v;
    }
    for (int i = 0; i < d * d; ++i) cov[i] = W[i] * dt;
    chol_lower(d, cov, L);
    double logdet = 0.0, quad = 0.0;
    for (int i = 0; i < d; ++i) {   /* u = L^-1 c */
      double v = c[i];
      for (int l = 0; l < i; ++l) v -= L[IDX(i, l, d)] * u[l];
      if (!(L[IDX(i, i, d)] > 0.0)) bad = 1;
      u[i] = v / L[IDX(i, i, d)];
      quad += u[i] * u[i];
      logdet += log(L[IDX(i, i, d)]);
    }
    ll += -0.5 * quad - (0.5 * d * 1.8378770664093453 + logdet);
  }
  free(cov); free(L); free(c); free(u);
  *out = bad ? NAN : ll;
  return bad;
}

/* ---- scalar AR(1) state (FilterAr.scala:15-82) -------------------------------------------------
 * alpha_t = mu + phi (alpha_{t-1} - mu) + eta_t, eta_t ~ N(0, sigma_eta^2);  y_t = alpha_t + eps_t, eps_t ~ N(0, v_t).
 * filterUnivariate (:34-49): m0 = mu, c0 = sigma_eta^2 / (1 - phi^2); stepUni (:17-32).  Outputs have T+1 records. */
void oracle_ar1_filter(int T, const double *y /* NaN = None */, const double *v, double phi, double mu,
                       double sigma_eta, double *m, double *c, double *a, double *r) {
  m[0] = a[0] = mu;
  c[0] = r[0] = sigma_eta * sigma_eta / (1.0 - phi * phi);
  for (int t = 0; t < T; ++t) {
    const double at = mu + phi * (m[t] - mu);
    const double rt = phi * phi * c[t] + sigma_eta * sigma_eta;
    a[t + 1] = at; r[t + 1] = rt;
    if (y[t] == y[t]) {
      const double kt = rt / (rt + v[t]);
      m[t + 1] = at + kt * (y[t] - at);
      c[t + 1] = kt * v[t];
    } else { m[t + 1] = at; c[t + 1] = rt; }
  }
}
/* univariateSample / backStepUni (:58-76): theta_T = m_T + sqrt(c_T) z_T;
 * mean = m_t + (c_t phi / r_{t+1}) (theta_{t+1} - a_{t+1}), cov = c_t - c_t^2 phi^2 / r_{t+1}. */
void oracle_ar1_backward_sample(int T, double phi, const double *m, const double *c, const double *a,
                                const double *r, const double *z, double *theta) {
  theta[T] = m[T] + sqrt(c[T]) * z[T];
  for (int t = T - 1; t >= 0; --t) {
    const double mean = m[t] + (c[t] * phi / r[t + 1]) * (theta[t + 1] - a[t + 1]);
    double cov = c[t] - (c[t] * c[t]) * (phi * phi) / r[t + 1];
    if (cov < 0.0) cov = 0.0;
    theta[t] = mean + sqrt(cov) * z[t];
  }
}

/* ---- scalar Ornstein-Uhlenbeck state on an irregular grid (FilterOu.scala:7-79) -----------------------------------
 * stepUni (:7-27): dt = t - st.time; variance = sigma^2 (1 - exp(-2 phi dt)) / (2 phi);
 *   a = mu + exp(-phi dt) (m - mu);  r = exp(-2 phi dt) c + variance;  update as FilterAr.
 * filterUnivariate (:34-46), LITERALLY: m0 = mu, c0 = sigma * sigma / phi * phi (= sigma^2 by operator precedence, the
 * stationary variance sigma^2 / (2 phi) was presumably meant), and the initial state carries the time of the FIRST
 * observation, so the first step has dt = 0.  parity unpinned: the reference holds no fixture for this filter. */
void oracle_ou_filter(int T, const double *times, const double *y, const double *v, double phi, double mu,
                      double sigma_eta, double *m, double *c, double *a, double *r) {
  m[0] = a[0] = mu;
  c[0] = r[0] = sigma_eta * sigma_eta / phi * phi;
  for (int t = 0; t < T; ++t) {
    const double dt = times[t] - (t == 0 ? times[0] : times[t - 1]);
    const double variance = (pow(sigma_eta, 2) * (1 - exp(-2 * phi * dt))) / (2 * phi);
    const double at = mu + exp(-phi * dt) * (m[t] - mu);
    const double rt = exp(-2 * phi * dt) * c[t] + variance;
    a[t + 1] = at; r[t + 1] = rt;
    if (y[t] == y[t]) {
      const double kt = rt / (rt + v[t]);
      m[t + 1] = at + kt * (y[t] - at);
      c[t + 1] = kt * v[t];
    } else { m[t + 1] = at; c[t + 1] = rt; }
  }
}
/* univariateSample / backStepUni (:48-71): phi(dt) = exp(-phi dt) with dt = ss.time - fs.time */
void oracle_ou_backward_sample(int T, const double *times, double phi, const double *m, const double *c,
                               const double *a, const double *r, const double *z, double *theta) {
  theta[T] = m[T] + sqrt(c[T]) * z[T];
  for (int t = T - 1; t >= 0; --t) {
    const double dt = times[t] - (t == 0 ? times[0] : times[t - 1]);   /* time(record t+1) - time(record t) */
    const double ph = exp(-phi * dt);
    const double mean = m[t] + (c[t] * ph / r[t + 1]) * (theta[t + 1] - a[t + 1]);
    double cov = c[t] - (pow(c[t], 2) * pow(ph, 2)) / r[t + 1];
    if (cov < 0.0) cov = 0.0;   /* dt = 0 (the first step, always): c - c^2/c at rounding level; the reference would draw NaN */
    theta[t] = mean + sqrt(cov) * z[t];
  }
}

static void mvn_draw(int d, const double *mu, const double *cov, const double *z, int factor,
                     double *out, double *w1, double *w2, double *w3) {
  if (factor == 0) {
    memcpy(w1, cov, sizeof(double) * d * d);
    eig_sym(d, w1, w3, w2); /* w3 = lambda ascending, w2 = E */
    for (int i = 0; i < d; ++i) {
      double s = mu[i];
      for (int k = 0; k < d; ++k) s += w2[IDX(i, k, d)] * sqrt(w3[k]) * z[k];
      out[i] = s;
    }
  } else {
    chol_lower(d, cov, w1);
    for (int i = 0; i < d; ++i) {
      double s = mu[i];
      for (int k = 0; k <= i; ++k) s += w1[IDX(i, k, d)] * z[k];
      out[i] = s;
    }
  }
}

static int backward_sample_core(int d, int T, const double *G, const int *g_index, const double *dts,
                           const double *W0, long w_tstride, const double *m, const double *C, const double *a,
                           const double *R, const double *z, int factor, double *theta,
                           double *h_out, double *H_out) {
  const int dd = d * d;
  double *J = malloc(sizeof(double) * dd), *s1 = malloc(sizeof(double) * dd);
  double *s2 = malloc(sizeof(double) * dd), *D = malloc(sizeof(double) * dd);
  double *H = malloc(sizeof(double) * dd), *h = malloc(sizeof(double) * d);
  double *u = malloc(sizeof(double) * d), *w3 = malloc(sizeof(double) * dd);
  int rc = 0;
  /* initialise: Smoothing.scala:105-109 */
  mvn_draw(d, m + (size_t)T * d, C + (size_t)T * dd, z + (size_t)T * d, factor,
           theta + (size_t)T * d, s1, s2, w3);
  if (h_out) memcpy(h_out + (size_t)T * d, m + (size_t)T * d, sizeof(double) * d);
  if (H_out) memcpy(H_out + (size_t)T * dd, C + (size_t)T * dd, sizeof(double) * dd);
  for (int t = T - 1; t >= 0; --t) {
    const double *W = W0 + (size_t)t * w_tstride;   /* W of the transition into record t+1 */
    const double *Gn = G + (size_t)(g_index ? g_index[t] : 0) * dd;
    const double dt = dts ? dts[t] : 1.0;
    const double *R1 = R + (size_t)(t + 1) * dd, *a1 = a + (size_t)(t + 1) * d;
    const double *Ct = C + (size_t)t * dd, *mt = m + (size_t)t * d;
    const double *thn = theta + (size_t)(t + 1) * d;
    int s = cgrinv(d, R1, Gn, Ct, J, s1, s2);
    if (s && !rc) rc = t + 1;
    for (int i = 0; i < d; ++i) u[i] = thn[i] - a1[i];
    mm(d, 1, d, J, d, 0, u, d, 0, h, d);
    for (int i = 0; i < d; ++i) h[i] += mt[i];
    /* diff = I - J G ; cov = diff C diff^T + J (W dt) J^T ; symmetrise  (:92-95) */
    mm(d, d, d, J, d, 0, Gn, d, 0, D, d);
    for (int j = 0; j < d; ++j)
      for (int i = 0; i < d; ++i) D[IDX(i, j, d)] = (i == j ? 1.0 : 0.0) - D[IDX(i, j, d)];
    mm(d, d, d, D, d, 0, Ct, d, 0, s1, d);
    mm(d, d, d, s1, d, 0, D, d, 1, H, d);
    mm(d, d, d, J, d, 0, W, d, 0, s1, d);
    mm(d, d, d, s1, d, 0, J, d, 1, s2, d);
    for (int i = 0; i < dd; ++i) H[i] += s2[i] * dt;
    for (int j = 0; j < d; ++j)
      for (int i = 0; i < j; ++i) {
        double v = (H[IDX(i, j, d)] + H[IDX(j, i, d)]) / 2.0;
        H[IDX(i, j, d)] = v; H[IDX(j, i, d)] = v;
      }
    if (h_out) memcpy(h_out + (size_t)t * d, h, sizeof(double) * d);
    if (H_out) memcpy(H_out + (size_t)t * dd, H, sizeof(double) * dd);
    mvn_draw(d, h, H, z + (size_t)t * d, factor, theta + (size_t)t * d, s1, s2, w3);
  }
  free(J); free(s1); free(s2); free(D); free(H); free(h); free(u); free(w3);
  return rc;
}

int oracle_backward_sample(int d, int T, const double *G, const int *g_index, const double *dts,
                           const double *W, const double *m, const double *C, const double *a,
                           const double *R, const double *z, int factor, double *theta,
                           double *h_out, double *H_out) {
  return backward_sample_core(d, T, G, g_index, dts, W, 0, m, C, a, R, z, factor, theta, h_out, H_out);
}
/* the same with a W_t [T][d*d] stream */
int oracle_backward_sample_tv(int d, int T, const double *G, const int *g_index, const double *dts,
                              const double *W, long w_tstride, const double *m, const double *C, const double *a,
                              const double *R, const double *z, int factor, double *theta,
                              double *h_out, double *H_out) {
  return backward_sample_core(d, T, G, g_index, dts, W, w_tstride, m, C, a, R, z, factor, theta, h_out, H_out);
}

/* ------------------------------------------------------------------ */
/* Gibbs sufficient statistics: Gibbs.scala:23-50 (ssy, n),              */
/* Gibbs.scala:56-78 (ss = sum diff^2/dt), GibbsWishart.scala:16-35       */
/* (outer = sum diff diff^T / dt).  theta is [T+1][d]; record 0 is the    */
/* state at t0-1.                                                         */
/* ------------------------------------------------------------------ */
void oracle_gibbs_stats(int d, int p, int T, const double *F, long f_stride, const double *G,
                        const int *g_index, const double *dts, const double *y,
                        const double *theta, double *ssy, double *nobs, double *ss,
                        double *outer /* may be NULL */) {
  model_t M = {d, p, T, F, f_stride, G, g_index, dts};
  double *ft = malloc(sizeof(double) * p), *df = malloc(sizeof(double) * d);
  for (int i = 0; i < p; ++i) { ssy[i] = 0.0; nobs[i] = 0.0; }
  for (int i = 0; i < d; ++i) ss[i] = 0.0;
  if (outer) for (int i = 0; i < d * d; ++i) outer[i] = 0.0;
  for (int t = 0; t < T; ++t) {
    const double *x0 = theta + (size_t)t * d, *x1 = theta + (size_t)(t + 1) * d;
    mm(p, 1, d, F_at(&M, t), d, 1, x1, d, 0, ft, p);
    for (int i = 0; i < p; ++i) {
      double yi = y[(size_t)t * p + i];
      if (yi == yi) { ssy[i] += (yi - ft[i]) * (yi - ft[i]); nobs[i] += 1.0; }
    }
    mm(d, 1, d, G_at(&M, t), d, 0, x0, d, 0, df, d);
    const double dt = dt_at(&M, t);
    for (int i = 0; i < d; ++i) df[i] = x1[i] - df[i];
    for (int i = 0; i < d; ++i) ss[i] += df[i] * df[i] / dt;
    if (outer)
      for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) outer[IDX(i, j, d)] += df[i] * df[j] / dt;
  }
  free(ft); free(df);
}

/* ------------------------------------------------------------------ */
/* SVD (square-root) filter: SvdFilter.scala:38-95, :183-236.            */
/* sqrtVinv = diag(lam^-1/2) Vt, sqrtW = diag(lam^1/2) Vt (:210-236).      */
/* raw_w_q2 != 0 reproduces filterDlm literally: the time update receives  */
/* the UNtransformed W where sqrt(W) is expected (SvdFilter.scala:158-161   */
/* vs :104-106).  raw_w_q2 == 0 passes sqrt(W).                  */
/* Outputs: m [T+1][d], dc [T+1][d], uc [T+1][d*d] (C = uc dc^2 uc^T),       */
/* optionally a, dr, ur.  Missing components are masked by row/column        */
/* selection of sqrtVinv (:51, only valid for diagonal V).        */
/* ------------------------------------------------------------------ */
void oracle_sqrt_svd(int n, const double *Mx, int inverse, double *out) {
  double *A = malloc(sizeof(double) * n * n), *sig = malloc(sizeof(double) * n);
  double *Vt = malloc(sizeof(double) * n * n);
  memcpy(A, Mx, sizeof(double) * n * n);
  svd_right(n, n, A, sig, Vt);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      double s = inverse ? 1.0 / sqrt(sig[i]) : sqrt(sig[i]);
      out[IDX(i, j, n)] = s * Vt[IDX(i, j, n)];
    }
  free(A); free(sig); free(Vt);
}

/* V_t / W_t streams (v_tstride / w_tstride != 0): step t runs with transformParams(p.copy(v = V_t)) or
 * transformParams(p.copy(w = W_t)) -- DlmFsv.ffbsSvd (DlmFsv.scala:208-228), DlmFsvSystem.ffbsSvd
 * (DlmFsvSystem.scala:176-208): observation t uses V + t v_tstride, the transition into it W + t w_tstride. */
void oracle_svd_filter_tv(int d, int p, int T, const double *F, long f_stride, const double *G,
                          const int *g_index, const double *dts, const double *V, long v_tstride, const double *W,
                          long w_tstride, const double *m0, const double *C0, const double *y, int raw_w_q2,
                          double *m_out, double *dc_out, double *uc_out, double *a_out,
                          double *dr_out, double *ur_out) {
  model_t M = {d, p, T, F, f_stride, G, g_index, dts};
  const int dd = d * d;
  double *sqrtVinv = malloc(sizeof(double) * p * p), *sqrtW = malloc(sizeof(double) * dd);
  oracle_sqrt_svd(p, V, 1, sqrtVinv);
  oracle_sqrt_svd(d, W, 0, sqrtW);
  const double *Wadv = raw_w_q2 ? W : sqrtW;
  double *stack = malloc(sizeof(double) * (2 * d > p + d ? 2 * d : p + d) * d);
  double *sig = malloc(sizeof(double) * d), *Vt = malloc(sizeof(double) * dd);
  double *a = malloc(sizeof(double) * d), *dr = malloc(sizeof(double) * d);
  double *ur = malloc(sizeof(double) * dd), *tmp = malloc(sizeof(double) * dd);
  double *Fm = malloc(sizeof(double) * d * p), *Vm = malloc(sizeof(double) * p * p);
  double *VF = malloc(sizeof(double) * p * d), *VFU = malloc(sizeof(double) * p * d);
  double *yo = malloc(sizeof(double) * p), *et = malloc(sizeof(double) * p);
  double *fv = malloc(sizeof(double) * d * p), *DU = malloc(sizeof(double) * dd);
  double *gain = malloc(sizeof(double) * d * p), *VtV = malloc(sizeof(double) * p * p);
  int *idx = malloc(sizeof(int) * p);

  /* initialiseState: SvdFilter.scala:83-95 */
  memcpy(tmp, C0, sizeof(double) * dd);
  svd_right(d, d, tmp, sig, Vt);
  memcpy(m_out, m0, sizeof(double) * d);
  for (int i = 0; i < d; ++i) dc_out[i] = sqrt(sig[i]);
  for (int i = 0; i < d; ++i)
    for (int j = 0; j < d; ++j) uc_out[IDX(i, j, d)] = Vt[IDX(j, i, d)];
  if (a_out) memcpy(a_out, m0, sizeof(double) * d);
  if (dr_out) memcpy(dr_out, dc_out, sizeof(double) * d);
  if (ur_out) memcpy(ur_out, uc_out, sizeof(double) * dd);

  for (int t = 0; t < T; ++t) {
    const double *m = m_out + (size_t)t * d, *dc = dc_out + (size_t)t * d;
    const double *uc = uc_out + (size_t)t * dd;
    double *mn = m_out + (size_t)(t + 1) * d, *dcn = dc_out + (size_t)(t + 1) * d;
    double *ucn = uc_out + (size_t)(t + 1) * dd;
    const double *Ft = F_at(&M, t), *Gt = G_at(&M, t);
    const double dt = dt_at(&M, t);
    if (w_tstride) {
      const double *Wt = W + (size_t)t * w_tstride;
      oracle_sqrt_svd(d, Wt, 0, sqrtW);
      Wadv = raw_w_q2 ? Wt : sqrtW;
    }
    if (v_tstride) oracle_sqrt_svd(p, V + (size_t)t * v_tstride, 1, sqrtVinv);
    /* advState: SvdFilter.scala:183-202 */
    if (dt == 0.0) {
      memcpy(a, m, sizeof(double) * d); memcpy(dr, dc, sizeof(double) * d);
      memcpy(ur, uc, sizeof(double) * dd);
    } else {
      mm(d, 1, d, Gt, d, 0, m, d, 0, a, d);
      /* stack (2d x d) = [diag(dc) uc^T G^T ; Wadv * sqrt(dt)] */
      mm(d, d, d, uc, d, 1, Gt, d, 1, tmp, d);
      for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) {
          stack[IDX(i, j, 2 * d)] = dc[i] * tmp[IDX(i, j, d)];
          stack[IDX(d + i, j, 2 * d)] = Wadv[IDX(i, j, d)] * sqrt(dt);
        }
      svd_right(2 * d, d, stack, dr, Vt);
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) ur[IDX(i, j, d)] = Vt[IDX(j, i, d)];
    }
    if (a_out) memcpy(a_out + (size_t)(t + 1) * d, a, sizeof(double) * d);
    if (dr_out) memcpy(dr_out + (size_t)(t + 1) * d, dr, sizeof(double) * d);
    if (ur_out) memcpy(ur_out + (size_t)(t + 1) * dd, ur, sizeof(double) * dd);
    /* updateState: SvdFilter.scala:38-68 */
    int pm = 0;
    for (int i = 0; i < p; ++i) {
      double yi = y[(size_t)t * p + i];
      if (yi == yi) { idx[pm] = i; yo[pm] = yi; ++pm; }
    }
    if (pm == 0) {
      memcpy(mn, a, sizeof(double) * d); memcpy(dcn, dr, sizeof(double) * d);
      memcpy(ucn, ur, sizeof(double) * dd);
      continue;
    }
    for (int j = 0; j < pm; ++j) {
      for (int i = 0; i < d; ++i) Fm[IDX(i, j, d)] = Ft[IDX(i, idx[j], d)];
      for (int i = 0; i < pm; ++i) Vm[IDX(i, j, pm)] = sqrtVinv[IDX(idx[i], idx[j], p)];
    }
    mm(pm, 1, d, Fm, d, 1, a, d, 0, et, pm);
    for (int i = 0; i < pm; ++i) et[i] = yo[i] - et[i];
    /* stack ((pm+d) x d) = [vm * fm.t * ur ; diag(1/dr)] */
    mm(pm, d, pm, Vm, pm, 0, Fm, d, 1, VF, pm);
    mm(pm, d, d, VF, pm, 0, ur, d, 0, VFU, pm);
    const int rows = pm + d;
    for (int j = 0; j < d; ++j) {
      for (int i = 0; i < pm; ++i) stack[IDX(i, j, rows)] = VFU[IDX(i, j, pm)];
      for (int i = 0; i < d; ++i) stack[IDX(pm + i, j, rows)] = (i == j) ? 1.0 / dr[i] : 0.0;
    }
    svd_right(rows, d, stack, sig, Vt);
    mm(d, d, d, ur, d, 0, Vt, d, 1, ucn, d);         /* uc = ur * rightVectors.t */
    for (int i = 0; i < d; ++i) dcn[i] = 1.0 / sig[i];
    /* fv = fm * vm.t * vm ; gain = (diag(dc) uc^T)^T (diag(dc) uc^T) fv */
    mm(pm, pm, pm, Vm, pm, 1, Vm, pm, 0, VtV, pm);
    mm(d, pm, pm, Fm, d, 0, VtV, pm, 0, fv, d);
    for (int j = 0; j < d; ++j)
      for (int i = 0; i < d; ++i) DU[IDX(i, j, d)] = dcn[i] * ucn[IDX(j, i, d)];
    mm(d, pm, d, DU, d, 0, fv, d, 0, VFU, d);         /* reuse: d x pm */
    mm(d, pm, d, DU, d, 1, VFU, d, 0, gain, d);
    mm(d, 1, pm, gain, d, 0, et, pm, 0, mn, d);
    for (int i = 0; i < d; ++i) mn[i] += a[i];
  }
  free(sqrtVinv); free(sqrtW); free(stack); free(sig); free(Vt); free(a); free(dr); free(ur);
  free(tmp); free(Fm); free(Vm); free(VF); free(VFU); free(yo); free(et); free(fv); free(DU);
  free(gain); free(VtV); free(idx);
}

void oracle_svd_filter(int d, int p, int T, const double *F, long f_stride, const double *G,
                       const int *g_index, const double *dts, const double *V, const double *W,
                       const double *m0, const double *C0, const double *y, int raw_w_q2,
                       double *m_out, double *dc_out, double *uc_out, double *a_out,
                       double *dr_out, double *ur_out) {
  oracle_svd_filter_tv(d, p, T, F, f_stride, G, g_index, dts, V, 0, W, 0, m0, C0, y, raw_w_q2, m_out, dc_out, uc_out, a_out, dr_out, ur_out);
}

/* ------------------------------------------------------------------ */
/* SVD backward sampler: SvdSampler.scala:15-60, rnorm :94-102.          */
/* Inputs: SVD filter records (m, dc, uc, a); sqrtW = sqrtSvd(W).         */
/* Draw = h + uh diag(dh) z.  h_out optional.                             */
/* Canonical factor (LAPACK leaves order and sign open): singular values  */
/* descending, and the largest-|component| of each column of the FINAL    */
/* factor (uc_T for the first draw, uh afterwards) positive.              */
/* Quirk Q9: the reference hands sqrt(W) (ps.w, SvdSampler.scala:71-73)    */
/* to a step whose algebra needs sqrt(W)^-1 (information form             */
/* H = (C^-1 + G^T W^-1 G)^-1); with W = I, as in the reference's test,     */
/* the two coincide.  literal_q9 != 0 restates the reference; 0 uses       */
/* sqrtInvSvd(W), the form that agrees with Smoothing.step.                */
/* ------------------------------------------------------------------ */
/* w_tstride != 0: the backward step from record t uses W + t w_tstride (the transition into observation t),
 * as DlmFsvSystem.ffbsSvd zips ps with filtered.init (DlmFsvSystem.scala:196-205). */
void oracle_svd_backward_sample_tv(int d, int T, const double *G, const int *g_index,
                                   const double *W, long w_tstride, const double *m, const double *dc,
                                   const double *uc, const double *a, const double *z,
                                   int literal_q9, double *theta, double *h_out, double *dh_out,
                                   double *uh_out) {
  const int dd = d * d;
  double *sqrtW = malloc(sizeof(double) * dd);
  oracle_sqrt_svd(d, W, literal_q9 ? 0 : 1, sqrtW);
  double *stack = malloc(sizeof(double) * 2 * dd), *sig = malloc(sizeof(double) * d);
  double *Vt = malloc(sizeof(double) * dd), *t1 = malloc(sizeof(double) * dd);
  double *t2 = malloc(sizeof(double) * dd), *uh = malloc(sizeof(double) * dd);
  double *dh = malloc(sizeof(double) * d), *u = malloc(sizeof(double) * d);
  double *v1 = malloc(sizeof(double) * d), *v2 = malloc(sizeof(double) * d);
  /* initialise: SvdSampler.scala:38-45 -> rnorm(m_T, dc_T, uc_T) */
  {
    const double *mT = m + (size_t)T * d, *dT = dc + (size_t)T * d;
    double *uT = t1;
    memcpy(uT, uc + (size_t)T * dd, sizeof(double) * dd);
    canon_cols(d, d, uT, d);
    for (int i = 0; i < d; ++i) {
      double s = mT[i];
      for (int k = 0; k < d; ++k) s += uT[IDX(i, k, d)] * dT[k] * z[(size_t)T * d + k];
      theta[(size_t)T * d + i] = s;
    }
    if (h_out) memcpy(h_out + (size_t)T * d, mT, sizeof(double) * d);
    if (dh_out) memcpy(dh_out + (size_t)T * d, dT, sizeof(double) * d);
    if (uh_out) memcpy(uh_out + (size_t)T * dd, uT, sizeof(double) * dd);
  }
  for (int t = T - 1; t >= 0; --t) {
    const double *Gn = G + (size_t)(g_index ? g_index[t] : 0) * dd;
    const double *mt = m + (size_t)t * d, *dct = dc + (size_t)t * d, *uct = uc + (size_t)t * dd;
    const double *a1 = a + (size_t)(t + 1) * d, *thn = theta + (size_t)(t + 1) * d;
    if (w_tstride) oracle_sqrt_svd(d, W + (size_t)t * w_tstride, literal_q9 ? 0 : 1, sqrtW);
    /* root = svd([sqrtW * g * uc ; diag(1/dc)]) : SvdSampler.scala:19-21 */
    mm(d, d, d, sqrtW, d, 0, Gn, d, 0, t1, d);
    mm(d, d, d, t1, d, 0, uct, d, 0, t2, d);
    for (int j = 0; j < d; ++j)
      for (int i = 0; i < d; ++i) {
        stack[IDX(i, j, 2 * d)] = t2[IDX(i, j, d)];
        stack[IDX(d + i, j, 2 * d)] = (i == j) ? 1.0 / dct[i] : 0.0;
      }
    svd_right(2 * d, d, stack, sig, Vt);
    mm(d, d, d, uct, d, 0, Vt, d, 1, uh, d);
    canon_cols(d, d, uh, d);
    for (int i = 0; i < d; ++i) dh[i] = 1.0 / sig[i];
    /* h = m + (du^T du) g^T sqrtW^T sqrtW (theta1 - a1), du = diag(dh) uh^T */
    for (int i = 0; i < d; ++i) u[i] = thn[i] - a1[i];
    mm(d, 1, d, sqrtW, d, 0, u, d, 0, v1, d);
    mm(d, 1, d, sqrtW, d, 1, v1, d, 0, v2, d);
    mm(d, 1, d, Gn, d, 1, v2, d, 0, v1, d);
    mm(d, 1, d, uh, d, 1, v1, d, 0, v2, d);
    for (int i = 0; i < d; ++i) v2[i] *= dh[i] * dh[i];
    mm(d, 1, d, uh, d, 0, v2, d, 0, v1, d);
    for (int i = 0; i < d; ++i) {
      double h = mt[i] + v1[i];
      if (h_out) h_out[(size_t)t * d + i] = h;
      double s = h;
      for (int k = 0; k < d; ++k) s += uh[IDX(i, k, d)] * dh[k] * z[(size_t)t * d + k];
      theta[(size_t)t * d + i] = s;
    }
    if (dh_out) memcpy(dh_out + (size_t)t * d, dh, sizeof(double) * d);
    if (uh_out) memcpy(uh_out + (size_t)t * dd, uh, sizeof(double) * dd);
  }
  free(sqrtW); free(stack); free(sig); free(Vt); free(t1); free(t2); free(uh); free(dh);
  free(u); free(v1); free(v2);
}

void oracle_svd_backward_sample(int d, int T, const double *G, const int *g_index,
                                const double *W, const double *m, const double *dc,
                                const double *uc, const double *a, const double *z,
                                int literal_q9, double *theta, double *h_out, double *dh_out,
                                double *uh_out) {
  oracle_svd_backward_sample_tv(d, T, G, g_index, W, 0, m, dc, uc, a, z, literal_q9, theta, h_out, dh_out, uh_out);
}

/* ------------------------------------------------------------------ */
/* Counter-based normals shared by oracle and engine: Philox4x32-10      */
/* (Salmon et al. 2011) + Box-Muller.  The reference draws from Breeze's  */
/* global MersenneTwister, which cannot be seeded at the draw sites       */
/* (Smoothing.scala:98,107), so the stream is ours to define:   */
/* normal #k of (series n, record t, component i) uses counter             */
/*   (lo32(n), hi32(n), t, i/2) and key (seed_lo, seed_hi); Box-Muller      */
/*   pair element i&1.                                                      */
/* ------------------------------------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

double oracle_normal(uint64_t seed, uint64_t series, uint32_t t, uint32_t i) {
  uint32_t c[4] = {(uint32_t)series, (uint32_t)(series >> 32), t, i >> 1};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  /* two 53-bit-ish uniforms in (0,1] and [0,1) from the four words */
  double u1 = ((double)c[0] * 4294967296.0 + (double)c[1] + 1.0) * (1.0 / 18446744073709551616.0);
  double u2 = ((double)c[2] * 4294967296.0 + (double)c[3]) * (1.0 / 18446744073709551616.0);
  double r = sqrt(-2.0 * log(u1));
  double ang = 6.283185307179586476925286766559 * u2;
  return (i & 1) ? r * sin(ang) : r * cos(ang);
}

/* The engine's d-Inverse-Gamma step (bayesian_dlms_amd/csrc/dlm_gibbs.hip), restated: Marsaglia-Tsang Gamma(a, 1) on
 * Philox draws keyed (seed, series, iteration, component, attempt); value = rate / gamma, i.e.
 * InverseGamma(shape, rate).draw = 1 / Gamma(shape, 1 / rate).draw (InverseGamma.scala:14).  The reference's own
 * generator cannot be seeded: this pins the engine's stream, the distribution is checked separately. */
static void gibbs_rand(uint64_t seed, uint64_t series, uint64_t iteration, uint32_t comp, uint32_t attempt,
                       uint32_t which, double *u1, double *u2) {
  uint32_t c[4] = {(uint32_t)series, (uint32_t)(series >> 32), (uint32_t)iteration, comp * 2048u + attempt * 2u + which};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32) ^ 0x47494242u);
  *u1 = ((double)c[0] * 4294967296.0 + (double)c[1] + 1.0) * (1.0 / 18446744073709551616.0);
  *u2 = ((double)c[2] * 4294967296.0 + (double)c[3]) * (1.0 / 18446744073709551616.0);
}
double oracle_gamma_unit(double a, uint64_t seed, uint64_t series, uint64_t iteration, uint32_t comp) {
  double boost = 1.0, u1, u2, w1, w2;
  if (a < 1.0) { gibbs_rand(seed, series, iteration, comp, 1023u, 0u, &u1, &u2); boost = pow(u1, 1.0 / a); a += 1.0; }
  const double dd = a - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
  for (uint32_t k = 0; k < 1023u; ++k) {
    gibbs_rand(seed, series, iteration, comp, k, 0u, &u1, &u2);
    const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
    double v = 1.0 + cc * x;
    if (v <= 0.0) continue;
    v = v * v * v;
    gibbs_rand(seed, series, iteration, comp, k, 1u, &w1, &w2);
    if (log(w1) < 0.5 * x * x + dd - dd * v + dd * log(v)) return dd * v * boost;
  }
  return dd * boost;
}
/* dinvGammaStep for one series (Gibbs.scala:41-48, :72-77): stats = [ssy (p) | n (p) | ss (d) | T]; vdiag [p], wdiag [d] */
void oracle_dinvgamma_step(int d, int p, const double *stats, double av, double bv, double aw, double bw, uint64_t seed,
                           uint64_t series, uint64_t iteration, double *vdiag, double *wdiag) {
  const int L = 2 * p + d + 1;
  for (int j = 0; j < p; ++j)
    vdiag[j] = (bv + 0.5 * stats[j]) / oracle_gamma_unit(av + 0.5 * stats[p + j], seed, series, iteration, (uint32_t)j);
  for (int i = 0; i < d; ++i)
    wdiag[i] = (bw + 0.5 * stats[2 * p + i]) / oracle_gamma_unit(aw + 0.5 * stats[L - 1], seed, series, iteration, (uint32_t)(p + i));
}

/* Dlm.simulateRegular / simStep (Dlm.scala:245-292) with lower-Cholesky draws on the engine's Philox convention
 * (record t, component i: i < d state noise, d <= i < d + p observation noise; record 0 = the initial state). */
int oracle_simulate(int d, int p, int T, const double *F, long f_stride, const double *G, const int *g_index,
                    const double *dts, const double *V, const double *W, const double *m0, const double *C0,
                    uint64_t seed, uint64_t series, double *x_out /* [T+1][d] */, double *y_out /* [T][p] */) {
  model_t M = {d, p, T, F, f_stride, G, g_index, dts};
  double *Lw = malloc(sizeof(double) * d * d), *Lc = malloc(sizeof(double) * d * d), *Lv = malloc(sizeof(double) * p * p);
  int rc = chol_lower(d, W, Lw) | chol_lower(d, C0, Lc) | chol_lower(p, V, Lv);
  for (int i = 0; i < d; ++i) {
    double acc = m0[i];
    for (int k = 0; k <= i; ++k) acc = fma(Lc[i + k * d], oracle_normal(seed, series, 0u, (uint32_t)k), acc);
    x_out[i] = acc;
  }
  for (int t = 0; t < T; ++t) {
    const double *Gt = G_at(&M, t), *Ft = F_at(&M, t), *x = x_out + (size_t)t * d;
    double *xn = x_out + (size_t)(t + 1) * d;
    const double sdt = sqrt(dt_at(&M, t));
    for (int i = 0; i < d; ++i) {
      double acc = 0.0;
      for (int k = 0; k < d; ++k) acc = fma(Gt[i + k * d], x[k], acc);
      for (int k = 0; k <= i; ++k) acc = fma(Lw[i + k * d] * sdt, oracle_normal(seed, series, (uint32_t)(t + 1), (uint32_t)k), acc);
      xn[i] = acc;
    }
    for (int j = 0; j < p; ++j) {
      double acc = 0.0;
      for (int k = 0; k < d; ++k) acc = fma(Ft[k + j * d], xn[k], acc);
      for (int k = 0; k <= j; ++k) acc = fma(Lv[j + k * p], oracle_normal(seed, series, (uint32_t)(t + 1), (uint32_t)(d + k)), acc);
      y_out[(size_t)t * p + j] = acc;
    }
  }
  free(Lw); free(Lc); free(Lv);
  return rc;
}

void oracle_normals(uint64_t seed, uint64_t series, int T1, int d, double *z) {
  for (int t = 0; t < T1; ++t)
    for (int i = 0; i < d; ++i) z[(size_t)t * d + i] = oracle_normal(seed, series, (uint32_t)t, (uint32_t)i);
}

/* ------------------------------------------------------------------ */
/* Batched filter + smoother used as bench.py's cpu_baseline ("port"):   */
/* same operation sequence as the Scala code per series, OpenMP over      */
/* series when compiled with -fopenmp.  Shared model and parameters.      */
/* ------------------------------------------------------------------ */
int oracle_filter_smooth_batch(int N, int d, int p, int T, const double *F, const double *G,
                               const double *V, const double *W, const double *m0,
                               const double *C0, const double *y, /* [N][T][p] */
                               double *filt, double *smooth /* [N][T+1][d+d*d] or NULL */) {
  int bad = 0;
  const size_t rec = (size_t)d + (size_t)d * d, T1 = (size_t)T + 1;
  /* one set of per-series arrays per THREAD (5.5 MB at d = 13, T = 1000): allocated inside the loop they were an mmap / munmap pair
   * per series and the batch scaled 26x on 256 threads; static schedule: equal series, equal work */
#pragma omp parallel reduction(+ : bad)
  {
    double *m = malloc(sizeof(double) * T1 * d), *C = malloc(sizeof(double) * T1 * d * d);
    double *a = malloc(sizeof(double) * T1 * d), *R = malloc(sizeof(double) * T1 * d * d);
    double *s = malloc(sizeof(double) * T1 * d), *S = malloc(sizeof(double) * T1 * d * d);
#pragma omp for schedule(static)
    for (int n = 0; n < N; ++n) {
      bad += oracle_kf_filter(d, p, T, F, 0, G, NULL, NULL, V, W, m0, C0, y + (size_t)n * T * p, m, C,
                              a, R, NULL, NULL) != 0;
      bad += oracle_smoother(d, T, G, NULL, m, C, a, R, 0, s, S) != 0;
      if (filt && smooth)
        for (size_t t = 0; t < T1; ++t) {
          double *fr = filt + ((size_t)n * T1 + t) * rec, *sr = smooth + ((size_t)n * T1 + t) * rec;
          memcpy(fr, m + t * d, sizeof(double) * d);
          memcpy(fr + d, C + t * d * d, sizeof(double) * d * d);
          memcpy(sr, s + t * d, sizeof(double) * d);
          memcpy(sr + d, S + t * d * d, sizeof(double) * d * d);
        }
    }
    free(m); free(C); free(a); free(R); free(s); free(S);
  }
  return bad;
}

/* threads the batch above runs on (0: leave it; returns the number in force) -- bench.py times one thread beside all of them */
#ifdef _OPENMP
#include <omp.h>
int oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int oracle_set_threads(int n) { (void)n; return 1; }
#endif
