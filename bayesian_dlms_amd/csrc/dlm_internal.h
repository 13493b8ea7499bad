// Internal declarations shared by the engine's translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dlm {

// Device-pointer view of one batched call.  All pointers are device pointers.
struct KArgs {
  int d, p, T, N;
  const double* F; long long f_stride;
  const double* G; int n_g; const int* g_index; const double* dt;
  const double* V; long long v_stride;
  const double* W; long long w_stride;
  long long v_tstride, w_tstride;   // per-time strides of V_t / W_t (0 = time-invariant)
  const double* m0; long long m0_stride;
  const double* C0; long long c0_stride;
  const double* y;     // [N][T][p]
  const double* z;     // [N][T+1][d] injected normals or nullptr
  const double* filt_in;  // [N][T+1][d+dd] (smoother / sampler input)
  double* filt;        // outputs (nullable where optional)
  double* prior;
  double* fq;
  double* smooth;
  double* theta;
  double* cond;
  double* stats;
  double* loglik;      // [N] prediction-error log-likelihood (nullable); with it, filt may be null (nothing stored)
  int* status;
  int packed;          // structured d <= 15 fast path only.  bit 0: filt / filt_in hold PACKED records (engine-internal workspace,
                       // or DLM_OPT_PACKED_SYM); bit 1: the smoothed records are written packed too (DLM_OPT_PACKED_SYM)
  const struct SparseBig* spb;   // tiled path: [2 n_g] row / column tables of a structured G, or nullptr (dense G)
  int spb_k;           // largest nonzero count per row / column over those tables (1..4)
  const struct SparseF* spf;     // structured (time-invariant) F of the tiled path: column / row tables, or nullptr
  int spf_k;           // largest nonzero count per column / row of F (1..4)
  unsigned flags;
  unsigned long long seed, series_offset;
  // Shared-covariance path (dlm_sparse16.hip): route[n] != 0 marks a series with a missing observation.  The
  // mean-only kernels set it and skip such a series; the per-series kernels launched behind them with route_take = 1 serve
  // exactly those series (and leave at once for the others).  nullptr: every series is served (the ordinary calls).
  unsigned char* route;
  int route_take;
  unsigned long long* counters;   // engine-owned [4], zeroed per call (nullable): steady (short) steps taken by the forward [0] / backward [1] kernel,
                                  // series served by the shared-covariance kernels [2], series sent to their own full recursion [3]
  const unsigned char* plain;     // structured d <= 15 path, regular grid (nullable): plain[n] = 1 -- series n misses enough observations (more than T / 256)
                                  // that its covariance recursion is not worth testing for convergence: it takes every step in full, in the
                                  // backward kernel's instantiation without the shortcut's machinery (k_count_gaps fills it from the data alone)
  const unsigned char* keep_cov;  // per-wave filter, 16 <= d <= 48 (nullable): a dlm_ffbs_batch call that keeps no records and draws against shared factors
                                  // reads only the MEANS of the series without a gap -- once such a series' covariance has settled its steady steps
                                  // store the mean alone (keep_cov[n] = 0); the series with a gap (1) write whole records for their own sampler
  double* ktab;                   // per-wave filter, with keep_cov: the steady gain K^T (16 PT rows of the LDS image) of the batch -- WRITTEN by the series
                                  // of zeros where its covariance recursion settles (with settle_step), and, when set for the batch's own filter, the
                                  // sign that its series without a gap leave at that step: k_steady_filter_w48 carries their means on from there
  int* leave_step;                // per-wave filter, with ktab set for the batch's own filter: [N] the step at which series n left k_filter_w48 (its own
                                  // convergence test; T: it never did) -- k_steady_filter_w48 carries series n on from THAT record, so that no assumption
                                  // about every series settling at the zero series' step is needed
  int stretches;                  // backward sampler: 1 = every stretch of steps starts from scratch (dlm_sampler16.hip: SF_STRETCH), set for the calls
                                  // whose parameters allow a shared-factor table -- its stretches are made side by side, and a series that computes
                                  // its own factors in such a call follows the same rule, so that the two agree bit for bit
  int* settle_step;               // the filter of the shared-factor tables' series of zeros (nullptr otherwise): the kernel stops when its covariance
                                  // recursion has settled and leaves here the last record it wrote -- every later record would repeat that covariance
};

// Packed record of the structured fast path's internal workspaces: [m (d) | lower triangle of C by rows], padded to a
// multiple of 16 B (the backward pass fetches records by 16-byte LDS-DMA pieces).
__host__ __device__ inline int packed_rec_bytes(int d) { return ((d + d * (d + 1) / 2) * 8 + 15) & ~15; }

__host__ __device__ inline int stats_len(int d, int p, unsigned flags) {
  return 2 * p + ((flags & (1u << 4)) ? d * d : d) + 1;  // DLM_OPT_STATS_OUTER
}

// ---- generic wave-per-series kernels (any d <= 64, p <= 64), dlm_generic.hip ----------
size_t generic_filter_lds_bytes(int d, int p);
size_t generic_smoother_lds_bytes(int d, int p);
size_t generic_sampler_lds_bytes(int d, int p);
hipError_t launch_generic_filter(const KArgs& a, hipStream_t s);
hipError_t launch_generic_smoother(const KArgs& a, hipStream_t s);
hipError_t launch_generic_sampler(const KArgs& a, hipStream_t s);
hipError_t launch_generic_simulate(const KArgs& a, hipStream_t s);   // a.theta = x (nullable), a.smooth = y
hipError_t launch_stats_pool(const double* stats, int N, int L, double* pooled, hipStream_t s);

// ---- specialised d <= 16, p == 1 kernels on the fp64 MFMA layout, dlm_mfma16.hip ------
bool mfma16_supported(const KArgs& a);   // dense-G kernels: regular grid only
bool fast_shape(const KArgs& a);         // d <= 15, p == 1, time-invariant F
// `side` [N][T+1][2] carries (e_t/Q_t, 1/Q_t) from the forward to the backward pass (NaN = no update)
hipError_t launch_mfma16_filter(const KArgs& a, double* side, hipStream_t s);
hipError_t launch_mfma16_smoother(const KArgs& a, const double* side, hipStream_t s);

// ---- structured-G variant of the fast path (<= 4 nonzeros per row/column), dlm_sparse16.hip
struct SparseT { int K; int pad; int idx[16][4]; double val[16][4]; };
int sparse16_analyse(const double* G_host, int d, SparseT* rows, SparseT* cols);
// sparse16 kernels take tabs[2 * gi + 0] = rows of G_gi, tabs[2 * gi + 1] = columns of G_gi; K = max count
// xplus != nullptr: simulation-smoother forward pass (also writes x+ [N][T+1][d])
hipError_t launch_sparse16_filter(const KArgs& a, int K, const SparseT* rows_dev, double* side, double* xplus, hipStream_t s);
// tabs_dev[0] = rows of G, tabs_dev[1] = columns of G
hipError_t launch_sparse16_simsmooth(const KArgs& a, int K, const SparseT* tabs_dev, const double* side, const double* xplus, hipStream_t s);
// the reference-form backward sampler (a.filt_in -> theta / cond / stats), dlm_sampler16.hip; tabs_dev as above
hipError_t launch_sparse16_sampler(const KArgs& a, int K, const SparseT* tabs_dev, hipStream_t s);
hipError_t launch_small_mv_sampler(const KArgs& a, hipStream_t s);   // the same for d <= 15, p >= 2 (a.spb tables)
// RTS smoother (Smoothing.smoothStep) from filter records alone on the same register tiles; textbook or literal Q1
hipError_t launch_sparse16_rts(const KArgs& a, int K, const SparseT* tabs_dev, hipStream_t s);
hipError_t launch_small_mv_rts(const KArgs& a, hipStream_t s);
hipError_t launch_sparse16_smoother(const KArgs& a, int K, const SparseT* cols_dev, const double* side, hipStream_t s);
hipError_t launch_sparse16_count_gaps(const KArgs& a, unsigned char* plain, hipStream_t s);   // KArgs::plain of the call from its observations
// ---- shared covariance sequence: with parameters shared by the batch and no missing observation C_t, R_t, K_t,
// Q_t, P_t and S_t do not depend on the data.  ONE wave runs the covariance recursions (the per-series kernels' own code on a
// series of zeros: the same arithmetic, bit for bit) into L2-resident tables, and every series runs only its mean recursions
// against them.  A series that meets a missing observation is marked in KArgs::route and served by the per-series kernels.
struct CovTabs {
  double* ftab;   // [T+1] rows of frow bytes: [filtered record of the covariance-only run (mean slots 0): C_t | 16 doubles: R_t F (full
                  //   step) or K_t (steady step) per component, [15] = +-1/Q_t (negative: steady step)]
  double* btab;   // [T+1] rows of brow bytes: [smoothed record of the covariance-only run: S_t | 16 doubles: K_t = C_t F / V per
                  //   component, [15] = 1 where the step was a steady one | C_t record]
  double* cside;  // [T+1][2]  side records of the covariance-only forward run (0, +-1/Q_t), read by its backward run
  double* eq;     // [N][T+1]  per series: e_t / Q_t, forward -> backward
  double* mc;     // fused call: the filtered means once more, compact -- [ceil(N / 4)][T+1][4][16] (512 bytes per wave and step) --
                  //   for the backward kernel (a sequential stream instead of 112 bytes out of every 1456-byte record); nullptr: dlm_filter_batch
  double* sc;     // (unused: the smoothed means of the record-writer experiment, profiles/r03_notes.md)
  int frow, brow;
  const int* skip;   // nullable: nonzero on the device -- the tables are not wanted after all (most series of the call have a gap): the covariance-only filter returns at once
};
size_t covtabs_doubles(int d, int T);   // doubles of the table block (everything but eq)
void covtabs_carve(double* base, int d, int T, CovTabs& t);
bool shared_cov_eligible(const KArgs& a);   // regular grid, time-invariant F / V / W, V, W, C0 shared by the batch, dense records, no prior / forecast records
hipError_t launch_sparse16_cov_filter(const KArgs& a, int K, const SparseT* rows_dev, const CovTabs& tabs, hipStream_t s);
hipError_t launch_sparse16_cov_smoother(const KArgs& a, int K, const SparseT* cols_dev, const CovTabs& tabs, hipStream_t s);
hipError_t launch_sparse16_mean_filter(const KArgs& a, int K, const SparseT* rows_dev, const CovTabs& tabs, hipStream_t s);
hipError_t launch_sparse16_mean_smoother(const KArgs& a, int K, const SparseT* cols_dev, const CovTabs& tabs, hipStream_t s);

// ---- shared factors of the reference-form backward sampler, dlm_sampler16.hip: J_t, H_t and chol(H_t) depend on
// the filtered covariances alone.  With parameters shared by the batch ONE wave runs k_sampler_sp16 on the records of a series of
// zeros into a table; every series without a missing observation draws with the mean-only kernel (four series per wave), the
// others with k_sampler_sp16 as always.  Draw for draw the per-series kernel's results, bit for bit.
struct SampTabs {
  double* rows;          // [T+1] rows of 16 x 34 doubles: per component c [ J_t^T[.][c] (16) | L_t[c][.] (16) | pad (2) ]
  double* zrec;          // [T+1][d + d^2] filter records of the series of zeros
  double* zeros;         // max(T, 16) zeros: its observations and initial mean
  unsigned char* need;   // [T+1] 1: row t was written (a full step); 0: the factors of the last row above it
  int* status;           // status of the zero series' two kernels, for every series served by the tables
  double* z4;            // the normals of the call, made while the batch is filtered.  d <= 15 (k_normals4): [ceil(N / 4)][T+1][4][16], 512 bytes per step
                         //   and wave of the draw kernel; 16 <= d <= 48 (k_normals_rows): [N][T+1][d].  nullptr: injected normals (KArgs::z)
  int zstride;           // d <= 15, table kernel: bytes between the records of the series of zeros (0: d + d^2 doubles; the rows of CovTabs::ftab otherwise)
  const double* mc4;     // d <= 15, draw kernel: the filtered means come from the mean-only forward kernel's compact stream ([ceil(N / 4)][T+1][4][16],
                         //   CovTabs::mc) instead of the filter records: a dlm_ffbs_batch call that does not want the records (filt_ws == NULL)
  int marked;            // KArgs::route holds this call's gap marks already (the draw launch does not mark again)
  double* ktab;          // 16 <= d <= 48: the steady gain K^T as the zero series' filter leaves it (KArgs::ktab), for k_steady_filter_w48
  int* settle;           // index of the last record of zrec that was written (KArgs::settle_step): the records above it repeat its covariance
};
bool sampler_shared_model_ok(const KArgs& a);   // V, W, C0 shared by the batch, regular grid, time-invariant model (what KArgs::stretches follows)
bool sampler_shared_eligible(const KArgs& a);   // ... and this call can use the table (no conditional-moment records, flags)
size_t sampler_shared_ws_bytes(const KArgs& a);
void sampler_shared_carve(void* ws, const KArgs& a, SampTabs& tb);
// the tables: filter on zeros, then the sampler with its export on (both one wave; stream s)
hipError_t launch_sampler_shared_tables(const KArgs& a, int K, const SparseT* tabs_dev, const SampTabs& tb, hipStream_t s);
// the same from covariance records that exist already (crec, stride bytes apart: the forward table of the shared-covariance kernels)
hipError_t launch_sampler_shared_tables_from(const KArgs& a, int K, const SparseT* tabs_dev, SampTabs tb, const double* crec, int stride, hipStream_t s);
size_t sampler_shared_normals_bytes(const KArgs& a);
hipError_t launch_sampler_shared_normals(const KArgs& a, double* z4, hipStream_t s);   // the Philox normals of every series and step, in the draw kernel's layout
// a.route [N] is filled here (series with a missing observation), the mean-only kernel draws for the others, k_sampler_sp16 for these
hipError_t launch_sampler_shared_draw(const KArgs& a, int K, const SparseT* tabs_dev, const SampTabs& tb, hipStream_t s);

// Shared factors of the RTS smoother (Smoothing.scala:38-47, textbook or literal Q1; dlm_sampler16.hip): k_smoother_rts16 runs once, on the
// filter records of a series of zeros, into these tables; every series without a missing observation runs the mean recursion against them
// (k_mean_rts16, four series per wave), the others k_smoother_rts16 as always.  Bit for bit the per-series kernel's records.
struct RtsTabs {
  double* jrows;         // [T+1] rows of 16 x 18 doubles: per component c [ J_t^T[.][c] (16) | pad (2) ], written where need[t]
  double* srec;          // [T+1][d + d^2] smoothed records of the series of zeros: S_t behind d doubles nobody reads
  unsigned char* need;   // [T+1] 1: row t of jrows was written (a full step); 0: J of the last row above it
  int* status;           // status of the zero series' two kernels, for every series served by the tables
  const double* crec;    // the covariances exist already: records crec_stride bytes apart (the forward table of the shared-covariance kernels, CovTabs::ftab)
  int crec_stride;
  int* gaps;             // the number of series with a missing observation (launch_rts_shared_mark)
  int* skip;             // 1: more than half of the series have one -- the tables are not made, every series is routed to the per-series kernel
};
bool rts_shared_eligible(const KArgs& a, bool textbook);   // textbook: the call asks for S = C - J (R+ - S+) J^T (the default): a larger batch is needed to pay for the tables
size_t rts_shared_ws_bytes(const KArgs& a);
void rts_shared_carve(void* ws, const KArgs& a, RtsTabs& tb);
// the tables: the covariance-only filter into ctb.ftab (launch_sparse16_cov_filter), then the smoother with its export on (both one wave)
hipError_t launch_rts_shared_cov(const KArgs& a, int K, const SparseT* tabs_dev, RtsTabs& tb, const CovTabs& ctb, hipStream_t s);
hipError_t launch_rts_shared_tables(const KArgs& a, int K, const SparseT* tabs_dev, const RtsTabs& tb, hipStream_t s);
hipError_t launch_rts_shared_mark(const KArgs& a, unsigned char* route, const RtsTabs& tb, hipStream_t s);   // KArgs::route of the call from its observations; tb.gaps, tb.skip
hipError_t launch_rts_shared_means(const KArgs& a, int K, const SparseT* tabs_dev, const RtsTabs& tb, bool own_rts, hipStream_t s);

// the same for 16 <= d <= 48 (dlm_wave48.hip): rows of 64 x (4 DT^2 + 16 DT) doubles, per lane [ J^T tiles | row `lane` of L ]
bool wave48_sampler_shared_model_ok(const KArgs& a);
bool wave48_sampler_shared_eligible(const KArgs& a);
size_t wave48_sampler_shared_ws_bytes(const KArgs& a);
void wave48_sampler_shared_carve(void* ws, const KArgs& a, SampTabs& tb);
hipError_t launch_wave48_sampler_shared_tables(const KArgs& a, const SampTabs& tb, hipStream_t s, hipEvent_t after_filter = nullptr);   // after_filter: recorded behind the zero series' filter (ktab, settle)
hipError_t launch_wave48_sampler_shared_draw(const KArgs& a, const SampTabs& tb, hipStream_t s);
size_t wave48_ktab_doubles(const KArgs& a);
// the steady steps of the series without a gap, from record *settle on: a = G m, e = y - F^T a, m = a + K e (the per-wave filter's steady step,
// operation for operation), the mean stored into the records' mean slots; k.keep_cov marks the series, ktab / settle come from the zero series
hipError_t launch_wave48_steady_filter(const KArgs& a, const double* ktab, const int* settle, hipStream_t s);
hipError_t launch_wave48_mark_gaps(const KArgs& a, unsigned char* route, hipStream_t s);   // route[n] = 1: series n has a missing observation component
size_t wave48_sampler_shared_normals_bytes(const KArgs& a);
hipError_t launch_wave48_sampler_shared_normals(const KArgs& a, double* z, hipStream_t s);   // rows [N][T+1][d], SampTabs::z4 of these calls

// ---- multivariate path: workgroup per series, MFMA-tiled GEMMs from LDS, dlm_tiled.hip --------
// Nonzeros of the rows (`rows`) and of the columns (`cols`) of a d x d G with at most 4 per row and column (every
// model the reference can build): the congruences G C G^T and G^T M G are then two gather passes instead of two
// dense MFMA products.  sparse48_analyse returns the largest count, or 99 if G is not that sparse.
struct SparseBig { int K; int pad; int idx[48][4]; double val[48][4]; };
int sparse48_analyse(const double* G_host, int d, SparseBig* rows, SparseBig* cols);
// The same for the observation matrix F (d x p): block models composed with |*| (Dlm.scala:197-208) have one component's
// few nonzeros in every column.  cidx/cval: nonzeros of column j (states loading on observation j); ridx/rval: nonzeros of
// row i (observations state i loads on).  sparsef_analyse returns the largest count, or 99 if F is not that sparse.
struct SparseF { int K; int pad; int cidx[32][4]; double cval[32][4]; int ridx[48][4]; double rval[48][4]; };
int sparsef_analyse(const double* F_host, int d, int p, SparseF* out);
bool tiled_supported(const KArgs& a);
// innov [N][T][p] (nullable for the filter): innovations y_t - f_t (NaN = missing) handed from the forward to the backward pass
hipError_t launch_tiled_filter(const KArgs& a, double* innov, hipStream_t s);
hipError_t launch_tiled_smoother(const KArgs& a, const double* innov, hipStream_t s);
// ---- one wavefront per series, register-resident tiles (structured G, 16 <= d <= 48), dlm_wave48.hip ----
bool wave48_small_shape(const KArgs& a);   // d <= 15 with 2 <= p <= 32: the per-wave kernels with one tile per dimension
bool wave48_small_ok(const KArgs& a);      // ... and a structured G
bool wave48_filter_supported(const KArgs& a);
hipError_t launch_wave48_filter(const KArgs& a, int K, double* innov, hipStream_t s);
bool wave48_smoother_supported(const KArgs& a);
hipError_t launch_wave48_smoother(const KArgs& a, int K, const double* innov, hipStream_t s);
bool wave48_simsmooth_supported(const KArgs& a);
hipError_t launch_wave48_simsmooth(const KArgs& a, int K, double* xplus, double* ystar, hipStream_t s);
bool wave48_sampler_supported(const KArgs& a);   // the reference-form backward sampler on register tiles (16 <= d <= 48, structured G)
hipError_t launch_wave48_sampler(const KArgs& a, hipStream_t s);
// simulation-smoother FFBS (forward SIM pass + mean-only backward pass); xplus [N][T+1][d], ystar [N][T][p]
hipError_t launch_tiled_simsmooth(const KArgs& a, double* xplus, double* ystar, hipStream_t s);

// ---- SVD filter / sampler (one-sided Jacobi in LDS), dlm_svd.hip ----------------------
bool svd_supported(const KArgs& a);      // d <= 48, p <= 32
size_t svd_filter_lds_bytes(int d, int p);
hipError_t launch_svd_filter(const KArgs& a, double* svd_rec, hipStream_t s);
hipError_t launch_svd_sampler(const KArgs& a, const double* svd_rec, hipStream_t s);
// shared factors (parameters shared by the batch, no missing observation): the decompositions once per call, a mean-only kernel per series
bool svd_shared_eligible(const KArgs& a);
size_t svd_shared_ws_doubles(const KArgs& a);
hipError_t launch_svd_filter_shared(const KArgs& a, double* svd_rec, double* ws, unsigned char* route, hipStream_t s);

// ---- one lane per series for d <= 3, p = 1 (dlm_lane.hip): filter (+ prior / forecast records, log-likelihood), RTS smoother
bool lane_supported(const KArgs& a);
hipError_t launch_lane_filter(const KArgs& a, hipStream_t s);
hipError_t launch_lane_smoother(const KArgs& a, hipStream_t s);   // a.filt_in -> a.smooth; no side buffer
hipError_t launch_lane_sampler(const KArgs& a, hipStream_t s);   // the literal backward sampler (a.filt_in -> theta / cond / stats)
hipError_t launch_lane_simsmooth(const KArgs& a, double* xplus, hipStream_t s);   // simulation-smoother FFBS + statistics (time-invariant V, W)

// ---- scalar AR(1) FFBS, one lane per series (FilterAr.scala:15-82), dlm_ar1.hip -----------------
// times != nullptr: the Ornstein-Uhlenbeck variant on that (shared, irregular) time grid (FilterOu.scala:7-79)
hipError_t launch_ar1_ffbs(int N, int T, const double* times, const double* y, const double* v, long long v_stride,
                           const double* sv, long long sv_stride, const double* z, unsigned long long seed,
                           unsigned long long series_offset, double* filt, double* theta, int* status, hipStream_t s);

// ---- d-Inverse-Gamma conjugate draws on the device (Gibbs.scala:23-78), dlm_gibbs.hip --------------
hipError_t launch_dinvgamma_step(int d, int p, int N, const double* stats, double av, double bv, double aw, double bw,
                                 unsigned long long seed, unsigned long long series_offset, unsigned long long iteration,
                                 double* Vout, double* Wout, hipStream_t s);

// ---- KalmanFilter.likelihood literally (transition density of the filtered means), dlm_loglik.hip ------
size_t loglik_q7_ws_bytes(const KArgs& a);
hipError_t launch_loglik_q7(const KArgs& a, const double* records, void* ws, hipStream_t s);   // a.loglik [N] <- records [N][T+1][d+dd]

// ---- when may a converged covariance recursion stop being recomputed? -------------------------------------------------
// The reference recomputes C_t, R_t, K_t, Q_t (KalmanFilter.scala:64-107) and the smoother's matrices (Smoothing.scala:31-47)
// at every step.  The kernels may stop once the recursion has converged -- but a small ONE-STEP change says little about the
// distance to the limit when the recursion contracts slowly (small W / V): what remains is the geometric tail
// delta * rho / (1 - rho).  settle_test bounds that tail.  `delta` is the largest change of an entry over one step and
// `scale` the largest entry of the matrix (both wave-uniform); tests come every `period` steps on an uninterrupted regular
// stretch (settle_reset otherwise).  With r = delta_k / delta_{k-1} (= rho^period) and 1 / (1 - rho) <= period / (1 - r):
//     settled  <=>  period * delta_k / (1 - r)  <=  DLM_SETTLE_TOL * scale,
// where r is the LARGER of the last two estimates, each taken from two consecutive tests whose changes stand clear of the
// rounding noise of the recursion's own arithmetic (`noise`, relative to scale: 32 eps for the covariance recursions; a
// change below it counts as that level and gives no new estimate).  At least three tests, r < 1.  A recursion too slow to
// be resolved above the noise floor (rho > ~0.993 per step at period 4) never settles: it is recomputed at every step like
// the reference's.  What the rule assumes: from the test on, successive changes shrink at least as fast as the larger of
// the two measured ratios (the convergence is geometric by then).  NumPy replay against the every-step recursion:
// tools/settle_replay.py (the frozen matrix stays within DLM_SETTLE_TOL * max|C| for W scaled by 1 .. 1e-6 and V = 1e4 at
// T up to 40 000; the round-2 rule -- one step's change <= 1e-13 Q -- was off by up to 6e-9 there); GPU tests of every
// kernel that freezes, in the slow regime: tests/test_settle_gpu.py.
constexpr double DLM_SETTLE_TOL = 1e-12;
// The state of the test -- the relative change at the previous test (< 0: none) and the last two estimates of rho^period
// (2: none yet) -- lives in three floats of the wave's LDS (`st`): kept in registers it cost the d <= 15 backward kernel its
// fifth wave per SIMD.  Every lane reads and writes the same values (the maxima are wave-uniform): no lane masking, and the
// LDS queue of a wave is in order.
constexpr int SETTLE_FLOATS = 4;
__device__ __forceinline__ float settle_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ void settle_reset(float* st) { st[0] = -1.f; st[1] = 2.f; st[2] = 2.f; }
// (single precision throughout -- the callers bring their maxima into float range with a power-of-two factor --: a double
// division at the point where a kernel holds the most live values costs it a dozen registers)
__device__ __forceinline__ bool settle_test(float* st, float delta, float scale, int period, float noise = 7.1e-15f) {
  const float x = delta * __builtin_amdgcn_rcpf(scale);
  float prev = st[0], rate = st[1], rate_prev = st[2];
  bool ok = false;
  if (!(x >= 0.f) || !(x < 1e30f)) { prev = -1.f; rate = 2.f; rate_prev = 2.f; }   // NaN / Inf / empty matrix: start over
  else {
    const float xe = fmaxf(x, noise);
    if (prev > 8.f * noise && x >= noise) { rate_prev = rate; rate = xe * __builtin_amdgcn_rcpf(prev); }   // a clean pair
    const float r = fmaxf(rate, rate_prev);
    ok = r < 1.f && (float)period * xe <= (float)DLM_SETTLE_TOL * (1.f - r);
    ok |= (x == 0.f && prev == 0.f);                             // a fixed point reached bit for bit
    prev = x > 0.f ? xe : 0.f;
  }
  st[0] = prev; st[1] = rate; st[2] = rate_prev;
  return __builtin_amdgcn_readfirstlane((int)ok) != 0;
}
// NaN-proof |a - b| for the maxima that feed settle_test (fmax drops a NaN operand)
__device__ __forceinline__ double settle_absdiff(double a, double b) { const double v = fabs(a - b); return v == v ? v : __builtin_inf(); }
// 2^-k for x = f * 2^k (f in [1, 2)), x a positive normal wave-uniform double: brings matrices of x's order of magnitude into float range
__device__ __forceinline__ double settle_pow2_inverse_of(double x) {
  const int e = (__builtin_amdgcn_readfirstlane(__double2hiint(x)) >> 20) & 0x7ff;
  return (e > 0 && e < 0x7fe) ? __hiloint2double((2046 - e) << 20, 0) : __builtin_nan("");   // NaN: the test starts over
}

// ---- counter-based normals (same stream as oracle_normal in oracle/dlm_oracle.c) ------
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ double philox_normal(unsigned long long seed, unsigned long long series,
                                                unsigned t, unsigned i) {
  unsigned c[4] = {(unsigned)series, (unsigned)(series >> 32), t, i >> 1};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  double u1 = ((double)c[0] * 4294967296.0 + (double)c[1] + 1.0) * (1.0 / 18446744073709551616.0);
  double u2 = ((double)c[2] * 4294967296.0 + (double)c[3]) * (1.0 / 18446744073709551616.0);
  double r = sqrt(-2.0 * log(u1));
  double ang = 6.283185307179586476925286766559 * u2;
  return (i & 1) ? r * sin(ang) : r * cos(ang);
}

// components 2 q and 2 q + 1 of record t share one Philox block and one Box-Muller pair: both at the cost of one
__device__ __forceinline__ void philox_normal2(unsigned long long seed, unsigned long long series, unsigned t, unsigned q,
                                               double& z_even, double& z_odd) {
  unsigned c[4] = {(unsigned)series, (unsigned)(series >> 32), t, q};
  philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
  double u1 = ((double)c[0] * 4294967296.0 + (double)c[1] + 1.0) * (1.0 / 18446744073709551616.0);
  double u2 = ((double)c[2] * 4294967296.0 + (double)c[3]) * (1.0 / 18446744073709551616.0);
  double r = sqrt(-2.0 * log(u1));
  double ang = 6.283185307179586476925286766559 * u2;
  z_even = r * cos(ang); z_odd = r * sin(ang);
}

}  // namespace dlm
