/*
 * dlm_engine.h -- C ABI of the MI355X batched Kalman filter / smoother / FFBS engine.
 *
 * This is the drop-in boundary.  The reference (jonnylaw/bayesian_dlms, Scala) has no
 * FFI/plugin interface: its seams are per-timestep closures (KalmanFilter.scala:32,
 * SvdFilter.scala:22, Smoothing.scala:114-116), far too fine for a JNI crossing.  The
 * boundary therefore sits one level up, at the whole-series calls, widened to a batch of
 * N independent series that share one model (`Dlm`) and one time grid.  Each entry point
 * names the reference call it replaces; the JNI / Scala binding is in INTEGRATION.md.
 *
 * Conventions (all citations relative to /root/reference/core/src/main/scala/dlm/model/):
 *  - fp64 everywhere; matrices column-major (Breeze `DenseMatrix.data`).
 *  - F is d x p and is used as F^T (KalmanFilter.scala:317).
 *  - A missing observation component (`None`, Dlm.scala:94) is NaN in `y`.
 *  - Outputs carry T+1 records; record 0 is the initial state at t0-1 (`.filter` keeps it,
 *    Filter.scala:41-45; KalmanFilter.initialiseState, KalmanFilter.scala:112-118).
 *  - A "state record" is d + d*d doubles: mean (d) then covariance (d x d, column-major).
 *  - The caller owns every buffer.  `opts->mem` says whether ALL data pointers of the call
 *    (descriptors' arrays, y, outputs, status) are device (HIP) pointers or host pointers;
 *    in host mode the engine stages H2D/D2H through its own workspace.
 *  - Return value: 0 = OK, negative = error (message via dlm_last_error).  Numerical trouble
 *    is reported per series in `status[N]` bit flags; the batch still completes.
 *  - An engine handle is not thread-safe; use one handle per thread / per GPU.
 */
#ifndef DLM_ENGINE_H
#define DLM_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dlm_engine dlm_engine;

enum {
  DLM_OK = 0,
  DLM_ERR_ARG = -1,         /* bad argument / shape                                   */
  DLM_ERR_HIP = -2,         /* HIP runtime error                                      */
  DLM_ERR_UNSUPPORTED = -3, /* shape outside what the kernels cover   */
  DLM_ERR_RCCL = -4         /* RCCL error                                             */
};

enum { DLM_MEM_DEVICE = 0, DLM_MEM_HOST = 1 };

/* dlm_options.flags */
enum {
  DLM_OPT_SMOOTHER_COMPAT_Q1 = 1u << 0, /* literal Smoothing.scala:44 (J X J, no transpose) */
  DLM_OPT_SVD_RAW_W_Q2 = 1u << 1,       /* literal SvdFilter.filterDlm: raw W as sqrt(W)    */
  DLM_OPT_SVD_SAMPLER_Q9 = 1u << 2,     /* literal SvdSampler.step: sqrt(W) for sqrt(W)^-1  */
  DLM_OPT_FORCE_GENERIC = 1u << 3,      /* disable the specialised (MFMA) kernels           */
  DLM_OPT_STATS_OUTER = 1u << 4,        /* Gibbs stats: full outer product (GibbsWishart)   */
  DLM_OPT_ASYNC = 1u << 5,              /* do not synchronise the stream before returning   */
  DLM_OPT_FFBS_SIMSMOOTH = 1u << 6,     /* draw with the Durbin-Koopman simulation smoother */
  DLM_OPT_PACKED_SYM = 1u << 7,         /* state records leave PACKED: [mean (d) | lower triangle of the covariance by rows],
                                           dlm_packed_record_doubles(d) doubles per record (see below)              */
  /* A PROMISE of the caller, not a tuning knob -- results are WRONG if it is broken.  Device-memory calls only: F, G and the
   * time grid are bit for bit those of this engine's previous model-taking call, so its analysis of their structure is reused
   * (two stream round trips per call less).  The engine checks what it can without reading device memory (d, p, n_g, f_stride
   * and the path taken must agree, else it analyses afresh) and, unless DLM_OPT_TRUST_MODEL_UNCHANGED is also set, verifies
   * the promise with a 64-bit checksum of F, G, g_index and dt computed on the device (one tiny kernel per call; a mismatch
   * returns DLM_ERR_ARG).  Host-memory calls never need it: the engine compares the tables itself. */
  DLM_OPT_MODEL_UNCHANGED = 1u << 8,
  DLM_OPT_COUNT_STEPS = 1u << 9,        /* count the steps that took a short path (dlm_last_counters); costs a 32-byte memset per call */
  DLM_OPT_TRUST_MODEL_UNCHANGED = 1u << 10, /* with DLM_OPT_MODEL_UNCHANGED: skip the device checksum (callers that stage the tables themselves
                                         * and compare them on the host, as bayesian_dlms_amd/engine.py does)                     */
  DLM_OPT_LOGLIK_LITERAL_Q7 = 1u << 11, /* dlm_loglik_batch: KalmanFilter.likelihood as written (KalmanFilter.scala:299-306, what
                                         * MetropolisHastings.dlm evaluates): the transition density of the filtered means        */
  /* Kernel-selection overrides: measurements and tests only, results do not depend on them.           */
  DLM_OPT_NO_LANE = 1u << 16,           /* no lane-per-series kernels (d <= 5, p = 1)                                  */
  DLM_OPT_NO_SAMPLER16 = 1u << 17,      /* no register-tile backward sampler: the generic kernel                       */
  DLM_OPT_NO_WAVE = 1u << 18,           /* 16 <= d <= 48: the workgroup-per-series kernels instead of wave-per-series  */
  DLM_OPT_FORCE_WAVE = 1u << 19,        /* 16 <= d <= 48: wave-per-series kernels also for batches of <= 256 series    */
  DLM_OPT_NO_SPARSE_F = 1u << 20,       /* treat F as dense                                                            */
  DLM_OPT_NO_SMALL_BATCH = 1u << 21,    /* d <= 15: the throughput kernels also for batches that leave SIMDs idle      */
  DLM_OPT_NO_STEADY = 1u << 22,         /* d <= 15: every step recomputes the covariance recursion, also once it has settled */
  DLM_OPT_NO_PIPE = 1u << 23,           /* d <= 15, small batches: the output product's MFMAs in one block (as at full occupancy) */
  DLM_OPT_SVD_PER_SERIES = 1u << 25,    /* SVD filter: every series runs its own decompositions, also when the batch shares V, W, C0 and has no missing
                                           observation (by default they are then done once per call and shared: the same bits) */
  DLM_OPT_SAMPLER_PER_SERIES = 1u << 26, /* dlm_ffbs_batch, reference-form sampler: every series computes its own J_t, H_t and factors, also when the batch
                                           shares V, W, C0 on a regular grid (by default one wave computes them once per call and the series draw against
                                           its table: the same draws, bit for bit) */
  DLM_OPT_DRAW_EIG = 1u << 27,          /* dlm_ffbs_batch / dlm_backward_sample_batch, reference-form sampler: draw with the REFERENCE's factor, theta = h + E sqrt(Lambda) z from
                                           the symmetric eigendecomposition of H (MultivariateGaussianSvd.scala:13-22; eigenvalues ascending, the largest-|.| entry of each
                                           eigenvector positive -- LAPACK leaves the sign open), instead of the engine's lower Cholesky factor.  The same distribution,
                                           other draws; served by the general LDS kernel (d <= 53): for parity with the literal operation sequence, not for speed */
  DLM_OPT_SMOOTHER_PER_SERIES = 1u << 28, /* dlm_filter_smooth_batch with DLM_OPT_SMOOTHER_COMPAT_Q1, structured d <= 15, p = 1: every series computes its own J_t and
                                           S_t, also when the batch shares V, W, C0 on a regular grid (by default they are computed once per call and every series
                                           without a missing observation runs only its mean recursion: the same records, bit for bit) */
  DLM_OPT_TEST_FAIL_AFTER_TABLES = 1u << 30, /* TEST HOOK (tests/test_shared_sampler_gpu.py, test_shared_rts_gpu.py): dlm_ffbs_batch / dlm_filter_smooth_batch return DLM_ERR_UNSUPPORTED right after they have started the
                                           shared-factor tables and normals on the engine's auxiliary streams -- the error path that must leave the engine usable */
  DLM_OPT_SHARED_COV = 1u << 24         /* d <= 15, p = 1, regular grid, V, W, C0 shared by the batch: ONE wave runs the covariance recursions, every series
                                           only its mean recursions against their tables; a series with a missing observation runs its own recursion
                                           as always.  Bit for bit the results of the default kernels (tests/test_shared_cov_gpu.py) -- and, measured,
                                           no faster than them: opt-in (profiles/r03_notes.md) */
};

/* per-series status bits */
enum {
  DLM_ST_NONFINITE = 1, /* a non-finite value appeared in the state                        */
  DLM_ST_NOT_PD = 2,    /* a matrix that must be positive definite was not (Q, R or H)     */
  DLM_ST_NOCONV = 4     /* Jacobi SVD did not converge                                     */
};

/* The model, materialised on the host from the reference's closures
 * (`Dlm(f, g)`, Dlm.scala:14-15): F_t = f(time_t), G_k = g(dt_k) per distinct dt. */
typedef struct {
  int32_t d, p, T, N;
  const double *F;        /* [nF][d*p]; nF = 1 if f_stride == 0 else T                     */
  int64_t f_stride;       /* 0: time-invariant F; else F_t = F + t * f_stride (doubles)    */
  const double *G;        /* [n_g][d*d]                                                   */
  int32_t n_g;
  const int32_t *g_index; /* [T] table index of step t; NULL: all 0                        */
  const double *dt;       /* [T] time increments; NULL: all 1.0.  dt == 0 means "no        */
                          /* advance" exactly as KalmanFilter.advState (:279-280)          */
} dlm_model_desc;

/* `DlmParameters(v, w, m0, c0)` (Dlm.scala:36-39); a stride of 0 shares the array between
 * all series, otherwise series n reads base + n * stride (doubles).
 * Time-varying variances (the per-step V_t / W_t streams of StudentT.filter, StudentTGibbs.scala:100-136,
 * and of DlmFsvSystem.ffbs, DlmFsvSystem.scala:137-208): with v_tstride / w_tstride != 0 observation t (0-based) uses
 * V + n * v_stride + t * v_tstride and the transition INTO observation t uses W + n * w_stride + t * w_tstride
 * (T matrices each).  0 = time-invariant.  The SVD entry points take them too: step t then runs with the square roots
 * of V_t / W_t (DlmFsv.ffbsSvd, DlmFsv.scala:208-228; DlmFsvSystem.ffbsSvd, DlmFsvSystem.scala:176-208). */
typedef struct {
  const double *V;  int64_t v_stride;   /* p x p */
  const double *W;  int64_t w_stride;   /* d x d */
  const double *m0; int64_t m0_stride;  /* d     */
  const double *C0; int64_t c0_stride;  /* d x d */
  int64_t v_tstride;                    /* 0 or p * p */
  int64_t w_tstride;                    /* 0 or d * d */
} dlm_params_desc;

typedef struct {
  uint32_t flags;     /* DLM_OPT_*                                                        */
  int32_t mem;        /* DLM_MEM_DEVICE or DLM_MEM_HOST                                   */
  uint64_t seed;      /* Philox key for FFBS draws                                        */
  uint64_t series_offset; /* global index of series 0 of this call (multi-GPU shards       */
                          /* draw the same normals as a single-GPU run)                   */
} dlm_options;

/* ---- lifecycle ------------------------------------------------------------------- */
int dlm_engine_create(int device, dlm_engine **out);
void dlm_engine_destroy(dlm_engine *e);
const char *dlm_last_error(const dlm_engine *e);
const char *dlm_version(void);
/* Launch on a caller-owned hipStream_t (NULL = the engine's own stream).  The engine's workspaces are shared by its calls: a
 * change of stream first drains the work still in flight on the old one (DLM_OPT_ASYNC calls). */
int dlm_engine_set_stream(dlm_engine *e, void *hip_stream);
int dlm_engine_sync(dlm_engine *e);
/* Name of the kernel variant the last call dispatched to ("generic", "mfma16", ...). */
const char *dlm_last_variant(const dlm_engine *e);

/* ---- ordering against the caller's streams ------------------------------------------
 * The engine launches on its own stream.  A caller that produces inputs (or frees / reuses buffers) on another
 * HIP stream orders the two with events, without a host synchronisation:
 *   dlm_engine_wait_stream   work submitted to `hip_stream` so far completes before any later engine work starts;
 *   dlm_stream_wait_engine   engine work submitted so far completes before later work on `hip_stream` starts
 *                            (needed only after DLM_OPT_ASYNC calls: synchronous calls have drained the engine stream).
 * hip_stream == NULL means the legacy default stream (what `torch.cuda.current_stream()` is unless changed). */
int dlm_engine_wait_stream(dlm_engine *e, void *hip_stream);
int dlm_stream_wait_engine(dlm_engine *e, void *hip_stream);

/* ---- engine-owned device buffers ------------------------------------------------------
 * For callers without a device allocator of their own (the JVM through JNI, plain C / C++): allocate once, keep
 * inputs and results in HBM across calls (opts->mem = DLM_MEM_DEVICE, the measured path) and move only what is
 * needed.  A buffer is an ordinary HIP device pointer (arithmetic on it is allowed: base + offset is what the
 * descriptors take).  Upload / download are synchronous on return and ordered after all engine work submitted
 * before them; offsets and sizes in bytes.  Buffers still allocated are released by dlm_engine_destroy. */
int dlm_buffer_alloc(dlm_engine *e, uint64_t bytes, void **dev_ptr);
int dlm_buffer_free(dlm_engine *e, void *dev_ptr);
int dlm_buffer_upload(dlm_engine *e, void *dst_dev, uint64_t dst_offset, const void *src_host, uint64_t bytes);
int dlm_buffer_download(dlm_engine *e, const void *src_dev, uint64_t src_offset, void *dst_host, uint64_t bytes);
int dlm_buffer_fill(dlm_engine *e, void *dst_dev, uint64_t dst_offset, int byte_value, uint64_t bytes);
int dlm_device_mem_info(dlm_engine *e, uint64_t *free_bytes, uint64_t *total_bytes);

/* ---- packed symmetric records (DLM_OPT_PACKED_SYM) --------------------------------------
 * A packed state record is [mean (d) | C(0,0) | C(1,0) C(1,1) | C(2,0) ... C(d-1,d-1)] -- the lower triangle by
 * rows -- in a slot of dlm_packed_record_doubles(d) = d + d (d + 1) / 2 rounded up to even doubles (104 at d = 13
 * against 182 dense; the padding double, when there is one, is never written).  With the flag dlm_filter_batch writes `filt` and
 * dlm_filter_smooth_batch writes `filt` and `smooth` packed; algorithmic traffic of the fused pass drops from
 * 8p + 24 (d + d^2) to 8p + 24 (d + d (d + 1) / 2) bytes per series-step (2504 instead of 4376 at d = 13).  Served by the
 * structured d <= 15, p = 1 kernels (every model the reference can build at those sizes); other shapes, the Q1 literal
 * mode, prior records and dlm_smooth_batch return DLM_ERR_UNSUPPORTED -- use dense records there.
 * dlm_unpack_records expands [count] packed records into dense ones (host or device memory according to opts->mem). */
int32_t dlm_packed_record_doubles(int32_t d);
int dlm_unpack_records(dlm_engine *e, int32_t d, int64_t count, const double *packed, const dlm_options *opts,
                       double *dense);

/* ---- Kalman filter ----------------------------------------------------------------
 * Replaces KalmanFilter(KalmanFilter.advanceState(p, mod.g)).filter(mod, ys, p)
 * (KalmanFilter.scala:262-294, Filter.scala:41-45) for N series.
 *   y      [N][T][p]
 *   filt   [N][T+1][d+d*d]   (m_t, C_t)
 *   prior  [N][T+1][d+d*d]   (a_t, R_t)            optional (NULL)
 *   fq     [N][T+1][p+p*p]   (f_t, Q_t; record 0 NaN) optional (NULL)
 *   status [N]                                      optional (NULL) */
int dlm_filter_batch(dlm_engine *e, const dlm_model_desc *model, const dlm_params_desc *params,
                     const double *y, const dlm_options *opts, double *filt, double *prior,
                     double *fq, int32_t *status);

/* ---- RTS smoother -----------------------------------------------------------------
 * Replaces Smoothing.backwardsSmoother(mod)(kfStates) (Smoothing.scala:31-64).
 * Takes the filter records and recomputes a_{t+1}, R_{t+1} from (m_t, C_t) instead of
 * reading them back.  smooth [N][T+1][d+d*d] = (s_t, S_t). */
int dlm_smooth_batch(dlm_engine *e, const dlm_model_desc *model, const dlm_params_desc *params,
                     const double *filt, const dlm_options *opts, double *smooth, int32_t *status);

/* Simulation from the model on the device: Dlm.simulateRegular / simStep (Dlm.scala:245-292) over the model's time grid,
 *   x_0 ~ N(m0, C0);  x_t = G_t x_{t-1} + w_t, w_t ~ N(0, W dt_t);  y_t = F_t^T x_t + v_t, v_t ~ N(0, V).
 * Lower-Cholesky factors of C0, W, V (the reference draws through an eigen-factor: same distribution) on the Philox
 * stream (opts->seed, opts->series_offset + n, record t, i), i < d state noise, d <= i < d + p observation noise.
 * x [N][T+1][d] (record 0 = x_0) nullable; y [N][T][p]; status [N] nullable. */
int dlm_simulate_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                       const dlm_options* opts, double* x, double* y, int32_t* status);

/* GibbsSampling.dinvGammaStep on the device (Gibbs.scala:23-78, :134-151) for per-series parameters: from the
 * statistics [N][2p + d + 1] = [ssy | n | ss | T] of an FFBS call (without DLM_OPT_STATS_OUTER) draw
 *   V_jj ~ InverseGamma(alpha_v + n_j / 2, beta_v + ssy_j / 2),   W_ii ~ InverseGamma(alpha_w + T / 2, beta_w + ss_i / 2)
 * and write the dense diagonal matrices V_out [N][p*p], W_out [N][d*d] -- directly usable as the next call's
 * per-series parameters (v_stride = p*p, w_stride = d*d), so that a Gibbs iteration never leaves the GPU.
 * Marsaglia-Tsang Gamma draws on the Philox stream (opts->seed, opts->series_offset + n, iteration, component):
 * reproducible and shard-invariant; the reference's generator cannot be seeded, only the distribution compares. */
int dlm_dinvgamma_step_batch(dlm_engine* e, int32_t d, int32_t p, int32_t N, const double* stats, double alpha_v,
                             double beta_v, double alpha_w, double beta_w, uint64_t iteration, const dlm_options* opts,
                             double* V_out, double* W_out);

/* Scalar AR(1) state-space FFBS, one GPU lane per series: FilterAr.filterUnivariate / univariateSample / ffbs
 * (FilterAr.scala:15-82), the filter of the stochastic-volatility samplers (StochasticVolatility.scala:142-162,
 * FactorSv.scala:415-512):
 *   alpha_t = mu + phi (alpha_{t-1} - mu) + eta_t, eta_t ~ N(0, sigma_eta^2);   y_t = alpha_t + eps_t, eps_t ~ N(0, v_t)
 * y [N][T] (NaN = None); v: per-step observation variances, [N][T] with v_stride = T or one shared stream [T] with
 * v_stride = 0; sv: (phi, mu, sigma_eta) in SvParameters order, [N][3] with sv_stride = 3 or one shared triple with
 * sv_stride = 0; z [N][T+1] injected normals or NULL (Philox stream (seed, series_offset + n, t, 0));
 * filt [N][T+1][2] = (m_t, c_t) with record 0 = (mu, sigma_eta^2 / (1 - phi^2)), nullable;
 * theta [N][T+1] one draw per state, NULL = filter only; status [N] nullable.  opts->flags: DLM_OPT_ASYNC only. */
int dlm_ar1_ffbs_batch(dlm_engine* e, int32_t N, int32_t T, const double* y, const double* v, int64_t v_stride,
                       const double* sv, int64_t sv_stride, const double* z, const dlm_options* opts,
                       double* filt, double* theta, int32_t* status);

/* The Ornstein-Uhlenbeck variant of the above on an irregular time grid: FilterOu.filterUnivariate / univariateSample /
 * ffbs (FilterOu.scala:7-79).  times [T] observation times shared by the batch; sv = (phi, mu, sigma_eta) with phi > 0
 * the mean-reversion rate.  Literal reference behaviour: c0 = sigma * sigma / phi * phi (= sigma^2) and the initial
 * state sits at the first observation time (first dt = 0).  Everything else as dlm_ar1_ffbs_batch. */
int dlm_ou_ffbs_batch(dlm_engine* e, int32_t N, int32_t T, const double* times, const double* y, const double* v,
                      int64_t v_stride, const double* sv, int64_t sv_stride, const double* z, const dlm_options* opts,
                      double* filt, double* theta, int32_t* status);

/* Per-series log-likelihood by the prediction-error decomposition,
 *   loglik[n] = sum_t log N(y_t^obs ; f_t^obs, Q_t^obs),
 * i.e. KalmanFilter.conditionalLikelihood (KalmanFilter.scala:138-153) summed over the series (steps with no observed
 * component contribute 0).  It is the filter recursion with one scalar reduction per series and no record output, for
 * the callers that evaluate a bank of parameter sets (MetropolisHastings.scala:126-137, RaoBlackwellFilter.scala:43-57;
 * SURVEY 8f #2): params strides select per-series parameters.  loglik [N]; status [N] as for dlm_filter_batch.
 *
 * DLM_OPT_LOGLIK_LITERAL_Q7 selects what the reference's `KalmanFilter.likelihood` (KalmanFilter.scala:299-306) -- the
 * function MetropolisHastings.dlm calls (MetropolisHastings.scala:134, :205) -- really computes: it filters and then
 * sums the TRANSITION density of the filtered means (KalmanFilter.logLikelihood, :175-183),
 *   loglik[n] = sum_{t=1..T} log N(m_t ; g(dt_t) m_{t-1}, W dt_t),     m_0 = the initial state at t0 - 1,
 * with Breeze's MultivariateGaussian.logPdf (Cholesky of W dt).  Every consecutive pair counts, also across a missing
 * observation.  W must be positive definite and every dt > 0 (Breeze's cholesky throws otherwise: here loglik[n] = NaN and
 * DLM_ST_NOT_PD); a W_t stream (w_tstride != 0) is DLM_ERR_UNSUPPORTED (the reference passes the time-invariant p.w).
 * The call filters into an engine workspace of N (T + 1) (d + d^2) doubles first. */
int dlm_loglik_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                     const double* y, const dlm_options* opts, double* loglik, int32_t* status);

/* ---- fused filter + smoother (the headline metric path) ----------------------------
 * KalmanFilter(...).filter followed by Smoothing.backwardsSmoother (KalmanFilter.scala:262-294, Smoothing.scala:57-64)
 * in one call.  filt may be NULL when only the smoothed moments are wanted: the filtered records then stay in an
 * engine workspace (stored packed -- mean + lower triangle -- on the structured d <= 15 path, which cuts the traffic
 * of both passes). */
int dlm_filter_smooth_batch(dlm_engine *e, const dlm_model_desc *model,
                            const dlm_params_desc *params, const double *y,
                            const dlm_options *opts, double *filt, double *smooth,
                            int32_t *status);

/* Device time of the forward (ms[0]) and backward (ms[1]) kernels of the LAST dlm_filter_smooth_batch, dlm_ffbs_batch,
 * dlm_backward_sample_batch, dlm_svd_filter_batch or dlm_svd_ffbs_batch call, from HIP events recorded on the
 * engine's stream around them (a part that did not run reads 0). */
int dlm_last_timing(dlm_engine *e, double ms[2]);

/* Step counters of the LAST call made with DLM_OPT_COUNT_STEPS (synchronises the engine's stream):
 *   out[0]  steps of the forward kernel that took its steady-state (mean-only) path, summed over the series
 *   out[1]  the same for the backward kernel
 *   out[2]  series served by the shared-covariance kernels
 *   out[3]  series of the same call that ran their own covariance recursion (a missing observation)
 * `bench.py` reports out[0..1] / (N T) as `steady_fraction`. */
int dlm_last_counters(dlm_engine *e, uint64_t out[4]);

/* ---- FFBS + Gibbs sufficient statistics --------------------------------------------
 * Replaces Smoothing.ffbsDlm (Smoothing.scala:173-180) and, when `stats` is given, the
 * sums inside GibbsSampling.sampleObservationMatrix / sampleSystemMatrix
 * (Gibbs.scala:23-78) or GibbsWishart.sampleSystemMatrix (GibbsWishart.scala:16-35).
 *   filt_ws [N][T+1][d+d*d]  the forward pass's records (the caller's to read afterwards); NULL: not wanted -- the engine keeps
 *                            them in a workspace of its own, and where the batch shares V, W, C0 on a regular grid (d <= 15) it
 *                            does not produce them at all: a mean-only forward pass against one covariance table
 *   z       [N][T+1][d]      injected standard normals; NULL = Philox4x32-10 stream
 *                            keyed by (opts->seed, opts->series_offset + n, t, i)
 *   theta   [N][T+1][d]      the draw                     optional
 *   cond    [N][T+1][d+d*d]  conditional (h_t, H_t)       optional
 *   stats   [N][L]           L = dlm_stats_len(d, p, flags):
 *                            [ssy(p) | n(p) | ss(d) | T]            (d-Inverse-Gamma)
 *                            [ssy(p) | n(p) | outer(d*d) | T]       (DLM_OPT_STATS_OUTER)
 * DLM_OPT_FFBS_SIMSMOOTH: draw theta = E[x | y - y+] + x+ with (x+, y+) simulated from the model
 * (Durbin & Koopman 2002): the same distribution as FFBS without a d x d factorisation per step
 * (fast paths: d <= 15, p = 1, structured G, regular grid; or 16 <= d <= 48, p <= 32; otherwise the
 * flag is ignored).  Normals: d + p per record (state noise, then observation noise), so injected z is
 * [N][T+1][d+p]; `cond` is not produced.
 * The draw uses the lower Cholesky factor of H_t (theta = h + L z); the reference's eigSym
 * factor has LAPACK-defined signs and an unseedable RNG, so draw-level parity with Breeze is
 * not defined -- see DESIGN.md. */
int dlm_ffbs_batch(dlm_engine *e, const dlm_model_desc *model, const dlm_params_desc *params,
                   const double *y, const double *z, const dlm_options *opts, double *filt_ws,
                   double *theta, double *cond, double *stats, int32_t *status);
int32_t dlm_stats_len(int32_t d, int32_t p, uint32_t flags);

/* Backward sampling only, from existing filter records (Smoothing.sampleDlm,
 * Smoothing.scala:164-165). */
int dlm_backward_sample_batch(dlm_engine *e, const dlm_model_desc *model,
                              const dlm_params_desc *params, const double *y, const double *filt,
                              const double *z, const dlm_options *opts, double *theta,
                              double *cond, double *stats, int32_t *status);

/* ---- SVD (square-root) filter / sampler -------------------------------------------
 * Replaces SvdFilter.filterDlm (SvdFilter.scala:158-161) and SvdSampler.ffbsDlm
 * (SvdSampler.scala:79-82).  svd_rec [N][T+1][d + d + d*d] = (m_t, dc_t, uc_t) with
 * C_t = uc diag(dc^2) uc^T.  d <= 48, p <= 32 (DLM_ERR_UNSUPPORTED beyond): one wavefront per series, the
 * decompositions in LDS; models with d, p <= 16 keep fourteen series per CU, larger ones one. */
int dlm_svd_filter_batch(dlm_engine *e, const dlm_model_desc *model,
                         const dlm_params_desc *params, const double *y,
                         const dlm_options *opts, double *svd_rec, int32_t *status);
int dlm_svd_ffbs_batch(dlm_engine *e, const dlm_model_desc *model, const dlm_params_desc *params,
                       const double *y, const double *z, const dlm_options *opts,
                       double *svd_ws, double *theta, double *stats, int32_t *status);

/* ---- pooled-parameter Gibbs: reduce over series, then over GPUs --------------------
 * dlm_stats_pool sums stats [N][L] over the N series of this shard into pooled[L] (device).
 * dlm_comm_* wrap RCCL (one rank per engine / GPU); dlm_gibbs_suffstats_allreduce is one
 * ncclAllReduce(sum, fp64) of `count` doubles in place.  The only collective on the path. */
int dlm_stats_pool(dlm_engine *e, const double *stats, int32_t N, int32_t L, double *pooled,
                   const dlm_options *opts);
#define DLM_COMM_ID_BYTES 128
int dlm_comm_unique_id(uint8_t id[DLM_COMM_ID_BYTES]);
int dlm_comm_init_rank(dlm_engine *e, int32_t nranks, int32_t rank,
                       const uint8_t id[DLM_COMM_ID_BYTES]);
int dlm_gibbs_suffstats_allreduce(dlm_engine *e, double *stats_dev, int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* DLM_ENGINE_H */
