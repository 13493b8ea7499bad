// C-ABI implementation (include/dlm_engine.h): argument checking, host<->device staging,
// kernel-variant dispatch, RCCL wrapper.  No torch types, no exceptions across the boundary.
#include "../../include/dlm_engine.h"
#include <cstdlib>
#include "dlm_internal.h"

#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <set>
#include <string>
#include <vector>

using dlm::KArgs;

struct dlm_engine {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  std::string err;
  const char* variant = "none";
  void* arena = nullptr;      // device staging arena for DLM_MEM_HOST calls
  size_t arena_bytes = 0;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // around the forward and backward kernels
  bool timed = false;
  dlm::SparseT* sp_dev = nullptr;  // [2 n_g]: row table, column table of every G (structured fast path)
  dlm::SparseBig* spb_dev = nullptr;   // the same for the tiled path (d up to 48)
  dlm::SparseF* spf_dev = nullptr;     // tables of a structured F (tiled path)
  size_t spb_count = 0;
  size_t sp_count = 0;
  int sparse_k = 0;           // 0: some G is not structured (dense MFMA path, regular grids only)
  double* fws = nullptr;      // filtered records of a fused call that does not want them (packed on the structured path)
  size_t fws_bytes = 0;
  double* side = nullptr;     // forward->backward innovations buffer of the fused fast path
  size_t side_bytes = 0;
  double* xplus = nullptr;    // simulated states x+ of the simulation smoother [N][T+1][d]
  size_t xplus_bytes = 0;
  double* ystar = nullptr;    // y - y+ of the simulation smoother (multivariate path) [N][T][p]
  size_t ystar_bytes = 0;
  ncclComm_t comm = nullptr;
  bool has_comm = false;
  std::set<void*> buffers;     // dlm_buffer_alloc allocations still owned by the caller (released at destroy)
  hipEvent_t order_ev = nullptr;   // dlm_engine_wait_stream / dlm_stream_wait_engine
  // The structure analysis of the last call (what analyse_g left behind), reused by DLM_OPT_MODEL_UNCHANGED and -- in host
  // mode, where the tables can be compared -- whenever G and F are bit for bit the same: two stream round trips per call less.
  struct Analysis {
    bool valid = false;
    int d = 0, p = 0, n_g = 0, branch = 0;      // branch: 0 none, 1 structured d <= 15 tables, 2 per-wave / tiled tables
    long long f_stride = 0;
    unsigned generic = 0;
    int sparse_k = 0, spb_k = 0, spf_k = 0;
    bool have_spb = false, have_spf = false;
    std::vector<double> g_host, f_host;         // host-mode calls: the tables analysed
  } an;
  bool an_touched = true;
  // shared-covariance path: the tables of the covariance-only run, one byte per series (route), and a second
  // stream + events so that the backward covariance run overlaps the forward mean kernel
  double* covws = nullptr;
  size_t covws_bytes = 0;
  unsigned char* route = nullptr;
  size_t route_bytes = 0;
  unsigned char* plainbuf = nullptr;   // KArgs::plain of the call (k_count_gaps): one byte per series
  size_t plainbuf_bytes = 0;
  void* sampws = nullptr;       // shared factors of the backward sampler: table, the zero series' records, flags
  size_t sampws_bytes = 0;
  double* zws = nullptr;        // ... and the call's normals in the draw kernel's layout, made on a third stream while the batch is filtered
  size_t zws_bytes = 0;
  hipStream_t rng_stream = nullptr;
  hipEvent_t rng_ev = nullptr;
  hipStream_t cov_stream = nullptr;
  hipEvent_t cov_ev[2] = {nullptr, nullptr};
  hipEvent_t rng_gate = nullptr;  // 16 <= d <= 48, records-free shared-factor FFBS: the end of the batch's forward pass, behind which the normals start
  hipEvent_t cov_ev2 = nullptr;   // behind the zero series' filter of a shared-factor table (its steady gain and settle step)
  // work of the CURRENT call is (or may be) in flight on the auxiliary streams and e->stream does not depend on it yet: set when the
  // first operation goes to the stream, cleared by aux_join / drain_all (the rules are written at aux_join)
  bool cov_busy = false, rng_busy = false;
  // DLM_OPT_COUNT_STEPS: [4] device counters the kernels add to (KArgs::counters), read by dlm_last_counters
  unsigned long long* counters = nullptr;
  // DLM_OPT_MODEL_UNCHANGED is verified on the device: model_sum[0] = checksum of (F, G, g_index, dt) at the last fresh
  // analysis, [1] = scratch; a mismatch raises *model_bad (pinned host memory the kernel writes through model_bad_dev)
  unsigned long long* model_sum = nullptr;
  int* model_bad = nullptr;
  int* model_bad_dev = nullptr;
  bool model_sum_valid = false;
};

namespace {

int fail(dlm_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  return code;
}

// the device checksum of a DLM_OPT_MODEL_UNCHANGED call did not match the model the engine analysed: its results are invalid
int promise_broken(dlm_engine* e) {
  *e->model_bad = 0;
  e->an.valid = false;
  return fail(e, DLM_ERR_ARG, "DLM_OPT_MODEL_UNCHANGED was set, but F, G or the time grid differ from the model of the previous call (device checksum): the results of this call are invalid -- drop the flag when the model changes");
}

#define HIP_TRY(e, expr)                                                                    \
  do {                                                                                      \
    hipError_t _err = (expr);                                                               \
    if (_err != hipSuccess)                                                                 \
      return fail((e), DLM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_err));   \
  } while (0)

// ---- the auxiliary streams (cov_stream: tables of the shared-covariance / shared-factor paths, raised priority; rng_stream: the
// normals of a shared-factor call) -- who waits for whom.  Two rules, kept by every entry point:
//   A. when an entry point returns -- success, error or "not eligible after all" -- e->stream depends on everything the call put on
//      the auxiliary streams (aux_join; AuxScope does it on every exit path).  A caller that synchronises e->stream (every call
//      without DLM_OPT_ASYNC does, dlm_engine_sync does) has therefore waited for the auxiliary work too, and the next call's work,
//      whose auxiliary launches wait for an event recorded on e->stream, is ordered behind it.
//   B. before a workspace any of the three streams may touch is freed or resized, before the engine moves to another stream and
//      before it is destroyed, all three streams are drained on the host (drain_all).
// (Round 3 kept neither on its error paths: a return between start_sampler_tables and the hipStreamWaitEvent of the draw left kernels
// on cov_stream / rng_stream reading the staging arena and e->sampws with nothing ordered behind them, the workspace re-size paths
// synchronised e->stream alone, and dlm_engine_destroy freed buffers and destroyed streams with auxiliary work possibly in flight.)
int aux_join(dlm_engine* e) {
  if (e->cov_busy) {
    e->cov_busy = false;   // (cleared first: a failing HIP call must not make every later exit path fail again)
    HIP_TRY(e, hipEventRecord(e->cov_ev[1], e->cov_stream));            // everything queued on the stream so far
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->cov_ev[1], 0));
  }
  if (e->rng_busy) {
    e->rng_busy = false;
    HIP_TRY(e, hipEventRecord(e->rng_ev, e->rng_stream));
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->rng_ev, 0));
  }
  return DLM_OK;
}
int drain_all(dlm_engine* e) {
  e->cov_busy = e->rng_busy = false;
  if (e->cov_stream) HIP_TRY(e, hipStreamSynchronize(e->cov_stream));
  if (e->rng_stream) HIP_TRY(e, hipStreamSynchronize(e->rng_stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return DLM_OK;
}
struct AuxScope {   // rule A on every exit path of an entry point that may use the auxiliary streams
  dlm_engine* e;
  explicit AuxScope(dlm_engine* e_) : e(e_) {}
  ~AuxScope() { if (e && (e->cov_busy || e->rng_busy)) (void)aux_join(e); }
};

// Collects the buffers of one call.  Device mode: pointers pass through.  Host mode: every
// buffer gets a slot in the engine's arena; inputs are copied up before the launch, outputs
// are copied back by finish().
class Stager {
 public:
  Stager(dlm_engine* e, bool host) : e_(e), host_(host) {}

  template <class T>
  void in(const T** field, const T* p, size_t count) { add((void**)field, (void*)p, count * sizeof(T), true, false); }
  template <class T>
  void out(T** field, T* p, size_t count) { add((void**)field, (void*)p, count * sizeof(T), false, true); }
  template <class T>
  void zeroed_out(T** field, T* p, size_t count) { add((void**)field, (void*)p, count * sizeof(T), false, true, true); }

  int commit() {
    if (host_) {
      size_t need = 0;
      for (auto& b : bufs_) if (b.user) { b.offset = need; need += (b.bytes + 255) & ~(size_t)255; }
      if (need > e_->arena_bytes) {
        if (e_->arena) { { const int rcd = drain_all(e_); if (rcd) return rcd; } HIP_TRY(e_, hipFree(e_->arena)); e_->arena = nullptr; e_->arena_bytes = 0; }
        HIP_TRY(e_, hipMalloc(&e_->arena, need));
        e_->arena_bytes = need;
      }
    }
    for (auto& b : bufs_) {
      if (!b.user) { *b.field = nullptr; continue; }
      void* dev = host_ ? (void*)((char*)e_->arena + b.offset) : b.user;
      b.dev = dev;
      *b.field = dev;
      if (host_ && b.is_in) HIP_TRY(e_, hipMemcpyAsync(dev, b.user, b.bytes, hipMemcpyHostToDevice, e_->stream));
      if (b.zero) HIP_TRY(e_, hipMemsetAsync(dev, 0, b.bytes, e_->stream));
    }
    return DLM_OK;
  }

  int finish(bool async) {
    if (host_) {
      for (auto& b : bufs_)
        if (b.user && b.is_out) HIP_TRY(e_, hipMemcpyAsync(b.user, b.dev, b.bytes, hipMemcpyDeviceToHost, e_->stream));
      HIP_TRY(e_, hipStreamSynchronize(e_->stream));
    } else if (!async) {
      HIP_TRY(e_, hipStreamSynchronize(e_->stream));
    }
    if ((host_ || !async) && e_->model_bad && *e_->model_bad) return promise_broken(e_);
    return DLM_OK;
  }

 private:
  struct Buf { void** field; void* user; size_t bytes; bool is_in, is_out, zero; size_t offset; void* dev; };
  void add(void** field, void* p, size_t bytes, bool is_in, bool is_out, bool zero = false) {
    bufs_.push_back(Buf{field, p, bytes, is_in, is_out, zero, 0, nullptr});
  }
  dlm_engine* e_;
  bool host_;
  std::vector<Buf> bufs_;
};

int check_common(dlm_engine* e, const dlm_model_desc* m, const dlm_params_desc* p, const dlm_options* o) {
  if (!e) return DLM_ERR_ARG;
  if (!e->an_touched) e->an.valid = false;   // the call before this one took a model but did not analyse it: DLM_OPT_MODEL_UNCHANGED speaks of ITS model
  e->an_touched = false;
  if (!m || !p || !o) return fail(e, DLM_ERR_ARG, "null descriptor");
  if (m->d < 1 || m->p < 1 || m->T < 1 || m->N < 1) return fail(e, DLM_ERR_ARG, "d, p, T, N must be >= 1 (the reference throws on empty input, KalmanFilter.scala:116-117)");
  if (!m->F || !m->G || m->n_g < 1) return fail(e, DLM_ERR_ARG, "model tables F/G missing");
  if (m->n_g > 1 && !m->g_index) return fail(e, DLM_ERR_ARG, "n_g > 1 needs g_index");
  if (m->f_stride != 0 && m->f_stride != (int64_t)m->d * m->p) return fail(e, DLM_ERR_ARG, "f_stride must be 0 or d*p");
  if (!p->V || !p->W || !p->m0 || !p->C0) return fail(e, DLM_ERR_ARG, "parameter arrays missing");
  if ((p->v_tstride != 0 && p->v_tstride != (int64_t)m->p * m->p) || (p->w_tstride != 0 && p->w_tstride != (int64_t)m->d * m->d))
    return fail(e, DLM_ERR_ARG, "v_tstride must be 0 or p*p, w_tstride 0 or d*d");
  if (o->mem != DLM_MEM_DEVICE && o->mem != DLM_MEM_HOST) return fail(e, DLM_ERR_ARG, "opts->mem");
  if (m->d > 64 || m->p > 64) return fail(e, DLM_ERR_UNSUPPORTED, "d and p are limited to 64 in this build (and by the LDS of one CU on the general kernels: see the entry point's message)");
  HIP_TRY(e, hipSetDevice(e->device));
  return DLM_OK;
}

// DLM_OPT_COUNT_STEPS: the kernels of this call count their short steps into the engine's counters
int want_counters(dlm_engine* e, KArgs& k) {
  k.counters = nullptr;
  if (!(k.flags & DLM_OPT_COUNT_STEPS)) return DLM_OK;
  if (!e->counters) HIP_TRY(e, hipMalloc((void**)&e->counters, 4 * sizeof(unsigned long long)));
  HIP_TRY(e, hipMemsetAsync(e->counters, 0, 4 * sizeof(unsigned long long), e->stream));
  k.counters = e->counters;
  return DLM_OK;
}

// model + params -> staged KArgs fields
void stage_model(Stager& st, KArgs& k, const dlm_model_desc* m, const dlm_params_desc* p, const dlm_options* o) {
  const size_t d = m->d, pp = m->p, T = m->T, N = m->N;
  k.d = m->d; k.p = m->p; k.T = m->T; k.N = m->N; k.n_g = m->n_g;
  k.f_stride = m->f_stride; k.v_stride = p->v_stride; k.w_stride = p->w_stride;
  k.v_tstride = p->v_tstride; k.w_tstride = p->w_tstride;
  k.m0_stride = p->m0_stride; k.c0_stride = p->c0_stride;
  k.flags = o->flags; k.seed = o->seed; k.series_offset = o->series_offset;
  st.in(&k.F, m->F, (m->f_stride ? T : 1) * d * pp);
  st.in(&k.G, m->G, (size_t)m->n_g * d * d);
  st.in(&k.g_index, (const int*)m->g_index, m->g_index ? T : 0);
  st.in(&k.dt, m->dt, m->dt ? T : 0);
  const size_t vblk = p->v_tstride ? (T - 1) * (size_t)p->v_tstride + pp * pp : pp * pp;   // one series' V (or V_0 .. V_{T-1})
  const size_t wblk = p->w_tstride ? (T - 1) * (size_t)p->w_tstride + d * d : d * d;
  st.in(&k.V, p->V, p->v_stride ? (N - 1) * (size_t)p->v_stride + vblk : vblk);
  st.in(&k.W, p->W, p->w_stride ? (N - 1) * (size_t)p->w_stride + wblk : wblk);
  st.in(&k.m0, p->m0, p->m0_stride ? (N - 1) * (size_t)p->m0_stride + d : d);
  st.in(&k.C0, p->C0, p->c0_stride ? (N - 1) * (size_t)p->c0_stride + d * d : d * d);
}

bool fast_shape_ok(const KArgs& k) { return !(k.flags & DLM_OPT_FORCE_GENERIC) && dlm::fast_shape(k); }
// d <= 15, p = 1 fast path: the structured kernels take any time grid, the dense-G MFMA kernels a regular one
bool use_fast(const dlm_engine* e, const KArgs& k) { return fast_shape_ok(k) && (e->sparse_k > 0 || dlm::mfma16_supported(k)); }
// d <= 5, p = 1: one lane per series (DLM_OPT_NO_LANE: A/B measurements)
bool use_lane(const KArgs& k) { return !(k.flags & (DLM_OPT_FORCE_GENERIC | DLM_OPT_NO_LANE)) && dlm::lane_supported(k); }
// the d >= 16 kernels, or -- d <= 15 with several observation components and a structured G -- the per-wave kernels
bool use_tiled(const KArgs& k) { return !(k.flags & DLM_OPT_FORCE_GENERIC) && (dlm::tiled_supported(k) || dlm::wave48_small_ok(k)); }
bool tiled_analysis_wanted(const KArgs& k) { return !(k.flags & DLM_OPT_FORCE_GENERIC) && (dlm::tiled_supported(k) || dlm::wave48_small_shape(k)); }

// Inspect every G of the table (fast-path shapes only) and upload the sparse tables when all of them are
// structured.  `G_user` is the caller's pointer (host or device according to host_mode).
int analyse_g_tiled(dlm_engine* e, KArgs& k, const double* G_user, bool host_mode) {
  const size_t dd = (size_t)k.d * k.d, ng = (size_t)k.n_g;
  std::vector<double> g(dd * ng);
  if (host_mode) memcpy(g.data(), G_user, g.size() * sizeof(double));
  else {
    HIP_TRY(e, hipMemcpyAsync(g.data(), G_user, g.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  std::vector<dlm::SparseBig> tabs(2 * ng);
  for (size_t q = 0; q < ng; ++q)
    if (dlm::sparse48_analyse(g.data() + q * dd, k.d, &tabs[2 * q], &tabs[2 * q + 1]) > 4) return DLM_OK;   // dense G
  int K = 1;
  for (auto& t : tabs) K = t.K > K ? t.K : K;
  for (auto& t : tabs) t.K = K;
  if (tabs.size() > e->spb_count) {
    if (e->spb_dev) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->spb_dev)); e->spb_dev = nullptr; }
    HIP_TRY(e, hipMalloc((void**)&e->spb_dev, tabs.size() * sizeof(dlm::SparseBig)));
    e->spb_count = tabs.size();
  }
  HIP_TRY(e, hipMemcpyAsync(e->spb_dev, tabs.data(), tabs.size() * sizeof(dlm::SparseBig), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));  // tabs lives on this stack frame
  k.spb = e->spb_dev;
  k.spb_k = K;
  k.spf = nullptr;
  if (!k.f_stride) {   // a time-invariant F: its few nonzeros per column / row turn the products with F into gathers
    std::vector<double> f((size_t)k.d * k.p);
    HIP_TRY(e, hipMemcpyAsync(f.data(), k.F, f.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
    dlm::SparseF tf;
    const int kf = dlm::sparsef_analyse(f.data(), k.d, k.p, &tf);
    if (kf <= 4) {
      if (!e->spf_dev) HIP_TRY(e, hipMalloc((void**)&e->spf_dev, sizeof(dlm::SparseF)));
      HIP_TRY(e, hipMemcpyAsync(e->spf_dev, &tf, sizeof(tf), hipMemcpyHostToDevice, e->stream));
      HIP_TRY(e, hipStreamSynchronize(e->stream));
      k.spf = e->spf_dev; k.spf_k = kf;
    }
  }
  return DLM_OK;
}

int analyse_g_fresh(dlm_engine* e, KArgs& k, const double* G_user, bool host_mode);

// 64-bit checksum of the model tables (F, G, g_index, dt) of a device-memory call: one 256-thread block, every 64-bit word
// mixed with its position (splitmix64 finaliser) and summed.  store != 0: remember it (fresh analysis); store == 0: compare with
// the remembered one and raise *bad (pinned host memory) on a mismatch -- the check behind DLM_OPT_MODEL_UNCHANGED.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
__global__ __launch_bounds__(256) void k_model_checksum(const double* F, long long nF, const double* G, long long nG, const int* gi,
                                                        const double* dt, long long T, unsigned long long* sums, int store, int* bad) {
  __shared__ unsigned long long part[256];
  unsigned long long acc = 0;
  const unsigned long long* f = (const unsigned long long*)F;
  const unsigned long long* g = (const unsigned long long*)G;
  const unsigned long long* q = (const unsigned long long*)dt;
  for (long long i = threadIdx.x; i < nF; i += 256) acc += mix64(f[i] + 0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1));
  for (long long i = threadIdx.x; i < nG; i += 256) acc += mix64(g[i] + 0xc2b2ae3d27d4eb4full * (unsigned long long)(i + 1));
  if (gi) for (long long i = threadIdx.x; i < T; i += 256) acc += mix64((unsigned long long)(unsigned)gi[i] + 0x165667b19e3779f9ull * (unsigned long long)(i + 1));
  if (dt) for (long long i = threadIdx.x; i < T; i += 256) acc += mix64(q[i] + 0x27d4eb2f165667c5ull * (unsigned long long)(i + 1));
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    const unsigned long long v = part[0] + mix64((unsigned long long)nF) + mix64((unsigned long long)nG * 3 + (gi ? 1 : 0) + (dt ? 2 : 0)) + mix64((unsigned long long)T * 7);
    if (store) sums[0] = v;
    else if (sums[0] != v) { __threadfence_system(); *bad = 1; }
  }
}
int model_checksum(dlm_engine* e, const KArgs& k, bool store) {
  if (!e->model_sum) {
    HIP_TRY(e, hipMalloc((void**)&e->model_sum, 2 * sizeof(unsigned long long)));
    HIP_TRY(e, hipHostMalloc((void**)&e->model_bad, sizeof(int), hipHostMallocMapped));
    *e->model_bad = 0;
    HIP_TRY(e, hipHostGetDevicePointer((void**)&e->model_bad_dev, e->model_bad, 0));
  }
  const long long nF = (long long)(k.f_stride ? k.T : 1) * k.d * k.p, nG = (long long)k.n_g * k.d * k.d;
  hipLaunchKernelGGL(k_model_checksum, dim3(1), dim3(256), 0, e->stream, k.F, nF, k.G, nG, k.g_index, k.dt, (long long)k.T,
                     e->model_sum, store ? 1 : 0, e->model_bad_dev);
  HIP_TRY(e, hipGetLastError());
  return DLM_OK;
}

// analyse_g with the cache of the last call's result in front of it
int analyse_g(dlm_engine* e, KArgs& k, const double* G_user, bool host_mode) {
  dlm_engine::Analysis& an = e->an;
  e->an_touched = true;
  { const int rcc = want_counters(e, k); if (rcc) return rcc; }   // (every kernel-launching entry point of the filter family passes through here)
  const int branch = tiled_analysis_wanted(k) ? 2 : (fast_shape_ok(k) ? 1 : 0);
  const unsigned generic = k.flags & DLM_OPT_FORCE_GENERIC;
  bool same = an.valid && an.d == k.d && an.p == k.p && an.n_g == k.n_g && an.branch == branch && an.f_stride == k.f_stride &&
              an.generic == generic;
  const size_t gn = (size_t)k.d * k.d * (size_t)k.n_g, fn = (size_t)k.d * k.p;
  if (same && host_mode) {   // host pointers: compare the tables themselves (the caller's F is staged by now: compare the user's copy of G and our copy of F's source)
    same = an.g_host.size() == gn && memcmp(an.g_host.data(), G_user, gn * sizeof(double)) == 0;
    if (same && branch == 2 && !k.f_stride) same = false;   // (F of a host-mode call is only known staged: analysed afresh, it is one small copy)
  } else if (same) {
    same = (k.flags & DLM_OPT_MODEL_UNCHANGED) != 0 && e->model_sum_valid;
    // the promise is verified on the device (no host round trip: a mismatch surfaces when the call synchronises)
    if (same && !(k.flags & DLM_OPT_TRUST_MODEL_UNCHANGED)) { const int rc = model_checksum(e, k, false); if (rc) return rc; }
  }
  if (same) {
    e->sparse_k = an.sparse_k;
    k.spb = an.have_spb ? e->spb_dev : nullptr;
    k.spb_k = an.spb_k;
    k.spf = an.have_spf ? e->spf_dev : nullptr;
    k.spf_k = an.spf_k;
    return DLM_OK;
  }
  an.valid = false;
  const int rc = analyse_g_fresh(e, k, G_user, host_mode);
  if (rc) return rc;
  an.valid = true; an.d = k.d; an.p = k.p; an.n_g = k.n_g; an.branch = branch; an.f_stride = k.f_stride; an.generic = generic;
  an.sparse_k = e->sparse_k; an.spb_k = k.spb_k; an.spf_k = k.spf_k; an.have_spb = k.spb != nullptr; an.have_spf = k.spf != nullptr;
  an.g_host.clear();
  if (host_mode) an.g_host.assign(G_user, G_user + gn);
  (void)fn;
  e->model_sum_valid = false;
  if (!host_mode) { const int rc2 = model_checksum(e, k, true); if (rc2) return rc2; e->model_sum_valid = true; }   // what a later DLM_OPT_MODEL_UNCHANGED call is checked against
  return DLM_OK;
}

int analyse_g_fresh(dlm_engine* e, KArgs& k, const double* G_user, bool host_mode) {
  e->sparse_k = 0;
  k.spb = nullptr;
  if (tiled_analysis_wanted(k)) return analyse_g_tiled(e, k, G_user, host_mode);
  if (!fast_shape_ok(k)) return DLM_OK;
  const size_t dd = (size_t)k.d * k.d, ng = (size_t)k.n_g;
  std::vector<double> g(dd * ng);
  if (host_mode) memcpy(g.data(), G_user, g.size() * sizeof(double));
  else {
    HIP_TRY(e, hipMemcpyAsync(g.data(), G_user, g.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(e, hipStreamSynchronize(e->stream));
  }
  std::vector<dlm::SparseT> tabs(2 * ng);
  int K = 1;
  for (size_t q = 0; q < ng; ++q) {
    const int kq = dlm::sparse16_analyse(g.data() + q * dd, k.d, &tabs[2 * q], &tabs[2 * q + 1]);
    if (kq > 4) return DLM_OK;
    K = kq > K ? kq : K;
  }
  if (tabs.size() > e->sp_count) {
    if (e->sp_dev) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->sp_dev)); e->sp_dev = nullptr; }
    HIP_TRY(e, hipMalloc((void**)&e->sp_dev, tabs.size() * sizeof(dlm::SparseT)));
    e->sp_count = tabs.size();
  }
  HIP_TRY(e, hipMemcpyAsync(e->sp_dev, tabs.data(), tabs.size() * sizeof(dlm::SparseT), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));  // tabs lives on this stack frame
  e->sparse_k = K;
  return DLM_OK;
}

// HIP events around the forward (0 -> 1) and backward (1 -> 2) part of a call, read back by dlm_last_timing
int mark(dlm_engine* e, int i) {
  HIP_TRY(e, hipEventRecord(e->ev[i], e->stream));
  if (i == 2) e->timed = true;
  return DLM_OK;
}

int ensure_fws(dlm_engine* e, size_t need) {
  if (need > e->fws_bytes) {
    if (e->fws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->fws)); e->fws = nullptr; e->fws_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->fws, need));
    e->fws_bytes = need;
  }
  return DLM_OK;
}

int ensure_side(dlm_engine* e, const KArgs& k) {
  // (e / Q, 1 / Q) per record for the per-series kernels, and behind them one double per record (e / Q) for the shared-covariance kernels
  const size_t need = sizeof(double) * 3 * (size_t)k.N * ((size_t)k.T + 1);
  if (need > e->side_bytes) {
    if (e->side) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->side)); e->side = nullptr; e->side_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->side, need));
    e->side_bytes = need;
  }
  return DLM_OK;
}

int ensure_xplus(dlm_engine* e, const KArgs& k) {
  const size_t need = sizeof(double) * (size_t)k.N * ((size_t)k.T + 1) * (size_t)k.d;
  if (need > e->xplus_bytes) {
    if (e->xplus) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->xplus)); e->xplus = nullptr; e->xplus_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->xplus, need));
    e->xplus_bytes = need;
  }
  return DLM_OK;
}

int ensure_ystar(dlm_engine* e, const KArgs& k) {
  // innovations [N][T][p], then one byte per record [N][T+1]: the marks "C_t is C_{t-1}" of the per-wave forward kernel's steady steps
  const size_t need = sizeof(double) * (size_t)k.N * (size_t)k.T * (size_t)k.p + (((size_t)k.N * (size_t)(k.T + 1) + 15) & ~(size_t)15);
  if (need > e->ystar_bytes) {
    if (e->ystar) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->ystar)); e->ystar = nullptr; e->ystar_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->ystar, need));
    e->ystar_bytes = need;
  }
  return DLM_OK;
}

// DLM_OPT_PACKED_SYM is served by the structured d <= 15, p = 1 kernels (the lane kernels step aside for it)
int packed_path(dlm_engine* e, KArgs& k) {
  k.flags |= DLM_OPT_NO_LANE;
  if (!(fast_shape_ok(k) && e->sparse_k > 0) || (k.flags & DLM_OPT_SMOOTHER_COMPAT_Q1))
    return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_PACKED_SYM needs the structured d <= 15, p = 1 path (time-invariant F, <= 4 nonzeros per row and column of G) without DLM_OPT_FORCE_GENERIC / DLM_OPT_SMOOTHER_COMPAT_Q1; use dense records otherwise");
  return DLM_OK;
}

// Shared covariance sequence (dlm_sparse16.hip): structured d <= 15, p = 1 path on a regular grid with V, W, C0
// shared by the batch.  One wave runs the covariance recursions into tables, every series its mean recursions against them; a
// series with a missing observation is marked by the forward mean kernel and served by the per-series kernels launched behind.
bool use_shared_cov(const dlm_engine* e, const KArgs& k) {
  return fast_shape_ok(k) && e->sparse_k > 0 && !dlm::lane_supported(k) && dlm::shared_cov_eligible(k);
}
// route: one byte per series; behind it (64-byte aligned) one int per series, the step a series left the per-wave filter at (KArgs::leave_step)
size_t route_pad(size_t N) { return (N + 63) & ~(size_t)63; }
int ensure_route(dlm_engine* e, size_t N) {
  if (N > e->route_bytes) {
    if (e->route) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->route)); e->route = nullptr; e->route_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->route, route_pad(N) + N * sizeof(int)));
    e->route_bytes = N;
  }
  return DLM_OK;
}
int* leave_of(dlm_engine* e) { return (int*)(e->route + route_pad(e->route_bytes)); }
int ensure_xplus_bytes(dlm_engine* e, size_t need) {
  if (need > e->xplus_bytes) {
    if (e->xplus) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->xplus)); e->xplus = nullptr; e->xplus_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->xplus, need));
    e->xplus_bytes = need;
  }
  return DLM_OK;
}
// Structured d <= 15 path on a regular grid: the series that miss more than T / 256 of their observations are marked (from the data
// alone: the choice does not depend on the batch) and take every step in full -- the forward kernel without its convergence test, the
// backward kernel in the instantiation without the shortcut's machinery.  With gaps nothing settles, and the machinery only costs.
int mark_plain(dlm_engine* e, KArgs& k) {
  k.plain = nullptr;
  if (!(fast_shape_ok(k) && e->sparse_k > 0) || use_lane(k) || !k.y || k.g_index || k.dt || k.f_stride || k.v_tstride || k.w_tstride ||
      (k.flags & DLM_OPT_NO_STEADY)) return DLM_OK;
  if ((size_t)k.N > e->plainbuf_bytes) {
    if (e->plainbuf) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->plainbuf)); e->plainbuf = nullptr; e->plainbuf_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->plainbuf, (size_t)k.N));
    e->plainbuf_bytes = (size_t)k.N;
  }
  HIP_TRY(e, dlm::launch_sparse16_count_gaps(k, e->plainbuf, e->stream));
  k.plain = e->plainbuf;
  return DLM_OK;
}
int ensure_cov_stream(dlm_engine* e) {
  if (!e->cov_stream) {
    int lo = 0, hi = 0;   // the few waves of the table kernels go first: they run beside a kernel that fills the device
    HIP_TRY(e, hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(e, hipStreamCreateWithPriority(&e->cov_stream, hipStreamNonBlocking, hi));
    for (auto& ev : e->cov_ev) HIP_TRY(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(e, hipEventCreateWithFlags(&e->cov_ev2, hipEventDisableTiming));
  }
  return DLM_OK;
}
// Shared factors of the reference-form backward sampler (dlm_sampler16.hip): the tables are made on the second
// stream -- a filter and a sampler run of ONE wave on a series of zeros -- while the batch is filtered on the first.
int start_sampler_normals(dlm_engine* e, const KArgs& k, dlm::SampTabs& tb, bool big, hipEvent_t gate);
int start_sampler_tables(dlm_engine* e, const KArgs& k, dlm::SampTabs& tb, bool big, const double* crec = nullptr, int crec_stride = 0, bool defer_normals = false) {
  const size_t need = big ? dlm::wave48_sampler_shared_ws_bytes(k) : dlm::sampler_shared_ws_bytes(k);
  if (need > e->sampws_bytes) {
    if (e->sampws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->sampws)); e->sampws = nullptr; e->sampws_bytes = 0; }
    HIP_TRY(e, hipMalloc(&e->sampws, need));
    e->sampws_bytes = need;
  }
  int rc = ensure_route(e, (size_t)k.N);
  if (rc || (rc = ensure_cov_stream(e))) return rc;
  if (big) dlm::wave48_sampler_shared_carve(e->sampws, k, tb); else dlm::sampler_shared_carve(e->sampws, k, tb);
  HIP_TRY(e, hipEventRecord(e->cov_ev[0], e->stream));               // the model and the tables of G are staged
  e->cov_busy = true;                                                 // (from here on every exit path joins the stream: AuxScope)
  HIP_TRY(e, hipStreamWaitEvent(e->cov_stream, e->cov_ev[0], 0));
  tb.zstride = 0; tb.mc4 = nullptr; tb.marked = 0;
  if (big) HIP_TRY(e, dlm::launch_wave48_sampler_shared_tables(k, tb, e->cov_stream, e->cov_ev2));
  else if (crec) HIP_TRY(e, dlm::launch_sampler_shared_tables_from(k, e->sparse_k, e->sp_dev, tb, crec, crec_stride, e->cov_stream));   // the covariances exist already
  else HIP_TRY(e, dlm::launch_sampler_shared_tables(k, e->sparse_k, e->sp_dev, tb, e->cov_stream));
  HIP_TRY(e, hipEventRecord(e->cov_ev[1], e->cov_stream));
  tb.z4 = nullptr;
  if (defer_normals) return DLM_OK;   // (start_sampler_normals, behind the forward pass's launch)
  return start_sampler_normals(e, k, tb, big, e->cov_ev[0]);
}
// The normals of the call (third stream).  `gate`: an event of the engine's stream the launch waits for -- the staging of the call (the draw kernel of the
// call before has read its normals) or, 16 <= d <= 48, the END of the batch's forward pass: k_filter_w48 is one wave per SIMD and launched behind a
// kernel of forty million threads it started 0.4 ms late (the normals are wanted last, by the draw kernel; they now run beside k_steady_filter_w48).
int start_sampler_normals(dlm_engine* e, const KArgs& k, dlm::SampTabs& tb, bool big, hipEvent_t gate) {
  if (!k.z) {
    const size_t zb = big ? dlm::wave48_sampler_shared_normals_bytes(k) : dlm::sampler_shared_normals_bytes(k);
    if (zb > e->zws_bytes) {
      if (e->zws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->zws)); e->zws = nullptr; e->zws_bytes = 0; }
      HIP_TRY(e, hipMalloc((void**)&e->zws, zb));
      e->zws_bytes = zb;
    }
    if (!e->rng_stream) {
      HIP_TRY(e, hipStreamCreateWithFlags(&e->rng_stream, hipStreamNonBlocking));
      HIP_TRY(e, hipEventCreateWithFlags(&e->rng_ev, hipEventDisableTiming));
    }
    e->rng_busy = true;
    HIP_TRY(e, hipStreamWaitEvent(e->rng_stream, gate, 0));
    if (big) HIP_TRY(e, dlm::launch_wave48_sampler_shared_normals(k, e->zws, e->rng_stream));
    else HIP_TRY(e, dlm::launch_sampler_shared_normals(k, e->zws, e->rng_stream));
    HIP_TRY(e, hipEventRecord(e->rng_ev, e->rng_stream));
    tb.z4 = e->zws;
  }
  return DLM_OK;
}
// Shared factors of the RTS smoother (literal Q1; dlm_sampler16.hip): the tables are made on the second stream -- a filter and a smoother
// run of ONE wave on a series of zeros -- while the batch is filtered on the first.  The workspace is the backward sampler's (e->sampws).
int start_rts_tables(dlm_engine* e, const KArgs& k, dlm::RtsTabs& tb) {
  const size_t need = dlm::rts_shared_ws_bytes(k);
  if (need > e->sampws_bytes) {
    if (e->sampws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->sampws)); e->sampws = nullptr; e->sampws_bytes = 0; }
    HIP_TRY(e, hipMalloc(&e->sampws, need));
    e->sampws_bytes = need;
  }
  const size_t cneed = sizeof(double) * dlm::covtabs_doubles(k.d, k.T);   // the forward covariance table of the shared-covariance kernels
  if (cneed > e->covws_bytes) {
    if (e->covws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->covws)); e->covws = nullptr; e->covws_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->covws, cneed));
    e->covws_bytes = cneed;
  }
  int rc = ensure_route(e, (size_t)k.N);
  if (rc || (rc = ensure_cov_stream(e))) return rc;
  dlm::rts_shared_carve(e->sampws, k, tb);
  dlm::CovTabs ctb{};
  dlm::covtabs_carve(e->covws, k.d, k.T, ctb);
  HIP_TRY(e, dlm::launch_rts_shared_mark(k, e->route, tb, e->stream));                     // the series with a gap; where they are the majority, no tables (tb.skip)
  HIP_TRY(e, dlm::launch_rts_shared_cov(k, e->sparse_k, e->sp_dev, tb, ctb, e->stream));   // the forward covariances: in front of the batch's forward pass
  HIP_TRY(e, hipEventRecord(e->cov_ev[0], e->stream));
  e->cov_busy = true;                                                 // (from here on every exit path joins the stream: AuxScope)
  HIP_TRY(e, hipStreamWaitEvent(e->cov_stream, e->cov_ev[0], 0));
  HIP_TRY(e, dlm::launch_rts_shared_tables(k, e->sparse_k, e->sp_dev, tb, e->cov_stream));   // J_t, S_t: beside it
  // The table run keeps a whole CU to itself (whole_cu_lds): it has to be resident before the forward pass fills every CU with its
  // workgroups, or it waits for that kernel's last wave.  The caller launches the gap count of the forward pass (mark_plain: a 30 us kernel)
  // in between on the engine's stream.
  HIP_TRY(e, hipEventRecord(e->cov_ev[1], e->cov_stream));
  return DLM_OK;
}
int ensure_shared(dlm_engine* e, const KArgs& k, dlm::CovTabs& tb, bool with_backward) {
  const size_t need = sizeof(double) * dlm::covtabs_doubles(k.d, k.T);
  if (need > e->covws_bytes) {
    if (e->covws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->covws)); e->covws = nullptr; e->covws_bytes = 0; }
    HIP_TRY(e, hipMalloc((void**)&e->covws, need));
    e->covws_bytes = need;
  }
  int rc = ensure_route(e, (size_t)k.N);
  if (rc) return rc;
  rc = ensure_side(e, k);
  if (rc) return rc;
  if ((rc = ensure_cov_stream(e))) return rc;
  dlm::covtabs_carve(e->covws, k.d, k.T, tb);
  tb.eq = e->side + 2 * (size_t)k.N * ((size_t)k.T + 1);
  tb.mc = tb.sc = nullptr;
  if (with_backward) {   // compact means of the mean-only kernels: 512 bytes per step and group of four series, filtered then smoothed
    const size_t one = 64 * (size_t)((k.N + 3) / 4) * ((size_t)k.T + 1);
    int rc2 = ensure_xplus_bytes(e, sizeof(double) * one);
    if (rc2) return rc2;
    tb.mc = e->xplus;
  }
  return DLM_OK;
}
// forward half: covariance-only run, mean kernel, then the per-series kernel for the series the mean kernel routed away.
// with_backward: the backward covariance run starts on the engine's second stream as soon as the forward tables exist and
// overlaps the forward mean kernel (it needs nothing from the data).
int run_shared_filter(dlm_engine* e, KArgs& k, dlm::CovTabs& tb, bool with_backward) {
  int rc = ensure_shared(e, k, tb, with_backward);
  if (rc) return rc;
  k.route = e->route; k.route_take = 0;
  e->variant = "sparse16";   // (the same kernel family; dlm_last_counters[2] tells the series served by the shared-covariance kernels)
  HIP_TRY(e, dlm::launch_sparse16_cov_filter(k, e->sparse_k, e->sp_dev, tb, e->stream));
  if (with_backward) {
    HIP_TRY(e, hipEventRecord(e->cov_ev[0], e->stream));
    e->cov_busy = true;
    HIP_TRY(e, hipStreamWaitEvent(e->cov_stream, e->cov_ev[0], 0));
    HIP_TRY(e, dlm::launch_sparse16_cov_smoother(k, e->sparse_k, e->sp_dev, tb, e->cov_stream));
    HIP_TRY(e, hipEventRecord(e->cov_ev[1], e->cov_stream));
  }
  HIP_TRY(e, dlm::launch_sparse16_mean_filter(k, e->sparse_k, e->sp_dev, tb, e->stream));
  KArgs kg = k;
  kg.route_take = 1;
  HIP_TRY(e, dlm::launch_sparse16_filter(kg, e->sparse_k, e->sp_dev, with_backward ? e->side : nullptr, nullptr, e->stream));
  return DLM_OK;
}
int run_shared_smoother(dlm_engine* e, KArgs& k, const dlm::CovTabs& tb) {
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->cov_ev[1], 0));   // the S_t table
  e->cov_busy = false;                                          // (cov_ev[1] was recorded behind the stream's last operation of this call)
  HIP_TRY(e, dlm::launch_sparse16_mean_smoother(k, e->sparse_k, e->sp_dev, tb, e->stream));
  KArgs kg = k;
  kg.route_take = 1;
  HIP_TRY(e, dlm::launch_sparse16_smoother(kg, e->sparse_k, e->sp_dev, e->side, e->stream));
  return DLM_OK;
}

// want_side: the caller will run the fast backward pass on this filter's output
int run_filter(dlm_engine* e, const KArgs& k, bool want_side) {
  if (use_lane(k)) {
    e->variant = "lane";
    HIP_TRY(e, dlm::launch_lane_filter(k, e->stream));
  } else if (use_fast(e, k) && (!k.prior || e->sparse_k)) {   // the dense-G MFMA kernel does not write (a, R) records
    if (want_side) { int rc = ensure_side(e, k); if (rc) return rc; }
    double* side = want_side ? e->side : nullptr;
    if (e->sparse_k) {
      e->variant = "sparse16";
      HIP_TRY(e, dlm::launch_sparse16_filter(k, e->sparse_k, e->sp_dev, side, nullptr, e->stream));
    } else {
      e->variant = "mfma16";
      HIP_TRY(e, dlm::launch_mfma16_filter(k, side, e->stream));
    }
  } else if (use_tiled(k)) {
    e->variant = dlm::wave48_filter_supported(k) ? "wave-mfma" : "tiled-mfma";
    if (want_side) { int rc = ensure_ystar(e, k); if (rc) return rc; }
    HIP_TRY(e, dlm::launch_tiled_filter(k, want_side ? e->ystar : nullptr, e->stream));
  } else {
    e->variant = "generic";
    if (dlm::generic_filter_lds_bytes(k.d, k.p) > 160 * 1024)   // 3 d + 4 d^2 + 2 d p + 2 p^2 + 3 p doubles of one CU's 160 KB of LDS
      return fail(e, DLM_ERR_UNSUPPORTED, "the general filter kernel keeps a series' matrices in the LDS of one CU: this d, p needs more than its 160 KB (d = 64 fits with p <= 12, p = 64 with d <= 26)");
    HIP_TRY(e, dlm::launch_generic_filter(k, e->stream));
  }
  return DLM_OK;
}

bool fast_smoother_ok(const dlm_engine* e, const KArgs& k) { return use_fast(e, k) && !(k.flags & DLM_OPT_SMOOTHER_COMPAT_Q1); }

// have_side: the preceding run_filter(.., want_side = true) of THIS call filled e->side
int run_smoother(dlm_engine* e, const KArgs& k, bool have_side) {
  if (use_lane(k) && !(k.flags & DLM_OPT_SMOOTHER_COMPAT_Q1) && !k.packed) {
    e->variant = "lane";   // needs nothing from the forward pass: also serves dlm_smooth_batch
    HIP_TRY(e, dlm::launch_lane_smoother(k, e->stream));
  } else if (have_side && fast_smoother_ok(e, k)) {
    if (e->sparse_k) {
      e->variant = "sparse16";
      HIP_TRY(e, dlm::launch_sparse16_smoother(k, e->sparse_k, e->sp_dev, e->side, e->stream));
    } else {
      e->variant = "mfma16";
      HIP_TRY(e, dlm::launch_mfma16_smoother(k, e->side, e->stream));
    }
  } else if (have_side && use_tiled(k) && !(k.flags & DLM_OPT_SMOOTHER_COMPAT_Q1)) {
    e->variant = dlm::wave48_smoother_supported(k) ? "wave-mfma" : "tiled-mfma";   // fused call only: the information-form pass takes the innovations of the forward pass
    HIP_TRY(e, dlm::launch_tiled_smoother(k, e->ystar, e->stream));
  } else if (fast_shape_ok(k) && e->sparse_k > 0 && !(k.packed & 1)) {
    // RTS recursion on register tiles from the records alone: dlm_smooth_batch, and the literal Q1 covariance (Smoothing.scala:44)
    e->variant = "sparse16-rts";
    HIP_TRY(e, dlm::launch_sparse16_rts(k, e->sparse_k, e->sp_dev, e->stream));
  } else if (!(k.flags & DLM_OPT_FORCE_GENERIC) && dlm::wave48_small_ok(k)) {
    e->variant = "sparse16-rts";   // d <= 15 with several observation components: the same kernel on the multivariate tables
    HIP_TRY(e, dlm::launch_small_mv_rts(k, e->stream));
  } else {
    e->variant = "generic";
    if (dlm::generic_smoother_lds_bytes(k.d, k.p) > 160 * 1024)
      return fail(e, DLM_ERR_UNSUPPORTED, "the general RTS smoother kernel keeps six d x d matrices in the LDS of one CU: d <= 58 (structured models with d <= 48 take the per-wave kernels)");
    HIP_TRY(e, dlm::launch_generic_smoother(k, e->stream));
  }
  return DLM_OK;
}

// packed [m | lower triangle by rows] -> dense [m | C column-major]; one thread per dense element
__global__ void k_unpack_records(int d, long long count, int prec, const double* __restrict__ packed, double* __restrict__ dense) {
  const int rec = d + d * d;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count * rec) return;
  const long long r = i / rec;
  const int e = (int)(i - r * rec);
  const double* src = packed + r * prec;
  if (e < d) { dense[i] = src[e]; return; }
  const int a = (e - d) % d, b = (e - d) / d;           // element (row a, column b) of C
  const int hi = a > b ? a : b, lo = a > b ? b : a;
  dense[i] = src[d + hi * (hi + 1) / 2 + lo];
}

}  // namespace

extern "C" {

const char* dlm_version(void) { return "bayesian_dlms_amd 0.1.0 (gfx950)"; }

int dlm_engine_create(int device, dlm_engine** out) {
  if (!out) return DLM_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return DLM_ERR_HIP;
  dlm_engine* e = new dlm_engine();
  e->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete e;
    return DLM_ERR_HIP;
  }
  e->stream = e->own_stream;
  for (auto& ev : e->ev) (void)hipEventCreate(&ev);
  (void)hipEventCreateWithFlags(&e->order_ev, hipEventDisableTiming);
  *out = e;
  return DLM_OK;
}

void dlm_engine_destroy(dlm_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)drain_all(e);   // rule B: nothing of a DLM_OPT_ASYNC call (or of a call that failed half-way) is in flight on any of the three streams
  if (e->has_comm) ncclCommDestroy(e->comm);
  if (e->arena) (void)hipFree(e->arena);
  if (e->side) (void)hipFree(e->side);
  if (e->xplus) (void)hipFree(e->xplus);
  if (e->ystar) (void)hipFree(e->ystar);
  if (e->sp_dev) (void)hipFree(e->sp_dev);
  if (e->fws) (void)hipFree(e->fws);
  if (e->spb_dev) (void)hipFree(e->spb_dev);
  if (e->spf_dev) (void)hipFree(e->spf_dev);
  if (e->covws) (void)hipFree(e->covws);
  if (e->route) (void)hipFree(e->route);
  if (e->plainbuf) (void)hipFree(e->plainbuf);
  if (e->sampws) (void)hipFree(e->sampws);
  if (e->zws) (void)hipFree(e->zws);
  if (e->rng_ev) (void)hipEventDestroy(e->rng_ev);
  if (e->rng_stream) (void)hipStreamDestroy(e->rng_stream);
  for (auto& ev : e->cov_ev) if (ev) (void)hipEventDestroy(ev);
  if (e->cov_ev2) (void)hipEventDestroy(e->cov_ev2);
  if (e->rng_gate) (void)hipEventDestroy(e->rng_gate);
  if (e->cov_stream) (void)hipStreamDestroy(e->cov_stream);
  if (e->counters) (void)hipFree(e->counters);
  if (e->model_sum) (void)hipFree(e->model_sum);
  if (e->model_bad) (void)hipHostFree(e->model_bad);
  for (void* b : e->buffers) (void)hipFree(b);
  if (e->order_ev) (void)hipEventDestroy(e->order_ev);
  for (auto& ev : e->ev) if (ev) (void)hipEventDestroy(ev);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

const char* dlm_last_error(const dlm_engine* e) { return e ? e->err.c_str() : "null engine"; }
const char* dlm_last_variant(const dlm_engine* e) { return e ? e->variant : "none"; }

int dlm_engine_set_stream(dlm_engine* e, void* hip_stream) {
  if (!e) return DLM_ERR_ARG;
  hipStream_t next = hip_stream ? (hipStream_t)hip_stream : e->own_stream;
  if (next != e->stream) {
    // the engine's workspaces (side records, tables, staged parameters) belong to whatever runs on its stream: work still in
    // flight from DLM_OPT_ASYNC calls on the old stream is drained before anything is launched on the new one
    HIP_TRY(e, hipSetDevice(e->device));
    { const int rcd = drain_all(e); if (rcd) return rcd; }
    e->stream = next;
  }
  return DLM_OK;
}

int dlm_engine_sync(dlm_engine* e) {
  if (!e) return DLM_ERR_ARG;
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  if (e->model_bad && *e->model_bad) return promise_broken(e);   // a DLM_OPT_ASYNC call whose DLM_OPT_MODEL_UNCHANGED did not hold
  return DLM_OK;
}

int dlm_last_counters(dlm_engine* e, uint64_t out[4]) {
  if (!e || !out) return DLM_ERR_ARG;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!e->counters) return fail(e, DLM_ERR_ARG, "no call with DLM_OPT_COUNT_STEPS has been made");
  HIP_TRY(e, hipSetDevice(e->device));
  unsigned long long h[4];
  HIP_TRY(e, hipMemcpyAsync(h, e->counters, sizeof(h), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  for (int i = 0; i < 4; ++i) out[i] = h[i];
  return DLM_OK;
}

int dlm_engine_wait_stream(dlm_engine* e, void* hip_stream) {
  if (!e) return DLM_ERR_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  if ((hipStream_t)hip_stream == e->stream) return DLM_OK;
  HIP_TRY(e, hipEventRecord(e->order_ev, (hipStream_t)hip_stream));
  HIP_TRY(e, hipStreamWaitEvent(e->stream, e->order_ev, 0));
  return DLM_OK;
}

int dlm_stream_wait_engine(dlm_engine* e, void* hip_stream) {
  if (!e) return DLM_ERR_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  if ((hipStream_t)hip_stream == e->stream) return DLM_OK;
  HIP_TRY(e, hipEventRecord(e->order_ev, e->stream));
  HIP_TRY(e, hipStreamWaitEvent((hipStream_t)hip_stream, e->order_ev, 0));
  return DLM_OK;
}

int dlm_buffer_alloc(dlm_engine* e, uint64_t bytes, void** dev_ptr) {
  if (!e) return DLM_ERR_ARG;
  if (!dev_ptr || bytes == 0) return fail(e, DLM_ERR_ARG, "dlm_buffer_alloc: dev_ptr and a non-zero size are required");
  *dev_ptr = nullptr;
  HIP_TRY(e, hipSetDevice(e->device));
  void* p = nullptr;
  HIP_TRY(e, hipMalloc(&p, (size_t)bytes));
  e->buffers.insert(p);
  *dev_ptr = p;
  return DLM_OK;
}

int dlm_buffer_free(dlm_engine* e, void* dev_ptr) {
  if (!e) return DLM_ERR_ARG;
  if (!dev_ptr) return DLM_OK;
  auto it = e->buffers.find(dev_ptr);
  if (it == e->buffers.end()) return fail(e, DLM_ERR_ARG, "dlm_buffer_free: not a buffer of this engine");
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipStreamSynchronize(e->stream));   // no engine work may still use it
  HIP_TRY(e, hipFree(dev_ptr));
  e->buffers.erase(it);
  return DLM_OK;
}

int dlm_buffer_upload(dlm_engine* e, void* dst_dev, uint64_t dst_offset, const void* src_host, uint64_t bytes) {
  if (!e) return DLM_ERR_ARG;
  if (!dst_dev || (!src_host && bytes)) return fail(e, DLM_ERR_ARG, "dlm_buffer_upload: null pointer");
  if (!bytes) return DLM_OK;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemcpyAsync((char*)dst_dev + dst_offset, src_host, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return DLM_OK;
}

int dlm_buffer_download(dlm_engine* e, const void* src_dev, uint64_t src_offset, void* dst_host, uint64_t bytes) {
  if (!e) return DLM_ERR_ARG;
  if (!src_dev || (!dst_host && bytes)) return fail(e, DLM_ERR_ARG, "dlm_buffer_download: null pointer");
  if (!bytes) return DLM_OK;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemcpyAsync(dst_host, (const char*)src_dev + src_offset, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return DLM_OK;
}

int dlm_buffer_fill(dlm_engine* e, void* dst_dev, uint64_t dst_offset, int byte_value, uint64_t bytes) {
  if (!e) return DLM_ERR_ARG;
  if (!dst_dev) return fail(e, DLM_ERR_ARG, "dlm_buffer_fill: null pointer");
  if (!bytes) return DLM_OK;
  HIP_TRY(e, hipSetDevice(e->device));
  HIP_TRY(e, hipMemsetAsync((char*)dst_dev + dst_offset, byte_value, (size_t)bytes, e->stream));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return DLM_OK;
}

int dlm_device_mem_info(dlm_engine* e, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (!e) return DLM_ERR_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  size_t f = 0, t = 0;
  HIP_TRY(e, hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return DLM_OK;
}

int dlm_last_timing(dlm_engine* e, double ms[2]) {
  if (!e || !ms) return DLM_ERR_ARG;
  if (!e->timed) return fail(e, DLM_ERR_ARG, "no forward / backward call has been timed yet");
  float f = 0.f, b = 0.f;
  HIP_TRY(e, hipEventSynchronize(e->ev[2]));
  HIP_TRY(e, hipEventElapsedTime(&f, e->ev[0], e->ev[1]));
  HIP_TRY(e, hipEventElapsedTime(&b, e->ev[1], e->ev[2]));
  ms[0] = f; ms[1] = b;
  return DLM_OK;
}

int32_t dlm_stats_len(int32_t d, int32_t p, uint32_t flags) { return dlm::stats_len(d, p, flags); }

int32_t dlm_packed_record_doubles(int32_t d) { return d < 1 ? 0 : dlm::packed_rec_bytes(d) / 8; }

int dlm_unpack_records(dlm_engine* e, int32_t d, int64_t count, const double* packed, const dlm_options* opts, double* dense) {
  if (!e) return DLM_ERR_ARG;
  if (!opts || (opts->mem != DLM_MEM_DEVICE && opts->mem != DLM_MEM_HOST)) return fail(e, DLM_ERR_ARG, "opts");
  if (d < 1 || d > 64 || count < 1 || !packed || !dense) return fail(e, DLM_ERR_ARG, "dlm_unpack_records arguments");
  HIP_TRY(e, hipSetDevice(e->device));
  const size_t prec = (size_t)dlm_packed_record_doubles(d), rec = (size_t)d + (size_t)d * d;
  const double* src = nullptr; double* dst = nullptr;
  Stager st(e, opts->mem == DLM_MEM_HOST);
  st.in(&src, packed, (size_t)count * prec);
  st.out(&dst, dense, (size_t)count * rec);
  int rc = st.commit();
  if (rc) return rc;
  const long long total = (long long)count * (long long)rec;
  hipLaunchKernelGGL(k_unpack_records, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, e->stream, (int)d, (long long)count, (int)prec, src, dst);
  HIP_TRY(e, hipGetLastError());
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_filter_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                     const double* y, const dlm_options* opts, double* filt, double* prior,
                     double* fq, int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!y || !filt) return fail(e, DLM_ERR_ARG, "y and filt are required");
  const size_t d = model->d, p = model->p, T = model->T, N = model->N, rec = d + d * d;
  const bool want_packed = (opts->flags & DLM_OPT_PACKED_SYM) != 0;
  if (want_packed && prior) return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_PACKED_SYM: prior records are not available packed");
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  st.in(&k.y, y, N * T * p);
  st.out(&k.filt, filt, N * (T + 1) * (want_packed ? (size_t)dlm_packed_record_doubles((int)d) : rec));
  st.out(&k.prior, prior, N * (T + 1) * rec);
  st.out(&k.fq, fq, N * (T + 1) * (p + p * p));
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  if ((rc = analyse_g(e, k, model->G, opts->mem == DLM_MEM_HOST))) return rc;
  if (want_packed) {
    if ((rc = packed_path(e, k))) return rc;
    k.packed = 1;
  }
  if ((rc = mark_plain(e, k))) return rc;
  if (use_shared_cov(e, k)) {
    dlm::CovTabs tb;
    if ((rc = run_shared_filter(e, k, tb, false))) return rc;
    return st.finish(opts->flags & DLM_OPT_ASYNC);
  }
  if ((rc = run_filter(e, k, false))) return rc;
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_simulate_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                       const dlm_options* opts, double* x, double* y, int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!y) return fail(e, DLM_ERR_ARG, "y is required");
  const size_t d = model->d, p = model->p, T = model->T, N = model->N;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  st.out(&k.theta, x, x ? N * (T + 1) * d : 0);
  st.out(&k.smooth, y, N * T * p);
  st.zeroed_out(&k.status, (int*)status, status ? N : 0);
  if ((rc = st.commit())) return rc;
  e->variant = "generic-simulate";
  HIP_TRY(e, dlm::launch_generic_simulate(k, e->stream));
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_dinvgamma_step_batch(dlm_engine* e, int32_t d, int32_t p, int32_t N, const double* stats, double alpha_v,
                             double beta_v, double alpha_w, double beta_w, uint64_t iteration, const dlm_options* opts,
                             double* V_out, double* W_out) {
  if (!e) return DLM_ERR_ARG;
  if (!opts || (opts->mem != DLM_MEM_DEVICE && opts->mem != DLM_MEM_HOST)) return fail(e, DLM_ERR_ARG, "opts");
  if (d < 1 || p < 1 || N < 1 || !stats || !V_out || !W_out) return fail(e, DLM_ERR_ARG, "d, p, N >= 1; stats, V_out, W_out required");
  if (!(alpha_v > 0.0 && beta_v > 0.0 && alpha_w > 0.0 && beta_w > 0.0)) return fail(e, DLM_ERR_ARG, "InverseGamma priors need positive shape and scale");
  HIP_TRY(e, hipSetDevice(e->device));
  struct { const double* stats; double *V, *W; } k{};
  const size_t n = N, L = 2 * (size_t)p + d + 1;
  Stager st(e, opts->mem == DLM_MEM_HOST);
  st.in(&k.stats, stats, n * L);
  st.out(&k.V, V_out, n * p * p);
  st.out(&k.W, W_out, n * d * d);
  int rc;
  if ((rc = st.commit())) return rc;
  e->variant = "dinvgamma-step";
  HIP_TRY(e, dlm::launch_dinvgamma_step(d, p, N, k.stats, alpha_v, beta_v, alpha_w, beta_w, opts->seed, opts->series_offset,
                                        iteration, k.V, k.W, e->stream));
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

static int ar1_common(dlm_engine* e, int32_t N, int32_t T, const double* times, bool ou, const double* y, const double* v,
                      int64_t v_stride, const double* sv, int64_t sv_stride, const double* z, const dlm_options* opts,
                      double* filt, double* theta, int32_t* status);

int dlm_ar1_ffbs_batch(dlm_engine* e, int32_t N, int32_t T, const double* y, const double* v, int64_t v_stride,
                       const double* sv, int64_t sv_stride, const double* z, const dlm_options* opts,
                       double* filt, double* theta, int32_t* status) {
  return ar1_common(e, N, T, nullptr, false, y, v, v_stride, sv, sv_stride, z, opts, filt, theta, status);
}

int dlm_ou_ffbs_batch(dlm_engine* e, int32_t N, int32_t T, const double* times, const double* y, const double* v,
                      int64_t v_stride, const double* sv, int64_t sv_stride, const double* z, const dlm_options* opts,
                      double* filt, double* theta, int32_t* status) {
  if (e && !times) return fail(e, DLM_ERR_ARG, "times are required");
  return ar1_common(e, N, T, times, true, y, v, v_stride, sv, sv_stride, z, opts, filt, theta, status);
}

static int ar1_common(dlm_engine* e, int32_t N, int32_t T, const double* times, bool ou, const double* y, const double* v,
                      int64_t v_stride, const double* sv, int64_t sv_stride, const double* z, const dlm_options* opts,
                      double* filt, double* theta, int32_t* status) {
  if (!e) return DLM_ERR_ARG;
  if (!opts || (opts->mem != DLM_MEM_DEVICE && opts->mem != DLM_MEM_HOST)) return fail(e, DLM_ERR_ARG, "opts");
  if (N < 1 || T < 1) return fail(e, DLM_ERR_ARG, "N and T must be >= 1 (the reference throws on empty input, FilterAr.scala:40)");
  if (!y || !v || !sv) return fail(e, DLM_ERR_ARG, "y, v and sv are required");
  if (!filt && !theta) return fail(e, DLM_ERR_ARG, "nothing to compute: filt and theta are both NULL");
  if ((v_stride != 0 && v_stride != T) || (sv_stride != 0 && sv_stride != 3)) return fail(e, DLM_ERR_ARG, "v_stride must be 0 or T, sv_stride 0 or 3");
  HIP_TRY(e, hipSetDevice(e->device));
  struct { const double *times, *y, *v, *sv, *z; double *filt, *theta; int* status; } k{};
  const size_t n = N, t = T;
  Stager st(e, opts->mem == DLM_MEM_HOST);
  st.in(&k.times, times, ou ? t : 0);
  st.in(&k.y, y, n * t);
  st.in(&k.v, v, v_stride ? n * t : t);
  st.in(&k.sv, sv, sv_stride ? n * 3 : 3);
  st.in(&k.z, z, z ? n * (t + 1) : 0);
  st.out(&k.filt, filt, filt ? n * (t + 1) * 2 : 0);
  st.out(&k.theta, theta, theta ? n * (t + 1) : 0);
  st.zeroed_out(&k.status, (int*)status, status ? n : 0);
  int rc;
  if ((rc = st.commit())) return rc;
  double* fws = k.filt;
  if (!fws) {   // the backward sampler reads the filter records: engine scratch when the caller does not want them
    KArgs ks{}; ks.N = N; ks.T = T;
    if ((rc = ensure_side(e, ks))) return rc;
    fws = e->side;
  }
  e->variant = ou ? "ou-lane" : "ar1-lane";
  HIP_TRY(e, dlm::launch_ar1_ffbs(N, T, k.times, k.y, k.v, v_stride, k.sv, sv_stride, k.z, opts->seed, opts->series_offset, fws,
                                  k.theta, k.status, e->stream));
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_loglik_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                     const double* y, const dlm_options* opts, double* loglik, int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!y || !loglik) return fail(e, DLM_ERR_ARG, "y and loglik are required");
  const size_t p = model->p, T = model->T, N = model->N;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  st.in(&k.y, y, N * T * p);
  st.out(&k.loglik, loglik, N);
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  if ((rc = analyse_g(e, k, model->G, opts->mem == DLM_MEM_HOST))) return rc;
  if (opts->flags & DLM_OPT_LOGLIK_LITERAL_Q7) {
    // KalmanFilter.likelihood as written (KalmanFilter.scala:299-306): filter into a workspace, then the transition density of
    // the filtered means (dlm_loglik.hip)
    if (params->w_tstride) return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_LOGLIK_LITERAL_Q7 takes a time-invariant W (KalmanFilter.likelihood is given p.w)");
    const size_t d = model->d, rec = d + d * d;
    const size_t recs = N * (T + 1) * rec * sizeof(double), wsb = dlm::loglik_q7_ws_bytes(k);
    if ((rc = ensure_fws(e, recs + wsb))) return rc;
    double* ll_out = k.loglik;
    k.filt = e->fws; k.loglik = nullptr;
    if ((rc = run_filter(e, k, false))) return rc;
    k.loglik = ll_out;
    HIP_TRY(e, dlm::launch_loglik_q7(k, e->fws, (char*)e->fws + recs, e->stream));
    return st.finish(opts->flags & DLM_OPT_ASYNC);
  }
  if ((rc = run_filter(e, k, false))) return rc;   // k.filt == nullptr: the forward kernels store nothing
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_smooth_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                     const double* filt, const dlm_options* opts, double* smooth, int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!filt || !smooth) return fail(e, DLM_ERR_ARG, "filt and smooth are required");
  if (opts->flags & DLM_OPT_PACKED_SYM) return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_PACKED_SYM: packed records are read by the fused call only (dlm_filter_smooth_batch); dlm_smooth_batch takes dense records (dlm_unpack_records)");
  const size_t d = model->d, T = model->T, N = model->N, rec = d + d * d;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  st.in(&k.filt_in, filt, N * (T + 1) * rec);
  st.out(&k.smooth, smooth, N * (T + 1) * rec);
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  if ((rc = analyse_g(e, k, model->G, opts->mem == DLM_MEM_HOST))) return rc;   // structure tables for the register-tile RTS kernel
  if ((rc = mark(e, 0)) || (rc = mark(e, 1))) return rc;
  if ((rc = run_smoother(e, k, false))) return rc;
  if ((rc = mark(e, 2))) return rc;
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_filter_smooth_batch(dlm_engine* e, const dlm_model_desc* model,
                            const dlm_params_desc* params, const double* y,
                            const dlm_options* opts, double* filt, double* smooth,
                            int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!y || !smooth) return fail(e, DLM_ERR_ARG, "y and smooth are required");
  AuxScope aux(e);   // (the shared-covariance path runs its backward covariance table on the second stream)
  const size_t d = model->d, p = model->p, T = model->T, N = model->N, rec = d + d * d;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  const bool want_packed = (opts->flags & DLM_OPT_PACKED_SYM) != 0;
  const size_t orec = want_packed ? (size_t)dlm_packed_record_doubles((int)d) : rec;
  st.in(&k.y, y, N * T * p);
  st.out(&k.filt, filt, filt ? N * (T + 1) * orec : 0);
  st.out(&k.smooth, smooth, N * (T + 1) * orec);
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  if ((rc = analyse_g(e, k, model->G, opts->mem == DLM_MEM_HOST))) return rc;
  if (want_packed && (rc = packed_path(e, k))) return rc;
  bool fused_fast = fast_smoother_ok(e, k) || use_tiled(k);
  if (want_packed) k.packed = 3;   // filtered and smoothed records both leave packed
  // Structured d <= 15 path, V, W, C0 shared by the batch: J_t, S_t of the RTS recursion once per call, every series its mean recursion
  // (k_smoother_rts16 with its export on + k_mean_rts16).  Literal Q1: from 2048 series; textbook: from 6144 series (a caller
  // that does not take the filtered records gets them written to the engine's workspace, dense: the mean kernel reads their first 128 bytes).
  const bool q1 = (k.flags & DLM_OPT_SMOOTHER_COMPAT_Q1) != 0;
  const bool rts_shared = fast_shape_ok(k) && e->sparse_k > 0 && !use_lane(k) && !k.packed && !(k.flags & DLM_OPT_SHARED_COV) &&
                          (q1 ? !fused_fast : fast_smoother_ok(e, k)) && dlm::rts_shared_eligible(k, !q1);
  // (textbook: the series with a missing observation keep the information-form kernel, k_smoother_sp16, and the forward pass its side records for
  //  them -- the RTS kernel per series is five times slower; literal Q1 has only that kernel)
  if (!filt) {   // smoothed moments only: the filtered records stay in an engine workspace, packed on the structured path
    k.packed |= (fast_smoother_ok(e, k) && e->sparse_k > 0 && !use_lane(k) && !rts_shared) ? 1 : 0;
    if ((rc = ensure_fws(e, (k.packed & 1) ? N * (T + 1) * (size_t)dlm::packed_rec_bytes((int)d) : N * (T + 1) * rec * sizeof(double)))) return rc;
    k.filt = e->fws;
  }
  if ((rc = mark(e, 0))) return rc;
  dlm::RtsTabs rtb{};
  if (rts_shared && (rc = start_rts_tables(e, k, rtb))) return rc;
  if (rts_shared && (k.flags & DLM_OPT_TEST_FAIL_AFTER_TABLES))   // test hook: an error exit with the second stream busy
    return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_TEST_FAIL_AFTER_TABLES");
  if ((rc = mark_plain(e, k))) return rc;
  if (filt && fast_smoother_ok(e, k) && use_shared_cov(e, k)) {
    dlm::CovTabs tb;
    if ((rc = run_shared_filter(e, k, tb, true))) return rc;
    if ((rc = mark(e, 1))) return rc;
    k.filt_in = k.filt;
    if ((rc = run_shared_smoother(e, k, tb))) return rc;
    if ((rc = mark(e, 2))) return rc;
    return st.finish(opts->flags & DLM_OPT_ASYNC);
  }
  if ((rc = run_filter(e, k, fused_fast))) return rc;
  if (rts_shared) {
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->cov_ev[1], 0));   // the tables (in front of the timing mark: the backward time is the mean kernel's)
    e->cov_busy = false;
  }
  if ((rc = mark(e, 1))) return rc;
  k.filt_in = k.filt;
  if (rts_shared) {
    k.route = e->route; k.route_take = 0;
    e->variant = "sparse16-rts-shared";
    HIP_TRY(e, dlm::launch_rts_shared_means(k, e->sparse_k, e->sp_dev, rtb, q1, e->stream));
    if (!q1) {
      KArgs kg = k;
      kg.route_take = 1;
      HIP_TRY(e, dlm::launch_sparse16_smoother(kg, e->sparse_k, e->sp_dev, e->side, e->stream));
    }
  } else if ((rc = run_smoother(e, k, fused_fast))) return rc;
  if ((rc = mark(e, 2))) return rc;
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

static int sampler_common(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                          const double* y, const double* z, const dlm_options* opts,
                          const double* filt_in, double* filt_ws, double* theta, double* cond,
                          double* stats, int32_t* status, bool forward) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  AuxScope aux(e);                            // rule A of the auxiliary streams on every exit path below
  const bool norec = forward && !filt_ws;     // dlm_ffbs_batch that does not want the filter records
  if (forward && !y) return fail(e, DLM_ERR_ARG, "y is required");
  if (!forward && !filt_in) return fail(e, DLM_ERR_ARG, "filter records are required");
  if (stats && !y) return fail(e, DLM_ERR_ARG, "sufficient statistics need y");
  const size_t d = model->d, p = model->p, T = model->T, N = model->N, rec = d + d * d;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  const bool simflag = forward && (opts->flags & DLM_OPT_FFBS_SIMSMOOTH) && !cond;
  const bool eig = (opts->flags & DLM_OPT_DRAW_EIG) != 0;   // the reference's eigen-factor for the draw: the general kernel's backward pass (any forward kernel)
  if (eig && simflag) return fail(e, DLM_ERR_ARG, "DLM_OPT_DRAW_EIG selects the factor of the reference-form backward sampler; the simulation smoother (DLM_OPT_FFBS_SIMSMOOTH) factors nothing per step");
  bool shared_factors = false, shared_big = false;
  dlm::SampTabs stb{};
  // On the structured d <= 15, p = 1 path a V_t stream is a scalar per step (the Student-t DLM, StudentTGibbs.scala:100-136) and a W_t
  // stream (DlmFsvSystem.scala:137-208 feeds one W per step) a 13-pivot Cholesky factor per step; elsewhere they would need the
  // factorisation of a p x p / 48 x 48 matrix per step: the reference-form sampler serves those callers
  if (simflag && params->v_tstride && !(model->p == 1 && model->d <= 15))
    return fail(e, DLM_ERR_UNSUPPORTED, "the simulation smoother takes a V_t stream only on the structured d <= 15, p = 1 path (drop DLM_OPT_FFBS_SIMSMOOTH otherwise)");
  st.in(&k.y, y, y ? N * T * p : 0);
  st.in(&k.z, z, z ? N * (T + 1) * (simflag ? d + p : d) : 0);
  if (forward && !norec) st.out(&k.filt, filt_ws, N * (T + 1) * rec);
  else if (!forward) st.in(&k.filt_in, filt_in, N * (T + 1) * rec);
  st.out(&k.theta, theta, N * (T + 1) * d);
  st.out(&k.cond, cond, N * (T + 1) * rec);
  st.out(&k.stats, stats, N * (size_t)dlm_stats_len(model->d, model->p, opts->flags));
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  // the stretches of the reference-form sampler (KArgs::stretches): wherever a shared-factor table is possible for these parameters,
  // whether or not this call uses one -- a series' draws do not depend on the route it takes
  k.stretches = (dlm::sampler_shared_model_ok(k) || dlm::wave48_sampler_shared_model_ok(k)) ? 1 : 0;
  if (norec) {   // the records of the forward pass are nobody's output: an engine workspace
    if ((rc = ensure_fws(e, N * (T + 1) * rec * sizeof(double)))) return rc;
    k.filt = e->fws;
  }
  if (forward) {
    if ((rc = analyse_g(e, k, model->G, opts->mem == DLM_MEM_HOST))) return rc;
    if ((rc = mark(e, 0))) return rc;
    if (simflag && use_lane(k) && !k.v_tstride && !k.w_tstride) {
      if ((rc = ensure_xplus(e, k))) return rc;
      e->variant = "lane-simsmooth";
      HIP_TRY(e, dlm::launch_lane_simsmooth(k, e->xplus, e->stream));
      if ((rc = mark(e, 1)) || (rc = mark(e, 2))) return rc;
      return st.finish(opts->flags & DLM_OPT_ASYNC);
    }
    if (simflag && e->sparse_k) {
      // Durbin-Koopman simulation smoother on the structured fast path
      if ((rc = ensure_side(e, k)) || (rc = ensure_xplus(e, k))) return rc;
      e->variant = "sparse16-simsmooth";
      HIP_TRY(e, dlm::launch_sparse16_filter(k, e->sparse_k, e->sp_dev, e->side, e->xplus, e->stream));
      if ((rc = mark(e, 1))) return rc;
      k.filt_in = k.filt;
      HIP_TRY(e, dlm::launch_sparse16_simsmooth(k, e->sparse_k, e->sp_dev, e->side, e->xplus, e->stream));
      if ((rc = mark(e, 2))) return rc;
      return st.finish(opts->flags & DLM_OPT_ASYNC);
    }
    if (simflag && (k.v_tstride || (k.w_tstride && !use_tiled(k))))
      return fail(e, DLM_ERR_UNSUPPORTED, "a V_t / W_t stream with DLM_OPT_FFBS_SIMSMOOTH needs a structured G (this one is dense); V_t also p = 1");
    if (simflag && use_tiled(k)) {
      if (k.w_tstride) {   // a W_t stream (DlmFsvSystem.ffbs): the per-wave kernels, whose simulation prologue factors W_t at every step -- at any batch size
        k.flags |= DLM_OPT_FORCE_WAVE;
        if (!dlm::wave48_simsmooth_supported(k))
          return fail(e, DLM_ERR_UNSUPPORTED, "a W_t stream with DLM_OPT_FFBS_SIMSMOOTH needs a structured G (at most four nonzeros per row and column); the reference-form sampler takes any model");
      }
      if ((rc = ensure_xplus(e, k)) || (rc = ensure_ystar(e, k))) return rc;
      e->variant = dlm::wave48_simsmooth_supported(k) ? "wave-simsmooth" : "tiled-simsmooth";
      HIP_TRY(e, dlm::launch_tiled_simsmooth(k, e->xplus, e->ystar, e->stream));
      if ((rc = mark(e, 1)) || (rc = mark(e, 2))) return rc;
      return st.finish(opts->flags & DLM_OPT_ASYNC);
    }
    if (simflag && z) return fail(e, DLM_ERR_UNSUPPORTED, "injected normals with DLM_OPT_FFBS_SIMSMOOTH need a fast path (structured d <= 15 or 16 <= d <= 48)");
    shared_factors = !simflag && !eig && !use_lane(k) && fast_shape_ok(k) && e->sparse_k > 0 && dlm::sampler_shared_eligible(k);
    shared_big = !simflag && !eig && !shared_factors && use_tiled(k) && dlm::wave48_sampler_shared_eligible(k);   // 16 <= d <= 48 on the per-wave kernels
    if (norec && shared_factors) {
      // No records AND shared factors (d <= 15): nothing of the filtered covariances is needed per series -- the draw kernel reads the
      // means, the table is made from the covariance recursion alone.  So the forward pass is section 4.9's: one wave's covariance
      // recursion into a table, a mean-only kernel per series (compact means, 128 instead of 1456 bytes per series-step at d = 13);
      // the table of factors is made from that covariance table while the mean kernel runs; a series with a missing observation is
      // marked by the mean kernel and runs its own filter and sampler on the workspace.
      dlm::CovTabs ctb;
      if ((rc = ensure_shared(e, k, ctb, true))) return rc;
      k.route = e->route; k.route_take = 0; k.filt = nullptr;
      HIP_TRY(e, dlm::launch_sparse16_cov_filter(k, e->sparse_k, e->sp_dev, ctb, e->stream));
      if ((rc = start_sampler_tables(e, k, stb, false, ctb.ftab, ctb.frow))) return rc;
      if (k.flags & DLM_OPT_TEST_FAIL_AFTER_TABLES) return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_TEST_FAIL_AFTER_TABLES");
      HIP_TRY(e, dlm::launch_sparse16_mean_filter(k, e->sparse_k, e->sp_dev, ctb, e->stream));
      stb.mc4 = ctb.mc;
      KArgs kg = k;
      kg.route_take = 1; kg.filt = e->fws;
      HIP_TRY(e, dlm::launch_sparse16_filter(kg, e->sparse_k, e->sp_dev, nullptr, nullptr, e->stream));
      k.filt_in = e->fws;
    } else {
      const bool defer_z = norec && shared_big;   // (the normals start behind the forward pass: 8.64 -> 8.26 ms per C4 Gibbs iteration on one box)
      if ((shared_factors || shared_big) && (rc = start_sampler_tables(e, k, stb, shared_big, nullptr, 0, defer_z))) return rc;
      if ((shared_factors || shared_big) && (k.flags & DLM_OPT_TEST_FAIL_AFTER_TABLES))   // test hook: an error exit with the auxiliary streams busy
        return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_TEST_FAIL_AFTER_TABLES");
      if (norec && shared_big) {
        // 16 <= d <= 48, no records wanted: the draw kernel reads only the means of the series without a gap, so those series'
        // steady steps store the mean alone (KArgs::keep_cov; the first, full steps still write what the convergence test re-reads)
        HIP_TRY(e, dlm::launch_wave48_mark_gaps(k, e->route, e->stream));
        k.keep_cov = e->route;
        k.ktab = stb.ktab;     // ... and they leave the filter where the recursion settles: k_steady_filter_w48 carries their means on
        k.leave_step = leave_of(e);
        HIP_TRY(e, hipMemsetD32Async((hipDeviceptr_t)k.leave_step, k.T, (size_t)k.N, e->stream));   // T: the series ran the whole filter
        stb.marked = 1;
      }
      if ((rc = run_filter(e, k, false))) return rc;
      if (norec && shared_big) {
        if (defer_z) {
          if (!e->rng_gate) HIP_TRY(e, hipEventCreateWithFlags(&e->rng_gate, hipEventDisableTiming));
          HIP_TRY(e, hipEventRecord(e->rng_gate, e->stream));
          if ((rc = start_sampler_normals(e, k, stb, true, e->rng_gate))) return rc;
        }
        HIP_TRY(e, hipStreamWaitEvent(e->stream, e->cov_ev2, 0));   // the zero series' filter: the steady gain and the step it settled at
        HIP_TRY(e, dlm::launch_wave48_steady_filter(k, stb.ktab, stb.settle, e->stream));
      }
      k.keep_cov = nullptr; k.ktab = nullptr; k.leave_step = nullptr;
      k.filt_in = k.filt;
    }
  }
  if (!forward && ((rc = analyse_g(e, k, model->G, opts->mem == DLM_MEM_HOST)) || (rc = mark(e, 0)))) return rc;   // the structure tables of G
  if ((rc = mark(e, 1))) return rc;
  auto done = [&]() { const int r = mark(e, 2); return r ? r : st.finish(opts->flags & DLM_OPT_ASYNC); };
  if (eig) {
    if (dlm::generic_sampler_lds_bytes(k.d, k.p) > 160 * 1024)
      return fail(e, DLM_ERR_UNSUPPORTED, "DLM_OPT_DRAW_EIG runs on the general backward sampler kernel: d <= 53");
    e->variant = "generic-eig";
    HIP_TRY(e, dlm::launch_generic_sampler(k, e->stream));
    return done();
  }
  if (use_lane(k)) {
    e->variant = "lane-sampler";
    HIP_TRY(e, dlm::launch_lane_sampler(k, e->stream));
    return done();
  }
  if (shared_factors) {
    // V, W, C0 shared by the batch on a regular grid: J_t, H_t and the factors once per call (the table of the second stream), the
    // series draw against it; a series with a missing observation computes its own as always
    e->variant = "sparse16-sampler-shared";
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->cov_ev[1], 0));
    e->cov_busy = false;
    if (stb.z4) { HIP_TRY(e, hipStreamWaitEvent(e->stream, e->rng_ev, 0)); e->rng_busy = false; }
    k.route = e->route;
    HIP_TRY(e, dlm::launch_sampler_shared_draw(k, e->sparse_k, e->sp_dev, stb, e->stream));
    return done();
  }
  if (fast_shape_ok(k) && e->sparse_k > 0 && !(k.flags & DLM_OPT_NO_SAMPLER16)) {
    e->variant = "sparse16-sampler";
    HIP_TRY(e, dlm::launch_sparse16_sampler(k, e->sparse_k, e->sp_dev, e->stream));
    return done();
  }
  if (!(k.flags & (DLM_OPT_FORCE_GENERIC | DLM_OPT_NO_SAMPLER16)) && dlm::wave48_small_ok(k)) {
    e->variant = "sparse16-sampler";
    HIP_TRY(e, dlm::launch_small_mv_sampler(k, e->stream));
    return done();
  }
  if (shared_big) {
    e->variant = "wave-sampler-shared";
    HIP_TRY(e, hipStreamWaitEvent(e->stream, e->cov_ev[1], 0));
    e->cov_busy = false;
    if (stb.z4) { HIP_TRY(e, hipStreamWaitEvent(e->stream, e->rng_ev, 0)); e->rng_busy = false; }
    k.route = e->route;
    HIP_TRY(e, dlm::launch_wave48_sampler_shared_draw(k, stb, e->stream));
    return done();
  }
  if (!(k.flags & (DLM_OPT_FORCE_GENERIC | DLM_OPT_NO_SAMPLER16)) && dlm::wave48_sampler_supported(k)) {
    e->variant = "wave-sampler";
    HIP_TRY(e, dlm::launch_wave48_sampler(k, e->stream));
    return done();
  }
  e->variant = "generic";
  if (dlm::generic_sampler_lds_bytes(k.d, k.p) > 160 * 1024)
    return fail(e, DLM_ERR_UNSUPPORTED, "the general backward sampler kernel keeps seven d x d matrices in the LDS of one CU: d <= 53 (structured models with d <= 48 take the per-wave kernels)");
  HIP_TRY(e, dlm::launch_generic_sampler(k, e->stream));
  return done();
}

int dlm_ffbs_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                   const double* y, const double* z, const dlm_options* opts, double* filt_ws,
                   double* theta, double* cond, double* stats, int32_t* status) {
  return sampler_common(e, model, params, y, z, opts, nullptr, filt_ws, theta, cond, stats, status, true);
}

int dlm_backward_sample_batch(dlm_engine* e, const dlm_model_desc* model,
                              const dlm_params_desc* params, const double* y, const double* filt,
                              const double* z, const dlm_options* opts, double* theta,
                              double* cond, double* stats, int32_t* status) {
  return sampler_common(e, model, params, y, z, opts, filt, nullptr, theta, cond, stats, status, false);
}

int dlm_svd_filter_batch(dlm_engine* e, const dlm_model_desc* model,
                         const dlm_params_desc* params, const double* y,
                         const dlm_options* opts, double* svd_rec, int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!y || !svd_rec) return fail(e, DLM_ERR_ARG, "y and svd_rec are required");
  if (model->d > 48 || model->p > 32) return fail(e, DLM_ERR_UNSUPPORTED, "the SVD (square-root) filter takes d <= 48, p <= 32 in this build");
  const size_t d = model->d, p = model->p, T = model->T, N = model->N, srec = 2 * d + d * d;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  double* rec_dev = nullptr;
  st.in(&k.y, y, N * T * p);
  st.out(&rec_dev, svd_rec, N * (T + 1) * srec);
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  e->variant = "svd-jacobi";
  if ((rc = want_counters(e, k))) return rc;
  if ((rc = mark(e, 0))) return rc;
  if (dlm::svd_shared_eligible(k)) {
    // parameters shared by the batch: the decompositions once per call (one wave), a mean-only kernel per series; a series with a
    // missing observation runs k_svd_filter as always
    if ((rc = ensure_route(e, (size_t)k.N))) return rc;
    const size_t need = sizeof(double) * dlm::svd_shared_ws_doubles(k);
    if (need > e->covws_bytes) {
      if (e->covws) { { const int rcd = drain_all(e); if (rcd) return rcd; } HIP_TRY(e, hipFree(e->covws)); e->covws = nullptr; e->covws_bytes = 0; }
      HIP_TRY(e, hipMalloc((void**)&e->covws, need));
      e->covws_bytes = need;
    }
    HIP_TRY(e, dlm::launch_svd_filter_shared(k, rec_dev, e->covws, e->route, e->stream));
  } else HIP_TRY(e, dlm::launch_svd_filter(k, rec_dev, e->stream));
  if ((rc = mark(e, 1)) || (rc = mark(e, 2))) return rc;
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_svd_ffbs_batch(dlm_engine* e, const dlm_model_desc* model, const dlm_params_desc* params,
                       const double* y, const double* z, const dlm_options* opts,
                       double* svd_ws, double* theta, double* stats, int32_t* status) {
  int rc = check_common(e, model, params, opts);
  if (rc) return rc;
  if (!y || !svd_ws) return fail(e, DLM_ERR_ARG, "y and svd_ws are required");
  if (model->d > 48 || model->p > 32) return fail(e, DLM_ERR_UNSUPPORTED, "the SVD (square-root) sampler takes d <= 48, p <= 32 in this build");
  const size_t d = model->d, p = model->p, T = model->T, N = model->N, srec = 2 * d + d * d;
  KArgs k{};
  Stager st(e, opts->mem == DLM_MEM_HOST);
  stage_model(st, k, model, params, opts);
  double* rec_dev = nullptr;
  st.in(&k.y, y, N * T * p);
  st.in(&k.z, z, z ? N * (T + 1) * d : 0);
  st.out(&rec_dev, svd_ws, N * (T + 1) * srec);
  st.out(&k.theta, theta, N * (T + 1) * d);
  st.out(&k.stats, stats, N * (size_t)dlm_stats_len(model->d, model->p, opts->flags));
  st.zeroed_out(&k.status, (int*)status, N);
  if ((rc = st.commit())) return rc;
  e->variant = "svd-jacobi";
  if ((rc = want_counters(e, k))) return rc;
  if ((rc = mark(e, 0))) return rc;
  HIP_TRY(e, dlm::launch_svd_filter(k, rec_dev, e->stream));
  if ((rc = mark(e, 1))) return rc;
  HIP_TRY(e, dlm::launch_svd_sampler(k, rec_dev, e->stream));
  if ((rc = mark(e, 2))) return rc;
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_stats_pool(dlm_engine* e, const double* stats, int32_t N, int32_t L, double* pooled,
                   const dlm_options* opts) {
  if (!e) return DLM_ERR_ARG;
  if (!stats || !pooled || !opts || N < 1 || L < 1) return fail(e, DLM_ERR_ARG, "dlm_stats_pool arguments");
  HIP_TRY(e, hipSetDevice(e->device));
  const double* s_dev = nullptr; double* p_dev = nullptr;
  Stager st(e, opts->mem == DLM_MEM_HOST);
  st.in(&s_dev, stats, (size_t)N * L);
  st.out(&p_dev, pooled, (size_t)L);
  int rc = st.commit();
  if (rc) return rc;
  HIP_TRY(e, dlm::launch_stats_pool(s_dev, N, L, p_dev, e->stream));
  return st.finish(opts->flags & DLM_OPT_ASYNC);
}

int dlm_comm_unique_id(uint8_t id[DLM_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= DLM_COMM_ID_BYTES, "id buffer too small");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return DLM_ERR_RCCL;
  memset(id, 0, DLM_COMM_ID_BYTES);
  memcpy(id, &u, sizeof(u));
  return DLM_OK;
}

int dlm_comm_init_rank(dlm_engine* e, int32_t nranks, int32_t rank, const uint8_t id[DLM_COMM_ID_BYTES]) {
  if (!e || !id || nranks < 1 || rank < 0 || rank >= nranks) return DLM_ERR_ARG;
  HIP_TRY(e, hipSetDevice(e->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclResult_t r = ncclCommInitRank(&e->comm, nranks, u, rank);
  if (r != ncclSuccess) return fail(e, DLM_ERR_RCCL, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  e->has_comm = true;
  return DLM_OK;
}

int dlm_gibbs_suffstats_allreduce(dlm_engine* e, double* stats_dev, int64_t count) {
  if (!e || !stats_dev || count < 1) return DLM_ERR_ARG;
  if (!e->has_comm) return fail(e, DLM_ERR_RCCL, "no communicator: call dlm_comm_init_rank first");
  HIP_TRY(e, hipSetDevice(e->device));
  ncclResult_t r = ncclAllReduce(stats_dev, stats_dev, (size_t)count, ncclDouble, ncclSum, e->comm, e->stream);
  if (r != ncclSuccess) return fail(e, DLM_ERR_RCCL, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
  HIP_TRY(e, hipStreamSynchronize(e->stream));
  return DLM_OK;
}

}  // extern "C"
