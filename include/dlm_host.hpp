// dlm_host.hpp -- header-only C++17 host layer above the C ABI (dlm_engine.h).
//
// The reference's host language is Scala; no JVM/scalac exists in this image, and the reference is
// compiled code, so the native host layer above the C ABI is C++.  It mirrors the reference's
// operator interface for the hot path -- same names, argument order and per-series semantics --
// widened to a batch (a vector of series on one time grid):
//
//   Dlm{f, g}, Dlm::polynomial, Dlm::seasonal, compose (|+|), outer (|*|)     Dlm.scala:14-31,107-243
//   DlmParameters{v, w, m0, c0}, Data{time, observation}                       Dlm.scala:36-39,94
//   KalmanFilter::filterDlm / filter                                          KalmanFilter.scala:262-294
//   KalmanFilter::logLikelihood (sum of conditionalLikelihood)                KalmanFilter.scala:138-153
//   KalmanFilter::likelihood (the literal one MetropolisHastings.dlm calls)    KalmanFilter.scala:175-183, :299-306
//   Smoothing::backwardsSmoother / filterSmooth / ffbsDlm                      Smoothing.scala:57-64,173-180
//   SvdFilter::filterDlm, SvdSampler::ffbsDlm                                  SvdFilter.scala:158-161, SvdSampler.scala:79-82
//   GibbsSampling::sample / sampleSvd, GibbsWishart::sample                    Gibbs.scala:165-180,203-217, GibbsWishart.scala:65-80
//
// Two ways to run every call:
//   * host mode (DLM_MEM_HOST): the functions taking std::vector<std::vector<Data>> -- the engine stages the
//     buffers; results come back as vectors of state objects, like the reference's.  Fine up to a few GB.
//   * device-resident (DLM_MEM_DEVICE): DeviceSeries / DeviceRecords on engine-owned buffers
//     (dlm_buffer_alloc/upload/download).  Observations are uploaded once, results stay in HBM and single
//     records or series are fetched on demand -- the measured path (10^4 x 10^3 x d = 13 produces 29 GB of
//     records; neither a JVM nor this layer wants them as 10^7 objects).
//
// Matrices are column-major (Breeze DenseMatrix.data).  Errors throw std::runtime_error with dlm_last_error().
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "dlm_engine.h"

namespace dlm_host {

struct Matrix {
  int rows = 0, cols = 0;
  std::vector<double> data;  // column-major
  Matrix() = default;
  Matrix(int r, int c, double fill = 0.0) : rows(r), cols(c), data((size_t)r * c, fill) {}
  double& operator()(int i, int j) { return data[(size_t)i + (size_t)j * rows]; }
  double operator()(int i, int j) const { return data[(size_t)i + (size_t)j * rows]; }
  static Matrix eye(int n) { Matrix m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1.0; return m; }
  static Matrix diag(const std::vector<double>& d) { Matrix m((int)d.size(), (int)d.size()); for (size_t i = 0; i < d.size(); ++i) m((int)i, (int)i) = d[i]; return m; }
  bool operator==(const Matrix& o) const { return rows == o.rows && cols == o.cols && data == o.data; }
};

inline Matrix blockDiagonal(const Matrix& a, const Matrix& b) {  // Dlm.scala:197-208
  Matrix m(a.rows + b.rows, a.cols + b.cols);
  for (int j = 0; j < a.cols; ++j) for (int i = 0; i < a.rows; ++i) m(i, j) = a(i, j);
  for (int j = 0; j < b.cols; ++j) for (int i = 0; i < b.rows; ++i) m(a.rows + i, a.cols + j) = b(i, j);
  return m;
}
inline Matrix vertcat(const Matrix& a, const Matrix& b) {
  Matrix m(a.rows + b.rows, a.cols);
  for (int j = 0; j < a.cols; ++j) { for (int i = 0; i < a.rows; ++i) m(i, j) = a(i, j); for (int i = 0; i < b.rows; ++i) m(a.rows + i, j) = b(i, j); }
  return m;
}

struct Dlm {
  std::function<Matrix(double)> f;  // time -> d x p (used as F^T)
  std::function<Matrix(double)> g;  // dt   -> d x d
  Dlm compose(const Dlm& y) const {  // |+|, Dlm.scala:107-111
    Dlm x = *this;
    return Dlm{[x, y](double t) { return vertcat(x.f(t), y.f(t)); }, [x, y](double dt) { return blockDiagonal(x.g(dt), y.g(dt)); }};
  }
  Dlm outer(const Dlm& y) const {    // |*|, Dlm.scala:117-122
    Dlm x = *this;
    return Dlm{[x, y](double t) { return blockDiagonal(x.f(t), y.f(t)); }, [x, y](double dt) { return blockDiagonal(x.g(dt), y.g(dt)); }};
  }
  static Dlm polynomial(int order) {  // Dlm.scala:139-153
    return Dlm{[order](double) { Matrix m(order, 1); m(0, 0) = 1.0; return m; },
               [order](double) { Matrix m = Matrix::eye(order); for (int i = 0; i + 1 < order; ++i) m(i, i + 1) = 1.0; return m; }};
  }
  static Dlm seasonal(int period, int harmonics) {  // Dlm.scala:213-243
    return Dlm{[harmonics](double) { Matrix m(2 * harmonics, 1); for (int h = 0; h < 2 * harmonics; h += 2) m(h, 0) = 1.0; return m; },
               [period, harmonics](double dt) {
                 Matrix m(2 * harmonics, 2 * harmonics);
                 const double base = 2.0 * M_PI * std::fmod(dt, (double)period) / period;
                 for (int h = 1; h <= harmonics; ++h) {
                   const double th = h * base; const int o = 2 * (h - 1);
                   m(o, o) = std::cos(th); m(o, o + 1) = -std::sin(th); m(o + 1, o) = std::sin(th); m(o + 1, o + 1) = std::cos(th);
                 }
                 return m;
               }};
  }
};

struct DlmParameters { Matrix v, w; std::vector<double> m0; Matrix c0; };
struct Data { double time; std::vector<std::optional<double>> observation; };

struct KfState { double time; std::vector<double> mt; Matrix ct; };                 // KalmanFilter.scala:22-30 (m, C)
struct SmoothingState { double time; std::vector<double> mean; Matrix covariance; };  // Smoothing.scala:18-22
struct SamplingState { double time; std::vector<double> sample; };                  // Smoothing.scala:10-15
struct SvdState { double time; std::vector<double> mt, dc; Matrix uc; };            // SvdFilter.scala:7-14: C = uc diag(dc^2) uc^T

// Reference quirk switches: 0 = the mathematically correct forms, or any of DLM_OPT_SMOOTHER_COMPAT_Q1,
// DLM_OPT_SVD_RAW_W_Q2, DLM_OPT_SVD_SAMPLER_Q9 for the literal reference arithmetic; DLM_OPT_FFBS_SIMSMOOTH etc. pass through.
using Flags = uint32_t;

class Engine {
 public:
  explicit Engine(int device = 0) {
    if (dlm_engine_create(device, &e_) != DLM_OK) throw std::runtime_error("dlm_engine_create failed: no usable HIP device");
  }
  ~Engine() { dlm_engine_destroy(e_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;
  dlm_engine* get() const { return e_; }
  void check(int rc) const { if (rc != DLM_OK) throw std::runtime_error(std::string("dlm engine: ") + dlm_last_error(e_)); }
  std::string lastVariant() const { return dlm_last_variant(e_); }
 private:
  dlm_engine* e_ = nullptr;
};

// An engine-owned device allocation (dlm_buffer_alloc); freed with the object.  LIFETIME: the Engine must outlive its DeviceBuffers
// (and everything built on them: DeviceSeries, DeviceParameters, DeviceRecords) -- dlm_engine_destroy releases every buffer still
// allocated, after which this destructor would hand a dangling handle to dlm_buffer_free.  Declare the Engine first (destroyed last).
class DeviceBuffer {
 public:
  DeviceBuffer(Engine& e, size_t bytes) : e_(&e), bytes_(bytes) { e.check(dlm_buffer_alloc(e.get(), bytes, &p_)); }
  ~DeviceBuffer() { if (p_) dlm_buffer_free(e_->get(), p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void* ptr() const { return p_; }
  double* doubles() const { return static_cast<double*>(p_); }
  size_t bytes() const { return bytes_; }
  void upload(const void* src, size_t nbytes, size_t offset = 0) {
    if (offset + nbytes > bytes_) throw std::out_of_range("DeviceBuffer::upload");
    e_->check(dlm_buffer_upload(e_->get(), p_, offset, src, nbytes));
  }
  void download(void* dst, size_t nbytes, size_t offset = 0) const {
    if (offset + nbytes > bytes_) throw std::out_of_range("DeviceBuffer::download");
    e_->check(dlm_buffer_download(e_->get(), p_, offset, dst, nbytes));
  }
 private:
  Engine* e_; void* p_ = nullptr; size_t bytes_;
};

namespace detail {
// closures -> flat tables on the shared time grid (the job of the Scala shim in INTEGRATION.md)
struct Tables {
  int d = 0, p = 0, T = 0, N = 0, nG = 0;
  std::vector<double> F, G, dt, y, times;
  std::vector<int32_t> gIndex;
  int64_t fStride = 0;
  bool unitDt = true;
  double timeOf(int record) const { return record == 0 ? times[0] - dt[0] : times[record - 1]; }   // record 0 sits at t0 - 1
};
inline Tables materialiseGrid(const Dlm& mod, const std::vector<double>& times) {
  if (times.empty()) throw std::invalid_argument("empty observation vector (KalmanFilter.scala:116-117 throws on t0.get)");
  Tables t;
  t.T = (int)times.size(); t.times = times;
  double prev = *std::min_element(times.begin(), times.end()) - 1.0;  // KalmanFilter.initialiseState
  std::vector<Matrix> gs, fs;
  for (int k = 0; k < t.T; ++k) {
    const double dt = times[k] - prev; prev = times[k];
    t.dt.push_back(dt);
    t.unitDt = t.unitDt && dt == 1.0;
    Matrix gm = mod.g(dt);
    int idx = -1;
    for (size_t q = 0; q < gs.size(); ++q) if (gs[q] == gm) idx = (int)q;
    if (idx < 0) { idx = (int)gs.size(); gs.push_back(gm); }
    t.gIndex.push_back(idx);
    fs.push_back(mod.f(times[k]));
  }
  t.d = fs[0].rows; t.p = fs[0].cols; t.nG = (int)gs.size();
  bool constF = true;
  for (const auto& f : fs) constF = constF && (f == fs[0]);
  if (constF) t.F = fs[0].data; else { t.fStride = (int64_t)t.d * t.p; for (const auto& f : fs) t.F.insert(t.F.end(), f.data.begin(), f.data.end()); }
  for (const auto& g : gs) t.G.insert(t.G.end(), g.data.begin(), g.data.end());
  return t;
}
inline Tables materialise(const Dlm& mod, const std::vector<std::vector<Data>>& ys) {
  if (ys.empty() || ys[0].empty()) throw std::invalid_argument("empty observation vector (KalmanFilter.scala:116-117 throws on t0.get)");
  std::vector<double> times;
  for (const auto& d : ys[0]) times.push_back(d.time);
  Tables t = materialiseGrid(mod, times);
  t.N = (int)ys.size();
  for (const auto& s : ys) {
    if ((int)s.size() != t.T) throw std::invalid_argument("all series of a batch must share one time grid");
    for (const auto& d : s) {
      if ((int)d.observation.size() != t.p) throw std::invalid_argument("observation length differs from the model's p");
      for (const auto& o : d.observation) t.y.push_back(o ? *o : std::numeric_limits<double>::quiet_NaN());
    }
  }
  return t;
}
inline dlm_model_desc modelDesc(const Tables& t) {
  dlm_model_desc m{};
  m.d = t.d; m.p = t.p; m.T = t.T; m.N = t.N; m.F = t.F.data(); m.f_stride = t.fStride;
  m.G = t.G.data(); m.n_g = t.nG; m.g_index = t.nG > 1 ? t.gIndex.data() : nullptr;
  m.dt = t.unitDt ? nullptr : t.dt.data();
  return m;
}
inline dlm_params_desc paramsDesc(const DlmParameters& p) {
  return dlm_params_desc{p.v.data.data(), 0, p.w.data.data(), 0, p.m0.data(), 0, p.c0.data.data(), 0, 0, 0};
}
inline Matrix matAt(const double* rec, int d) { Matrix m(d, d); m.data.assign(rec, rec + (size_t)d * d); return m; }
inline Matrix unpackLower(const double* tri, int d) {   // packed lower triangle by rows -> dense symmetric
  Matrix m(d, d);
  for (int i = 0; i < d; ++i) for (int j = 0; j <= i; ++j) { const double v = tri[i * (i + 1) / 2 + j]; m(i, j) = v; m(j, i) = v; }
  return m;
}
inline dlm_options opts(Flags flags, int mem, uint64_t seed = 0, uint64_t offset = 0) { return dlm_options{flags, mem, seed, offset}; }
}  // namespace detail

// ------------------------------------------------------------------------------------------------------------
// host mode
// ------------------------------------------------------------------------------------------------------------
namespace KalmanFilter {
// KalmanFilter(advanceState(p, mod.g)).filter(mod, ys, p): T+1 states per series, the first being the
// initial state at t0 - 1 (Filter.scala:41-45)
inline std::vector<std::vector<KfState>> filter(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(0, DLM_MEM_HOST);
  const int rec = t.d + t.d * t.d;
  std::vector<double> filt((size_t)t.N * (t.T + 1) * rec);
  e.check(dlm_filter_batch(e.get(), &m, &q, t.y.data(), &o, filt.data(), nullptr, nullptr, nullptr));
  std::vector<std::vector<KfState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const double* r = filt.data() + ((size_t)n * (t.T + 1) + k) * rec;
      out[n].push_back(KfState{t.timeOf(k), std::vector<double>(r, r + t.d), detail::matAt(r + t.d, t.d)});
    }
  return out;
}
// sum over the series of KalmanFilter.conditionalLikelihood (KalmanFilter.scala:138-153): log p(y_1:T | V, W) per series
inline std::vector<double> logLikelihood(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(0, DLM_MEM_HOST);
  std::vector<double> ll((size_t)t.N);
  e.check(dlm_loglik_batch(e.get(), &m, &q, t.y.data(), &o, ll.data(), nullptr));
  return ll;
}
// KalmanFilter.likelihood(mod, ys)(p) as the reference writes it (KalmanFilter.scala:299-306; what MetropolisHastings.dlm
// evaluates, MetropolisHastings.scala:134, :205): the transition density of the filtered means,
// sum_t log N(m_t; g(dt_t) m_{t-1}, W dt_t) (KalmanFilter.logLikelihood, :175-183) -- NOT the prediction-error likelihood above
inline std::vector<double> likelihood(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(DLM_OPT_LOGLIK_LITERAL_Q7, DLM_MEM_HOST);
  std::vector<double> ll((size_t)t.N);
  e.check(dlm_loglik_batch(e.get(), &m, &q, t.y.data(), &o, ll.data(), nullptr));
  return ll;
}
// KalmanFilter.filterDlm: `filterTraverse` drops the initial state (Filter.scala:32-36)
inline std::vector<std::vector<KfState>> filterDlm(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p) {
  auto out = filter(e, mod, ys, p);
  for (auto& s : out) s.erase(s.begin());
  return out;
}
}  // namespace KalmanFilter

namespace Smoothing {
// Smoothing.backwardsSmoother(mod)(kfStates) for N series: takes the T+1 states of `.filter` (initial state included,
// core/src/test/scala/Smoothing.scala:31-40).  The reference's smoothStep reads only (time, m_t, C_t) plus (a, R) that the
// engine recomputes; `w` is the system noise the filter ran with.  flags: DLM_OPT_SMOOTHER_COMPAT_Q1 for the literal J X J.
inline std::vector<std::vector<SmoothingState>> backwardsSmoother(Engine& e, const Dlm& mod, const std::vector<std::vector<KfState>>& kfStates,
                                                                  const DlmParameters& p, Flags flags = 0) {
  if (kfStates.empty() || kfStates[0].size() < 2) throw std::invalid_argument("backwardsSmoother needs the T+1 states of .filter");
  std::vector<double> times;
  for (size_t k = 1; k < kfStates[0].size(); ++k) times.push_back(kfStates[0][k].time);
  detail::Tables t = detail::materialiseGrid(mod, times);
  t.N = (int)kfStates.size();
  const int rec = t.d + t.d * t.d;
  std::vector<double> filt((size_t)t.N * (t.T + 1) * rec), sm(filt.size());
  for (int n = 0; n < t.N; ++n) {
    if ((int)kfStates[n].size() != t.T + 1) throw std::invalid_argument("all series of a batch must share one time grid");
    for (int k = 0; k <= t.T; ++k) {
      double* r = filt.data() + ((size_t)n * (t.T + 1) + k) * rec;
      std::copy(kfStates[n][k].mt.begin(), kfStates[n][k].mt.end(), r);
      std::copy(kfStates[n][k].ct.data.begin(), kfStates[n][k].ct.data.end(), r + t.d);
    }
  }
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(flags, DLM_MEM_HOST);
  e.check(dlm_smooth_batch(e.get(), &m, &q, filt.data(), &o, sm.data(), nullptr));
  std::vector<std::vector<SmoothingState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const double* r = sm.data() + ((size_t)n * (t.T + 1) + k) * rec;
      out[n].push_back(SmoothingState{kfStates[n][k].time, std::vector<double>(r, r + t.d), detail::matAt(r + t.d, t.d)});
    }
  return out;
}
// KalmanFilter(...).filter followed by Smoothing.backwardsSmoother, fused on the device
inline std::vector<std::vector<SmoothingState>> filterSmooth(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p,
                                                             std::vector<std::vector<KfState>>* filtered = nullptr, Flags flags = 0) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(flags, DLM_MEM_HOST);
  const int rec = t.d + t.d * t.d;
  std::vector<double> filt((size_t)t.N * (t.T + 1) * rec), sm(filt.size());
  e.check(dlm_filter_smooth_batch(e.get(), &m, &q, t.y.data(), &o, filt.data(), sm.data(), nullptr));
  std::vector<std::vector<SmoothingState>> out(t.N);
  if (filtered) filtered->assign(t.N, {});
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const size_t off = ((size_t)n * (t.T + 1) + k) * rec;
      const double time = t.timeOf(k);
      out[n].push_back(SmoothingState{time, std::vector<double>(sm.data() + off, sm.data() + off + t.d), detail::matAt(sm.data() + off + t.d, t.d)});
      if (filtered) (*filtered)[n].push_back(KfState{time, std::vector<double>(filt.data() + off, filt.data() + off + t.d), detail::matAt(filt.data() + off + t.d, t.d)});
    }
  return out;
}
// Smoothing.ffbsDlm: one draw of the T+1 states per series, a pure function of (seed, series index)
inline std::vector<std::vector<SamplingState>> ffbsDlm(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p,
                                                       uint64_t seed, Flags flags = 0) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(flags, DLM_MEM_HOST, seed);
  const int rec = t.d + t.d * t.d;
  std::vector<double> ws((size_t)t.N * (t.T + 1) * rec), th((size_t)t.N * (t.T + 1) * t.d);
  e.check(dlm_ffbs_batch(e.get(), &m, &q, t.y.data(), nullptr, &o, ws.data(), th.data(), nullptr, nullptr, nullptr));
  std::vector<std::vector<SamplingState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const double* r = th.data() + ((size_t)n * (t.T + 1) + k) * t.d;
      out[n].push_back(SamplingState{t.timeOf(k), std::vector<double>(r, r + t.d)});
    }
  return out;
}
}  // namespace Smoothing

namespace SvdFilter {
// SvdFilter.filterDlm (SvdFilter.scala:158-161): T states (m_t, dc_t, uc_t) per series, C_t = uc diag(dc^2) uc^T.
// flags = DLM_OPT_SVD_RAW_W_Q2 reproduces the reference's raw-W time update.
inline std::vector<std::vector<SvdState>> filterDlm(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p, Flags flags = 0) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(flags, DLM_MEM_HOST);
  const int srec = 2 * t.d + t.d * t.d;
  std::vector<double> rec((size_t)t.N * (t.T + 1) * srec);
  e.check(dlm_svd_filter_batch(e.get(), &m, &q, t.y.data(), &o, rec.data(), nullptr));
  std::vector<std::vector<SvdState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 1; k <= t.T; ++k) {   // filterTraverse drops the initial state
      const double* r = rec.data() + ((size_t)n * (t.T + 1) + k) * srec;
      out[n].push_back(SvdState{t.timeOf(k), std::vector<double>(r, r + t.d), std::vector<double>(r + t.d, r + 2 * t.d), detail::matAt(r + 2 * t.d, t.d)});
    }
  return out;
}
}  // namespace SvdFilter

namespace SvdSampler {
// SvdSampler.ffbsDlm (SvdSampler.scala:79-82).  flags: DLM_OPT_SVD_RAW_W_Q2 | DLM_OPT_SVD_SAMPLER_Q9 for the literal reference.
inline std::vector<std::vector<SamplingState>> ffbsDlm(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p,
                                                       uint64_t seed, Flags flags = 0) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o = detail::opts(flags, DLM_MEM_HOST, seed);
  const int srec = 2 * t.d + t.d * t.d;
  std::vector<double> ws((size_t)t.N * (t.T + 1) * srec), th((size_t)t.N * (t.T + 1) * t.d);
  e.check(dlm_svd_ffbs_batch(e.get(), &m, &q, t.y.data(), nullptr, &o, ws.data(), th.data(), nullptr, nullptr));
  std::vector<std::vector<SamplingState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const double* r = th.data() + ((size_t)n * (t.T + 1) + k) * t.d;
      out[n].push_back(SamplingState{t.timeOf(k), std::vector<double>(r, r + t.d)});
    }
  return out;
}
}  // namespace SvdSampler

// ------------------------------------------------------------------------------------------------------------
// device-resident mode
// ------------------------------------------------------------------------------------------------------------
// The model tables and the observations of N series, uploaded once.
class DeviceSeries {
 public:
  // from the reference's data type
  DeviceSeries(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys) : e_(&e), t_(detail::materialise(mod, ys)) { upload(t_.y.data()); t_.y.clear(); t_.y.shrink_to_fit(); }
  // from a flat [N][T][p] array (NaN = missing) on the time grid `times` -- for callers that never had Data objects
  DeviceSeries(Engine& e, const Dlm& mod, const std::vector<double>& times, const double* y, int N) : e_(&e), t_(detail::materialiseGrid(mod, times)) { t_.N = N; upload(y); }
  Engine& engine() const { return *e_; }
  const detail::Tables& tables() const { return t_; }
  int d() const { return t_.d; } int p() const { return t_.p; } int T() const { return t_.T; } int N() const { return t_.N; }
  double timeOf(int record) const { return t_.timeOf(record); }
  const double* y() const { return y_->doubles(); }
  dlm_model_desc modelDesc() const {
    dlm_model_desc m{};
    m.d = t_.d; m.p = t_.p; m.T = t_.T; m.N = t_.N; m.f_stride = t_.fStride; m.n_g = t_.nG;
    char* base = static_cast<char*>(tab_->ptr());
    m.F = reinterpret_cast<const double*>(base + offF_); m.G = reinterpret_cast<const double*>(base + offG_);
    m.g_index = t_.nG > 1 ? reinterpret_cast<const int32_t*>(base + offGi_) : nullptr;
    m.dt = t_.unitDt ? nullptr : reinterpret_cast<const double*>(base + offDt_);
    return m;
  }
 private:
  static size_t pad(size_t b) { return (b + 255) & ~(size_t)255; }
  void upload(const double* y) {
    y_ = std::make_unique<DeviceBuffer>(*e_, (size_t)t_.N * t_.T * t_.p * sizeof(double));
    y_->upload(y, y_->bytes());
    offF_ = 0; offG_ = offF_ + pad(t_.F.size() * 8); offGi_ = offG_ + pad(t_.G.size() * 8); offDt_ = offGi_ + pad(t_.gIndex.size() * 4);
    tab_ = std::make_unique<DeviceBuffer>(*e_, offDt_ + pad(t_.dt.size() * 8));
    tab_->upload(t_.F.data(), t_.F.size() * 8, offF_); tab_->upload(t_.G.data(), t_.G.size() * 8, offG_);
    tab_->upload(t_.gIndex.data(), t_.gIndex.size() * 4, offGi_); tab_->upload(t_.dt.data(), t_.dt.size() * 8, offDt_);
  }
  Engine* e_; detail::Tables t_;
  std::unique_ptr<DeviceBuffer> y_, tab_;
  size_t offF_ = 0, offG_ = 0, offGi_ = 0, offDt_ = 0;
};

// DlmParameters on the device: one shared set, or one per series (the reference's semantics for N independent series).
class DeviceParameters {
 public:
  DeviceParameters(Engine& e, const DlmParameters& p) : DeviceParameters(e, std::vector<DlmParameters>{p}, true) {}
  DeviceParameters(Engine& e, const std::vector<DlmParameters>& ps, bool shared = false) : d_((int)ps.at(0).m0.size()), p_(ps[0].v.rows), n_(ps.size()), shared_(shared) {
    const size_t pp = (size_t)p_ * p_, dd = (size_t)d_ * d_, per = pp + dd + d_ + dd;
    std::vector<double> flat; flat.reserve(per * n_);
    for (const auto& q : ps) flat.insert(flat.end(), q.v.data.begin(), q.v.data.end());
    for (const auto& q : ps) flat.insert(flat.end(), q.w.data.begin(), q.w.data.end());
    for (const auto& q : ps) flat.insert(flat.end(), q.m0.begin(), q.m0.end());
    for (const auto& q : ps) flat.insert(flat.end(), q.c0.data.begin(), q.c0.data.end());
    buf_ = std::make_unique<DeviceBuffer>(e, flat.size() * 8);
    buf_->upload(flat.data(), flat.size() * 8);
  }
  double* V() const { return buf_->doubles(); }
  double* W() const { return V() + n_ * (size_t)p_ * p_; }
  double* m0() const { return W() + n_ * (size_t)d_ * d_; }
  double* C0() const { return m0() + n_ * (size_t)d_; }
  void setVW(const std::vector<double>& v, const std::vector<double>& w) {   // [n][p*p], [n][d*d]
    buf_->upload(v.data(), v.size() * 8, 0); buf_->upload(w.data(), w.size() * 8, n_ * (size_t)p_ * p_ * 8);
  }
  dlm_params_desc desc() const {
    const bool s = shared_;
    return dlm_params_desc{V(), s ? 0 : (int64_t)p_ * p_, W(), s ? 0 : (int64_t)d_ * d_, m0(), s ? 0 : (int64_t)d_, C0(), s ? 0 : (int64_t)d_ * d_, 0, 0};
  }
 private:
  int d_, p_; size_t n_; bool shared_;
  std::unique_ptr<DeviceBuffer> buf_;
};

// [N][T+1][record] state records living in HBM; records and series are fetched on demand.
class DeviceRecords {
 public:
  DeviceRecords(const DeviceSeries& s, bool packed) : s_(&s), packed_(packed), rec_(packed ? dlm_packed_record_doubles(s.d()) : s.d() + s.d() * s.d()),
                                                      buf_(std::make_shared<DeviceBuffer>(s.engine(), (size_t)s.N() * (s.T() + 1) * rec_ * sizeof(double))) {}
  double* ptr() const { return buf_->doubles(); }
  int recordDoubles() const { return rec_; }
  bool packed() const { return packed_; }
  size_t bytes() const { return buf_->bytes(); }
  // (mean, covariance) of series n at record k (0 = the initial state at t0 - 1): one small D2H copy
  KfState at(int n, int k) const {
    std::vector<double> r((size_t)rec_);
    buf_->download(r.data(), r.size() * 8, ((size_t)n * (s_->T() + 1) + k) * rec_ * 8);
    return toState(r.data(), k);
  }
  // all T+1 records of series n: one contiguous D2H copy
  std::vector<KfState> series(int n) const {
    std::vector<double> r((size_t)(s_->T() + 1) * rec_);
    buf_->download(r.data(), r.size() * 8, (size_t)n * (s_->T() + 1) * rec_ * 8);
    std::vector<KfState> out;
    for (int k = 0; k <= s_->T(); ++k) out.push_back(toState(r.data() + (size_t)k * rec_, k));
    return out;
  }
 private:
  KfState toState(const double* r, int k) const {
    const int d = s_->d();
    return KfState{s_->timeOf(k), std::vector<double>(r, r + d), packed_ ? detail::unpackLower(r + d, d) : detail::matAt(r + d, d)};
  }
  const DeviceSeries* s_; bool packed_; int rec_;
  std::shared_ptr<DeviceBuffer> buf_;
};

struct DeviceFilterSmooth {
  DeviceRecords filtered, smoothed;      // KfState-like (m_t, C_t) and SmoothingState-like (s_t, S_t) records
  std::vector<int32_t> status;           // per-series DLM_ST_* flags
  double forwardMs = 0.0, backwardMs = 0.0;   // device time of the two kernels (HIP events)
  std::shared_ptr<DeviceBuffer> statusDev;    // [N] int32 on the device (kept for repeated calls)
};

namespace KalmanFilter {
// KalmanFilter(...).filter on device-resident observations; records stay on the device
inline DeviceRecords filter(const DeviceSeries& ys, const DeviceParameters& p, std::vector<int32_t>* status = nullptr, Flags flags = 0) {
  Engine& e = ys.engine();
  DeviceRecords filt(ys, (flags & DLM_OPT_PACKED_SYM) != 0);
  DeviceBuffer st(e, (size_t)ys.N() * 4);
  const dlm_model_desc m = ys.modelDesc(); const dlm_params_desc q = p.desc();
  const dlm_options o = detail::opts(flags, DLM_MEM_DEVICE);
  e.check(dlm_filter_batch(e.get(), &m, &q, ys.y(), &o, filt.ptr(), nullptr, nullptr, static_cast<int32_t*>(st.ptr())));
  if (status) { status->resize(ys.N()); st.download(status->data(), st.bytes()); }
  return filt;
}
inline std::vector<double> logLikelihood(const DeviceSeries& ys, const DeviceParameters& p, Flags flags = 0) {
  Engine& e = ys.engine();
  DeviceBuffer ll(e, (size_t)ys.N() * 8), st(e, (size_t)ys.N() * 4);
  const dlm_model_desc m = ys.modelDesc(); const dlm_params_desc q = p.desc();
  const dlm_options o = detail::opts(flags, DLM_MEM_DEVICE);
  e.check(dlm_loglik_batch(e.get(), &m, &q, ys.y(), &o, ll.doubles(), static_cast<int32_t*>(st.ptr())));
  std::vector<double> out((size_t)ys.N());
  ll.download(out.data(), ll.bytes());
  return out;
}
// KalmanFilter.likelihood as written (KalmanFilter.scala:299-306) on device-resident series: the transition density of the filtered means
inline std::vector<double> likelihood(const DeviceSeries& ys, const DeviceParameters& p, Flags flags = 0) {
  return logLikelihood(ys, p, flags | DLM_OPT_LOGLIK_LITERAL_Q7);
}
}  // namespace KalmanFilter

namespace Smoothing {
// filter + backwardsSmoother fused (dlm_filter_smooth_batch), everything resident in HBM: the metric path
inline DeviceFilterSmooth filterSmooth(const DeviceSeries& ys, const DeviceParameters& p, Flags flags = 0) {
  Engine& e = ys.engine();
  const bool packed = (flags & DLM_OPT_PACKED_SYM) != 0;
  DeviceFilterSmooth r{DeviceRecords(ys, packed), DeviceRecords(ys, packed), std::vector<int32_t>((size_t)ys.N())};
  r.statusDev = std::make_shared<DeviceBuffer>(e, (size_t)ys.N() * 4);
  const dlm_model_desc m = ys.modelDesc(); const dlm_params_desc q = p.desc();
  const dlm_options o = detail::opts(flags, DLM_MEM_DEVICE);
  e.check(dlm_filter_smooth_batch(e.get(), &m, &q, ys.y(), &o, r.filtered.ptr(), r.smoothed.ptr(), static_cast<int32_t*>(r.statusDev->ptr())));
  r.statusDev->download(r.status.data(), r.statusDev->bytes());
  double ms[2] = {0.0, 0.0};
  if (dlm_last_timing(e.get(), ms) == DLM_OK) { r.forwardMs = ms[0]; r.backwardMs = ms[1]; }
  return r;
}
// the same into records the caller already holds (repeated calls: no allocation inside the timed region)
inline void filterSmoothInto(const DeviceSeries& ys, const DeviceParameters& p, DeviceFilterSmooth& r, Flags flags = 0) {
  Engine& e = ys.engine();
  const dlm_model_desc m = ys.modelDesc(); const dlm_params_desc q = p.desc();
  const dlm_options o = detail::opts(flags | (r.filtered.packed() ? DLM_OPT_PACKED_SYM : 0), DLM_MEM_DEVICE);
  e.check(dlm_filter_smooth_batch(e.get(), &m, &q, ys.y(), &o, r.filtered.ptr(), r.smoothed.ptr(), static_cast<int32_t*>(r.statusDev->ptr())));
  r.statusDev->download(r.status.data(), r.statusDev->bytes());
  double ms[2] = {0.0, 0.0};
  if (dlm_last_timing(e.get(), ms) == DLM_OK) { r.forwardMs = ms[0]; r.backwardMs = ms[1]; }
}
// Smoothing.backwardsSmoother on filter records already on the device
inline DeviceRecords backwardsSmoother(const DeviceSeries& ys, const DeviceParameters& p, const DeviceRecords& kfStates, Flags flags = 0) {
  Engine& e = ys.engine();
  // dlm_smooth_batch takes DENSE records (packed ones are read by the fused call only): packed input is expanded first
  if (kfStates.packed()) {
    DeviceRecords dense(ys, false);
    const dlm_options ou = detail::opts(0, DLM_MEM_DEVICE);
    e.check(dlm_unpack_records(e.get(), ys.d(), (int64_t)ys.N() * (ys.T() + 1), kfStates.ptr(), &ou, dense.ptr()));
    return backwardsSmoother(ys, p, dense, flags);
  }
  DeviceRecords sm(ys, false);
  const dlm_model_desc m = ys.modelDesc(); const dlm_params_desc q = p.desc();
  const dlm_options o = detail::opts(flags & ~(Flags)DLM_OPT_PACKED_SYM, DLM_MEM_DEVICE);
  e.check(dlm_smooth_batch(e.get(), &m, &q, kfStates.ptr(), &o, sm.ptr(), nullptr));
  return sm;
}
}  // namespace Smoothing

// ------------------------------------------------------------------------------------------------------------
// Gibbs samplers (Gibbs.scala:134-217, GibbsWishart.scala:40-80) for N independent series: FFBS + sufficient
// statistics on the device every iteration, conjugate draws on the host (tiny), per-series parameters back up.
// ------------------------------------------------------------------------------------------------------------
struct InverseGamma {   // InverseGamma(shape, scale): draw = 1 / Gamma(shape, 1 / scale).draw (InverseGamma.scala:14)
  double shape, scale;
  template <class Rng> double draw(Rng& r) const { return 1.0 / std::gamma_distribution<double>(shape, 1.0 / scale)(r); }
};
struct InverseWishart {   // InverseWishart(nu, psi), drawn through the Bartlett factor (InverseWishart.scala:17-25, Wishart.scala:34-43)
  double nu; Matrix psi;
  template <class Rng> Matrix draw(Rng& r) const {
    const int d = psi.rows;
    // W ~ Wishart(nu, psi^-1) = L A A^T L^T with L = chol(psi^-1), A the Bartlett factor; the draw is W^-1
    auto chol = [](const Matrix& a) { const int n = a.rows; Matrix l(n, n);
      for (int j = 0; j < n; ++j) { double s = a(j, j); for (int k = 0; k < j; ++k) s -= l(j, k) * l(j, k); if (!(s > 0.0)) throw std::runtime_error("InverseWishart: scale not positive definite");
        l(j, j) = std::sqrt(s); for (int i = j + 1; i < n; ++i) { double t = a(i, j); for (int k = 0; k < j; ++k) t -= l(i, k) * l(j, k); l(i, j) = t / l(j, j); } }
      return l; };
    auto invLower = [](const Matrix& l) { const int n = l.rows; Matrix x(n, n);
      for (int j = 0; j < n; ++j) { x(j, j) = 1.0 / l(j, j); for (int i = j + 1; i < n; ++i) { double s = 0.0; for (int k = j; k < i; ++k) s += l(i, k) * x(k, j); x(i, j) = -s / l(i, i); } }
      return x; };
    auto mul = [](const Matrix& a, const Matrix& b, bool ta, bool tb) { const int n = a.rows; Matrix c(n, n);
      for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0.0; for (int k = 0; k < n; ++k) s += (ta ? a(k, i) : a(i, k)) * (tb ? b(j, k) : b(k, j)); c(i, j) = s; }
      return c; };
    // psi^-1 = Lp^-T Lp^-1 with Lp = chol(psi); its Cholesky factor is needed: chol(psi^-1)
    const Matrix lpInv = invLower(chol(psi));
    const Matrix L = chol(mul(lpInv, lpInv, true, false));
    Matrix A(d, d);
    std::normal_distribution<double> nrm;
    for (int i = 0; i < d; ++i) { A(i, i) = std::sqrt(std::chi_squared_distribution<double>(nu - i)(r)); for (int j = 0; j < i; ++j) A(i, j) = nrm(r); }
    const Matrix LA = mul(L, A, false, false);           // W = LA LA^T
    const Matrix laInv = invLower(LA);                   // LA is lower triangular
    return mul(laInv, laInv, true, false);               // W^-1 = LA^-T LA^-1
  }
};

namespace GibbsSampling {
struct State { std::vector<DlmParameters> p; };   // GibbsSampling.State (Gibbs.scala:8-11) per series; the state draw stays on the device

// One chain per series, advanced together.  next() = dinvGammaStep (Gibbs.scala:134-151) for every series:
//   theta ~ FFBS (device), V_jj ~ IG(a + n_j / 2, b + ssy_j / 2), W_ii ~ IG(a + T / 2, b + ss_i / 2) (Gibbs.scala:41-48,72-77).
// svd = true is stepSvd / sampleSvd (Gibbs.scala:182-217): the state draw comes from the SVD filter / sampler.
// wishart != nullptr is GibbsWishart.wishartStep (GibbsWishart.scala:40-53): W ~ InverseWishart(nu + T, psi + sum outer), then V.
class Chain {
 public:
  Chain(const DeviceSeries& ys, InverseGamma priorV, InverseGamma priorW, const DlmParameters& init, uint64_t seed, Flags flags, bool svd,
        const InverseWishart* wishart = nullptr)
      : ys_(&ys), pv_(priorV), pw_(priorW), svd_(svd), flags_(flags | (wishart ? DLM_OPT_STATS_OUTER : 0)), seed_(seed), rng_(seed),
        params_((size_t)ys.N(), init), dev_(ys.engine(), params_), L_(dlm_stats_len(ys.d(), ys.p(), flags_)) {
    if (wishart) wish_ = std::make_unique<InverseWishart>(*wishart);
    const size_t rec = svd ? 2 * (size_t)ys.d() + (size_t)ys.d() * ys.d() : (size_t)ys.d() + (size_t)ys.d() * ys.d();
    ws_ = std::make_unique<DeviceBuffer>(ys.engine(), (size_t)ys.N() * (ys.T() + 1) * rec * 8);
    stats_ = std::make_unique<DeviceBuffer>(ys.engine(), (size_t)ys.N() * L_ * 8);
    theta_ = std::make_unique<DeviceBuffer>(ys.engine(), (size_t)ys.N() * (ys.T() + 1) * ys.d() * 8);
  }
  const State& next() {
    Engine& e = ys_->engine();
    const int d = ys_->d(), p = ys_->p(), N = ys_->N();
    const dlm_model_desc m = ys_->modelDesc(); const dlm_params_desc q = dev_.desc();
    const dlm_options o = detail::opts(flags_, DLM_MEM_DEVICE, seed_ * 1000003ull + iter_);
    if (svd_) e.check(dlm_svd_ffbs_batch(e.get(), &m, &q, ys_->y(), nullptr, &o, ws_->doubles(), theta_->doubles(), stats_->doubles(), nullptr));
    else e.check(dlm_ffbs_batch(e.get(), &m, &q, ys_->y(), nullptr, &o, ws_->doubles(), theta_->doubles(), nullptr, stats_->doubles(), nullptr));
    std::vector<double> st((size_t)N * L_);
    stats_->download(st.data(), st.size() * 8);
    std::vector<double> V((size_t)N * p * p, 0.0), W((size_t)N * d * d, 0.0);
    for (int n = 0; n < N; ++n) {
      const double* s = st.data() + (size_t)n * L_;
      const double tcount = s[L_ - 1];
      DlmParameters& pn = params_[n];
      Matrix w(d, d);
      if (wish_) {   // order theta, W, V as wishartStep
        Matrix scale = wish_->psi;
        for (int k = 0; k < d * d; ++k) scale.data[k] += s[2 * p + k];
        w = InverseWishart{wish_->nu + tcount, scale}.draw(rng_);
      }
      Matrix v(p, p);
      for (int j = 0; j < p; ++j) v(j, j) = InverseGamma{pv_.shape + 0.5 * s[p + j], pv_.scale + 0.5 * s[j]}.draw(rng_);
      if (!wish_) for (int i = 0; i < d; ++i) w(i, i) = InverseGamma{pw_.shape + 0.5 * tcount, pw_.scale + 0.5 * s[2 * p + i]}.draw(rng_);
      pn.v = v; pn.w = w;
      std::copy(v.data.begin(), v.data.end(), V.begin() + (size_t)n * p * p);
      std::copy(w.data.begin(), w.data.end(), W.begin() + (size_t)n * d * d);
    }
    state_.p = params_;
    dev_.setVW(V, W);
    ++iter_;
    return state_;
  }
  // the state draw of the last iteration for one series (theta_0 .. theta_T)
  std::vector<SamplingState> theta(int n) const {
    const int d = ys_->d(), T = ys_->T();
    std::vector<double> r((size_t)(T + 1) * d);
    theta_->download(r.data(), r.size() * 8, (size_t)n * (T + 1) * d * 8);
    std::vector<SamplingState> out;
    for (int k = 0; k <= T; ++k) out.push_back(SamplingState{ys_->timeOf(k), std::vector<double>(r.begin() + (size_t)k * d, r.begin() + (size_t)(k + 1) * d)});
    return out;
  }
 private:
  const DeviceSeries* ys_; InverseGamma pv_, pw_; bool svd_; Flags flags_; uint64_t seed_; std::mt19937_64 rng_;
  std::vector<DlmParameters> params_; DeviceParameters dev_; int L_; uint64_t iter_ = 0;
  std::unique_ptr<InverseWishart> wish_;
  std::unique_ptr<DeviceBuffer> ws_, stats_, theta_;
  State state_;
};

// GibbsSampling.sample(mod, priorV, priorW, initParams, observations) (Gibbs.scala:165-180): the model travels inside `observations`
inline Chain sample(InverseGamma priorV, InverseGamma priorW, const DlmParameters& initParams, const DeviceSeries& observations, uint64_t seed = 0, Flags flags = 0) {
  return Chain(observations, priorV, priorW, initParams, seed, flags, false);
}
// GibbsSampling.sampleSvd (Gibbs.scala:203-217)
inline Chain sampleSvd(InverseGamma priorV, InverseGamma priorW, const DlmParameters& initParams, const DeviceSeries& observations, uint64_t seed = 0, Flags flags = 0) {
  return Chain(observations, priorV, priorW, initParams, seed, flags, true);
}
}  // namespace GibbsSampling

namespace GibbsWishart {
// GibbsWishart.sample(mod, priorV, priorW, initParams, observations) (GibbsWishart.scala:65-80)
inline GibbsSampling::Chain sample(InverseGamma priorV, const InverseWishart& priorW, const DlmParameters& initParams, const DeviceSeries& observations,
                                   uint64_t seed = 0, Flags flags = 0) {
  return GibbsSampling::Chain(observations, priorV, InverseGamma{1.0, 1.0}, initParams, seed, flags, false, &priorW);
}
}  // namespace GibbsWishart

}  // namespace dlm_host
