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
//   Smoothing::backwardsSmoother / filterSmooth / ffbsDlm                      Smoothing.scala:57-64,173-180
//
// Matrices are column-major (Breeze DenseMatrix.data).  All entry points run the HIP engine in
// DLM_MEM_HOST mode; errors throw std::runtime_error with dlm_last_error().
#pragma once
#include <cmath>
#include <functional>
#include <limits>
#include <optional>
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
 private:
  dlm_engine* e_ = nullptr;
};

namespace detail {
// closures -> flat tables on the shared time grid (the job of the Scala shim in INTEGRATION.md)
struct Tables {
  int d = 0, p = 0, T = 0, N = 0, nG = 0;
  std::vector<double> F, G, dt, y, times;
  std::vector<int32_t> gIndex;
  int64_t fStride = 0;
};
inline Tables materialise(const Dlm& mod, const std::vector<std::vector<Data>>& ys) {
  if (ys.empty() || ys[0].empty()) throw std::invalid_argument("empty observation vector (KalmanFilter.scala:116-117 throws on t0.get)");
  Tables t;
  t.N = (int)ys.size(); t.T = (int)ys[0].size();
  double t0 = ys[0][0].time;
  for (const auto& d : ys[0]) { t.times.push_back(d.time); t0 = std::min(t0, d.time); }
  double prev = t0 - 1.0;  // KalmanFilter.initialiseState
  std::vector<Matrix> gs, fs;
  std::vector<double> seen;
  for (int k = 0; k < t.T; ++k) {
    const double dt = t.times[k] - prev; prev = t.times[k];
    t.dt.push_back(dt);
    Matrix gm = mod.g(dt);
    int idx = -1;
    for (size_t q = 0; q < gs.size(); ++q) if (gs[q] == gm) idx = (int)q;
    if (idx < 0) { idx = (int)gs.size(); gs.push_back(gm); }
    t.gIndex.push_back(idx);
    fs.push_back(mod.f(t.times[k]));
  }
  t.d = fs[0].rows; t.p = fs[0].cols; t.nG = (int)gs.size();
  bool constF = true;
  for (const auto& f : fs) constF = constF && (f == fs[0]);
  if (constF) t.F = fs[0].data; else { t.fStride = (int64_t)t.d * t.p; for (const auto& f : fs) t.F.insert(t.F.end(), f.data.begin(), f.data.end()); }
  for (const auto& g : gs) t.G.insert(t.G.end(), g.data.begin(), g.data.end());
  for (const auto& s : ys) {
    if ((int)s.size() != t.T) throw std::invalid_argument("all series of a batch must share one time grid");
    for (const auto& d : s) for (const auto& o : d.observation) t.y.push_back(o ? *o : std::numeric_limits<double>::quiet_NaN());
  }
  return t;
}
inline dlm_model_desc modelDesc(const Tables& t) {
  dlm_model_desc m{};
  m.d = t.d; m.p = t.p; m.T = t.T; m.N = t.N; m.F = t.F.data(); m.f_stride = t.fStride;
  m.G = t.G.data(); m.n_g = t.nG; m.g_index = t.nG > 1 ? t.gIndex.data() : nullptr;
  bool unit = true; for (double x : t.dt) unit = unit && (x == 1.0);
  m.dt = unit ? nullptr : t.dt.data();
  return m;
}
inline dlm_params_desc paramsDesc(const DlmParameters& p) {
  return dlm_params_desc{p.v.data.data(), 0, p.w.data.data(), 0, p.m0.data(), 0, p.c0.data.data(), 0};
}
inline Matrix matAt(const double* rec, int d) { Matrix m(d, d); m.data.assign(rec, rec + (size_t)d * d); return m; }
}  // namespace detail

namespace KalmanFilter {
// KalmanFilter(advanceState(p, mod.g)).filter(mod, ys, p): T+1 states per series, the first being the
// initial state at t0 - 1 (Filter.scala:41-45)
inline std::vector<std::vector<KfState>> filter(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o{0, DLM_MEM_HOST, 0, 0};
  const int rec = t.d + t.d * t.d;
  std::vector<double> filt((size_t)t.N * (t.T + 1) * rec);
  e.check(dlm_filter_batch(e.get(), &m, &q, t.y.data(), &o, filt.data(), nullptr, nullptr, nullptr));
  std::vector<std::vector<KfState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const double* r = filt.data() + ((size_t)n * (t.T + 1) + k) * rec;
      out[n].push_back(KfState{k == 0 ? t.times[0] - t.dt[0] : t.times[k - 1], std::vector<double>(r, r + t.d), detail::matAt(r + t.d, t.d)});
    }
  return out;
}
// sum over the series of KalmanFilter.conditionalLikelihood (KalmanFilter.scala:138-153): log p(y_1:T | V, W) per series
inline std::vector<double> logLikelihood(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o{0, DLM_MEM_HOST, 0, 0};
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
// KalmanFilter(...).filter followed by Smoothing.backwardsSmoother, fused on the device
inline std::vector<std::vector<SmoothingState>> filterSmooth(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p,
                                                             std::vector<std::vector<KfState>>* filtered = nullptr) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o{0, DLM_MEM_HOST, 0, 0};
  const int rec = t.d + t.d * t.d;
  std::vector<double> filt((size_t)t.N * (t.T + 1) * rec), sm(filt.size());
  e.check(dlm_filter_smooth_batch(e.get(), &m, &q, t.y.data(), &o, filt.data(), sm.data(), nullptr));
  std::vector<std::vector<SmoothingState>> out(t.N);
  if (filtered) filtered->assign(t.N, {});
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const size_t off = ((size_t)n * (t.T + 1) + k) * rec;
      const double time = k == 0 ? t.times[0] - t.dt[0] : t.times[k - 1];
      out[n].push_back(SmoothingState{time, std::vector<double>(sm.data() + off, sm.data() + off + t.d), detail::matAt(sm.data() + off + t.d, t.d)});
      if (filtered) (*filtered)[n].push_back(KfState{time, std::vector<double>(filt.data() + off, filt.data() + off + t.d), detail::matAt(filt.data() + off + t.d, t.d)});
    }
  return out;
}
// Smoothing.ffbsDlm: one draw of the T+1 states per series, a pure function of (seed, series index)
inline std::vector<std::vector<SamplingState>> ffbsDlm(Engine& e, const Dlm& mod, const std::vector<std::vector<Data>>& ys, const DlmParameters& p, uint64_t seed) {
  detail::Tables t = detail::materialise(mod, ys);
  const dlm_model_desc m = detail::modelDesc(t); const dlm_params_desc q = detail::paramsDesc(p);
  const dlm_options o{0, DLM_MEM_HOST, seed, 0};
  const int rec = t.d + t.d * t.d;
  std::vector<double> ws((size_t)t.N * (t.T + 1) * rec), th((size_t)t.N * (t.T + 1) * t.d);
  e.check(dlm_ffbs_batch(e.get(), &m, &q, t.y.data(), nullptr, &o, ws.data(), th.data(), nullptr, nullptr, nullptr));
  std::vector<std::vector<SamplingState>> out(t.N);
  for (int n = 0; n < t.N; ++n)
    for (int k = 0; k <= t.T; ++k) {
      const double* r = th.data() + ((size_t)n * (t.T + 1) + k) * t.d;
      out[n].push_back(SamplingState{k == 0 ? t.times[0] - t.dt[0] : t.times[k - 1], std::vector<double>(r, r + t.d)});
    }
  return out;
}
}  // namespace Smoothing

}  // namespace dlm_host
