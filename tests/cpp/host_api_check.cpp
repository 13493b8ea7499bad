// C++ host-layer check (include/dlm_host.hpp): reads like the reference's own tests.
//   1. KalmanFilterTest table (core/src/test/scala/KalmanFilter.scala:78-189), tol 1e-4
//   2. first_order_dlm golden CSVs (examples/src/main/scala/dlm/FirstOrderDlm.scala:52-77,237-255)
//   3. seasonal d = 13 batch: filterDlm drops the initial state, smoothed == filtered at T
// usage: host_api_check <tests/golden dir>; prints "HOST API OK" and returns 0 on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include "dlm_host.hpp"
using namespace dlm_host;

static int fails = 0;
static void expect(bool ok, const char* what) { if (!ok) { ++fails; std::printf("FAIL: %s\n", what); } }
static bool close(double a, double b, double tol) { return std::fabs(a - b) <= tol * (1.0 + std::fabs(b)); }

static std::vector<std::vector<double>> read_csv(const std::string& path) {
  std::ifstream in(path);
  if (!in) { std::printf("cannot open %s\n", path.c_str()); std::exit(2); }
  std::vector<std::vector<double>> rows; std::string line; std::getline(in, line);
  while (std::getline(in, line)) {
    std::vector<double> r; std::stringstream ss(line); std::string cell;
    while (std::getline(ss, cell, ',')) r.push_back(std::stod(cell));
    rows.push_back(r);
  }
  return rows;
}

int main(int argc, char** argv) {
  const std::string golden = argc > 1 ? argv[1] : "tests/golden";
  Engine eng(0);

  {  // 1. bivariate local level with missing data and a dt = 2 step
    Dlm model = Dlm::polynomial(1).outer(Dlm::polynomial(1));
    DlmParameters p{Matrix::diag({3.0, 3.0}), Matrix::diag({1.0, 1.0}), {0.0, 0.0}, Matrix::diag({1.0, 1.0})};
    std::vector<Data> data = {{1.0, {4.5, 4.5}}, {2.0, {3.0, 3.0}}, {3.0, {6.3, 6.3}}, {4.0, {std::nullopt, std::nullopt}},
                              {5.0, {10.1, std::nullopt}}, {7.0, {15.2, 15.2}}};
    auto st = KalmanFilter::filterDlm(eng, model, {data}, p)[0];
    expect(st.size() == data.size(), "filterDlm returns one state per observation");
    const double tol = 1e-4;
    expect(close(st[1].mt[0], 2.307692, tol) && close(st[1].ct(0, 0), 1.269231, tol), "time step 2");
    expect(close(st[2].mt[0], 4.027007, tol) && close(st[2].ct(1, 1), 1.291971, tol), "time step 3");
    expect(close(st[3].mt[0], 4.027007, tol) && close(st[3].ct(0, 0), 2.291971, tol), "time step 4, missing data");
    expect(close(st[4].mt[0], 7.204408, tol) && close(st[4].mt[1], 4.027007, tol) && close(st[4].ct(0, 0), 1.569606, tol) &&
               close(st[4].ct(1, 1), 3.291971, tol), "time step 5, partially observed");
    expect(close(st[5].mt[0], 11.54883, tol) && close(st[5].ct(0, 0), 1.630055, tol), "time step 7 (commented expectations)");
  }
  {  // 2. config C1 golden CSVs
    auto obs = read_csv(golden + "/first_order_dlm.csv");
    auto fr = read_csv(golden + "/first_order_dlm_filtered.csv");
    auto sr = read_csv(golden + "/first_order_dlm_smoothed.csv");
    std::vector<Data> ys;
    for (auto& r : obs) ys.push_back(Data{r[0], {r[1]}});
    DlmParameters p{Matrix::diag({2.0}), Matrix::diag({3.0}), {0.0}, Matrix::diag({10.0})};
    std::vector<std::vector<KfState>> filt;
    auto sm = Smoothing::filterSmooth(eng, Dlm::polynomial(1), {ys}, p, &filt)[0];
    expect(sm.size() == 1001 && filt[0].size() == 1001, "T+1 states including the initial state");
    double ef = 0, es = 0;
    for (size_t t = 0; t < 1001; ++t) {
      ef = std::max(ef, std::fabs(filt[0][t].mt[0] - fr[t][1])); ef = std::max(ef, std::fabs(filt[0][t].ct(0, 0) - fr[t][2]));
      es = std::max(es, std::fabs(sm[t].mean[0] - sr[t][1])); es = std::max(es, std::fabs(sm[t].covariance(0, 0) - sr[t][2]));
    }
    std::printf("first_order_dlm: max |filtered - golden| = %.3g, max |smoothed - golden| = %.3g\n", ef, es);
    expect(ef < 1e-11 && es < 1e-10, "golden CSVs reproduced");
    expect(sm[0].time == 0.0 && sm[1000].time == 1000.0, "times");
    // log-likelihood against the forecasts (f, Q) the reference itself wrote into the golden CSV
    double want = 0.0;
    for (size_t t = 1; t < 1001; ++t) {
      const double e = obs[t - 1][1] - fr[t][3], Q = fr[t][4];
      want -= 0.5 * (std::log(2.0 * M_PI) + std::log(Q) + e * e / Q);
    }
    const double got = KalmanFilter::logLikelihood(eng, Dlm::polynomial(1), {ys}, p)[0];
    std::printf("first_order_dlm: log-likelihood %.10f (from the golden forecasts %.10f)\n", got, want);
    expect(std::fabs(got - want) < 1e-8 * std::fabs(want), "log-likelihood from the reference's forecasts");
  }
  {  // 3. seasonal d = 13 batch + FFBS shape
    Dlm mod = Dlm::polynomial(1).compose(Dlm::seasonal(24, 6));
    DlmParameters p{Matrix::diag({1.0}), Matrix::diag({0.01, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4, 0.2, 0.4, 0.5, 0.2, 0.1, 0.4}),
                    std::vector<double>(13, 0.0), Matrix::eye(13)};
    std::vector<std::vector<Data>> ys(3);
    for (int n = 0; n < 3; ++n) for (int t = 1; t <= 50; ++t) ys[n].push_back(Data{(double)t, {std::sin(0.3 * t + n) + 0.1 * n}});
    std::vector<std::vector<KfState>> filt;
    auto sm = Smoothing::filterSmooth(eng, mod, ys, p, &filt);
    expect(sm.size() == 3 && sm[0].size() == 51 && sm[0][0].mean.size() == 13, "batch shapes");
    double dlast = 0;
    for (int i = 0; i < 13; ++i) dlast = std::max(dlast, std::fabs(sm[2][50].mean[i] - filt[2][50].mt[i]));
    expect(dlast < 1e-12, "smoothed state at T equals the filtered state (Smoothing.scala:59-61)");
    auto th = Smoothing::ffbsDlm(eng, mod, ys, p, 42);
    auto th2 = Smoothing::ffbsDlm(eng, mod, ys, p, 42);
    expect(th.size() == 3 && th[0].size() == 51 && th[0][0].sample.size() == 13, "ffbs shapes");
    expect(th[1][7].sample == th2[1][7].sample, "draws are a pure function of the seed");
  }
  try {  // empty input is an error, as in the reference (t0.get on None)
    KalmanFilter::filterDlm(eng, Dlm::polynomial(1), {}, DlmParameters{Matrix::diag({1.0}), Matrix::diag({1.0}), {0.0}, Matrix::diag({1.0})});
    expect(false, "empty input must throw");
  } catch (const std::invalid_argument&) {}
  std::printf(fails ? "HOST API FAILED (%d)\n" : "HOST API OK\n", fails);
  return fails != 0;
}
