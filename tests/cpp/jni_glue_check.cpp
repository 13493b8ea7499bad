// Drives EVERY native method of integration/jni/dlm_jni.cpp on the GPU through a functional stand-in for the JNI calls it
// makes (tests/cpp/jni_stub/jni.h; there is no JDK in this image): what Batched.scala would do, written in C++.
//   * config C1 (the reference's golden CSVs) device-resident through bufferAlloc / bufferUpload / filterSmooth / bufferDownload
//   * the same in host mode through Native.address of "direct buffers"
//   * every other entry point once, cross-checked against the fused call or for shape / finiteness
//   * error mapping: bad descriptor arrays -> IllegalArgumentException, engine errors -> RuntimeException(dlm_last_error)
// usage: jni_glue_check <tests/golden dir>; prints "JNI GLUE OK" and returns 0 on success.
// Every native call goes through CALL(), which notes the entry point in flight; a watchdog thread reports it on stderr every 20 s
// and ends the process (exit code 3) after 150 s in ONE call -- a hang names its entry point instead of dying silently at the test
// runner's timeout (round 3, gpurun_out/r03s/full7.txt: 300 s without a byte of output; stdout was a fully buffered pipe).
#include <jni.h>   // tests/cpp/jni_stub/jni.h
#include "../../integration/jni/dlm_jni.cpp"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <thread>
#include <unistd.h>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include "dlm_host.hpp"
using namespace dlm_host;

#define N_(name) Java_com_github_jonnylaw_dlm_gpu_Native_##name
static std::atomic<const char*> g_stage{"start-up"};
static std::atomic<long> g_calls{0};
#define CALL(name, ...) (g_stage.store(#name), g_calls.fetch_add(1), N_(name)(__VA_ARGS__))
static void watchdog() {
  long seen = -1; int quiet = 0;
  for (;;) {
    std::this_thread::sleep_for(std::chrono::seconds(5));
    const long c = g_calls.load();
    if (c != seen) { seen = c; quiet = 0; continue; }
    quiet += 5;
    if (quiet % 20 == 0) { std::fprintf(stderr, "WATCHDOG: %d s inside native call #%ld, %s\n", quiet, c, g_stage.load()); std::fflush(stderr); }
    if (quiet >= 150) { std::fprintf(stderr, "WATCHDOG: giving up -- hung in %s\n", g_stage.load()); std::fflush(stderr); std::fflush(stdout); _exit(3); }
  }
}

static int fails = 0;
static void expect(bool ok, const char* what) { if (!ok) { ++fails; std::printf("FAIL: %s\n", what); } }
static jlong A(const void* p) { return static_cast<jlong>(reinterpret_cast<uintptr_t>(p)); }

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

struct Dev {   // a device buffer through the glue
  JNIEnv* env; jlong h; jlong p; size_t bytes;
  Dev(JNIEnv* e, jlong h_, size_t b) : env(e), h(h_), p(CALL(bufferAlloc, e, nullptr, h_, (jlong)b)), bytes(b) {}
  ~Dev() { CALL(bufferFree, env, nullptr, h, p); }
  void up(const void* src, size_t b, size_t off = 0) { CALL(bufferUpload, env, nullptr, h, p, (jlong)off, CALL(address, env, nullptr, env->direct(const_cast<void*>(src))), (jlong)b); }
  void down(void* dst, size_t b, size_t off = 0) { CALL(bufferDownload, env, nullptr, h, p, (jlong)off, CALL(address, env, nullptr, env->direct(dst)), (jlong)b); }
};

int main(int argc, char** argv) {
  const std::string golden = argc > 1 ? argv[1] : "tests/golden";
  std::setvbuf(stdout, nullptr, _IOLBF, 0);   // (a pipe would otherwise hold every line back until exit)
  std::thread(watchdog).detach();
  JNIEnv envObj; JNIEnv* env = &envObj;
  const jlong h = CALL(engineCreate, env, nullptr, 0);
  expect(h != 0 && !env->pending, "engineCreate");
  if (!h) { std::printf("JNI GLUE FAILED (no engine)\n"); return 1; }
  expect(CALL(version, env, nullptr)->str.find("gfx950") != std::string::npos, "version");

  // ---- config C1 from the reference's files -------------------------------------------------------------------
  auto obs = read_csv(golden + "/first_order_dlm.csv");
  auto fr = read_csv(golden + "/first_order_dlm_filtered.csv");
  auto sr = read_csv(golden + "/first_order_dlm_smoothed.csv");
  const int T = (int)obs.size(), d = 1, p = 1, rec = 2;
  std::vector<double> y(T), V{2.0}, W{3.0}, m0{0.0}, C0{10.0}, F{1.0}, G{1.0};
  for (int t = 0; t < T; ++t) y[t] = obs[t][1];
  std::vector<double> filt((size_t)(T + 1) * rec), sm(filt.size()), filtH(filt.size()), smH(filt.size());
  std::vector<int32_t> status(1, -1);
  {
    Dev dy(env, h, y.size() * 8), dtab(env, h, 6 * 256), dfilt(env, h, filt.size() * 8), dsm(env, h, sm.size() * 8), dst(env, h, 256);
    dy.up(y.data(), y.size() * 8);
    const double tabs[6] = {F[0], G[0], V[0], W[0], m0[0], C0[0]};
    for (int k = 0; k < 6; ++k) dtab.up(&tabs[k], 8, (size_t)k * 256);
    CALL(bufferFill, env, nullptr, h, dst.p, 0, 0xff, 256);
    jlongArray model = env->longs({d, p, T, 1, dtab.p, 0, dtab.p + 256, 1, 0, 0});
    jlongArray params = env->longs({dtab.p + 512, 0, dtab.p + 768, 0, dtab.p + 1024, 0, dtab.p + 1280, 0, 0, 0});
    jlongArray optsDev = env->longs({0, DLM_MEM_DEVICE, 0, 0});
    CALL(filterSmooth, env, nullptr, h, model, params, optsDev, dy.p, dfilt.p, dsm.p, dst.p);
    expect(!env->pending, "filterSmooth (device mode) raised no exception");
    dfilt.down(filt.data(), filt.size() * 8); dsm.down(sm.data(), sm.size() * 8); dst.down(status.data(), 4);
    double ef = 0, es = 0;
    for (int t = 0; t <= T; ++t) {
      ef = std::max(ef, std::max(std::fabs(filt[2 * t] - fr[t][1]), std::fabs(filt[2 * t + 1] - fr[t][2])));
      es = std::max(es, std::max(std::fabs(sm[2 * t] - sr[t][1]), std::fabs(sm[2 * t + 1] - sr[t][2])));
    }
    std::printf("JNI glue, device-resident first_order_dlm: max |filtered - golden| = %.3g, |smoothed - golden| = %.3g, variant %s\n", ef, es,
                CALL(lastVariant, env, nullptr, h)->str.c_str());
    expect(ef < 1e-11 && es < 1e-10 && status[0] == 0, "golden CSVs reproduced through the JNI glue (device mode)");
    jdoubleArray ms = CALL(lastTiming, env, nullptr, h);
    expect(ms && ms->doubles.size() == 2 && ms->doubles[0] > 0.0 && ms->doubles[1] > 0.0, "lastTiming");
    // separate filter and smoother calls on the device give the same records
    Dev df2(env, h, filt.size() * 8), ds2(env, h, sm.size() * 8);
    CALL(filter, env, nullptr, h, model, params, optsDev, dy.p, df2.p, 0, 0, dst.p);
    CALL(smooth, env, nullptr, h, model, params, optsDev, df2.p, ds2.p, dst.p);
    std::vector<double> f2(filt.size()), s2(sm.size());
    df2.down(f2.data(), f2.size() * 8); ds2.down(s2.data(), s2.size() * 8);
    double e2 = 0; for (size_t k = 0; k < f2.size(); ++k) e2 = std::max(e2, std::max(std::fabs(f2[k] - filt[k]), std::fabs(s2[k] - sm[k])));
    expect(!env->pending && e2 < 1e-10, "filter + smooth equal the fused call");
  }
  // ---- host mode: addresses of direct buffers ----------------------------------------------------------------
  jlongArray modelH = env->longs({d, p, T, 1, A(F.data()), 0, A(G.data()), 1, 0, 0});
  jlongArray paramsH = env->longs({A(V.data()), 0, A(W.data()), 0, A(m0.data()), 0, A(C0.data()), 0, 0, 0});
  jlongArray optsH = env->longs({0, DLM_MEM_HOST, 42, 0});
  CALL(filterSmooth, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), A(filtH.data()), A(smH.data()), A(status.data()));
  expect(!env->pending && filtH == filt && smH == sm, "host mode equals device mode bit for bit");
  {
    std::vector<double> ll(1), fq((size_t)(T + 1) * 2);
    CALL(loglik, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), A(ll.data()), A(status.data()));
    CALL(filter, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), A(filtH.data()), 0, A(fq.data()), A(status.data()));
    double want = 0.0;
    for (int t = 1; t <= T; ++t) { const double e = y[t - 1] - fq[2 * t], Q = fq[2 * t + 1]; want -= 0.5 * (std::log(2.0 * M_PI) + std::log(Q) + e * e / Q); }
    expect(!env->pending && std::fabs(ll[0] - want) < 1e-8 * std::fabs(want), "loglik equals the sum over the forecasts of filter");
    double eq = 0; for (int t = 1; t <= T; ++t) eq = std::max(eq, std::max(std::fabs(fq[2 * t] - fr[t][3]), std::fabs(fq[2 * t + 1] - fr[t][4])));
    expect(eq < 1e-11, "forecast records equal the golden CSV's f, Q columns");
  }
  {  // FFBS, statistics, backward sampling, pooling, conjugate step
    const int L = CALL(statsLen, env, nullptr, d, p, 0);
    expect(L == 2 * p + d + 1, "statsLen");
    std::vector<double> ws(filt.size()), th(T + 1), th2(T + 1), cond(filt.size()), st(L), st2(L), pooled(L), Vo(1), Wo(1);
    CALL(ffbs, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), 0, A(ws.data()), A(th.data()), A(cond.data()), A(st.data()), A(status.data()));
    expect(!env->pending && ws == filt && std::isfinite(th[T / 2]) && st[L - 1] == (double)T && st[1] == (double)T, "ffbs: workspace = filter records, statistics");
    CALL(backwardSample, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), A(filt.data()), 0, A(th2.data()), 0, A(st2.data()), A(status.data()));
    expect(!env->pending && th2 == th && st2 == st, "backwardSample from the filter records equals ffbs (same seed)");
    CALL(statsPool, env, nullptr, h, A(st.data()), 1, L, A(pooled.data()), optsH);
    expect(!env->pending && pooled == st, "statsPool over one series");
    CALL(dinvgammaStep, env, nullptr, h, d, p, 1, A(st.data()), 4.0, 6.0, 4.0, 9.0, 0, optsH, A(Vo.data()), A(Wo.data()));
    expect(!env->pending && Vo[0] > 0.0 && Wo[0] > 0.0, "dinvgammaStep");
  }
  {  // SVD filter / sampler: U D^2 U^T equals the Kalman filter's C
    std::vector<double> sv((size_t)(T + 1) * 3), th(T + 1), st(4);
    CALL(svdFilter, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), A(sv.data()), A(status.data()));
    double e = 0; for (int t = 0; t <= T; ++t) e = std::max(e, std::max(std::fabs(sv[3 * t] - filt[2 * t]), std::fabs(sv[3 * t + 1] * sv[3 * t + 1] * sv[3 * t + 2] * sv[3 * t + 2] - filt[2 * t + 1])));
    expect(!env->pending && e < 1e-8, "svdFilter equals the Kalman filter");
    CALL(svdFfbs, env, nullptr, h, modelH, paramsH, optsH, A(y.data()), 0, A(sv.data()), A(th.data()), A(st.data()), A(status.data()));
    expect(!env->pending && std::isfinite(th[3]) && st[3] == (double)T, "svdFfbs");
  }
  {  // simulation, AR(1) / OU lanes
    std::vector<double> x(T + 1), ys(T);
    CALL(simulate, env, nullptr, h, modelH, paramsH, optsH, A(x.data()), A(ys.data()), A(status.data()));
    expect(!env->pending && std::isfinite(x[T]) && std::isfinite(ys[T - 1]), "simulate");
    std::vector<double> v(T, 1.5), svp{0.8, 0.2, 0.4}, f2((size_t)(T + 1) * 2), th(T + 1), times(T);
    for (int t = 0; t < T; ++t) times[t] = 1.0 + 0.5 * t;
    CALL(ar1Ffbs, env, nullptr, h, 1, T, A(y.data()), A(v.data()), 0, A(svp.data()), 0, 0, optsH, A(f2.data()), A(th.data()), A(status.data()));
    expect(!env->pending && f2[0] == 0.2 && std::isfinite(th[T]), "ar1Ffbs (record 0 is the stationary prior mean mu)");
    CALL(ouFfbs, env, nullptr, h, 1, T, A(times.data()), A(y.data()), A(v.data()), 0, A(svp.data()), 0, 0, optsH, A(f2.data()), A(th.data()), A(status.data()));
    expect(!env->pending && std::isfinite(th[T]), "ouFfbs");
  }
  {  // RCCL wrapper (one rank), streams, memory info, packed records
    jbyteArray id = CALL(commUniqueId, env, nullptr);
    expect(id && id->bytes.size() == DLM_COMM_ID_BYTES, "commUniqueId");
    CALL(commInitRank, env, nullptr, h, 1, 0, id);
    Dev ds(env, h, 4 * 8);
    const double four[4] = {1, 2, 3, 4}; double back[4];
    ds.up(four, 32);
    CALL(gibbsSuffstatsAllreduce, env, nullptr, h, ds.p, 4);
    ds.down(back, 32);
    expect(!env->pending && back[0] == 1 && back[3] == 4, "commInitRank + gibbsSuffstatsAllreduce (one rank: identity)");
    CALL(engineWaitStream, env, nullptr, h, 0); CALL(streamWaitEngine, env, nullptr, h, 0); CALL(engineSetStream, env, nullptr, h, 0); CALL(engineSync, env, nullptr, h);
    expect(!env->pending, "stream ordering calls");
    jlongArray mi = CALL(deviceMemInfo, env, nullptr, h);
    expect(mi && mi->longs[1] > (jlong)100e9 && mi->longs[0] <= mi->longs[1], "deviceMemInfo (an MI355X has 288 GB)");
    expect(CALL(packedRecordDoubles, env, nullptr, 13) == 104 && CALL(packedRecordDoubles, env, nullptr, 2) == 6, "packedRecordDoubles");
    const double packed[6] = {1, 2, 10, 11, 12, 0}; double dense[6];   // d = 2: m = (1, 2), C = [[10, 11], [11, 12]]
    CALL(unpackRecords, env, nullptr, h, 2, 1, A(packed), optsH, A(dense));
    expect(!env->pending && dense[0] == 1 && dense[1] == 2 && dense[2] == 10 && dense[3] == 11 && dense[4] == 11 && dense[5] == 12, "unpackRecords");
  }
  {  // error mapping
    CALL(filterSmooth, env, nullptr, h, env->longs({1, 1, 1}), paramsH, optsH, A(y.data()), A(filtH.data()), A(smH.data()), 0);
    expect(env->pending && env->pendingClass == "java/lang/IllegalArgumentException", "short model array -> IllegalArgumentException");
    env->clearException();
    CALL(filterSmooth, env, nullptr, h, env->longs({d, p, 0, 1, A(F.data()), 0, A(G.data()), 1, 0, 0}), paramsH, optsH, A(y.data()), A(filtH.data()), A(smH.data()), 0);
    expect(env->pending && env->pendingClass == "java/lang/RuntimeException" && env->pendingMessage.find("T") != std::string::npos, "T = 0 -> RuntimeException with dlm_last_error");
    expect(CALL(lastError, env, nullptr, h)->str == env->pendingMessage, "lastError");
    env->clearException();
    CALL(address, env, nullptr, env->longs({1}));
    expect(env->pending, "address of a non-direct buffer throws");
    env->clearException();
  }
  CALL(engineDestroy, env, nullptr, h);
  std::printf(fails ? "JNI GLUE FAILED (%d)\n" : "JNI GLUE OK\n", fails);
  return fails != 0;
}
