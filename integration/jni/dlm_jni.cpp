// JNI glue between the Scala shim (integration/scala/Batched.scala) and the C ABI
// (include/dlm_engine.h).  SOURCE ONLY in this repository: neither a JDK nor <jni.h> exists in
// the build container or on the GPU box, so the file compiles to nothing there.  Build where a
// JDK is present:
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
//       integration/jni/dlm_jni.cpp -Lbayesian_dlms_amd -ldlm_engine -o libdlm_jni.so
//
// All bulk arrays cross as direct java.nio.DoubleBuffer / IntBuffer (GetDirectBufferAddress);
// nothing is copied through Get<Primitive>ArrayElements.  The engine is called in DLM_MEM_HOST
// mode (it stages H2D/D2H itself) unless the shim passes device addresses obtained elsewhere.
#if __has_include(<jni.h>)
#include <jni.h>
#include "dlm_engine.h"

namespace {
template <class T> T* addr(JNIEnv* env, jobject buf) {
  return buf ? static_cast<T*>(env->GetDirectBufferAddress(buf)) : nullptr;
}
void throw_if(JNIEnv* env, dlm_engine* e, int rc) {
  if (rc == DLM_OK) return;
  jclass cls = env->FindClass("java/lang/RuntimeException");
  env->ThrowNew(cls, e ? dlm_last_error(e) : "dlm engine error");
}
dlm_model_desc model(JNIEnv* env, jint d, jint p, jint T, jint N, jobject F, jlong fStride, jobject G,
                     jint nG, jobject gIndex, jobject dt) {
  dlm_model_desc m{};
  m.d = d; m.p = p; m.T = T; m.N = N;
  m.F = addr<double>(env, F); m.f_stride = fStride;
  m.G = addr<double>(env, G); m.n_g = nG;
  m.g_index = addr<int32_t>(env, gIndex); m.dt = addr<double>(env, dt);
  return m;
}
dlm_params_desc params(JNIEnv* env, jobject V, jlong vs, jobject W, jlong ws, jobject m0, jlong ms,
                       jobject C0, jlong cs) {
  dlm_params_desc q{};
  q.V = addr<double>(env, V); q.v_stride = vs; q.W = addr<double>(env, W); q.w_stride = ws;
  q.m0 = addr<double>(env, m0); q.m0_stride = ms; q.C0 = addr<double>(env, C0); q.c0_stride = cs;
  return q;
}
}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_create(JNIEnv* env, jclass, jint device) {
  dlm_engine* e = nullptr;
  throw_if(env, nullptr, dlm_engine_create(device, &e));
  return reinterpret_cast<jlong>(e);
}

JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_destroy(JNIEnv*, jclass, jlong h) {
  dlm_engine_destroy(reinterpret_cast<dlm_engine*>(h));
}

// replaces KalmanFilter(...).filter + Smoothing.backwardsSmoother  (KalmanFilter.scala:262-294, Smoothing.scala:57-64)
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_filterSmooth(
    JNIEnv* env, jclass, jlong h, jint d, jint p, jint T, jint N, jobject F, jlong fStride, jobject G, jint nG,
    jobject gIndex, jobject dt, jobject V, jlong vs, jobject W, jlong ws, jobject m0, jlong ms, jobject C0, jlong cs,
    jobject y, jint flags, jobject filt, jobject smooth, jobject status) {
  dlm_engine* e = reinterpret_cast<dlm_engine*>(h);
  dlm_model_desc m = model(env, d, p, T, N, F, fStride, G, nG, gIndex, dt);
  dlm_params_desc q = params(env, V, vs, W, ws, m0, ms, C0, cs);
  dlm_options o{static_cast<uint32_t>(flags), DLM_MEM_HOST, 0, 0};
  throw_if(env, e, dlm_filter_smooth_batch(e, &m, &q, addr<double>(env, y), &o, addr<double>(env, filt),
                                           addr<double>(env, smooth), addr<int32_t>(env, status)));
}

// sum over each series of KalmanFilter.conditionalLikelihood  (KalmanFilter.scala:138-153)
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_logLikelihood(
    JNIEnv* env, jclass, jlong h, jint d, jint p, jint T, jint N, jobject F, jlong fStride, jobject G, jint nG,
    jobject gIndex, jobject dt, jobject V, jlong vs, jobject W, jlong ws, jobject m0, jlong ms, jobject C0, jlong cs,
    jobject y, jint flags, jobject loglik, jobject status) {
  dlm_engine* e = reinterpret_cast<dlm_engine*>(h);
  dlm_model_desc m = model(env, d, p, T, N, F, fStride, G, nG, gIndex, dt);
  dlm_params_desc q = params(env, V, vs, W, ws, m0, ms, C0, cs);
  dlm_options o{static_cast<uint32_t>(flags), DLM_MEM_HOST, 0, 0};
  throw_if(env, e, dlm_loglik_batch(e, &m, &q, addr<double>(env, y), &o, addr<double>(env, loglik), addr<int32_t>(env, status)));
}

// replaces KalmanFilter.filterDlm  (KalmanFilter.scala:291-294)
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_filter(
    JNIEnv* env, jclass, jlong h, jint d, jint p, jint T, jint N, jobject F, jlong fStride, jobject G, jint nG,
    jobject gIndex, jobject dt, jobject V, jlong vs, jobject W, jlong ws, jobject m0, jlong ms, jobject C0, jlong cs,
    jobject y, jint flags, jobject filt, jobject prior, jobject fq, jobject status) {
  dlm_engine* e = reinterpret_cast<dlm_engine*>(h);
  dlm_model_desc m = model(env, d, p, T, N, F, fStride, G, nG, gIndex, dt);
  dlm_params_desc q = params(env, V, vs, W, ws, m0, ms, C0, cs);
  dlm_options o{static_cast<uint32_t>(flags), DLM_MEM_HOST, 0, 0};
  throw_if(env, e, dlm_filter_batch(e, &m, &q, addr<double>(env, y), &o, addr<double>(env, filt),
                                    addr<double>(env, prior), addr<double>(env, fq), addr<int32_t>(env, status)));
}

// replaces Smoothing.ffbsDlm and the sums of Gibbs.scala:23-78 / GibbsWishart.scala:16-35
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_ffbs(
    JNIEnv* env, jclass, jlong h, jint d, jint p, jint T, jint N, jobject F, jlong fStride, jobject G, jint nG,
    jobject gIndex, jobject dt, jobject V, jlong vs, jobject W, jlong ws, jobject m0, jlong ms, jobject C0, jlong cs,
    jobject y, jint flags, jlong seed, jlong seriesOffset, jobject filtWs, jobject theta, jobject stats,
    jobject status) {
  dlm_engine* e = reinterpret_cast<dlm_engine*>(h);
  dlm_model_desc m = model(env, d, p, T, N, F, fStride, G, nG, gIndex, dt);
  dlm_params_desc q = params(env, V, vs, W, ws, m0, ms, C0, cs);
  dlm_options o{static_cast<uint32_t>(flags), DLM_MEM_HOST, static_cast<uint64_t>(seed),
                static_cast<uint64_t>(seriesOffset)};
  throw_if(env, e, dlm_ffbs_batch(e, &m, &q, addr<double>(env, y), nullptr, &o, addr<double>(env, filtWs),
                                  addr<double>(env, theta), nullptr, addr<double>(env, stats),
                                  addr<int32_t>(env, status)));
}

}  // extern "C"
#endif  // __has_include(<jni.h>)
