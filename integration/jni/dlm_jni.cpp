// JNI glue between the Scala shim (integration/scala/Batched.scala) and the C ABI (include/dlm_engine.h): one native
// method of class com.github.jonnylaw.dlm.gpu.Native per export, nothing else.
//
// Neither a JDK nor <jni.h> exists in the build container or on the GPU box.  The file is therefore checked two ways:
// tests/cpp/jni_glue_check.cpp compiles it against a small functional stand-in for the JNI calls it uses
// (tests/cpp/jni_stub/jni.h) and drives every entry point on the GPU; tests/test_integration_sources.py checks that every
// dlm_* export is bound here and declared @native in Batched.scala.  Build where a JDK is present:
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude integration/jni/dlm_jni.cpp
//       -Lbayesian_dlms_amd -ldlm_engine -o libdlm_jni.so
//
// Conventions
//   * Every bulk array crosses as a raw ADDRESS (jlong): a device pointer from bufferAlloc (opts.mem = DLM_MEM_DEVICE, the
//     measured path -- results stay in HBM and are fetched with bufferDownload) or the address of a direct
//     java.nio buffer obtained with Native.address (opts.mem = DLM_MEM_HOST).  0 = NULL.  Nothing is copied through
//     Get<Primitive>ArrayElements.
//   * The three descriptors cross as small long[] arrays in the field order of the C structs:
//       model  = {d, p, T, N, F, fStride, G, nG, gIndex, dt}                                   (dlm_model_desc)
//       params = {V, vStride, W, wStride, m0, m0Stride, C0, c0Stride, vTStride, wTStride}      (dlm_params_desc)
//       opts   = {flags, mem, seed, seriesOffset}                                              (dlm_options)
//   * A non-zero return code becomes a RuntimeException carrying dlm_last_error (Breeze throws on singular systems, the
//     engine reports numerical trouble per series in `status` instead; argument and HIP errors throw).
#if __has_include(<jni.h>)
#include <jni.h>
#include <cstdint>
#include <cstring>
#include "dlm_engine.h"

namespace {
template <class T> T* ptr(jlong a) { return reinterpret_cast<T*>(static_cast<uintptr_t>(a)); }
dlm_engine* eng(jlong h) { return ptr<dlm_engine>(h); }

// returns true when an exception is now pending (the caller returns at once)
bool throw_if(JNIEnv* env, dlm_engine* e, int rc) {
  if (rc == DLM_OK) return false;
  jclass cls = env->FindClass("java/lang/RuntimeException");
  if (cls) env->ThrowNew(cls, e ? dlm_last_error(e) : "dlm engine error");
  return true;
}
bool throw_arg(JNIEnv* env, const char* msg) {
  jclass cls = env->FindClass("java/lang/IllegalArgumentException");
  if (cls) env->ThrowNew(cls, msg);
  return true;
}

struct Desc { dlm_model_desc m; dlm_params_desc q; dlm_options o; };

bool read_opts(JNIEnv* env, jlongArray opts, dlm_options& o) {
  if (!opts || env->GetArrayLength(opts) != 4) return !throw_arg(env, "opts must be long[4] = {flags, mem, seed, seriesOffset}");
  jlong v[4];
  env->GetLongArrayRegion(opts, 0, 4, v);
  o.flags = static_cast<uint32_t>(v[0]); o.mem = static_cast<int32_t>(v[1]);
  o.seed = static_cast<uint64_t>(v[2]); o.series_offset = static_cast<uint64_t>(v[3]);
  return true;
}
bool read_desc(JNIEnv* env, jlongArray model, jlongArray params, jlongArray opts, Desc& d) {
  std::memset(&d, 0, sizeof d);
  if (!model || env->GetArrayLength(model) != 10) return !throw_arg(env, "model must be long[10] = {d, p, T, N, F, fStride, G, nG, gIndex, dt}");
  if (!params || env->GetArrayLength(params) != 10) return !throw_arg(env, "params must be long[10] = {V, vStride, W, wStride, m0, m0Stride, C0, c0Stride, vTStride, wTStride}");
  jlong m[10], q[10];
  env->GetLongArrayRegion(model, 0, 10, m);
  env->GetLongArrayRegion(params, 0, 10, q);
  d.m.d = static_cast<int32_t>(m[0]); d.m.p = static_cast<int32_t>(m[1]); d.m.T = static_cast<int32_t>(m[2]); d.m.N = static_cast<int32_t>(m[3]);
  d.m.F = ptr<const double>(m[4]); d.m.f_stride = m[5]; d.m.G = ptr<const double>(m[6]); d.m.n_g = static_cast<int32_t>(m[7]);
  d.m.g_index = ptr<const int32_t>(m[8]); d.m.dt = ptr<const double>(m[9]);
  d.q.V = ptr<const double>(q[0]); d.q.v_stride = q[1]; d.q.W = ptr<const double>(q[2]); d.q.w_stride = q[3];
  d.q.m0 = ptr<const double>(q[4]); d.q.m0_stride = q[5]; d.q.C0 = ptr<const double>(q[6]); d.q.c0_stride = q[7];
  d.q.v_tstride = q[8]; d.q.w_tstride = q[9];
  return read_opts(env, opts, d.o);
}
}  // namespace

extern "C" {

// ---- lifecycle --------------------------------------------------------------------------------------------------
JNIEXPORT jlong JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_engineCreate(JNIEnv* env, jobject, jint device) {
  dlm_engine* e = nullptr;
  if (dlm_engine_create(device, &e) != DLM_OK) {
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, "dlm_engine_create failed: no usable HIP device");
    return 0;
  }
  return static_cast<jlong>(reinterpret_cast<uintptr_t>(e));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_engineDestroy(JNIEnv*, jobject, jlong h) { dlm_engine_destroy(eng(h)); }
JNIEXPORT jstring JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_lastError(JNIEnv* env, jobject, jlong h) { return env->NewStringUTF(dlm_last_error(eng(h))); }
JNIEXPORT jstring JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_version(JNIEnv* env, jobject) { return env->NewStringUTF(dlm_version()); }
JNIEXPORT jstring JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_lastVariant(JNIEnv* env, jobject, jlong h) { return env->NewStringUTF(dlm_last_variant(eng(h))); }
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_engineSetStream(JNIEnv* env, jobject, jlong h, jlong stream) {
  throw_if(env, eng(h), dlm_engine_set_stream(eng(h), ptr<void>(stream)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_engineSync(JNIEnv* env, jobject, jlong h) { throw_if(env, eng(h), dlm_engine_sync(eng(h))); }
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_engineWaitStream(JNIEnv* env, jobject, jlong h, jlong stream) {
  throw_if(env, eng(h), dlm_engine_wait_stream(eng(h), ptr<void>(stream)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_streamWaitEngine(JNIEnv* env, jobject, jlong h, jlong stream) {
  throw_if(env, eng(h), dlm_stream_wait_engine(eng(h), ptr<void>(stream)));
}
// {forward ms, backward ms} of the last fused call
JNIEXPORT jdoubleArray JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_lastTiming(JNIEnv* env, jobject, jlong h) {
  double ms[2] = {0.0, 0.0};
  if (throw_if(env, eng(h), dlm_last_timing(eng(h), ms))) return nullptr;
  jdoubleArray out = env->NewDoubleArray(2);
  if (out) env->SetDoubleArrayRegion(out, 0, 2, ms);
  return out;
}
// {forward steady steps, backward steady steps, series on the shared-covariance path, series on their own recursion} of the last
// call made with DLM_OPT_COUNT_STEPS
JNIEXPORT jlongArray JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_lastCounters(JNIEnv* env, jobject, jlong h) {
  uint64_t c[4] = {0, 0, 0, 0};
  if (throw_if(env, eng(h), dlm_last_counters(eng(h), c))) return nullptr;
  jlong v[4] = {static_cast<jlong>(c[0]), static_cast<jlong>(c[1]), static_cast<jlong>(c[2]), static_cast<jlong>(c[3])};
  jlongArray out = env->NewLongArray(4);
  if (out) env->SetLongArrayRegion(out, 0, 4, v);
  return out;
}

// ---- engine-owned device buffers ----------------------------------------------------------------------------------
// address of a direct java.nio buffer (host mode, and the host side of upload / download)
JNIEXPORT jlong JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_address(JNIEnv* env, jobject, jobject directBuffer) {
  void* a = directBuffer ? env->GetDirectBufferAddress(directBuffer) : nullptr;
  if (directBuffer && !a) { throw_arg(env, "not a direct buffer"); return 0; }
  return static_cast<jlong>(reinterpret_cast<uintptr_t>(a));
}
JNIEXPORT jlong JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_bufferAlloc(JNIEnv* env, jobject, jlong h, jlong bytes) {
  void* p = nullptr;
  if (bytes <= 0) { throw_arg(env, "bufferAlloc: bytes must be positive"); return 0; }
  if (throw_if(env, eng(h), dlm_buffer_alloc(eng(h), static_cast<uint64_t>(bytes), &p))) return 0;
  return static_cast<jlong>(reinterpret_cast<uintptr_t>(p));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_bufferFree(JNIEnv* env, jobject, jlong h, jlong dev) {
  throw_if(env, eng(h), dlm_buffer_free(eng(h), ptr<void>(dev)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_bufferUpload(JNIEnv* env, jobject, jlong h, jlong dstDev, jlong dstOffset, jlong srcHost, jlong bytes) {
  if (dstOffset < 0 || bytes < 0) { throw_arg(env, "bufferUpload: negative offset or size"); return; }
  throw_if(env, eng(h), dlm_buffer_upload(eng(h), ptr<void>(dstDev), static_cast<uint64_t>(dstOffset), ptr<const void>(srcHost), static_cast<uint64_t>(bytes)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_bufferDownload(JNIEnv* env, jobject, jlong h, jlong srcDev, jlong srcOffset, jlong dstHost, jlong bytes) {
  if (srcOffset < 0 || bytes < 0) { throw_arg(env, "bufferDownload: negative offset or size"); return; }
  throw_if(env, eng(h), dlm_buffer_download(eng(h), ptr<const void>(srcDev), static_cast<uint64_t>(srcOffset), ptr<void>(dstHost), static_cast<uint64_t>(bytes)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_bufferFill(JNIEnv* env, jobject, jlong h, jlong dstDev, jlong dstOffset, jint byteValue, jlong bytes) {
  if (dstOffset < 0 || bytes < 0) { throw_arg(env, "bufferFill: negative offset or size"); return; }
  throw_if(env, eng(h), dlm_buffer_fill(eng(h), ptr<void>(dstDev), static_cast<uint64_t>(dstOffset), byteValue, static_cast<uint64_t>(bytes)));
}
// {free bytes, total bytes}
JNIEXPORT jlongArray JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_deviceMemInfo(JNIEnv* env, jobject, jlong h) {
  uint64_t f = 0, t = 0;
  if (throw_if(env, eng(h), dlm_device_mem_info(eng(h), &f, &t))) return nullptr;
  jlong v[2] = {static_cast<jlong>(f), static_cast<jlong>(t)};
  jlongArray out = env->NewLongArray(2);
  if (out) env->SetLongArrayRegion(out, 0, 2, v);
  return out;
}
JNIEXPORT jint JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_packedRecordDoubles(JNIEnv*, jobject, jint d) { return dlm_packed_record_doubles(d); }
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_unpackRecords(JNIEnv* env, jobject, jlong h, jint d, jlong count, jlong packed, jlongArray opts, jlong dense) {
  dlm_options o{};
  if (!read_opts(env, opts, o)) return;
  throw_if(env, eng(h), dlm_unpack_records(eng(h), d, count, ptr<const double>(packed), &o, ptr<double>(dense)));
}

// ---- Kalman filter: KalmanFilter(advanceState(p, mod.g)).filter / filterDlm (KalmanFilter.scala:262-294) ---------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_filter(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                      jlong y, jlong filt, jlong prior, jlong fq, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_filter_batch(eng(h), &d.m, &d.q, ptr<const double>(y), &d.o, ptr<double>(filt), ptr<double>(prior), ptr<double>(fq), ptr<int32_t>(status)));
}
// ---- RTS smoother: Smoothing.backwardsSmoother (Smoothing.scala:31-64) --------------------------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_smooth(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                      jlong filt, jlong smooth, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_smooth_batch(eng(h), &d.m, &d.q, ptr<const double>(filt), &d.o, ptr<double>(smooth), ptr<int32_t>(status)));
}
// ---- fused filter + smoother (the metric path) --------------------------------------------------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_filterSmooth(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                            jlong y, jlong filt, jlong smooth, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_filter_smooth_batch(eng(h), &d.m, &d.q, ptr<const double>(y), &d.o, ptr<double>(filt), ptr<double>(smooth), ptr<int32_t>(status)));
}
// ---- log-likelihood: sum of KalmanFilter.conditionalLikelihood (KalmanFilter.scala:138-153) --------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_loglik(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                      jlong y, jlong loglik, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_loglik_batch(eng(h), &d.m, &d.q, ptr<const double>(y), &d.o, ptr<double>(loglik), ptr<int32_t>(status)));
}
// ---- Dlm.simulateRegular (Dlm.scala:245-292) -------------------------------------------------------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_simulate(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                        jlong x, jlong y, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_simulate_batch(eng(h), &d.m, &d.q, &d.o, ptr<double>(x), ptr<double>(y), ptr<int32_t>(status)));
}
// ---- FFBS + Gibbs sufficient statistics: Smoothing.ffbsDlm (Smoothing.scala:173-180), sums of Gibbs.scala:23-78 ---------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_ffbs(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                    jlong y, jlong z, jlong filtWs, jlong theta, jlong cond, jlong stats, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_ffbs_batch(eng(h), &d.m, &d.q, ptr<const double>(y), ptr<const double>(z), &d.o, ptr<double>(filtWs), ptr<double>(theta),
                                       ptr<double>(cond), ptr<double>(stats), ptr<int32_t>(status)));
}
JNIEXPORT jint JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_statsLen(JNIEnv*, jobject, jint d, jint p, jint flags) {
  return dlm_stats_len(d, p, static_cast<uint32_t>(flags));
}
// ---- Smoothing.sampleDlm (Smoothing.scala:164-165): backward sampling from existing filter records ---------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_backwardSample(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                              jlong y, jlong filt, jlong z, jlong theta, jlong cond, jlong stats, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_backward_sample_batch(eng(h), &d.m, &d.q, ptr<const double>(y), ptr<const double>(filt), ptr<const double>(z), &d.o, ptr<double>(theta),
                                                  ptr<double>(cond), ptr<double>(stats), ptr<int32_t>(status)));
}
// ---- SvdFilter.filterDlm (SvdFilter.scala:158-161), SvdSampler.ffbsDlm (SvdSampler.scala:79-82) ---------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_svdFilter(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                         jlong y, jlong svdRec, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_svd_filter_batch(eng(h), &d.m, &d.q, ptr<const double>(y), &d.o, ptr<double>(svdRec), ptr<int32_t>(status)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_svdFfbs(JNIEnv* env, jobject, jlong h, jlongArray model, jlongArray params, jlongArray opts,
                                                                       jlong y, jlong z, jlong svdWs, jlong theta, jlong stats, jlong status) {
  Desc d;
  if (!read_desc(env, model, params, opts, d)) return;
  throw_if(env, eng(h), dlm_svd_ffbs_batch(eng(h), &d.m, &d.q, ptr<const double>(y), ptr<const double>(z), &d.o, ptr<double>(svdWs), ptr<double>(theta),
                                           ptr<double>(stats), ptr<int32_t>(status)));
}
// ---- GibbsSampling.dinvGammaStep on the device (Gibbs.scala:23-78, :134-151) -------------------------------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_dinvgammaStep(JNIEnv* env, jobject, jlong h, jint d, jint p, jint n, jlong stats, jdouble alphaV, jdouble betaV,
                                                                             jdouble alphaW, jdouble betaW, jlong iteration, jlongArray opts, jlong vOut, jlong wOut) {
  dlm_options o{};
  if (!read_opts(env, opts, o)) return;
  throw_if(env, eng(h), dlm_dinvgamma_step_batch(eng(h), d, p, n, ptr<const double>(stats), alphaV, betaV, alphaW, betaW, static_cast<uint64_t>(iteration), &o,
                                                 ptr<double>(vOut), ptr<double>(wOut)));
}
// ---- scalar AR(1) / OU FFBS: FilterAr (FilterAr.scala:15-82), FilterOu (FilterOu.scala:7-79) ------------------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_ar1Ffbs(JNIEnv* env, jobject, jlong h, jint n, jint t, jlong y, jlong v, jlong vStride, jlong sv, jlong svStride,
                                                                       jlong z, jlongArray opts, jlong filt, jlong theta, jlong status) {
  dlm_options o{};
  if (!read_opts(env, opts, o)) return;
  throw_if(env, eng(h), dlm_ar1_ffbs_batch(eng(h), n, t, ptr<const double>(y), ptr<const double>(v), vStride, ptr<const double>(sv), svStride, ptr<const double>(z), &o,
                                           ptr<double>(filt), ptr<double>(theta), ptr<int32_t>(status)));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_ouFfbs(JNIEnv* env, jobject, jlong h, jint n, jint t, jlong times, jlong y, jlong v, jlong vStride, jlong sv,
                                                                      jlong svStride, jlong z, jlongArray opts, jlong filt, jlong theta, jlong status) {
  dlm_options o{};
  if (!read_opts(env, opts, o)) return;
  throw_if(env, eng(h), dlm_ou_ffbs_batch(eng(h), n, t, ptr<const double>(times), ptr<const double>(y), ptr<const double>(v), vStride, ptr<const double>(sv), svStride,
                                          ptr<const double>(z), &o, ptr<double>(filt), ptr<double>(theta), ptr<int32_t>(status)));
}
// ---- pooled-parameter Gibbs: reduce over series, then over GPUs (RCCL) ---------------------------------------------------------------
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_statsPool(JNIEnv* env, jobject, jlong h, jlong stats, jint n, jint l, jlong pooled, jlongArray opts) {
  dlm_options o{};
  if (!read_opts(env, opts, o)) return;
  throw_if(env, eng(h), dlm_stats_pool(eng(h), ptr<const double>(stats), n, l, ptr<double>(pooled), &o));
}
JNIEXPORT jbyteArray JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_commUniqueId(JNIEnv* env, jobject) {
  uint8_t id[DLM_COMM_ID_BYTES];
  if (dlm_comm_unique_id(id) != DLM_OK) {
    jclass cls = env->FindClass("java/lang/RuntimeException");
    if (cls) env->ThrowNew(cls, "dlm_comm_unique_id failed (RCCL)");
    return nullptr;
  }
  jbyteArray out = env->NewByteArray(DLM_COMM_ID_BYTES);
  if (out) env->SetByteArrayRegion(out, 0, DLM_COMM_ID_BYTES, reinterpret_cast<const jbyte*>(id));
  return out;
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_commInitRank(JNIEnv* env, jobject, jlong h, jint nranks, jint rank, jbyteArray id) {
  if (!id || env->GetArrayLength(id) != DLM_COMM_ID_BYTES) { throw_arg(env, "commInitRank: id must be the byte[128] of commUniqueId"); return; }
  uint8_t raw[DLM_COMM_ID_BYTES];
  env->GetByteArrayRegion(id, 0, DLM_COMM_ID_BYTES, reinterpret_cast<jbyte*>(raw));
  throw_if(env, eng(h), dlm_comm_init_rank(eng(h), nranks, rank, raw));
}
JNIEXPORT void JNICALL Java_com_github_jonnylaw_dlm_gpu_Native_gibbsSuffstatsAllreduce(JNIEnv* env, jobject, jlong h, jlong statsDev, jlong count) {
  throw_if(env, eng(h), dlm_gibbs_suffstats_allreduce(eng(h), ptr<double>(statsDev), count));
}

}  // extern "C"
#endif  // __has_include(<jni.h>)
