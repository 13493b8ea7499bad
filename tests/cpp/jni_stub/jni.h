// NOT a JDK header.  A small FUNCTIONAL stand-in for the handful of JNI calls integration/jni/dlm_jni.cpp makes, so that
// the glue can be compiled and driven from a C++ test (tests/cpp/jni_glue_check.cpp) in an image without a JDK: arrays,
// strings and direct buffers are plain C++ objects, ThrowNew records the pending exception.  Type names and call
// signatures follow the JNI specification; nothing here is used by the product.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef int8_t jbyte;
typedef int32_t jsize;

struct _jstub_object {
  enum Kind { CLASS, STRING, LONGS, DOUBLES, BYTES, DIRECT } kind = CLASS;
  std::vector<jlong> longs; std::vector<jdouble> doubles; std::vector<jbyte> bytes; std::string str;
  void* addr = nullptr;   // DIRECT: the buffer's address
};
typedef _jstub_object* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jarray;
typedef jobject jlongArray;
typedef jobject jdoubleArray;
typedef jobject jbyteArray;

struct JNIEnv {
  bool pending = false;
  std::string pendingClass, pendingMessage;
  std::vector<std::unique_ptr<_jstub_object>> owned;

  jobject make(_jstub_object::Kind k) { owned.emplace_back(new _jstub_object()); owned.back()->kind = k; return owned.back().get(); }
  void clearException() { pending = false; pendingClass.clear(); pendingMessage.clear(); }

  jclass FindClass(const char* name) { jobject o = make(_jstub_object::CLASS); o->str = name; return o; }
  jint ThrowNew(jclass cls, const char* msg) { pending = true; pendingClass = cls->str; pendingMessage = msg ? msg : ""; return 0; }
  jstring NewStringUTF(const char* s) { jobject o = make(_jstub_object::STRING); o->str = s ? s : ""; return o; }
  jsize GetArrayLength(jarray a) {
    switch (a->kind) {
      case _jstub_object::LONGS: return (jsize)a->longs.size();
      case _jstub_object::DOUBLES: return (jsize)a->doubles.size();
      case _jstub_object::BYTES: return (jsize)a->bytes.size();
      default: return 0;
    }
  }
  jlongArray NewLongArray(jsize n) { jobject o = make(_jstub_object::LONGS); o->longs.assign((size_t)n, 0); return o; }
  jdoubleArray NewDoubleArray(jsize n) { jobject o = make(_jstub_object::DOUBLES); o->doubles.assign((size_t)n, 0.0); return o; }
  jbyteArray NewByteArray(jsize n) { jobject o = make(_jstub_object::BYTES); o->bytes.assign((size_t)n, 0); return o; }
  void GetLongArrayRegion(jlongArray a, jsize start, jsize len, jlong* buf) { std::memcpy(buf, a->longs.data() + start, (size_t)len * sizeof(jlong)); }
  void SetLongArrayRegion(jlongArray a, jsize start, jsize len, const jlong* buf) { std::memcpy(a->longs.data() + start, buf, (size_t)len * sizeof(jlong)); }
  void SetDoubleArrayRegion(jdoubleArray a, jsize start, jsize len, const jdouble* buf) { std::memcpy(a->doubles.data() + start, buf, (size_t)len * sizeof(jdouble)); }
  void GetByteArrayRegion(jbyteArray a, jsize start, jsize len, jbyte* buf) { std::memcpy(buf, a->bytes.data() + start, (size_t)len); }
  void SetByteArrayRegion(jbyteArray a, jsize start, jsize len, const jbyte* buf) { std::memcpy(a->bytes.data() + start, buf, (size_t)len); }
  void* GetDirectBufferAddress(jobject b) { return b->kind == _jstub_object::DIRECT ? b->addr : nullptr; }

  // test-side helpers (no JNI counterpart)
  jlongArray longs(std::initializer_list<jlong> v) { jobject o = make(_jstub_object::LONGS); o->longs.assign(v); return o; }
  jobject direct(void* p) { jobject o = make(_jstub_object::DIRECT); o->addr = p; return o; }
};
