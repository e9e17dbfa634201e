/* Minimal stand-in for the JDK's <jni.h>, written for this repo's CPU tests only.
 *
 * The build image has no JDK, so jni/dsgd_jni.cpp cannot be compiled against the real header here.  This file
 * declares exactly the slice of the JNI C++ API the shim uses (same names, same signatures as the JNI
 * specification, chapter 4) so that tests/test_jni_shim.py can (1) compile the shim, (2) compare its exported
 * Java_* symbols with the @native declarations of scala/NativeSVM.scala, and (3) drive a few entry points with a
 * recording fake JNIEnv.  It is NOT a JNI implementation: arrays are plain heap blocks with a length header.
 */
#ifndef DSGD_TEST_JNI_STUB_H
#define DSGD_TEST_JNI_STUB_H

#include <stdint.h>
#include <string.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
#define JNI_COMMIT 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

/* a fake array object: the tests build these through ctypes (length, element size, data pointer).  The reference
 * types mirror the REAL header's C++ class hierarchy (JNI specification, chapter 3: `class _jarray : public _jobject`,
 * `class _jlongArray : public _jarray`, ...): a plain jarray does NOT convert to jlongArray implicitly, and code that
 * relies on such a conversion must fail to compile here exactly as it would against a JDK. */
class _jobject {
 public:
  jsize length;
  jint elem_size;
  void* data;
};
class _jclass : public _jobject {};
class _jthrowable : public _jobject {};
class _jarray : public _jobject {};
class _jlongArray : public _jarray {};
class _jintArray : public _jarray {};
class _jfloatArray : public _jarray {};
class _jdoubleArray : public _jarray {};
class _jbyteArray : public _jarray {};
class _jobjectArray : public _jarray {};
typedef _jobject* jobject;
typedef _jclass* jclass;
typedef _jthrowable* jthrowable;
typedef _jarray* jarray;
typedef _jlongArray* jlongArray;
typedef _jintArray* jintArray;
typedef _jfloatArray* jfloatArray;
typedef _jdoubleArray* jdoubleArray;
typedef _jbyteArray* jbyteArray;
typedef _jobjectArray* jobjectArray;

/* what the fake environment records (read back by the tests) */
struct JniStubLog {
  char thrown_class[128];
  char thrown_message[512];
  int n_get;      /* Get<Type>ArrayElements calls            */
  int n_release;  /* Release<Type>ArrayElements calls        */
  int n_critical; /* GetPrimitiveArrayCritical calls: must stay 0 */
};

struct JNIEnv_ {
  JniStubLog log;

  jclass FindClass(const char* name) {
    strncpy(log.thrown_class, name, sizeof(log.thrown_class) - 1);
    return reinterpret_cast<jclass>(this);
  }
  jint ThrowNew(jclass, const char* msg) {
    strncpy(log.thrown_message, msg ? msg : "", sizeof(log.thrown_message) - 1);
    return 0;
  }
  jsize GetArrayLength(jarray a) { return a ? a->length : 0; }
  jobject GetObjectArrayElement(jobjectArray a, jsize i) { return static_cast<jobject*>(a->data)[i]; }
  void DeleteLocalRef(jobject) {}
  void* GetPrimitiveArrayCritical(jarray a, jboolean*) {
    log.n_critical++;
    return a ? a->data : nullptr;
  }
  void ReleasePrimitiveArrayCritical(jarray, void*, jint) {}

#define DSGD_STUB_ARRAY(T, Name, A)                                                         \
  T* Get##Name##ArrayElements(A a, jboolean* is_copy) {                                     \
    if (is_copy) *is_copy = 0;                                                              \
    log.n_get++;                                                                            \
    return static_cast<T*>(a->data);                                                        \
  }                                                                                         \
  void Release##Name##ArrayElements(A, T*, jint) { log.n_release++; }                       \
  void Get##Name##ArrayRegion(A a, jsize start, jsize len, T* buf) {                        \
    memcpy(buf, static_cast<T*>(a->data) + start, sizeof(T) * static_cast<size_t>(len));    \
  }                                                                                         \
  void Set##Name##ArrayRegion(A a, jsize start, jsize len, const T* buf) {                  \
    memcpy(static_cast<T*>(a->data) + start, buf, sizeof(T) * static_cast<size_t>(len));    \
  }
  DSGD_STUB_ARRAY(jlong, Long, jlongArray)
  DSGD_STUB_ARRAY(jint, Int, jintArray)
  DSGD_STUB_ARRAY(jfloat, Float, jfloatArray)
  DSGD_STUB_ARRAY(jdouble, Double, jdoubleArray)
  DSGD_STUB_ARRAY(jbyte, Byte, jbyteArray)
#undef DSGD_STUB_ARRAY
};
typedef JNIEnv_ JNIEnv;

#endif
