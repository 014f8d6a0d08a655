/* jni.h - MINIMAL stand-in for the JDK's header, for images without a JDK (this repo's build image has none).
 *
 * It exists so that csrc/jni/jni_query_stream.cc is compiled, linked and symbol-checked in every build instead of staying
 * source only.  It is NOT a general JNI header: only the types and the JNIEnv members the glue uses are declared.  What makes
 * the result loadable by a real JVM is the layout rule of the JNI specification ("JNI Functions", Interface Function Table):
 * JNIEnv points to a table of function pointers at FIXED indices - FindClass 6, ThrowNew 14, GetStringUTFChars 169,
 * ReleaseStringUTFChars 170, GetArrayLength 171, SetByteArrayRegion 208, ExceptionCheck 228 - and the wrappers below call
 * through exactly those slots.  genomicsdb_amd/build.py prefers $JAVA_HOME/include/jni.h whenever a JDK is present.
 */
#ifndef GDBAMD_STUB_JNI_H
#define GDBAMD_STUB_JNI_H
#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNIIMPORT
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1
#define JNI_OK 0
#define GDBAMD_STUB_JNI 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

#ifdef __cplusplus
class _jobject {};
class _jclass : public _jobject {};
class _jstring : public _jobject {};
class _jthrowable : public _jobject {};
class _jarray : public _jobject {};
class _jbyteArray : public _jarray {};
typedef _jobject* jobject;
typedef _jclass* jclass;
typedef _jstring* jstring;
typedef _jthrowable* jthrowable;
typedef _jarray* jarray;
typedef _jbyteArray* jbyteArray;

struct JNIEnv_;
typedef JNIEnv_ JNIEnv;
/* the interface function table: 4 reserved slots, then the functions in specification order (indices 4 .. 234) */
struct JNINativeInterface_ { void* slot[235]; };

struct JNIEnv_ {
  const JNINativeInterface_* functions;
  jclass FindClass(const char* name) { return ((jclass(*)(JNIEnv*, const char*))functions->slot[6])(this, name); }
  jint ThrowNew(jclass cls, const char* msg) { return ((jint(*)(JNIEnv*, jclass, const char*))functions->slot[14])(this, cls, msg); }
  const char* GetStringUTFChars(jstring s, jboolean* is_copy) { return ((const char* (*)(JNIEnv*, jstring, jboolean*))functions->slot[169])(this, s, is_copy); }
  void ReleaseStringUTFChars(jstring s, const char* chars) { ((void (*)(JNIEnv*, jstring, const char*))functions->slot[170])(this, s, chars); }
  jsize GetArrayLength(jarray a) { return ((jsize(*)(JNIEnv*, jarray))functions->slot[171])(this, a); }
  void SetByteArrayRegion(jbyteArray a, jsize start, jsize len, const jbyte* buf) {
    ((void (*)(JNIEnv*, jbyteArray, jsize, jsize, const jbyte*))functions->slot[208])(this, a, start, len, buf);
  }
  jboolean ExceptionCheck() { return ((jboolean(*)(JNIEnv*))functions->slot[228])(this); }
};
#else
#error "the stand-in jni.h serves the C++ glue only"
#endif
#endif
