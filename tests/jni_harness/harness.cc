// harness.cc - TEST infrastructure: drives the JNI glue (genomicsdb_amd/csrc/jni/jni_query_stream.cc) the way the JVM would,
// without a JVM: a JNIEnv whose interface function table holds plain C++ stand-ins for the handful of functions the glue uses
// (at the specification's table indices, as csrc/jni/stub/jni.h declares them), Java strings / byte arrays as small C++ objects.
// Call order = GenomicsDBLibLoader.loadLibrary() + GenomicsDBQueryStream (reference
// src/main/java/com/intel/genomicsdb/reader/GenomicsDBQueryStream.java): OneTimeInitialize, Init, Read ... until 0, Close.
#include <jni.h>

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef GDBAMD_STUB_JNI
#error "the harness is built against csrc/jni/stub/jni.h"
#endif

extern "C" {
jint Java_com_intel_genomicsdb_GenomicsDBLibLoader_jniGenomicsDBOneTimeInitialize(JNIEnv*, jclass);
jlong Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBInit(JNIEnv*, jobject, jstring, jstring, jstring, jint, jint, jint, jlong, jlong, jboolean,
                                                                                 jboolean, jboolean, jboolean);
jlong Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBClose(JNIEnv*, jobject, jlong);
jlong Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBGetNumBytesAvailable(JNIEnv*, jobject, jlong);
jbyte Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBReadNextByte(JNIEnv*, jobject, jlong);
jint Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBRead(JNIEnv*, jobject, jlong, jbyteArray, jint, jint);
jlong Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBSkip(JNIEnv*, jobject, jlong, jlong);
}

namespace {
struct FakeString : _jstring { std::string s; };
struct FakeBytes : _jbyteArray { std::vector<jbyte> v; };
struct FakeClass : _jclass { std::string name; };
struct FakeEnv : JNIEnv_ {
  JNINativeInterface_ table;
  bool exception_pending = false;
  std::string exception_text;
  FakeClass io_exception;
};
jclass f_FindClass(JNIEnv* e, const char* name) { FakeEnv* fe = static_cast<FakeEnv*>(e); fe->io_exception.name = name; return &fe->io_exception; }
jint f_ThrowNew(JNIEnv* e, jclass, const char* msg) { FakeEnv* fe = static_cast<FakeEnv*>(e); fe->exception_pending = true; fe->exception_text = msg ? msg : ""; return 0; }
const char* f_GetStringUTFChars(JNIEnv*, jstring s, jboolean* is_copy) { if (is_copy) *is_copy = JNI_FALSE; return static_cast<FakeString*>(s)->s.c_str(); }
void f_ReleaseStringUTFChars(JNIEnv*, jstring, const char*) {}
jsize f_GetArrayLength(JNIEnv*, jarray a) { return (jsize) static_cast<FakeBytes*>(static_cast<_jbyteArray*>(a))->v.size(); }
void f_SetByteArrayRegion(JNIEnv* e, jbyteArray a, jsize start, jsize len, const jbyte* buf) {
  FakeBytes* b = static_cast<FakeBytes*>(a);
  if (start < 0 || len < 0 || (size_t)start + (size_t)len > b->v.size()) { f_ThrowNew(e, nullptr, "ArrayIndexOutOfBoundsException"); return; }
  memcpy(b->v.data() + start, buf, (size_t)len);
}
jboolean f_ExceptionCheck(JNIEnv* e) { return static_cast<FakeEnv*>(e)->exception_pending ? JNI_TRUE : JNI_FALSE; }
}  // namespace

// returns 0 and the whole stream in *out (malloc'ed), or -1 with the pending "Java exception" text in err
extern "C" int jni_harness_read_stream(const char* loader_json, const char* query_json, const char* chr, int start, int end, int is_bcf, int array_len, int use_read_next_byte_first,
                                       unsigned char** out, unsigned long long* nbytes, char* err, unsigned long long err_cap) {
  FakeEnv env;
  memset(&env.table, 0, sizeof(env.table));
  env.table.slot[6] = (void*)f_FindClass; env.table.slot[14] = (void*)f_ThrowNew; env.table.slot[169] = (void*)f_GetStringUTFChars;
  env.table.slot[170] = (void*)f_ReleaseStringUTFChars; env.table.slot[171] = (void*)f_GetArrayLength; env.table.slot[208] = (void*)f_SetByteArrayRegion;
  env.table.slot[228] = (void*)f_ExceptionCheck;
  env.functions = &env.table;
  auto fail = [&](const std::string& why) { if (err && err_cap) { strncpy(err, why.c_str(), (size_t)err_cap - 1); err[err_cap - 1] = 0; } return -1; };
  if (Java_com_intel_genomicsdb_GenomicsDBLibLoader_jniGenomicsDBOneTimeInitialize(&env, nullptr) != 0) return fail("OneTimeInitialize != 0");
  FakeString l, q, c;
  l.s = loader_json ? loader_json : ""; q.s = query_json; c.s = chr ? chr : "";
  const jlong h = Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBInit(&env, nullptr, &l, &q, &c, start, end, 0, 1048576, 1048576, is_bcf ? 1 : 0, 0, 0, 1);
  if (env.exception_pending || !h) return fail("Init: " + env.exception_text);
  if (Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBGetNumBytesAvailable(&env, nullptr, h) != 1048576) return fail("GetNumBytesAvailable");
  std::vector<unsigned char> all;
  if (use_read_next_byte_first) all.push_back((unsigned char)Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBReadNextByte(&env, nullptr, h));
  FakeBytes arr;
  arr.v.resize((size_t)array_len + 7);
  for (;;) {
    const jint got = Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBRead(&env, nullptr, h, &arr, 7, array_len);   // (offset 7: a read into the middle of the array)
    if (env.exception_pending) { Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBClose(&env, nullptr, h); return fail("Read: " + env.exception_text); }
    if (got <= 0) break;
    all.insert(all.end(), (unsigned char*)arr.v.data() + 7, (unsigned char*)arr.v.data() + 7 + got);
  }
  Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBClose(&env, nullptr, h);
  *out = (unsigned char*)malloc(all.size() ? all.size() : 1);
  memcpy(*out, all.data(), all.size());
  *nbytes = all.size();
  return 0;
}
extern "C" void jni_harness_free(void* p) { free(p); }
