// jni_query_stream.cc - the six native methods of com.intel.genomicsdb.reader.GenomicsDBQueryStream on top of the C ABI
// (include/genomicsdb_amd.h).  Reference: src/main/jni/src/genomicsdb_GenomicsDBQueryStream.cc:29-111 (the method names and
// argument lists are fixed by the Java class, src/main/java/com/intel/genomicsdb/reader/GenomicsDBQueryStream.java:197-211).
//
// Built only where a JDK is present (genomicsdb_amd/build.py looks for $JAVA_HOME/include/jni.h) into libtiledbgenomicsdb.so,
// the library name the reference's Java loader asks for; this image has no JDK, so here the file is source only.
#include <jni.h>

#include <cstdint>

#include "../../../include/genomicsdb_amd.h"

namespace {
inline void* handle_of(jlong h) { return reinterpret_cast<void*>(static_cast<std::uintptr_t>(h)); }
void throw_io(JNIEnv* env, const char* what) {
  jclass cls = env->FindClass("java/io/IOException");
  if (cls) env->ThrowNew(cls, what);
}
struct UtfChars {          // GetStringUTFChars / ReleaseStringUTFChars as a scope
  JNIEnv* env; jstring s; const char* p;
  UtfChars(JNIEnv* e, jstring js) : env(e), s(js), p(js ? e->GetStringUTFChars(js, NULL) : NULL) {}
  ~UtfChars() { if (p) env->ReleaseStringUTFChars(s, p); }
};
}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBInit(
    JNIEnv* env, jobject, jstring loader_json, jstring query_json, jstring chr, jint start, jint end, jint rank, jlong buffer_capacity,
    jlong segment_size, jboolean is_bcf, jboolean produce_header_only, jboolean use_missing_values_only_not_vector_end,
    jboolean keep_idx_fields_in_bcf_header) {
  UtfChars l(env, loader_json), q(env, query_json), c(env, chr);
  if (!q.p || !c.p) { throw_io(env, "GenomicsDBQueryStream: null query JSON / contig name"); return 0; }
  void* h = gdb_mi355_init(l.p ? l.p : "", q.p, c.p, (int)start, (int)end, (int)rank, (uint64_t)buffer_capacity, (uint64_t)segment_size, is_bcf ? 1 : 0,
                           produce_header_only ? 1 : 0, use_missing_values_only_not_vector_end ? 1 : 0, keep_idx_fields_in_bcf_header ? 1 : 0);
  if (!h) throw_io(env, gdb_mi355_last_error());     // (the reference lets its C++ exception escape; the JVM dies - an IOException is kinder)
  return static_cast<jlong>(reinterpret_cast<std::uintptr_t>(h));
}

JNIEXPORT jlong JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBClose(JNIEnv*, jobject, jlong handle) {
  return (jlong)gdb_mi355_close(handle_of(handle));
}

JNIEXPORT jlong JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBGetNumBytesAvailable(JNIEnv*, jobject, jlong handle) {
  return (jlong)gdb_mi355_get_num_bytes_available(handle_of(handle));
}

JNIEXPORT jbyte JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBReadNextByte(JNIEnv*, jobject, jlong handle) {
  return (jbyte)gdb_mi355_read_next_byte(handle_of(handle));
}

JNIEXPORT jint JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBRead(JNIEnv* env, jobject, jlong handle, jbyteArray java_byte_array,
                                                                                                 jint offset, jint n) {
  if (n <= 0) return 0;
  // no JNI calls between Get- and ReleasePrimitiveArrayCritical (same discipline as the reference, :83-98)
  jbyte* dst = static_cast<jbyte*>(env->GetPrimitiveArrayCritical(java_byte_array, NULL));
  if (!dst) return 0;
  const int64_t got = gdb_mi355_read(handle_of(handle), reinterpret_cast<uint8_t*>(dst), (uint64_t)offset, (uint64_t)n);
  env->ReleasePrimitiveArrayCritical(java_byte_array, dst, 0);
  if (got < 0) { throw_io(env, gdb_mi355_last_error()); return 0; }
  return (jint)got;
}

JNIEXPORT jlong JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBSkip(JNIEnv* env, jobject, jlong handle, jlong n) {
  const int64_t skipped = gdb_mi355_skip(handle_of(handle), (uint64_t)(n > 0 ? n : 0));
  if (skipped < 0) { throw_io(env, gdb_mi355_last_error()); return 0; }
  return (jlong)skipped;
}

}  // extern "C"
