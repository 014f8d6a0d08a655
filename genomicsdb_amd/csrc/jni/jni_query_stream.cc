// jni_query_stream.cc - the native methods GATK4's GenomicsDBFeatureReader reaches, on top of the C ABI
// (include/genomicsdb_amd.h): the six of com.intel.genomicsdb.reader.GenomicsDBQueryStream (reference
// src/main/jni/src/genomicsdb_GenomicsDBQueryStream.cc:29-111; names and argument lists are fixed by the Java class,
// src/main/java/com/intel/genomicsdb/reader/GenomicsDBQueryStream.java:197-211) and the one-time initialiser that
// GenomicsDBLibLoader.loadLibrary() calls right after System.loadLibrary (src/main/jni/src/genomicsdb_jni_init.cc:28-34,
// src/main/java/com/intel/genomicsdb/GenomicsDBLibLoader.java:36-53).
//
// Built into libtiledbgenomicsdb.so, the library name the reference's Java loader asks for (src/main/CMakeLists.txt:66):
// against $JAVA_HOME/include/jni.h where a JDK is present, else against csrc/jni/stub/jni.h (the JNI function-table indices
// of the specification, nothing else), so the glue is compiled, linked and symbol-checked in every build.
#include <jni.h>

#include <algorithm>
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

// The reference initialises MPI here (JNIMpiInit); this build's ranks are launcher processes, there is nothing to set up.
JNIEXPORT jint JNICALL Java_com_intel_genomicsdb_GenomicsDBLibLoader_jniGenomicsDBOneTimeInitialize(JNIEnv*, jclass) { return 0; }

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

// Like the reference (:83-107): the bytes go from the stream's current batch - here a pinned ring buffer the copy engine fills -
// into the Java array with SetByteArrayRegion, batch by batch.  No critical section: the GPU pipeline may run inside peek().
JNIEXPORT jint JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBRead(JNIEnv* env, jobject, jlong handle, jbyteArray java_byte_array,
                                                                                                 jint offset, jint n) {
  void* h = handle_of(handle);
  if (!h || n <= 0) return 0;
  jint total = 0;
  while (total < n) {
    const uint8_t* p = NULL;
    uint64_t avail = 0;
    const int rc = gdb_mi355_peek(h, &p, &avail);
    if (rc < 0) { throw_io(env, gdb_mi355_last_error()); return total; }
    if (rc == 0 || avail == 0) break;                 // end of the stream
    const jint k = (jint)std::min<uint64_t>(avail, (uint64_t)(n - total));
    env->SetByteArrayRegion(java_byte_array, offset + total, k, reinterpret_cast<const jbyte*>(p));
    if (env->ExceptionCheck()) return total;          // ArrayIndexOutOfBoundsException is pending
    if (gdb_mi355_skip(h, (uint64_t)k) != (int64_t)k) { throw_io(env, gdb_mi355_last_error()); return total; }
    total += k;
  }
  return total;
}

JNIEXPORT jlong JNICALL Java_com_intel_genomicsdb_reader_GenomicsDBQueryStream_jniGenomicsDBSkip(JNIEnv* env, jobject, jlong handle, jlong n) {
  const int64_t skipped = gdb_mi355_skip(handle_of(handle), (uint64_t)(n > 0 ? n : 0));
  if (skipped < 0) { throw_io(env, gdb_mi355_last_error()); return 0; }
  return (jlong)skipped;
}

}  // extern "C"
