// gdb_core.hpp - per-cell / per-record / per-entry device functions of the MI355X variant-combine path.
//
// Everything here is GDB_HD (__host__ __device__ under hipcc): the HIP kernels in
// kernels/gdb_kernels.hip are thin grid wrappers around these functions, and tests/hostsim compiles the
// very same functions with g++ so the index arithmetic can be debugged without a GPU.  This is NOT a CPU
// path of the product: the shipped library only contains the HIP kernels.
//
// What replaces what (reference paths under /root/reference/src/main/cpp):
//   classify_cell          gt_fill_row flags + field validity   src/genomicsdb/query_variants.cc:1014-1117,
//                                                               include/genomicsdb/variant_field_data.h:365-384
//   site_emit              handle_deletions + merge_reference_allele + merge_alt_alleles (+LUT) + INFO reducers +
//                          fixed VCF columns                    src/query_operations/broad_combined_gvcf.cc:523-601,
//                                                               :765-901, :912-1078; variant_operations.cc:73-228
//   entry_emit             remap_GT_field / remap_data_based_on_alleles / _genotype_{haploid,diploid} +
//                          collect_and_extend_fields + htslib FORMAT text
//                                                               src/genomicsdb/variant_field_handler.cc:41-191, :804-871
// The sweep itself (VariantCallEndPQ, handle_gvcf_ranges) has no per-thread counterpart: it becomes the
// event sort + scans of the pipeline.
#pragma once
#include "gdb_types.h"

// fetch-and-add on a 64-bit counter (device: atomic; serial host harness: plain)
#if defined(__HIP_DEVICE_COMPILE__)
#define GDB_FETCH_ADD_U64(p, v) atomicAdd((unsigned long long*)(p), (unsigned long long)(v))
#else
static inline unsigned long long gdb_fetch_add_u64(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o + v; return o; }
#define GDB_FETCH_ADD_U64(p, v) gdb_fetch_add_u64((unsigned long long*)(p), (unsigned long long)(v))
#endif

// ---- cell flags --------------------------------------------------------------------------------------
#define GDB_CF_REFBLOCK 1u
#define GDB_CF_DELETION 2u
#define GDB_CF_HAS_NR 4u
#define GDB_CF_HEAVY 8u
#define GDB_CF_IN_WINDOW 16u
#define GDB_CF_NALT(f) (((f) >> 8) & 0xFFu)
#define GDB_CF_PLOIDY(f) (((f) >> 16) & 0xFu)

// ---- record flags ------------------------------------------------------------------------------------
#define GDB_RF_NON_REF_EXISTS 1u
#define GDB_RF_REMAPPING_NEEDED 2u
#define GDB_RF_SKIP_G_FIELDS 4u   // too many ALT alleles for genotype-length fields

// incidence flags
#define GDB_IF_SPANNING 1u        // spanning deletion (began before this record)
#define GDB_IF_GT_OVERRIDE 2u     // GT replaced by the min-PL genotype
#define GDB_IF_NO_NR 4u           // call has no <NON_REF>: unmapped GT alleles become '.'

struct CellMeta {       // SoA, one entry per cell of the staged fragment
  uint64_t* vmask;      // bit f: plan field f is valid for this cell
  uint32_t* cflags;
  int32_t* dpval;       // INFO DP else MIN_DP else DP_FORMAT else 0 (broad_combined_gvcf.cc:696-712)
  int64_t* eff_end;     // END truncated by the next cell of the same row (overlap override, query_variants.cc:512-543)
  int32_t* k_lo;        // first / last record this cell is live in (-1: none)
  int32_t* k_hi;
};

struct AlleleRef { const char* p; int len; int suffix_from; uint32_t hash; };  // allele text = p[0..len) + mergedREF[suffix_from..)

struct RecordTable {    // one entry per output record (VCF line)
  int64_t npos;         // P
  const int64_t* start; // column interval of the record
  const int64_t* end;
};

struct SiteOut {        // per-record results of the site kernel
  uint8_t* num_alleles;    // A_k, merged alleles incl. REF
  uint8_t* rflags;
  uint32_t* fmt_mask;      // bit i: plan.format_field[i] is emitted for this record
  uint32_t* prefix_len;    // bytes of the fixed columns incl. trailing FORMAT keys, excl. sample columns
};

struct HeavyLists {     // incidences (record, heavy cell) sorted by (record, row)
  const int64_t* base;       // [P+1]
  const int64_t* cell;       // [T] cell idx
  const uint32_t* i2m_off;   // [T+1]
  int8_t* i2m;               // input allele idx -> merged allele idx (-1: none)
  uint8_t* iflags;           // [T]
  int8_t* gt_override;       // [T*GDB_MAX_PLOIDY] min-PL genotype of a spanning deletion (reduced allele indices)
};

struct PresenceCounts {  // per record, from difference arrays + scan
  const int32_t* fmt_cnt;  // [n_format][P]  #live calls with a valid value for format_field[i]
  const int32_t* dp_sum;   // [P]
  const int32_t* nr_cnt;   // [P]  #live calls whose ALT has <NON_REF>
  int64_t stride;          // P
};

struct NameTables {      // small text tables in device memory
  const char* text;
  const int32_t* field_name_off;   // per plan field: vcf name
  const int32_t* field_name_len;
  const int32_t* filter_name_off;  // per vid field idx (FILTER ids); len 0 = not in header
  const int32_t* filter_name_len;
  int32_t n_filter_names;
  const int32_t* filter_bcf_id;    // per vid field idx: index in the BCF header dictionary (-1: not in the header)
};

// ---- sinks -----------------------------------------------------------------------------------------
// (pos() / patch_u32(): the BCF emitters write a count first and fill it in once the values behind it are known)
struct CountSink {
  uint64_t n;
  GDB_HD CountSink() : n(0) {}
  GDB_HD void put(char) { ++n; }
  GDB_HD void write(const char*, int len) { n += (uint64_t)len; }
  GDB_HD uint32_t pos() const { return (uint32_t)n; }
  GDB_HD void patch_u32(uint32_t, uint32_t) {}
};
struct ByteSink {
  char* p;
  char* base;
  GDB_HD explicit ByteSink(char* q) : p(q), base(q) {}
  GDB_HD void put(char c) { *p++ = c; }
  GDB_HD void write(const char* s, int len) { for (int i = 0; i < len; ++i) *p++ = s[i]; }
  GDB_HD uint32_t pos() const { return (uint32_t)(p - base); }
  GDB_HD void patch_u32(uint32_t at, uint32_t v) { for (int i = 0; i < 4; ++i) base[at + i] = (char)((v >> (8 * i)) & 0xFFu); }
};

#if defined(__HIPCC__)
typedef __attribute__((address_space(3))) char gdb_lds_char;
struct LdsCapSink {  // counts every byte, stores the first `cap` of them in LDS: length and (short) text in ONE emitter pass
  gdb_lds_char* p;
  uint32_t n, cap;
  __device__ __forceinline__ LdsCapSink(gdb_lds_char* q, uint32_t c) : p(q), n(0), cap(c) {}
  __device__ __forceinline__ void put(char c) { if (n < cap) p[n] = c; ++n; }
  __device__ __forceinline__ void write(const char* s, int len) { for (int i = 0; i < len; ++i) put(s[i]); }
  // up to 8 characters held in a register (first character in the low byte) with ONE unaligned 8-byte LDS store: the bytes
  // behind the len valid ones are overwritten by what is emitted next (gfx950 executes DS accesses at any byte address)
  __device__ __forceinline__ void put_word(uint64_t w, int len) {
    if (n + 8u <= cap) { const uint32_t a = (uint32_t)(uintptr_t)p + n; asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(w) : "memory"); n += (uint32_t)len; }
    else for (int i = 0; i < len; ++i) { put((char)(w & 0xFFu)); w >>= 8; }
  }
  __device__ __forceinline__ uint32_t pos() const { return n; }
  __device__ __forceinline__ void patch_u32(uint32_t at, uint32_t v) { for (int i = 0; i < 4; ++i) if (at + i < cap) p[at + i] = (char)((v >> (8 * i)) & 0xFFu); }
};
// The same with a second tier for the longer texts (fixed columns with long allele lists or many INFO values): what does not fit the
// LDS strip goes to the RECORD'S OWN tail slot in global memory (kSpillTail bytes per record, written a word at a time: no pool, no
// atomic - a pool of chunks handed out by one counter made the site pass 3 x slower as soon as a third of the records spilled:
// ~10^5 atomics on one address, and the records behind the pool's end formatted a second time).  n counts every byte; the text is
// complete when n <= cap + kSpillTail.
constexpr uint32_t kSpillTail = 512;
struct SpillPool { char* buf; };                     // [records][kSpillTail]
struct LdsSpillSink {
  gdb_lds_char* p;
  uint32_t n, cap;
  uint32_t acc;           // the spilled bytes of the word under way
  char* tail;             // the record's tail slot
  __device__ __forceinline__ LdsSpillSink(gdb_lds_char* q, uint32_t c, const SpillPool& sp, int64_t record) : p(q), n(0), cap(c), acc(0), tail(sp.buf + (size_t)record * kSpillTail) {}
  __device__ __forceinline__ void put(char c) {
    if (n < cap) p[n] = c;
    else {
      const uint32_t at = n - cap;
      acc |= (uint32_t)(uint8_t)c << (8u * (at & 3u));
      if ((at & 3u) == 3u) { if (at < kSpillTail) *reinterpret_cast<uint32_t*>(tail + (at & ~3u)) = acc; acc = 0; }
    }
    ++n;
  }
  // the last, partial word (call once, when the text is complete)
  __device__ __forceinline__ void flush() { if (n > cap && ((n - cap) & 3u) && n - cap < kSpillTail) { *reinterpret_cast<uint32_t*>(tail + ((n - cap) & ~3u)) = acc; acc = 0; } }
  __device__ __forceinline__ void write(const char* s, int len) { for (int i = 0; i < len; ++i) put(s[i]); }
  __device__ __forceinline__ void put_word(uint64_t w, int len) {
    if (n + 8u <= cap) { const uint32_t a = (uint32_t)(uintptr_t)p + n; asm volatile("ds_write_b64 %0, %1" ::"v"(a), "v"(w) : "memory"); n += (uint32_t)len; }
    else for (int i = 0; i < len; ++i) { put((char)(w & 0xFFu)); w >>= 8; }
  }
  __device__ __forceinline__ bool complete() const { return n <= cap + kSpillTail; }
  __device__ __forceinline__ uint32_t pos() const { return n; }
  __device__ __forceinline__ void patch_u32(uint32_t at, uint32_t v) { for (int i = 0; i < 4; ++i) if (at + i < cap) p[at + i] = (char)((v >> (8 * i)) & 0xFFu); }   // (only ever the record header)
};
#endif

// ---- small helpers -----------------------------------------------------------------------------------
GDB_HD uint32_t gdb_f2u(float f) { union { float f; uint32_t u; } x; x.f = f; return x.u; }
GDB_HD bool gdb_int_valid(int32_t v) { return v != GDB_BCF_INT32_MISSING && v != GDB_BCF_INT32_VECTOR_END; }
GDB_HD bool gdb_float_valid(float v) { uint32_t u = gdb_f2u(v); return u != GDB_BCF_FLOAT_MISSING_BITS && u != GDB_BCF_FLOAT_VECTOR_END_BITS; }
GDB_HD int gdb_alleles2gt(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }
// General ploidy (KnownFieldInfo::get_number_of_genotypes / VariantOperations::get_genotype_index,
// src/main/cpp/src/utils/known_field_info.cc:130-162, variant_operations.cc:300-321): genotypes are the non-decreasing
// allele tuples a[0] <= ... <= a[p-1] in colexicographic order; index = sum_i C(a[i] + i, i + 1).
GDB_HD int64_t gdb_choose(int n, int k) {
  if (k < 0 || k > n) return 0;
  int64_t r = 1;
  for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
  return r;
}
GDB_HD int64_t gdb_genotype_index_sorted(const int* a, int p) {
  int64_t g = 0;
  for (int i = 0; i < p; ++i) g += gdb_choose(a[i] + i, i + 1);
  return g;
}
GDB_HD int64_t gdb_genotype_index(const int* a, int p) {  // a[] in any order, p <= GDB_MAX_PLOIDY
  int t[GDB_MAX_PLOIDY];
  for (int i = 0; i < p; ++i) { int v = a[i], j = i; while (j > 0 && t[j - 1] > v) { t[j] = t[j - 1]; --j; } t[j] = v; }
  return gdb_genotype_index_sorted(t, p);
}
// next tuple in VCF (colex) order over alleles 0..nalleles-1; false after the last one
GDB_HD bool gdb_next_genotype(int* a, int p, int nalleles) {
  for (int i = 0; i < p; ++i) {
    const int bound = i + 1 < p ? a[i + 1] : nalleles - 1;
    if (a[i] < bound) { ++a[i]; for (int j = 0; j < i; ++j) a[j] = 0; return true; }
  }
  return false;
}

template <class Sink> GDB_HD void put_u64(Sink& s, uint64_t v) {
  char buf[20];
  int n = 0;
  do { buf[n++] = (char)('0' + (v % 10)); v /= 10; } while (v);
  while (n) s.put(buf[--n]);
}
template <class Sink> GDB_HD void put_i64(Sink& s, int64_t v) {
  if (v < 0) { s.put('-'); put_u64(s, (uint64_t)(-(v + 1)) + 1u); } else put_u64(s, (uint64_t)v);
}
// 32-bit only (no 64-bit division on the hot path): every FORMAT integer goes through here.  The digits are gathered in
// a register (8 characters in a 64-bit word, first character in the low byte) - no private-memory array.
GDB_HD uint64_t gdb_pack_digits(uint32_t v, int& n) {   // v < 10^8
  uint64_t w = 0;
  n = 0;
  do { const uint32_t q = v / 10u; w = (w << 8) | (uint64_t)('0' + (v - q * 10u)); v = q; ++n; } while (v);
  return w;
}
template <class Sink> GDB_HD void put_packed(Sink& s, uint64_t w, int n) {
  for (int i = 0; i < n; ++i) { s.put((char)(w & 0xFFu)); w >>= 8; }
}
#if defined(__HIPCC__)
__device__ __forceinline__ void put_packed(LdsCapSink& s, uint64_t w, int n) { s.put_word(w, n); }
__device__ __forceinline__ void put_packed(LdsSpillSink& s, uint64_t w, int n) { s.put_word(w, n); }
#endif
template <class Sink> GDB_HD void put_u32(Sink& s, uint32_t v) {
  if (v < 10u) { s.put((char)('0' + v)); return; }
  if (v < 100u) { const uint32_t q = v / 10u; s.put((char)('0' + q)); s.put((char)('0' + (v - q * 10u))); return; }
  int n = 0;
  if (v >= 100000000u) {   // 9 or 10 digits: leading one or two, then exactly eight
    const uint32_t hi = v / 100000000u;
    v -= hi * 100000000u;
    const uint64_t wh = gdb_pack_digits(hi, n);   // (its own statement: the order in which call arguments are evaluated is unspecified)
    put_packed(s, wh, n);
    uint64_t w = gdb_pack_digits(v, n);
    for (; n < 8; ++n) w = (w << 8) | (uint64_t)'0';
    put_packed(s, w, 8);
    return;
  }
  const uint64_t w = gdb_pack_digits(v, n);
  put_packed(s, w, n);
}
template <class Sink> GDB_HD void put_i32(Sink& s, int32_t v) {
  if (v < 0) { s.put('-'); put_u32(s, 0u - (uint32_t)v); } else put_u32(s, (uint32_t)v);
}

// ---- "%g" of a float, exactly --------------------------------------------------------------------------------------------------
// htslib's kputd hands every value outside [0.0001, 999999] to printf("%g") (vcf.c / kstring.c; the oracle calls snprintf).  "%g" is
// fully specified by the C standard: the value - here a float, i.e. exactly M * 2^e2 with M < 2^24 - is rounded to P = 6 significant
// decimal digits (glibc rounds the EXACT value, ties to even); if the decimal exponent X of the rounded value satisfies
// -4 <= X < P the style is fixed, else scientific with at least two exponent digits; trailing zeros of the fraction (and a
// point with nothing behind it) are removed.  All arithmetic below is integer: a 224-bit register holds M * 10^k (k <= 50 for the
// smallest subnormal, 2^-149 = 1.4e-45) or M * 2^e2 (e2 <= 104: 3.4e38 < 2^128).
struct GdbBig { uint32_t w[7]; };
GDB_HD void gdb_big_mul10(GdbBig& b) {
  uint64_t c = 0;
  for (int i = 0; i < 7; ++i) { const uint64_t t = (uint64_t)b.w[i] * 10u + c; b.w[i] = (uint32_t)t; c = t >> 32; }
}
GDB_HD uint32_t gdb_big_divmod10(GdbBig& b) {
  uint64_t r = 0;
  for (int i = 6; i >= 0; --i) { const uint64_t t = (r << 32) | b.w[i]; b.w[i] = (uint32_t)(t / 10u); r = t % 10u; }
  return (uint32_t)r;
}
GDB_HD bool gdb_big_is_zero(const GdbBig& b) { uint32_t o = 0; for (int i = 0; i < 7; ++i) o |= b.w[i]; return o == 0; }
GDB_HD uint32_t gdb_big_bit(const GdbBig& b, int i) { return i >= 0 && i < 224 ? (b.w[i >> 5] >> (i & 31)) & 1u : 0u; }
GDB_HD bool gdb_big_any_below(const GdbBig& b, int n) {   // any of bits [0, n)
  uint32_t o = 0;
  for (int i = 0; i < 7; ++i) {
    const int lo = i * 32;
    if (n >= lo + 32) o |= b.w[i]; else if (n > lo) o |= b.w[i] & ((1u << (n - lo)) - 1u);
  }
  return o != 0;
}
GDB_HD GdbBig gdb_big_shr(const GdbBig& b, int n) {
  GdbBig r;
  const int ws = n >> 5, bs = n & 31;
  for (int i = 0; i < 7; ++i) {
    const uint32_t lo = i + ws < 7 ? b.w[i + ws] : 0u, hi = i + ws + 1 < 7 ? b.w[i + ws + 1] : 0u;
    r.w[i] = bs ? (lo >> bs) | (hi << (32 - bs)) : lo;
  }
  return r;
}
GDB_HD bool gdb_big_below(const GdbBig& b, uint32_t v) { uint32_t o = 0; for (int i = 1; i < 7; ++i) o |= b.w[i]; return o == 0 && b.w[0] < v; }
// q = the six significant digits (100000 .. 999999), X = decimal exponent of the rounded value, for M * 2^e2 (M > 0)
GDB_HD void gdb_round_6_digits(uint32_t M, int e2, uint32_t& q, int& X) {
  GdbBig N;
  for (int i = 0; i < 7; ++i) N.w[i] = 0;
  N.w[0] = M;
  bool up;
  if (e2 >= 0 || !gdb_big_below(gdb_big_shr(N, -e2), 100000u)) {
    // at least six integer digits: I = floor(value); digits are dropped from I, the fraction only breaks ties
    int s = 0;
    if (e2 >= 0) { const int ws = e2 >> 5, bs = e2 & 31; N.w[ws] = M << bs; N.w[0] = ws ? 0u : N.w[0]; if (bs && ws + 1 < 7) N.w[ws + 1] = M >> (32 - bs); }
    else s = -e2;
    GdbBig I = gdb_big_shr(N, s);
    const bool half_bit = s > 0 && gdb_big_bit(N, s - 1);
    bool sticky = s > 1 && gdb_big_any_below(N, s - 1);
    int dropped = 0;
    uint32_t last = 0;
    while (!gdb_big_below(I, 1000000u)) { if (dropped) sticky = sticky || last != 0; else sticky = sticky || half_bit; last = gdb_big_divmod10(I); ++dropped; }
    q = I.w[0];
    if (dropped) up = last > 5u || (last == 5u && (sticky || (q & 1u)));
    else up = half_bit && (sticky || (q & 1u));
    X = 5 + dropped;
  } else {
    // value < 100000: scale by 10^k until floor(value * 10^k) has six digits; what lies below the binary point decides the rounding
    const int s = -e2;
    int k = 0;
    while (gdb_big_below(gdb_big_shr(N, s), 100000u)) { gdb_big_mul10(N); ++k; }
    q = gdb_big_shr(N, s).w[0];
    const bool half_bit = gdb_big_bit(N, s - 1) != 0, sticky = s > 1 && gdb_big_any_below(N, s - 1);
    up = half_bit && (sticky || (q & 1u));
    X = 5 - k;
  }
  if (up && ++q == 1000000u) { q = 100000u; ++X; }
}
// |f| as "%g" prints it (f finite, not zero)
template <class Sink> GDB_HD void put_float_g(Sink& s, float f) {
  const uint32_t u = gdb_f2u(f) & 0x7FFFFFFFu;
  const uint32_t ex = u >> 23, man = u & 0x7FFFFFu;
  uint32_t q; int X;
  gdb_round_6_digits(ex ? man | 0x800000u : man, ex ? (int)ex - 150 : -149, q, X);
  char dg[6];
  for (int i = 5; i >= 0; --i) { dg[i] = (char)('0' + q % 10u); q /= 10u; }
  int last = 5;
  while (last > 0 && dg[last] == '0') --last;
  if (X < -4 || X >= 6) {
    s.put(dg[0]);
    if (last > 0) { s.put('.'); for (int i = 1; i <= last; ++i) s.put(dg[i]); }
    s.put('e');
    uint32_t ax = (uint32_t)(X < 0 ? -X : X);
    s.put(X < 0 ? '-' : '+');
    if (ax < 10u) s.put('0');
    put_u64(s, ax);
  } else if (X >= 0) {
    for (int i = 0; i <= X; ++i) s.put(dg[i]);
    if (last > X) { s.put('.'); for (int i = X + 1; i <= last; ++i) s.put(dg[i]); }
  } else {
    s.put('0'); s.put('.');
    for (int i = 0; i < -X - 1; ++i) s.put('0');
    for (int i = 0; i <= last; ++i) s.put(dg[i]);
  }
}

// Float text as htslib's kputd writes it (see oracle/gdb_oracle_combine.hpp format_float): its own six-digit rule inside
// [0.0001, 999999], printf("%g") of the magnitude outside - every float has a text, none is refused.
template <class Sink> GDB_HD bool put_float(Sink& s, float f) {
  const uint32_t bits = gdb_f2u(f);
  if ((bits & 0x7F800000u) == 0x7F800000u) {   // kputd: "d < 0" writes the sign of -inf itself; a NaN fails every comparison and reaches "%g" as it is (glibc: "-nan" with the sign bit)
    if (bits >> 31) s.put('-');
    if (bits & 0x7FFFFFu) { s.put('n'); s.put('a'); s.put('n'); } else { s.put('i'); s.put('n'); s.put('f'); }
    return true;
  }
  double d = (double)f;
  if (d == 0) { if (bits >> 31) s.put('-'); s.put('0'); return true; }
  if (d < 0) { s.put('-'); d = -d; }
  if (!(d >= 0.0001 && d <= 999999)) { put_float_g(s, f); return true; }
  uint64_t i = (uint64_t)(d * 10000000000.0);
  if (d < 0.001) i += 5; else if (d < 0.01) i += 50; else if (d < 0.1) i += 500; else if (d < 1) i += 5000;
  else if (d < 10) i += 50000; else if (d < 100) i += 500000; else if (d < 1000) i += 5000000;
  else if (d < 10000) i += 50000000; else if (d < 100000) i += 500000000; else i += 5000000000ULL;
  char dg[24];
  int p = 0;
  { char tmp[24]; int n = 0; do { tmp[n++] = (char)('0' + i % 10); i /= 10; } while (i); while (n) dg[p++] = tmp[--n]; }
  char out[40];
  int m = 0, dot = -1;
  if (p <= 10) {
    out[m++] = '0'; dot = m; out[m++] = '.';
    for (int z = 0; z < 10 - p; ++z) out[m++] = '0';
    int take = p < 6 ? p : 6;
    for (int z = 0; z < take; ++z) out[m++] = dg[z];
  } else {
    int ip = p - 10;
    for (int z = 0; z < ip; ++z) out[m++] = dg[z];
    if (ip < 6) { dot = m; out[m++] = '.'; for (int z = ip; z < 6; ++z) out[m++] = dg[z]; }
  }
  if (dot >= 0) while (m > dot + 2 && out[m - 1] == '0') --m;
  s.write(out, m);
  return true;
}

// ---- fragment accessors --------------------------------------------------------------------------------
template <class T> GDB_HD const T* cell_field(const FragmentView& fr, const CombinePlan& pl, int f, int64_t c, int& n) {
  const GdbColumn& col = fr.col[f];
  if (col.off) { uint32_t a = col.off[c], b = col.off[c + 1]; n = (int)(b - a); return (const T*)col.data + a; }
  n = pl.field[f].fixed_num;
  return (const T*)col.data + (int64_t)c * n;
}
GDB_HD bool field_valid(const CellMeta& cm, int64_t c, int f) { return (cm.vmask[c] >> f) & 1ull; }

// i-th '|' separated token of an ALT string (empty tokens skipped like strtok_r); false when exhausted
GDB_HD bool alt_token(const char* s, int len, int idx, const char*& tok, int& tlen) {
  int i = 0, k = -1;
  while (i < len) {
    while (i < len && s[i] == '|') ++i;
    if (i >= len) break;
    int j = i;
    while (j < len && s[j] != '|') ++j;
    if (++k == idx) { tok = s + i; tlen = j - i; return true; }
    i = j;
  }
  return false;
}
GDB_HD bool allele_is_symbolic(const char* a, int n) {  // VariantUtils::is_symbolic_allele
  if (n > 0 && a[0] == '&') return true;
  if (n == 1 && a[0] == '*') return true;
  if (n > 0 && a[0] == '<' && a[n - 1] == '>') return true;
  for (int i = 0; i < n; ++i) if (a[i] == '[' || a[i] == ']') return true;
  return false;
}
GDB_HD bool allele_is_deletion(int ref_len, const char* a, int n) {  // VariantUtils::is_deletion
  return ref_len > 1 && ((n == 1 && a[0] == '*') || (!allele_is_symbolic(a, n) && n < ref_len));
}

// ---- K0: per-cell classification ---------------------------------------------------------------------
GDB_HD void classify_cell(const FragmentView& fr, const CombinePlan& pl, const CellMeta& cm, int64_t c, uint32_t* err) {
  uint64_t vmask = 0;
  for (int f = 0; f < pl.nfields; ++f) {
    const GdbFieldDesc& fd = pl.field[f];
    int n;
    bool valid = false;
    if (fd.elem == GDB_ET_INT) {
      const int32_t* p = cell_field<int32_t>(fr, pl, f, c, n);
      for (int i = 0; i < n; ++i) if (p[i] != GDB_TILEDB_NULL_INT32) { valid = true; break; }
    } else if (fd.elem == GDB_ET_FLOAT) {
      const float* p = cell_field<float>(fr, pl, f, c, n);
      for (int i = 0; i < n; ++i) if (gdb_f2u(p[i]) != GDB_TILEDB_NULL_FLOAT_BITS) { valid = true; break; }
    } else {
      const char* p = cell_field<char>(fr, pl, f, c, n);
      for (int i = 0; i < n; ++i) if (p[i] != GDB_TILEDB_NULL_CHAR) { valid = true; break; }
    }
    if (valid) vmask |= 1ull << f;
  }
  uint32_t flags = 0;
  int nalt = 0, ref_len = 0;
  if (pl.f_REF >= 0 && pl.f_ALT >= 0 && ((vmask >> pl.f_REF) & 1) && ((vmask >> pl.f_ALT) & 1)) {
    int alt_len;
    cell_field<char>(fr, pl, pl.f_REF, c, ref_len);
    const char* alt = cell_field<char>(fr, pl, pl.f_ALT, c, alt_len);
    bool deletion = false, has_nr = false;
    const char* tok; int tl;
    for (int i = 0; alt_token(alt, alt_len, i, tok, tl); ++i) {
      ++nalt;
      if (tl > 0 && tok[0] == '&') has_nr = true;
      if (ref_len > 1 && !allele_is_symbolic(tok, tl) && tl < ref_len) deletion = true;  // contains_deletion
    }
    if (deletion) flags |= GDB_CF_DELETION;
    if (has_nr) flags |= GDB_CF_HAS_NR;
    if (ref_len == 1 && nalt == 1 && has_nr) flags |= GDB_CF_REFBLOCK;  // is_reference_block
  }
  if (nalt + 1 > GDB_MAX_INPUT_ALLELES) { *err |= GDB_ERR_TOO_MANY_INPUT_ALLELES; nalt = GDB_MAX_INPUT_ALLELES - 1; }
  int ploidy = 0;
  if (pl.f_GT >= 0 && ((vmask >> pl.f_GT) & 1)) {
    int n;
    cell_field<int32_t>(fr, pl, pl.f_GT, c, n);
    ploidy = pl.field[pl.f_GT].length == GDB_VL_PP ? (n + 1) >> 1 : n;
    if (ploidy > 15) ploidy = 15;
  }
  // heavy = needs the per-record site logic: not a plain reference block, or carries a value an INFO reducer /
  // QUAL / FILTER rule would read
  bool heavy = !(flags & GDB_CF_REFBLOCK);
  for (int i = 0; i < pl.n_info; ++i) if ((vmask >> pl.info_field[i]) & 1) heavy = true;
  for (int i = 0; i < pl.n_histogram; ++i) if ((vmask >> pl.histogram_bin_field[i]) & 1) heavy = true;
  if (pl.qual_combine_op != GDB_OP_UNKNOWN && pl.f_QUAL >= 0 && ((vmask >> pl.f_QUAL) & 1)) heavy = true;
  if (pl.produce_FILTER_field && pl.f_FILTER >= 0 && ((vmask >> pl.f_FILTER) & 1)) heavy = true;
  if (pl.f_ID >= 0 && ((vmask >> pl.f_ID) & 1)) heavy = true;   // ID union (broad_combined_gvcf.cc:730-763)
  if (heavy) flags |= GDB_CF_HEAVY;
  flags |= ((uint32_t)nalt & 0xFFu) << 8;
  flags |= ((uint32_t)ploidy & 0xFu) << 16;
  // DP contribution of this call to the INFO DP sum
  int32_t dp = 0;
  {
    int n; bool got = false;
    if (pl.f_DP >= 0 && ((vmask >> pl.f_DP) & 1)) { const int32_t* p = cell_field<int32_t>(fr, pl, pl.f_DP, c, n); if (n > 0 && gdb_int_valid(p[0])) { dp = p[0]; got = true; } }
    if (!got && pl.f_MIN_DP >= 0 && ((vmask >> pl.f_MIN_DP) & 1)) { const int32_t* p = cell_field<int32_t>(fr, pl, pl.f_MIN_DP, c, n); if (n > 0 && gdb_int_valid(p[0])) { dp = p[0]; got = true; } }
    if (!got && pl.f_DP_FORMAT >= 0 && ((vmask >> pl.f_DP_FORMAT) & 1)) { const int32_t* p = cell_field<int32_t>(fr, pl, pl.f_DP_FORMAT, c, n); if (n > 0 && gdb_int_valid(p[0])) { dp = p[0]; got = true; } }
  }
  cm.vmask[c] = vmask;
  cm.cflags[c] = flags;
  cm.dpval[c] = dp;
}

// ---- site (per record) ---------------------------------------------------------------------------------
// hash of the text an AlleleRef stands for (FNV-1a): the merge compares hashes first, strings only on a hash match - a site
// with thousands of variant calls would otherwise re-read every merged allele string for every call
GDB_HD uint32_t allele_hash(const char* p, int len, int suffix_from, const char* mref, int mref_len) {
  uint32_t h = 2166136261u;
  for (int i = 0; i < len; ++i) h = (h ^ (uint32_t)(unsigned char)p[i]) * 16777619u;
  if (suffix_from >= 0) for (int i = suffix_from; i < mref_len; ++i) h = (h ^ (uint32_t)(unsigned char)mref[i]) * 16777619u;
  return h;
}

GDB_HD bool allele_equal(const AlleleRef& a, const AlleleRef& b, const char* mref, int mref_len) {
  if (a.hash != b.hash) return false;
  int la = a.len + (a.suffix_from >= 0 ? mref_len - a.suffix_from : 0);
  int lb = b.len + (b.suffix_from >= 0 ? mref_len - b.suffix_from : 0);
  if (la != lb) return false;
  for (int i = 0; i < la; ++i) {
    char ca = i < a.len ? a.p[i] : mref[a.suffix_from + (i - a.len)];
    char cb = i < b.len ? b.p[i] : mref[b.suffix_from + (i - b.len)];
    if (ca != cb) return false;
  }
  return true;
}

// Median fields through a device-wide sort: key = record << 33 | invalid << 32 | order-preserving value bits, value = incidence.
// After the (stable) sort the valid values of record k are the first n_valid entries of [base[k], base[k+1]) in ascending
// order, equal values in row order.  slot[f] = index of plan field f among the sorted fields or -1; arrays of slot s start at
// s * stride.
struct MedianOrder {
  const uint64_t* keys;      // sorted
  const uint32_t* inc;       // incidence index per sorted key
  int64_t stride;            // T
  int8_t slot[GDB_MAX_FIELDS];
  int32_t enabled;
};
GDB_HD uint32_t gdb_orderable_bits(float v) {
  uint32_t u = gdb_f2u(v);
  if (u == 0x80000000u) u = 0;               // -0.0 and +0.0 compare equal: same key, row order decides
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
GDB_HD uint32_t gdb_orderable_bits(int32_t v) { return (uint32_t)v ^ 0x80000000u; }

// Medians of the few records with very many variant calls (the first record of a partition: every sample has a cell starting
// there), computed by one workgroup each before the site pass; index[k] = row of the table or -1; tables of field slot s
// start at s * stride.
struct BigMedians {
  const int32_t* index;      // [P]
  const uint32_t* value;     // raw 32-bit pattern of the median element
  const uint8_t* ok;         // 0: no valid value in the record
  int64_t stride;
  int8_t slot[GDB_MAX_FIELDS];
  int32_t enabled;
};

// std::nth_element as the reference's C++ library (libstdc++, bits/stl_algo.h: __introselect) runs it, restated on a plain
// float array: the reference takes medians with it (variant_field_handler.cc:529-607) and prints the selected element with
// "%g", so when -0 and +0 tie at the middle WHICH of them is selected shows in the VCF ("-0" / "0") and depends on the
// permutation the algorithm leaves behind, not only on the order of the values.  a[0..n) = valid values in call (row) order.
GDB_HD void gdb_swapf(float* a, int64_t i, int64_t j) { const float t = a[i]; a[i] = a[j]; a[j] = t; }
GDB_HD void gdb_heap_adjust(float* a, int64_t hole, int64_t len, float value) {   // __adjust_heap + __push_heap
  const int64_t top = hole;
  int64_t child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (a[child] < a[child - 1]) --child;
    a[hole] = a[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[hole] = a[child - 1];
    hole = child - 1;
  }
  int64_t parent = (hole - 1) / 2;
  while (hole > top && a[parent] < value) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; }
  a[hole] = value;
}
GDB_HD float gdb_introselect_libstdcxx(float* a, int64_t n, int64_t nth, int depth) {
  int64_t first = 0, last = n;
  while (last - first > 3) {
    if (depth == 0) {                                // __heap_select(first, nth + 1, last), then swap(first, nth)
      const int64_t middle = nth + 1, len = middle - first;
      if (len >= 2) for (int64_t parent = (len - 2) / 2;; --parent) { gdb_heap_adjust(a + first, parent, len, a[first + parent]); if (parent == 0) break; }
      for (int64_t i = middle; i < last; ++i)
        if (a[i] < a[first]) { const float v = a[i]; a[i] = a[first]; gdb_heap_adjust(a + first, 0, len, v); }
      gdb_swapf(a, first, nth);
      return a[nth];
    }
    --depth;
    // __unguarded_partition_pivot: median of (first + 1, mid, last - 1) to first, then partition (first + 1, last) around it
    const int64_t mid = first + (last - first) / 2, x = first + 1, y = mid, z = last - 1;
    if (a[x] < a[y]) {
      if (a[y] < a[z]) gdb_swapf(a, first, y);
      else if (a[x] < a[z]) gdb_swapf(a, first, z);
      else gdb_swapf(a, first, x);
    } else if (a[x] < a[z]) gdb_swapf(a, first, x);
    else if (a[y] < a[z]) gdb_swapf(a, first, z);
    else gdb_swapf(a, first, y);
    int64_t lo = first + 1, hi = last;
    const float pivot = a[first];                    // (the pivot slot is outside [first + 1, last): its value does not move)
    for (;;) {
      while (a[lo] < pivot) ++lo;
      --hi;
      while (pivot < a[hi]) --hi;
      if (!(lo < hi)) break;
      gdb_swapf(a, lo, hi);
      ++lo;
    }
    if (lo <= nth) first = lo; else last = lo;
  }
  for (int64_t i = first + 1; i < last; ++i) {       // __insertion_sort
    const float v = a[i];
    if (v < a[first]) { for (int64_t j = i; j > first; --j) a[j] = a[j - 1]; a[first] = v; }
    else { int64_t j = i; while (v < a[j - 1]) { a[j] = a[j - 1]; --j; } a[j] = v; }
  }
  return a[nth];
}
// The same selection with every partition step written as DATA-PARALLEL operations - what one workgroup of k_site_huge runs for a record
// of tens of thousands of calls whose median is a zero of either sign (one thread needs seconds for 2 000 such records).  The Hoare sweep
// of __unguarded_partition moves two pointers towards each other and swaps what they stop at; neither pointer ever reads a position the
// other has written before they cross, so the sweep is determined by two lists taken from the array as it stands: L = positions
// (ascending) the left pointer stops at, !(a[i] < pivot), and R = positions (descending) the right pointer stops at, !(pivot < a[i]).
// Pair k is swapped while L[k] < R[k]; with m such pairs the sweep returns L[m] if the left pointer reaches it before the swapped
// region (L[m] < R[m - 1]), else R[m - 1], where a value that is not below the pivot now lies.  posL / posR: scratch of n entries each.
// Here the lists are built by plain loops (the host-side statement the tests compare with the library, permutation for permutation);
// the kernel builds them with workgroup scans.
GDB_HD int64_t gdb_partition_by_lists(float* a, int64_t first, int64_t last, uint32_t* posL, uint32_t* posR) {
  const float pivot = a[first];
  int64_t nl = 0, nr = 0;
  for (int64_t i = first + 1; i < last; ++i) if (!(a[i] < pivot)) posL[nl++] = (uint32_t)i;
  for (int64_t i = last - 1; i > first; --i) if (!(pivot < a[i])) posR[nr++] = (uint32_t)i;
  int64_t m = 0;
  const int64_t both = nl < nr ? nl : nr;
  while (m < both && posL[m] < posR[m]) ++m;
  for (int64_t k = 0; k < m; ++k) gdb_swapf(a, posL[k], posR[k]);
  return (m < nl && (m == 0 || posL[m] < posR[m - 1])) ? (int64_t)posL[m] : (int64_t)posR[m - 1];
}
GDB_HD void gdb_median3_to_first(float* a, int64_t first, int64_t last) {   // __move_median_to_first(first, first + 1, mid, last - 1)
  const int64_t mid = first + (last - first) / 2, x = first + 1, y = mid, z = last - 1;
  if (a[x] < a[y]) {
    if (a[y] < a[z]) gdb_swapf(a, first, y);
    else if (a[x] < a[z]) gdb_swapf(a, first, z);
    else gdb_swapf(a, first, x);
  } else if (a[x] < a[z]) gdb_swapf(a, first, x);
  else if (a[y] < a[z]) gdb_swapf(a, first, z);
  else gdb_swapf(a, first, y);
}
GDB_HD float gdb_nth_element_by_lists(float* a, int64_t n, int64_t nth, uint32_t* posL, uint32_t* posR) {
  int depth = 0;
  for (int64_t m = n; m > 1; m >>= 1) ++depth;
  depth *= 2;
  int64_t first = 0, last = n;
  while (last - first > 3) {
    if (depth == 0) return gdb_introselect_libstdcxx(a + first, last - first, nth - first, 0);   // (heap select: sequential by nature, and never reached by real data)
    --depth;
    gdb_median3_to_first(a, first, last);
    const int64_t cut = gdb_partition_by_lists(a, first, last, posL, posR);
    if (cut <= nth) first = cut; else last = cut;
  }
  return gdb_introselect_libstdcxx(a + first, last - first, nth - first, 1);   // (at most three elements left: the insertion sort)
}
GDB_HD float gdb_nth_element_libstdcxx(float* a, int64_t n, int64_t nth) {
  int depth = 0;
  for (int64_t m = n; m > 1; m >>= 1) ++depth;     // std::__lg(n) * 2
  return gdb_introselect_libstdcxx(a, n, nth, 2 * depth);
}
// ---- FILTER union: the iteration order of libstdc++'s std::unordered_set<int> ------------------------------------------------------
// The reference unites the FILTER ids of the live calls in a std::unordered_set<int> (one RANGE insert per call) and writes them
// in the set's iteration order (broad_combined_gvcf.cc:846-874), so the order of several different ids in a record is a property
// of the C++ library - like the tied medians above.  Restated here for libstdc++ (GCC 11, hashtable_policy.h / hashtable.h /
// hashtable_c++0x.cc), checked insertion-sequence-for-insertion-sequence against the library on the host
// (tests/hostsim: hostsim_uset_order, tests/test_filter_union_order.py):
//   * std::hash<int> is the identity, bucket = (size_t)key % bucket_count;
//   * the elements form one singly linked list in which the elements of a bucket are adjacent; a new element goes IN FRONT of
//     its bucket's first element, or to the head of the whole list when its bucket is empty (_M_insert_bucket_begin);
//   * a rehash re-inserts the elements in list order by the same rule (_M_rehash_aux, unique keys);
//   * bucket counts come from _Prime_rehash_policy (max load factor 1): _M_need_rehash(n_bkt, n_elt, 1) per new element,
//     "start with 11" when nothing is allocated, growth factor 2, _M_next_bkt = small table below 14, else the next prime.
struct GdbUSetOrder {
  int32_t key[GDB_MAX_FILTER_IDS];   // list order = iteration order
  int32_t n, nbkt;
  uint64_t next_resize;
  bool overflow;
};
GDB_HD void gdb_uset_init(GdbUSetOrder& s) { s.n = 0; s.nbkt = 1; s.next_resize = 0; s.overflow = false; }
GDB_HD uint64_t gdb_uset_next_bkt(GdbUSetOrder& s, uint64_t n) {
  const unsigned char fast[14] = {2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13};
  if (n < 14) { if (n == 0) return 1; s.next_resize = fast[n]; return fast[n]; }
  const uint16_t primes[] = {17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113, 127, 137, 139, 149, 157, 167, 179, 193, 199,
                             211, 227, 241, 257, 277, 281, 283, 293, 307, 311, 313, 337, 347, 353, 359, 379, 389, 401, 409, 419, 431, 433, 449, 457, 461, 463, 467, 479, 487, 491, 499,
                             503, 509, 521, 523, 541};
  for (unsigned i = 0; i < sizeof(primes) / sizeof(primes[0]); ++i) if (primes[i] >= n) { s.next_resize = primes[i]; return primes[i]; }
  s.overflow = true;             // a range hint beyond anything a FILTER vector can have
  return n;
}
GDB_HD void gdb_uset_place(int32_t* list, int n, int32_t key, uint64_t nbkt) {   // n elements in list; key is not among them
  const uint64_t b = (uint64_t)(int64_t)key % nbkt;
  int at = 0;
  for (int i = 0; i < n; ++i) if ((uint64_t)(int64_t)list[i] % nbkt == b) { at = i; break; }
  for (int j = n; j > at; --j) list[j] = list[j - 1];
  list[at] = key;
}
GDB_HD void gdb_uset_insert_range(GdbUSetOrder& s, const int32_t* p, int count) {
  const uint64_t hint = 1;       // (GCC 11: a range insert into a container with unique keys is a loop of single inserts, hashtable_policy.h:900-906)
  for (int i = 0; i < count; ++i) {
    const int32_t k = p[i];
    bool found = false;
    for (int j = 0; j < s.n; ++j) found |= s.key[j] == k;
    if (found) continue;
    if (s.n >= GDB_MAX_FILTER_IDS) { s.overflow = true; return; }
    if ((uint64_t)s.n + hint > s.next_resize) {     // _M_need_rehash
      const uint64_t want = (uint64_t)s.n + hint;
      const uint64_t min_bkts = s.next_resize ? want : (want > 11 ? want : 11);
      if (min_bkts >= (uint64_t)s.nbkt) {
        const uint64_t nb = gdb_uset_next_bkt(s, (min_bkts + 1 > (uint64_t)s.nbkt * 2) ? min_bkts + 1 : (uint64_t)s.nbkt * 2);
        int32_t old[GDB_MAX_FILTER_IDS];
        for (int j = 0; j < s.n; ++j) old[j] = s.key[j];
        for (int j = 0; j < s.n; ++j) gdb_uset_place(s.key, j, old[j], nb);
        s.nbkt = (int32_t)nb;
      } else s.next_resize = (uint64_t)s.nbkt;
    }
    gdb_uset_place(s.key, s.n, k, (uint64_t)s.nbkt);
    ++s.n;
  }
}

// ---- ID union of a Release build: the iteration order of libstdc++'s std::unordered_set<std::string> -----------------------------
// merge_ID_field keeps the ';'-separated tokens of the live calls in a std::set only #ifdef DEBUG (the build the goldens come
// from); every other build - the one GATK ships - uses a std::unordered_set<std::string> and writes the tokens in ITS iteration
// order (broad_combined_gvcf.cc:730-763).  Restated for libstdc++ on a 64-bit target: std::hash<std::string> = _Hash_bytes
// (libsupc++ hash_bytes.cc: the 64-bit Murmur-style mix, seed 0xc70f6907), bucket = hash % bucket_count, list / rehash rules as
// for the FILTER ids above (single inserts: 13 buckets for the first 13 tokens, 29 from the 14th).  Checked token-sequence-for-
// token-sequence against the library on the host (tests/hostsim: hostsim_id_union_order, tests/test_id_union_order.py).
GDB_HD uint64_t gdb_libstdcxx_hash_bytes(const char* p, int len) {
  const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
  const int full = len & ~7;
  uint64_t hash = (uint64_t)0xc70f6907UL ^ ((uint64_t)len * mul);
  for (int i = 0; i < full; i += 8) {
    uint64_t w = 0;
    for (int b = 7; b >= 0; --b) w = (w << 8) | (uint64_t)(unsigned char)p[i + b];     // little-endian unaligned load
    w *= mul; w ^= w >> 47; w *= mul;
    hash ^= w; hash *= mul;
  }
  if (len & 7) {
    uint64_t w = 0;
    for (int b = (len & 7) - 1; b >= 0; --b) w = (w << 8) + (uint64_t)(unsigned char)p[full + b];
    hash ^= w; hash *= mul;
  }
  hash ^= hash >> 47; hash *= mul; hash ^= hash >> 47;
  return hash;
}
struct GdbUSetHashOrder {
  uint64_t hash[GDB_MAX_ID_TOKENS];  // list order = iteration order
  int32_t tok[GDB_MAX_ID_TOKENS];    // the caller's token number of each list element
  int32_t n, nbkt;
  uint64_t next_resize;
};
GDB_HD void gdb_useth_init(GdbUSetHashOrder& s) { s.n = 0; s.nbkt = 1; s.next_resize = 0; }
GDB_HD void gdb_useth_place(GdbUSetHashOrder& s, int n, uint64_t h, int32_t tok, uint64_t nbkt) {   // n elements in the list
  const uint64_t b = h % nbkt;
  int at = 0;
  for (int i = 0; i < n; ++i) if (s.hash[i] % nbkt == b) { at = i; break; }
  for (int j = n; j > at; --j) { s.hash[j] = s.hash[j - 1]; s.tok[j] = s.tok[j - 1]; }
  s.hash[at] = h; s.tok[at] = tok;
}
GDB_HD void gdb_useth_insert_new(GdbUSetHashOrder& s, uint64_t h, int32_t tok) {   // a key that is not in the set; s.n < GDB_MAX_ID_TOKENS
  if ((uint64_t)s.n + 1 > s.next_resize) {        // _Prime_rehash_policy::_M_need_rehash(n_bkt, n_elt, 1)
    const uint64_t want = (uint64_t)s.n + 1;
    const uint64_t min_bkts = s.next_resize ? want : (want > 11 ? want : 11);
    if (min_bkts >= (uint64_t)s.nbkt) {
      GdbUSetOrder pol;                            // (only its bucket-count policy is used)
      pol.next_resize = s.next_resize; pol.overflow = false;
      const uint64_t nb = gdb_uset_next_bkt(pol, (min_bkts + 1 > (uint64_t)s.nbkt * 2) ? min_bkts + 1 : (uint64_t)s.nbkt * 2);
      s.next_resize = pol.next_resize;
      uint64_t oh[GDB_MAX_ID_TOKENS]; int32_t ot[GDB_MAX_ID_TOKENS];
      for (int j = 0; j < s.n; ++j) { oh[j] = s.hash[j]; ot[j] = s.tok[j]; }
      for (int j = 0; j < s.n; ++j) gdb_useth_place(s, j, oh[j], ot[j], nb);
      s.nbkt = (int32_t)nb;
    } else s.next_resize = (uint64_t)s.nbkt;
  }
  gdb_useth_place(s, s.n, h, tok, (uint64_t)s.nbkt);
  ++s.n;
}
// Order of the ID tokens of one record.  release_order == 0: sorted (std::set<std::string>, the DEBUG build);
// != 0: libstdc++'s unordered_set.  tp / tn: the distinct tokens, written in the order they are to be printed.
GDB_HD void gdb_id_union_add(const char* p, int len, const char** tp, int* tn, int& nt, GdbUSetHashOrder& us, int release_order, uint32_t* err) {
  if (release_order) {
    for (int i = 0; i < nt; ++i) {                // (find: a token that is already in the set changes nothing)
      if (tn[i] != len) continue;
      bool same = true;
      for (int j = 0; j < len && same; ++j) same = tp[i][j] == p[j];
      if (same) return;
    }
    if (nt >= GDB_MAX_ID_TOKENS) { *err |= GDB_ERR_TOO_MANY_ID_TOKENS; return; }
    tp[nt] = p; tn[nt] = len;
    gdb_useth_insert_new(us, gdb_libstdcxx_hash_bytes(p, len), nt);
    ++nt;
    return;
  }
  int pos = 0, cmp = 1;                           // insertion point in the sorted token list
  for (; pos < nt; ++pos) {
    const int m = len < tn[pos] ? len : tn[pos];
    cmp = 0;
    for (int j = 0; j < m && cmp == 0; ++j) cmp = (int)(unsigned char)p[j] - (int)(unsigned char)tp[pos][j];
    if (cmp == 0) cmp = len - tn[pos];
    if (cmp <= 0) break;
  }
  if (pos == nt || cmp != 0) {
    if (nt >= GDB_MAX_ID_TOKENS) { *err |= GDB_ERR_TOO_MANY_ID_TOKENS; }
    else { for (int j = nt; j > pos; --j) { tp[j] = tp[j - 1]; tn[j] = tn[j - 1]; } tp[pos] = p; tn[pos] = len; ++nt; }
  }
}
// the tokens of one ID value (merge_ID_field's loop: a token in front of every ';', and what is left behind the last one)
GDB_HD void gdb_id_union_value(const char* p, int n, const char** tp, int* tn, int& nt, GdbUSetHashOrder& us, int release_order, uint32_t* err) {
  int last = 0;
  for (int i = 0; i <= n; ++i) {
    if (i < n && p[i] != ';') continue;
    const int len = i - last;
    if (i == n && len == 0) break;                // nothing behind a trailing ';'
    gdb_id_union_add(p + last, len, tp, tn, nt, us, release_order, err);
    last = i + 1;
  }
}
GDB_HD int gdb_id_union_at(const GdbUSetHashOrder& us, int release_order, int i) { return release_order ? us.tok[i] : i; }

// scratch of the (rare) tied-zero medians: reduce_scalar takes n_valid floats per use; capacity = passes x float median
// fields x heavy incidences, so it cannot run out
struct TieScratch {
  float* buf;
  unsigned long long* used;
  uint64_t capacity;
};

// scalar INFO-like values per (record, variant call) incidence, gathered once by a kernel before the site pass:
// val[slot * stride + t] = raw 32 bits of the value call t contributes to the field, or the bcf missing pattern of the field's
// type when the call does not count (field absent, value missing / vector end, spanning deletion for the INFO fields)
struct ScalarPre {
  const uint32_t* val;
  int64_t stride;
  int8_t slot[GDB_MAX_FIELDS];
  int32_t enabled;
};

// Records with very many variant calls (the hot sites of a dense region, the first record of a partition): everything in the
// record logic that walks the calls - allele merge + LUTs, scalar reducers, FILTER - is done beforehand by one WORKGROUP per
// record (k_site_huge: lane = call) and left here; site_emit then only formats.
struct HugeScalar { int64_t nvalid, nbelow, nneg0, npos0; uint32_t sum_bits, median_bits; int32_t median_ok, pad; };   // median_ok: 0 none, 1 value, 2 zeros of both signs tie
struct HugeSiteOut {
  AlleleRef merged[GDB_MAX_MERGED_ALLELES];
  const char* mref;            // longest REF among the calls starting at the record (null: take the reference base)
  int32_t mref_len, nmerged;
  int32_t filter_first_id, filter_multi, any_id, fallback;   // fallback != 0: a hash collision between two alleles - use the serial walk
  HugeScalar scalar[GDB_MAX_INFO_FIELDS + 2];                // by ScalarPre slot
};
struct HugeSites {
  const int32_t* index;        // [P] row of `out` or -1
  const HugeSiteOut* out;
  int32_t enabled;
};

struct SiteCtx {
  FragmentView fr;
  CombinePlan pl;
  CellMeta cm;
  RecordTable rec;
  HeavyLists hl;
  PresenceCounts pc;
  NameTables names;
  QueryWindow qw;
  SiteOut so;
  // optional (records with many variant calls): per median field, the incidences of every record ordered by value
  MedianOrder med;
  BigMedians big;
  TieScratch tie;
  ScalarPre pre;
  HugeSites huge;
};

// value of a scalar INFO-like field over the heavy list: median / sum / mean (variant_field_handler.cc:529-607).
// Spanning-deletion calls have their INFO fields invalidated (broad_combined_gvcf.cc:1068-1075) unless keep_spanning.
GDB_HD bool inc_is_spanning(const SiteCtx& cx, int64_t t, int64_t s_k) {
  int64_t c = cx.hl.cell[t];
  return (cx.cm.cflags[c] & GDB_CF_DELETION) && s_k > cx.fr.begin[c];
}
// the value call t of record k contributes to scalar field f, false when it does not count (field absent, value missing,
// spanning deletion unless keep_spanning).  With the gathered table (ScalarPre) this is one load from a contiguous array;
// without it (CPU harness) the cell is consulted.
// `alt_merged` > 0: an A-length field of a record whose alleles were merged (alt_merged = merged alleles).  The reference's scalar reducers
// read element 0 of the REMAPPED vector (handle_VCF_field_combine_operation picks m_remapped_variant for allele-dependent fields,
// broad_combined_gvcf.cc:386-390; get_valid_sum / median read ptr->get()[0], variant_field_handler.cc:529-607): the call's value for the
// FIRST MERGED ALT allele - its own allele of that name, else its <NON_REF>, else missing.  (Element 0 of an R- or G-length vector is the
// REF allele / the all-REF genotype in both orders.)
template <class T> GDB_HD bool scalar_at(const SiteCtx& cx, const uint32_t* pre, int64_t t, int64_t s_k, int f, bool keep_spanning, bool is_float, T& v,
                                         int alt_merged = 0, bool non_ref_exists = false) {
  if (pre) {
    union { uint32_t u; T v; } x;
    x.u = pre[t];
    if (is_float ? x.u == GDB_BCF_FLOAT_MISSING_BITS : x.u == (uint32_t)GDB_BCF_INT32_MISSING) return false;
    v = x.v;
    return true;
  }
  if (!keep_spanning && inc_is_spanning(cx, t, s_k)) return false;
  const int64_t c = cx.hl.cell[t];
  if (!field_valid(cx.cm, c, f)) return false;
  int n;
  const T* p = cell_field<T>(cx.fr, cx.pl, f, c, n);
  int idx = 0;
  if (alt_merged > 0) {
    const int8_t* lut = cx.hl.i2m + cx.hl.i2m_off[t];
    const int nal = (int)GDB_CF_NALT(cx.cm.cflags[c]) + 1;
    int in = -1;
    for (int a = 0; a < nal; ++a) if (lut[a] == 1) { in = a; break; }
    if (in < 0 && non_ref_exists) for (int a = 0; a < nal; ++a) if (lut[a] == alt_merged - 1) in = a;
    idx = in - 1;
    if (in < 0 || idx < 0 || idx >= n) return false;
  }
  v = p[idx];
  return is_float ? gdb_float_valid((float)v) : gdb_int_valid((int32_t)v);
}
template <class T> GDB_HD bool reduce_scalar(const SiteCtx& cx, int64_t k, int f, int op, bool keep_spanning, T& result, uint32_t* err, int num_merged = 0,
                                             bool non_ref_exists = false, bool remapping_needed = false) {
  const int alt_merged = (remapping_needed && cx.pl.field[f].length == GDB_VL_A) ? num_merged : 0;   // (such fields have none of the precomputed helpers: combine_plan / prepare_interval)
  const int64_t b = cx.hl.base[k], e = cx.hl.base[k + 1];
  if (op == GDB_OP_MEDIAN && cx.big.enabled && cx.big.slot[f] >= 0 && cx.big.index[k] >= 0) {
    const int64_t at = (int64_t)cx.big.slot[f] * cx.big.stride + cx.big.index[k];
    if (!cx.big.ok[at]) return false;
    if (cx.big.ok[at] == 1) {                       // (2: -0 and +0 tie at the middle, resolved below)
      union { uint32_t u; T v; } x;
      x.u = cx.big.value[at];
      result = x.v;
      return true;
    }
  }
  const int64_t s_k = cx.rec.start[k];
  const bool is_float = cx.pl.field[f].elem == GDB_ET_FLOAT;
  const uint32_t* pre = (cx.pre.enabled && cx.pre.slot[f] >= 0) ? cx.pre.val + (int64_t)cx.pre.slot[f] * cx.pre.stride : nullptr;
  int64_t nvalid = 0, nbelow = 0, nneg0 = 0, npos0 = 0;
  T sum = 0;
  const HugeSiteOut* hs = (cx.huge.enabled && pre && cx.huge.index[k] >= 0 && !cx.huge.out[cx.huge.index[k]].fallback) ? &cx.huge.out[cx.huge.index[k]] : nullptr;
  if (hs) {   // a record with very many calls: counts, sum and median were taken by its workgroup (k_site_huge)
    const HugeScalar& hv = hs->scalar[cx.pre.slot[f]];
    nvalid = hv.nvalid; nbelow = hv.nbelow; nneg0 = hv.nneg0; npos0 = hv.npos0;
    union { uint32_t u; T v; } x;
    x.u = hv.sum_bits; sum = x.v;
    if (nvalid && op == GDB_OP_MEDIAN && hv.median_ok == 1) { x.u = hv.median_bits; result = x.v; return true; }
  } else
  for (int64_t t = b; t < e; ++t) {
    T v;
    if (!scalar_at<T>(cx, pre, t, s_k, f, keep_spanning, is_float, v, alt_merged, non_ref_exists)) continue;
    sum += v;
    ++nvalid;
    if (is_float) {
      const uint32_t u = gdb_f2u((float)v);
      nneg0 += u == 0x80000000u;
      npos0 += u == 0u;
      nbelow += v < (T)0;
    }
  }
  if (!nvalid) return false;
  if (op == GDB_OP_SUM) { result = sum; return true; }
  if (op == GDB_OP_MEAN) {
    // get_valid_mean divides the sum by an `unsigned` count (variant_field_handler.cc:596-607): for an int field the usual
    // arithmetic conversions make the SUM unsigned too, so a negative sum is divided as 2^32 + sum ((-6) / 3 prints 1431655763)
    if (is_float) result = sum / (T)nvalid;
    else result = (T)(int32_t)((uint32_t)(int32_t)sum / (uint32_t)nvalid);
    return true;
  }
  if (is_float && nneg0 && npos0 && nbelow <= nvalid / 2 && nvalid / 2 < nbelow + nneg0 + npos0) {
    // the median is a zero and both signs are present: which one the reference prints is decided by its nth_element
    const uint64_t at = cx.tie.buf ? GDB_FETCH_ADD_U64(cx.tie.used, (uint64_t)nvalid) : 0;
    if (!cx.tie.buf || at + (uint64_t)nvalid > cx.tie.capacity) { *err |= GDB_ERR_INTERNAL; return false; }
    float* a = cx.tie.buf + at;
    int64_t m = 0;
    for (int64_t t = b; t < e; ++t) {
      T v;
      if (scalar_at<T>(cx, pre, t, s_k, f, keep_spanning, is_float, v, alt_merged, non_ref_exists)) a[m++] = (float)v;
    }
    result = (T)gdb_nth_element_libstdcxx(a, m, m / 2);
    return true;
  }
  if (cx.med.enabled && cx.med.slot[f] >= 0) {
    // sorted flavour: entry of rank nvalid/2 among the valid ones; among equal values the first in row order, like the scan below
    const uint64_t* keys = cx.med.keys + (int64_t)cx.med.slot[f] * cx.med.stride;
    const uint32_t* inc = cx.med.inc + (int64_t)cx.med.slot[f] * cx.med.stride;
    const uint64_t want = keys[b + nvalid / 2];
    int64_t lo = b, hi = b + nvalid / 2;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
    const int64_t t = inc[lo];
    int n;
    const T* p = cell_field<T>(cx.fr, cx.pl, f, cx.hl.cell[t], n);
    result = p[0];
    return true;
  }
  // median = element of rank nvalid/2 in ascending order (std::nth_element at mid_point)
  int64_t mid = nvalid / 2;
  for (int64_t t = b; t < e; ++t) {
    T v;
    if (!scalar_at<T>(cx, pre, t, s_k, f, keep_spanning, is_float, v, alt_merged, non_ref_exists)) continue;
    int64_t less = 0, leq = 0;
    for (int64_t u = b; u < e; ++u) {
      T w;
      if (!scalar_at<T>(cx, pre, u, s_k, f, keep_spanning, is_float, w, alt_merged, non_ref_exists)) continue;
      if (w < v) ++less;
      if (w <= v) ++leq;
    }
    if (less <= mid && mid < leq) { result = v; return true; }
  }
  return false;
}

// element helpers shared by the int32 and float flavours of the vector emitters
GDB_HD bool elem_is_vector_end(int32_t v) { return v == GDB_BCF_INT32_VECTOR_END; }
GDB_HD bool elem_is_vector_end(float v) { return gdb_f2u(v) == GDB_BCF_FLOAT_VECTOR_END_BITS; }
GDB_HD bool elem_is_missing(int32_t v) { return v == GDB_BCF_INT32_MISSING; }
GDB_HD bool elem_is_missing(float v) { return gdb_f2u(v) == GDB_BCF_FLOAT_MISSING_BITS; }
GDB_HD void elem_set_missing(int32_t& v) { v = GDB_BCF_INT32_MISSING; }
GDB_HD void elem_set_missing(float& v) { union { uint32_t u; float f; } x; x.u = GDB_BCF_FLOAT_MISSING_BITS; v = x.f; }
// text of an element in a register (first character in the low byte), or 0 when it does not fit the packed form
GDB_HD int elem_pack_text(int32_t v, uint64_t& w) {
  if (v == GDB_BCF_INT32_MISSING) { w = (uint64_t)'.'; return 1; }
  if (v < 0 || v >= 100000000) return 0;
  int n;
  w = gdb_pack_digits((uint32_t)v, n);
  return n;
}
GDB_HD int elem_pack_text(float, uint64_t&) { return 0; }
template <class Sink> GDB_HD void put_elem(Sink& s, int32_t v, uint32_t*) { if (elem_is_missing(v)) s.put('.'); else put_i32(s, v); }
template <class Sink> GDB_HD void put_elem(Sink& s, float v, uint32_t* err) {
  if (elem_is_missing(v)) s.put('.');
  else put_float(s, v);
}

// The elements of one call's INFO vector as the reducers below see them: as stored (mode 0), or - a record whose alleles were
// merged - through the call's allele LUT like the reference's remapped variant: R / A length in merged-allele order (modes 1 / 2,
// remap_data_based_on_alleles, variant_field_handler.cc:104-132), G length in merged-genotype order (mode 3,
// remap_data_based_on_genotype, :134-297: haploid by allele, diploid through bcf_alleles2gt(input j, input k) - in this argument order -
// any other ploidy by enumeration).  A merged allele the call does not have reads its <NON_REF>; no such input: missing.
// fn(i, v) -> false stops the walk.
template <class T, class Fn> GDB_HD void info_for_each_element(const T* p, int n_in, int mode, const int8_t* lut, int nal, int num_merged, bool non_ref_exists, int ploidy,
                                                             uint32_t* err, Fn&& fn) {
  if (mode == 0) { for (int i = 0; i < n_in; ++i) if (!fn(i, p[i])) return; return; }
  int nr_in = -1;
  if (non_ref_exists) for (int a = 0; a < nal; ++a) if (lut[a] == num_merged - 1) nr_in = a;
  const auto look = [&](int aj) -> int { for (int a = 0; a < nal; ++a) if (lut[a] == aj) return a; return nr_in; };
  T miss; elem_set_missing(miss);
  if (mode == 1 || mode == 2) {
    const bool alt_only = mode == 2;
    const int n = alt_only ? num_merged - 1 : num_merged;
    for (int i = 0; i < n; ++i) {
      const int in = look(alt_only ? i + 1 : i);
      const int idx = alt_only ? in - 1 : in;
      if (!fn(i, (in >= 0 && idx >= 0 && idx < n_in) ? p[idx] : miss)) return;
    }
    return;
  }
  if (ploidy == 1) {
    for (int j = 0; j < num_merged; ++j) { const int in = look(j); if (!fn(j, (in >= 0 && in < n_in) ? p[in] : miss)) return; }
  } else if (ploidy == 2) {
    int i = 0;
    for (int kk = 0; kk < num_merged; ++kk) {
      const int in_k = look(kk);
      for (int j = 0; j <= kk; ++j, ++i) {
        const int in_j = look(j);
        const bool both = in_j >= 0 && in_k >= 0;
        const int gi = both ? gdb_alleles2gt(in_j, in_k) : 0;
        if (!fn(i, (both && gi < n_in) ? p[gi] : miss)) return;
      }
    }
  } else if (ploidy >= 3 && ploidy <= GDB_MAX_PLOIDY) {
    int g[GDB_MAX_PLOIDY], in[GDB_MAX_PLOIDY];
    for (int q = 0; q < ploidy; ++q) g[q] = 0;
    int i = 0;
    do {
      bool missing = false;
      for (int q = 0; q < ploidy; ++q) { in[q] = look(g[q]); if (in[q] < 0) missing = true; }
      T v = miss;
      if (!missing) { const int64_t gi = gdb_genotype_index(in, ploidy); if (gi < n_in) v = p[gi]; }
      if (!fn(i++, v)) return;
    } while (gdb_next_genotype(g, ploidy, num_merged));
  } else *err |= GDB_ERR_UNSUPPORTED_PLOIDY;      // (no GT in the call: the reference sizes the vector for ploidy 0; not on the device path)
}
GDB_HD int info_element_mode(const GdbFieldDesc& fd, bool remapping_needed) {
  if (!remapping_needed) return 0;
  return fd.length == GDB_VL_R ? 1 : fd.length == GDB_VL_A ? 2 : fd.length == GDB_VL_G ? 3 : 0;
}

// element_wise_sum / concatenate INFO combiners (handle_VCF_field_combine_operation, broad_combined_gvcf.cc:374-429;
// compute_valid_element_wise_sum, variant_field_handler.cc:618-664).  Allele-dependent fields (A / R length) are read
// through the call's allele LUT, i.e. in merged-allele order with the <NON_REF> fallback, like the reference reads them from
// its remapped variant.  Values live in a small per-record array: more than GDB_MAX_INFO_VECTOR elements raise an error bit.
template <class T, class Sink> GDB_HD bool info_vector_combine(const SiteCtx& cx, int64_t k, int f, int op, int num_merged, bool non_ref_exists, bool remapping_needed,
                                                             Sink& sink, bool& any, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  const GdbFieldDesc& fd = pl.field[f];
  const int64_t b = cx.hl.base[k], e = cx.hl.base[k + 1];
  const int64_t s_k = cx.rec.start[k];
  const int mode = info_element_mode(fd, remapping_needed);
  T r[GDB_MAX_INFO_VECTOR];
  int num_valid = 0, nconcat = 0;
  bool name_written = false;
  for (int64_t t = b; t < e; ++t) {
    if (inc_is_spanning(cx, t, s_k)) continue;           // INFO of spanning deletions is invalidated
    const int64_t c = cx.hl.cell[t];
    if (!field_valid(cx.cm, c, f)) continue;
    int n_in;
    const T* p = cell_field<T>(cx.fr, pl, f, c, n_in);
    bool stop = false;
    info_for_each_element<T>(p, n_in, mode, cx.hl.i2m + cx.hl.i2m_off[t], (int)GDB_CF_NALT(cx.cm.cflags[c]) + 1, num_merged, non_ref_exists, (int)GDB_CF_PLOIDY(cx.cm.cflags[c]), err,
                             [&](int i, T v) -> bool {
      if (op == GDB_OP_CONCATENATE) {
        if (!name_written) {
          if (any) sink.put(';');
          sink.write(cx.names.text + cx.names.field_name_off[f], cx.names.field_name_len[f]);
          sink.put('=');
          name_written = true; any = true;
        }
        if (elem_is_vector_end(v)) { stop = true; return false; }           // htslib stops printing a vector at vector_end
        if (nconcat++) sink.put(',');
        put_elem(sink, v, err);
        return true;
      }
      if (elem_is_missing(v) || elem_is_vector_end(v)) return true;
      if (i >= GDB_MAX_INFO_VECTOR) { *err |= GDB_ERR_INFO_VECTOR_TOO_LONG; return false; }
      if (i < num_valid && !elem_is_missing(r[i])) r[i] += v;
      else { r[i] = v; if (i >= num_valid) { for (int j = num_valid; j < i; ++j) elem_set_missing(r[j]); num_valid = i + 1; } }
      return true;
    });
    if (stop) return true;
  }
  if (op == GDB_OP_CONCATENATE) return name_written;
  if (num_valid == 0) return false;
  if (any) sink.put(';');
  sink.write(cx.names.text + cx.names.field_name_off[f], cx.names.field_name_len[f]);
  sink.put('=');
  for (int i = 0; i < num_valid; ++i) { if (i) sink.put(','); put_elem(sink, r[i], err); }
  any = true;
  return true;
}

// ---- BCF2 typed values (BCFv2.2 specification 6.3; htslib vcf.c: bcf_enc_size, bcf_enc_int1, bcf_enc_vint, bcf_enc_vchar) ---------
#define GDB_BT_NULL 0
#define GDB_BT_INT8 1
#define GDB_BT_INT16 2
#define GDB_BT_INT32 3
#define GDB_BT_FLOAT 5
#define GDB_BT_CHAR 7
template <class Sink> GDB_HD void bcf_put_u16(Sink& s, uint32_t v) { s.put((char)(v & 0xFFu)); s.put((char)((v >> 8) & 0xFFu)); }
template <class Sink> GDB_HD void bcf_put_u32(Sink& s, uint32_t v) { for (int i = 0; i < 4; ++i) s.put((char)((v >> (8 * i)) & 0xFFu)); }
// smallest integer type that holds every value of [mn, mx]; the 8 lowest codes of every type are reserved (missing, vector end, ...)
GDB_HD int bcf_int_type(int32_t mn, int32_t mx) {
  if (mx <= 127 && mn >= -120) return GDB_BT_INT8;
  if (mx <= 32767 && mn >= -32760) return GDB_BT_INT16;
  return GDB_BT_INT32;
}
GDB_HD int bcf_type_width(int t) { return t == GDB_BT_INT8 || t == GDB_BT_CHAR ? 1 : t == GDB_BT_INT16 ? 2 : t == GDB_BT_NULL ? 0 : 4; }
// one int32 (with the int32 missing / vector-end patterns) as an element of type t
template <class Sink> GDB_HD void bcf_put_int(Sink& s, int32_t v, int t) {
  if (t == GDB_BT_INT8) s.put(v == GDB_BCF_INT32_MISSING ? (char)0x80 : v == GDB_BCF_INT32_VECTOR_END ? (char)0x81 : (char)v);
  else if (t == GDB_BT_INT16) bcf_put_u16(s, v == GDB_BCF_INT32_MISSING ? 0x8000u : v == GDB_BCF_INT32_VECTOR_END ? 0x8001u : (uint32_t)v);
  else bcf_put_u32(s, (uint32_t)v);
}
template <class Sink> GDB_HD void bcf_enc_int1(Sink& s, int32_t x) {
  const int t = (x == GDB_BCF_INT32_MISSING || x == GDB_BCF_INT32_VECTOR_END) ? GDB_BT_INT8 : bcf_int_type(x, x);
  s.put((char)((1 << 4) | t));
  bcf_put_int(s, x, t);
}
template <class Sink> GDB_HD void bcf_enc_size(Sink& s, int size, int type) {
  if (size >= 15) { s.put((char)((15 << 4) | type)); bcf_enc_int1(s, size); }
  else s.put((char)((size << 4) | type));
}
GDB_HD int bcf_enc_size_bytes(int size) { return size < 15 ? 1 : size <= 127 ? 3 : size <= 32767 ? 4 : 6; }
GDB_HD int bcf_enc_int1_bytes(int32_t x) { return 1 + bcf_type_width((x == GDB_BCF_INT32_MISSING || x == GDB_BCF_INT32_VECTOR_END) ? GDB_BT_INT8 : bcf_int_type(x, x)); }
// bcf_enc_vint: n values, one common type chosen over the values that are neither missing nor vector end
template <class Sink> GDB_HD void bcf_enc_vint(Sink& s, const int32_t* a, int n) {
  if (n <= 0) { s.put((char)0); return; }
  if (n == 1) { bcf_enc_int1(s, a[0]); return; }
  int32_t mx = INT32_MIN + 1, mn = INT32_MAX;
  for (int i = 0; i < n; ++i) { if (a[i] == GDB_BCF_INT32_MISSING || a[i] == GDB_BCF_INT32_VECTOR_END) continue; if (mx < a[i]) mx = a[i]; if (mn > a[i]) mn = a[i]; }
  const int t = bcf_int_type(mn, mx);
  bcf_enc_size(s, n, t);
  for (int i = 0; i < n; ++i) bcf_put_int(s, a[i], t);
}

// the record's FILTER ids in the order the reference writes them: one range insert per live call with a valid FILTER field, in
// call order (broad_combined_gvcf.cc:852-861)
template <class Ctx>
GDB_HD_NOINLINE void filter_union_order(const Ctx& cx, int64_t hb, int64_t he, GdbUSetOrder& us, uint32_t* err) {
  gdb_uset_init(us);
  for (int64_t t = hb; t < he; ++t) {
    const int64_t c = cx.hl.cell[t];
    if (!field_valid(cx.cm, c, cx.pl.f_FILTER)) continue;
    int n;
    const int32_t* p = cell_field<int32_t>(cx.fr, cx.pl, cx.pl.f_FILTER, c, n);
    gdb_uset_insert_range(us, p, n);
  }
  if (us.overflow) *err |= GDB_ERR_TOO_MANY_FILTER_IDS;
}

GDB_HD int find_contig(const QueryWindow& qw, int64_t pos) {  // VidMapper::get_contig_location
  int lo = 0, hi = qw.ncontigs;  // last contig with offset <= pos
  while (lo < hi) { int mid = (lo + hi) >> 1; if (qw.contigs[mid].offset <= pos) lo = mid + 1; else hi = mid; }
  int idx = lo - 1;
  if (idx < 0) return -1;
  if (pos >= qw.contigs[idx].offset && pos < qw.contigs[idx].offset + qw.contigs[idx].length) return idx;
  return -1;
}

// the same combiners as typed BCF INFO values: key, then the vector (element_wise_sum: the summed elements; concatenate: every
// element of every valid call in call order - vector-end elements included, a text writer stops at the first one)
template <class T, class Sink> GDB_HD bool info_vector_combine_bcf(const SiteCtx& cx, int64_t k, int f, int op, int num_merged, bool non_ref_exists, bool remapping_needed,
                                                                 Sink& sink, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  const GdbFieldDesc& fd = pl.field[f];
  const int64_t b = cx.hl.base[k], e = cx.hl.base[k + 1];
  const int64_t s_k = cx.rec.start[k];
  const int mode = info_element_mode(fd, remapping_needed);
  const bool is_float = fd.elem == GDB_ET_FLOAT;
  T r[GDB_MAX_INFO_VECTOR];
  int num_valid = 0;
  // pass 0 (concatenate only): element count and integer range; pass 1: values
  int n_total = 0;
  int32_t mn = INT32_MAX, mx = INT32_MIN + 1;
  int type = is_float ? GDB_BT_FLOAT : GDB_BT_INT8;
  for (int pass = (op == GDB_OP_CONCATENATE ? 0 : 1); pass < 2; ++pass) {
    if (pass == 1 && op == GDB_OP_CONCATENATE) {
      if (n_total == 0) return false;
      bcf_enc_int1(sink, pl.bcf_id[f]);
      if (!is_float) type = bcf_int_type(mn, mx);
      if (!is_float && n_total == 1) {}   // (bcf_enc_vint encodes a single value as a typed scalar: same bytes as a vector of one)
      bcf_enc_size(sink, n_total, type);
    }
    for (int64_t t = b; t < e; ++t) {
      if (inc_is_spanning(cx, t, s_k)) continue;
      const int64_t c = cx.hl.cell[t];
      if (!field_valid(cx.cm, c, f)) continue;
      int n_in;
      const T* p = cell_field<T>(cx.fr, pl, f, c, n_in);
      info_for_each_element<T>(p, n_in, mode, cx.hl.i2m + cx.hl.i2m_off[t], (int)GDB_CF_NALT(cx.cm.cflags[c]) + 1, num_merged, non_ref_exists, (int)GDB_CF_PLOIDY(cx.cm.cflags[c]), err,
                               [&](int i, T v) -> bool {
        if (op == GDB_OP_CONCATENATE) {
          if (pass == 0) {
            ++n_total;
            if (!is_float && !elem_is_missing(v) && !elem_is_vector_end(v)) { const int32_t iv = (int32_t)v; if (iv < mn) mn = iv; if (iv > mx) mx = iv; }
          } else if (is_float) { union { T t; uint32_t u; } x; x.u = 0; x.t = v; bcf_put_u32(sink, x.u); }
          else bcf_put_int(sink, (int32_t)v, type);
          return true;
        }
        if (elem_is_missing(v) || elem_is_vector_end(v)) return true;
        if (i >= GDB_MAX_INFO_VECTOR) { *err |= GDB_ERR_INFO_VECTOR_TOO_LONG; return false; }
        if (i < num_valid && !elem_is_missing(r[i])) r[i] += v;
        else { r[i] = v; if (i >= num_valid) { for (int j = num_valid; j < i; ++j) elem_set_missing(r[j]); num_valid = i + 1; } }
        return true;
      });
    }
  }
  if (op == GDB_OP_CONCATENATE) return true;
  if (num_valid == 0) return false;
  bcf_enc_int1(sink, pl.bcf_id[f]);
  if (is_float) {
    bcf_enc_size(sink, num_valid, GDB_BT_FLOAT);
    for (int i = 0; i < num_valid; ++i) { union { T t; uint32_t u; } x; x.u = 0; x.t = r[i]; bcf_put_u32(sink, x.u); }
  } else {
    int32_t tmp[GDB_MAX_INFO_VECTOR];
    for (int i = 0; i < num_valid; ++i) tmp[i] = (int32_t)r[i];
    bcf_enc_vint(sink, tmp, num_valid);
  }
  return true;
}

#include "gdb_asa.hpp"

// Per-record site logic.  PASS 0 (Sink = CountSink, write_luts = false) sizes the prefix; PASS 1 writes the prefix
// text and the per-incidence allele LUTs / flags.  One call handles one record.
// Position of `cand` in the merged allele list, appended when it is new (CombineAllelesLUT / merge_alt_alleles order: first
// appearance).  Short lists are scanned; from kMergeScan entries on a 256-entry open-addressing index over the allele hashes
// (built when the list first grows past the limit) replaces the scan - sites where thousands of calls share a few dozen
// alleles would otherwise spend their time walking the list once per call.
constexpr int kMergeScan = 8;
struct MergeIndex { uint8_t slot[256]; bool built; };
GDB_HD int merged_find_or_add(AlleleRef* merged, int& nmerged, MergeIndex& ix, const AlleleRef& cand, const char* mref, int mref_len, uint32_t* err) {
  if (nmerged <= kMergeScan) {
    for (int j = 1; j < nmerged; ++j) if (allele_equal(merged[j], cand, mref, mref_len)) return j;
  } else {
    if (!ix.built) {
      for (int h = 0; h < 256; ++h) ix.slot[h] = 0;
      for (int j = 1; j < nmerged; ++j) { uint32_t h = merged[j].hash & 255u; while (ix.slot[h]) h = (h + 1) & 255u; ix.slot[h] = (uint8_t)j; }
      ix.built = true;
    }
    for (uint32_t h = cand.hash & 255u; ix.slot[h]; h = (h + 1) & 255u)
      if (allele_equal(merged[ix.slot[h]], cand, mref, mref_len)) return ix.slot[h];
  }
  if (nmerged >= GDB_MAX_MERGED_ALLELES - 1) { *err |= GDB_ERR_TOO_MANY_MERGED_ALLELES; return nmerged - 1; }
  merged[nmerged] = cand;
  if (ix.built) { uint32_t h = cand.hash & 255u; while (ix.slot[h]) h = (h + 1) & 255u; ix.slot[h] = (uint8_t)nmerged; }
  return nmerged++;
}

// REF of the lowest-row cell starting at s_k (cells are (col,row) sorted; a plain reference block counts), or null
GDB_HD const char* site_first_ref(const SiteCtx& cx, int64_t s_k, int& len) {
  int64_t lo = 0, hi = cx.fr.ncells;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (cx.fr.begin[mid] < s_k) lo = mid + 1; else hi = mid; }
  len = 0;
  if (lo < cx.fr.ncells && cx.fr.begin[lo] == s_k && field_valid(cx.cm, lo, cx.pl.f_REF)) return cell_field<char>(cx.fr, cx.pl, cx.pl.f_REF, lo, len);
  return nullptr;
}
// What one variant call of a record feeds into the allele merge, in the order the reference's merge_alt_alleles sees it
// (spanning deletions folded in as handle_deletions does): find_or_add(candidate) returns the candidate's merged index; with
// write_luts the call's input -> merged LUT, its flags and the min-PL genotype of a spanning deletion are stored.
GDB_HD const char* gdb_star_allele() { return "*"; }
template <class F> GDB_HD void site_merge_call(const SiteCtx& cx, int64_t t, int64_t s_k, const char* mref, int mref_len, bool write_luts, F& find_or_add, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
    int64_t c = cx.hl.cell[t];
    uint32_t cf = cx.cm.cflags[c];
    int ref_len = 0, alt_len = 0;
    cell_field<char>(cx.fr, pl, pl.f_REF, c, ref_len);
    const char* alt = cell_field<char>(cx.fr, pl, pl.f_ALT, c, alt_len);
    int nalt = (int)GDB_CF_NALT(cf);
    int8_t* lut = write_luts ? cx.hl.i2m + cx.hl.i2m_off[t] : nullptr;
    bool spanning = (cf & GDB_CF_DELETION) && s_k > cx.fr.begin[c];
    uint8_t iflag = 0;
    if (lut) { lut[0] = 0; for (int a = 1; a <= nalt; ++a) lut[a] = -1; }
    if (spanning) {
      iflag |= GDB_IF_SPANNING;
      if (!(cf & GDB_CF_HAS_NR)) iflag |= GDB_IF_NO_NR;
      // deletion allele with the lowest hom PL maps to '*' (broad_combined_gvcf.cc:960-989)
      int ploidy = (int)GDB_CF_PLOIDY(cf);
      int npl = 0;
      const int32_t* plv = nullptr;
      bool pl_exists = pl.f_PL >= 0 && field_valid(cx.cm, c, pl.f_PL);
      if (pl_exists) plv = cell_field<int32_t>(cx.fr, pl, pl.f_PL, c, npl);
      int lowest = -1;
      int lowest_pl = 0x7FFFFFFF;
      const char* tok; int tl;
      for (int i = 0; alt_token(alt, alt_len, i, tok, tl); ++i) {
        int aidx = i + 1;
        if (allele_is_deletion(ref_len, tok, tl)) {
          if (lowest < 0) lowest = aidx;
          if (pl_exists) {
            int64_t gt_idx;
            if (ploidy == 0) gt_idx = 0;
            else if (ploidy == 1) gt_idx = aidx;
            else if (ploidy == 2) gt_idx = gdb_alleles2gt(aidx, aidx);
            else if (ploidy <= GDB_MAX_PLOIDY) { int hom[GDB_MAX_PLOIDY]; for (int q = 0; q < ploidy; ++q) hom[q] = aidx; gt_idx = gdb_genotype_index_sorted(hom, ploidy); }
            else { *err |= GDB_ERR_UNSUPPORTED_PLOIDY; gt_idx = 0x7FFFFFFF; }
            if (gt_idx < npl && plv[gt_idx] < lowest_pl) { lowest_pl = plv[gt_idx]; lowest = aidx; }
          }
        }
      }
      if (lowest < 0) { *err |= GDB_ERR_INTERNAL; lowest = 1; }
      AlleleRef cand{gdb_star_allele(), 1, -1, 0u};
      cand.hash = allele_hash(cand.p, cand.len, cand.suffix_from, mref, mref_len);
      const int found = find_or_add(cand);
      if (lut) lut[lowest] = (int8_t)found;
      if (lut && pl.min_PL_GT_for_spanning_deletions && pl.produce_GT_field && pl_exists && pl.f_GT >= 0 && field_valid(cx.cm, c, pl.f_GT)) {
        // update_GT_to_correspond_to_min_PL_value on the REDUCED PL (alleles REF,*,[<NON_REF>])
        int nr_idx = -1;
        for (int i = 0; alt_token(alt, alt_len, i, tok, tl); ++i) if (tl > 0 && tok[0] == '&') nr_idx = i + 1;
        int red2in[3] = {0, lowest, nr_idx};
        int nred = nr_idx >= 0 ? 3 : 2;
        int best_min = 0x7FFFFFFF, best_a = -1, best_b = -1;
        if (ploidy == 1) {
          for (int a = 0; a < nred; ++a) {
            int64_t gi = red2in[a];
            int32_t v = gi < npl ? plv[gi] : GDB_BCF_INT32_MISSING;
            if (gdb_int_valid(v) && v < best_min) { best_min = v; best_a = a; }
          }
        } else if (ploidy == 2) {
          for (int a = 0; a < nred; ++a) for (int b2 = a; b2 < nred; ++b2) {
            int64_t gi = gdb_alleles2gt(red2in[a], red2in[b2]);
            int32_t v = gi < npl ? plv[gi] : GDB_BCF_INT32_MISSING;
            if (gdb_int_valid(v) && v < best_min) { best_min = v; best_a = a; best_b = b2; }
          }
        } else if (ploidy <= GDB_MAX_PLOIDY) {   // general ploidy: VCF (colex) genotype order, first minimum wins (remap_..._general + tracker)
          int g[GDB_MAX_PLOIDY], in[GDB_MAX_PLOIDY];
          for (int q = 0; q < ploidy; ++q) g[q] = 0;
          do {
            for (int q = 0; q < ploidy; ++q) in[q] = red2in[g[q]];
            const int64_t gi = gdb_genotype_index(in, ploidy);
            const int32_t v = gi < npl ? plv[gi] : GDB_BCF_INT32_MISSING;
            if (gdb_int_valid(v) && v < best_min) {
              best_min = v; best_a = g[0];
              for (int q = 0; q < ploidy; ++q) cx.hl.gt_override[GDB_MAX_PLOIDY * t + q] = (int8_t)g[q];
            }
          } while (gdb_next_genotype(g, ploidy, nred));
        } else *err |= GDB_ERR_UNSUPPORTED_PLOIDY;
        if (best_a >= 0) {
          iflag |= GDB_IF_GT_OVERRIDE;
          if (ploidy <= 2) {
            cx.hl.gt_override[GDB_MAX_PLOIDY * t] = (int8_t)best_a;      // reduced allele idx: 0 REF, 1 '*', 2 <NON_REF>
            cx.hl.gt_override[GDB_MAX_PLOIDY * t + 1] = (int8_t)best_b;
          }
        }
      }
    } else {
      bool suffix_needed = ref_len < mref_len;
      const char* tok; int tl;
      for (int i = 0; alt_token(alt, alt_len, i, tok, tl); ++i) {
        if (tl > 0 && tok[0] == '&') continue;  // <NON_REF> goes last
        AlleleRef cand{tok, tl, (suffix_needed && !allele_is_symbolic(tok, tl)) ? ref_len : -1, 0u};
        cand.hash = allele_hash(cand.p, cand.len, cand.suffix_from, mref, mref_len);
        const int found = find_or_add(cand);
        if (lut && i + 1 <= nalt) lut[i + 1] = (int8_t)found;
      }
    }
    if (write_luts) cx.hl.iflags[t] = iflag;
}

template <class Sink> GDB_HD void site_emit(const SiteCtx& cx, int64_t k, Sink& sink, bool write_luts, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  const int64_t s_k = cx.rec.start[k], e_k = cx.rec.end[k];
  const int64_t hb = cx.hl.base[k], he = cx.hl.base[k + 1];
  // -- merged REF: longest REF among calls that START here (merge_reference_allele) -------------------
  const char* mref = nullptr;
  int mref_len = 0;
  char ref_base = 'N';
  const HugeSiteOut* hsite = (cx.huge.enabled && cx.huge.index[k] >= 0 && !cx.huge.out[cx.huge.index[k]].fallback) ? &cx.huge.out[cx.huge.index[k]] : nullptr;
  {
    if (hsite) { mref = hsite->mref; mref_len = hsite->mref_len; }
    else mref = site_first_ref(cx, s_k, mref_len);
    if (!hsite) for (int64_t t = hb; t < he; ++t) {
      int64_t c = cx.hl.cell[t];
      if (cx.fr.begin[c] != s_k || !field_valid(cx.cm, c, pl.f_REF)) continue;
      int n;
      const char* r = cell_field<char>(cx.fr, pl, pl.f_REF, c, n);
      if (n > mref_len) { mref = r; mref_len = n; }
    }
    if (mref_len == 0 || (mref_len == 1 && mref[0] == 'N')) {
      // nobody starts here: base from the reference genome (broad_combined_gvcf.cc:825-830)
      int64_t off = s_k - cx.qw.ref_begin;
      char b = (cx.qw.ref_bases && off >= 0 && off < cx.qw.ref_len) ? cx.qw.ref_bases[off] : 'N';
      ref_base = (b == 'A' || b == 'C' || b == 'G' || b == 'T') ? b : 'N';
      mref = &ref_base;
      mref_len = 1;
    }
  }
  // -- merged ALT list + LUTs (merge_alt_alleles), spanning deletions folded in (handle_deletions) ------
  AlleleRef merged[GDB_MAX_MERGED_ALLELES];
  MergeIndex mix;
  mix.built = false;
  int nmerged = 1;  // index 0 = REF
  bool non_ref_exists = cx.pc.nr_cnt[k] > 0;
  if (hsite) {
    nmerged = hsite->nmerged;
    for (int j = 1; j < nmerged; ++j) merged[j] = hsite->merged[j];
  } else {
    struct SerialMerge {
      AlleleRef* merged; int* nmerged; MergeIndex* mix; const char* mref; int mref_len; uint32_t* err;
      GDB_HD int operator()(const AlleleRef& cand) { return merged_find_or_add(merged, *nmerged, *mix, cand, mref, mref_len, err); }
    } fa{merged, &nmerged, &mix, mref, mref_len, err};
    for (int64_t t = hb; t < he; ++t) site_merge_call(cx, t, s_k, mref, mref_len, write_luts, fa, err);
  }
  const int num_merged = nmerged + (non_ref_exists ? 1 : 0);
  if (write_luts && non_ref_exists && !hsite) {
    for (int64_t t = hb; t < he; ++t) {
      int64_t c = cx.hl.cell[t];
      uint32_t cf = cx.cm.cflags[c];
      if (!(cf & GDB_CF_HAS_NR)) continue;
      int alt_len;
      const char* alt = cell_field<char>(cx.fr, pl, pl.f_ALT, c, alt_len);
      const char* tok; int tl;
      for (int i = 0; alt_token(alt, alt_len, i, tok, tl); ++i)
        if (tl > 0 && tok[0] == '&' && i + 1 <= (int)GDB_CF_NALT(cf)) cx.hl.i2m[cx.hl.i2m_off[t] + i + 1] = (int8_t)(num_merged - 1);
    }
  }
  const bool ref_block_only = (mref_len == 1 && nmerged == 1 && non_ref_exists);
  const bool skip_G = (num_merged - 1) > pl.max_diploid_alt_alleles;
  // -- FORMAT presence -------------------------------------------------------------------------------
  uint32_t fmt_mask = 0;
  if (!pl.sites_only_query)
    for (int i = 0; i < pl.n_format; ++i) {
      int f = pl.format_field[i];
      if (cx.pc.fmt_cnt[(int64_t)i * cx.pc.stride + k] <= 0) continue;
      if (pl.field[f].length == GDB_VL_G && skip_G) continue;
      fmt_mask |= 1u << i;
    }
  if (write_luts) {
    cx.so.num_alleles[k] = (uint8_t)num_merged;
    cx.so.rflags[k] = (uint8_t)((non_ref_exists ? GDB_RF_NON_REF_EXISTS : 0) | (!ref_block_only ? GDB_RF_REMAPPING_NEEDED : 0) | (skip_G ? GDB_RF_SKIP_G_FIELDS : 0));
    cx.so.fmt_mask[k] = fmt_mask;
  }
  // -- fixed columns ----------------------------------------------------------------------------------
  int ci = find_contig(cx.qw, s_k);
  if (ci < 0) { *err |= GDB_ERR_INTERNAL; ci = 0; }
  const GdbContig& ctg = cx.qw.contigs[ci];
  if (pl.bcf_mode) {
    // BCF2 shared block (BCFv2.2 6.3.1; the order of the reference's bcf_update_* calls, broad_combined_gvcf.cc:795-880):
    // CHROM POS rlen QUAL n_allele|n_info n_fmt|n_sample ID alleles FILTER INFO...; l_shared / l_indiv and the FORMAT block
    // are written by the page assembly.  rlen = END - POS for interval records, else the length of REF.
    int n_fmt = 0;
    for (uint32_t m = fmt_mask; m; m &= m - 1) ++n_fmt;
    bcf_put_u32(sink, (uint32_t)ctg.rid);
    bcf_put_u32(sink, (uint32_t)(s_k - ctg.offset));
    bcf_put_u32(sink, (uint32_t)(e_k > s_k ? e_k - s_k + 1 : (int64_t)mref_len));
    float qv;
    uint32_t qbits = GDB_BCF_FLOAT_MISSING_BITS;
    if (pl.qual_combine_op != GDB_OP_UNKNOWN && pl.f_QUAL >= 0 && reduce_scalar<float>(cx, k, pl.f_QUAL, pl.qual_combine_op, true, qv, err)) qbits = gdb_f2u(qv);
    bcf_put_u32(sink, qbits);
    const uint32_t at_counts = sink.pos();
    bcf_put_u32(sink, (uint32_t)num_merged << 16);          // n_info patched below
    bcf_put_u32(sink, ((uint32_t)n_fmt << 24) | ((uint32_t)pl.bcf_n_sample & 0xFFFFFFu));
    // ID (the same token union as the text flavour)
    {
      const char* tp[GDB_MAX_ID_TOKENS]; int tn[GDB_MAX_ID_TOKENS]; int nt = 0;
      GdbUSetHashOrder us;
      gdb_useth_init(us);
      if (pl.f_ID >= 0 && !(hsite && !hsite->any_id))
        for (int64_t t = hb; t < he; ++t) {
          const int64_t c = cx.hl.cell[t];
          if (!field_valid(cx.cm, c, pl.f_ID)) continue;
          int n;
          const char* p = cell_field<char>(cx.fr, pl, pl.f_ID, c, n);
          gdb_id_union_value(p, n, tp, tn, nt, us, pl.id_order_unordered_set, err);
        }
      int total = nt ? nt - 1 : 0;
      for (int i = 0; i < nt; ++i) total += tn[i];
      bcf_enc_size(sink, total, GDB_BT_CHAR);
      for (int i = 0; i < nt; ++i) { const int w = gdb_id_union_at(us, pl.id_order_unordered_set, i); if (i) sink.put(';'); sink.write(tp[w], tn[w]); }
    }
    // alleles
    bcf_enc_size(sink, mref_len, GDB_BT_CHAR);
    sink.write(mref, mref_len);
    for (int j = 1; j < nmerged; ++j) {
      const int sl = merged[j].suffix_from >= 0 ? mref_len - merged[j].suffix_from : 0;
      bcf_enc_size(sink, merged[j].len + sl, GDB_BT_CHAR);
      sink.write(merged[j].p, merged[j].len);
      if (sl) sink.write(mref + merged[j].suffix_from, sl);
    }
    if (non_ref_exists) { const char nr[] = "<NON_REF>"; bcf_enc_size(sink, 9, GDB_BT_CHAR); sink.write(nr, 9); }
    // FILTER
    {
      int first_id = -1;
      bool multi = false;
      if (hsite) { first_id = hsite->filter_first_id; multi = hsite->filter_multi != 0; }
      else if (pl.produce_FILTER_field && pl.f_FILTER >= 0)
        for (int64_t t = hb; t < he; ++t) {
          int64_t c = cx.hl.cell[t];
          if (!field_valid(cx.cm, c, pl.f_FILTER)) continue;
          int n;
          const int32_t* p = cell_field<int32_t>(cx.fr, pl, pl.f_FILTER, c, n);
          for (int i = 0; i < n; ++i) { if (first_id < 0) first_id = p[i]; else if (p[i] != first_id) multi = true; }
        }
      if (multi) {   // typed vector of the header ids in the set's order (bcf_update_filter -> bcf_enc_vint)
        GdbUSetOrder us;
        filter_union_order(cx, hb, he, us, err);
        int32_t hids[GDB_MAX_FILTER_IDS];
        int nh = 0;
        for (int i = 0; i < us.n; ++i) {
          const int32_t id = us.key[i];
          const int32_t hid = (id >= 0 && id < cx.names.n_filter_names) ? cx.names.filter_bcf_id[id] : -1;
          if (hid >= 0) hids[nh++] = hid;
        }
        bcf_enc_vint(sink, hids, nh);
      } else {
        const int32_t hid = (first_id >= 0 && first_id < cx.names.n_filter_names) ? cx.names.filter_bcf_id[first_id] : -1;
        if (hid >= 0) bcf_enc_int1(sink, hid); else sink.put((char)0);
      }
    }
    // INFO: END, reducers in query order, DP
    uint32_t n_info = 0;
    if (e_k > s_k) { bcf_enc_int1(sink, pl.bcf_end_id); bcf_enc_int1(sink, (int32_t)(e_k - ctg.offset + 1)); ++n_info; }
    for (int i = 0; i < pl.n_info; ++i) {
      const int f = pl.info_field[i];
      const GdbFieldDesc& fd = pl.field[f];
      if (fd.length == GDB_VL_G && (num_merged - 1) > pl.max_diploid_alt_alleles) continue;   // (handle_VCF_field_combine_operation, broad_combined_gvcf.cc:380-385)
      if (fd.ndim == 2) { bool any2 = false; if (info_asa_sum(cx, k, f, num_merged, non_ref_exists, !ref_block_only, true, sink, any2, err)) ++n_info; continue; }
      if (fd.combine_op == GDB_OP_ELEMENT_WISE_SUM || fd.combine_op == GDB_OP_CONCATENATE) {
        bool found;
        if (fd.elem == GDB_ET_FLOAT) found = info_vector_combine_bcf<float>(cx, k, f, fd.combine_op, num_merged, non_ref_exists, !ref_block_only, sink, err);
        else found = info_vector_combine_bcf<int32_t>(cx, k, f, fd.combine_op, num_merged, non_ref_exists, !ref_block_only, sink, err);
        if (found) ++n_info;
        continue;
      }
      if (fd.elem == GDB_ET_FLOAT) {
        float v;
        if (!reduce_scalar<float>(cx, k, f, fd.combine_op, false, v, err, num_merged, non_ref_exists, !ref_block_only)) continue;
        bcf_enc_int1(sink, pl.bcf_id[f]);
        sink.put((char)((1 << 4) | GDB_BT_FLOAT));
        bcf_put_u32(sink, gdb_f2u(v));
      } else {
        int32_t v;
        if (!reduce_scalar<int32_t>(cx, k, f, fd.combine_op, false, v, err, num_merged, non_ref_exists, !ref_block_only)) continue;
        bcf_enc_int1(sink, pl.bcf_id[f]);
        bcf_enc_int1(sink, v);
      }
      ++n_info;
    }
    for (int i = 0; i < pl.n_histogram; ++i) {
      bool any2 = false;
      if (info_asa_histogram(cx, k, pl.histogram_bin_field[i], pl.histogram_count_field[i], num_merged, non_ref_exists, !ref_block_only, true, sink, any2, err)) ++n_info;
    }
    {
      const int32_t dp = cx.pc.dp_sum[k];
      if ((pl.f_DP >= 0 || pl.f_DP_FORMAT >= 0) && dp > 0 && !ref_block_only) { bcf_enc_int1(sink, pl.bcf_dp_id); bcf_enc_int1(sink, dp); ++n_info; }
    }
    sink.patch_u32(at_counts, ((uint32_t)num_merged << 16) | (n_info & 0xFFFFu));
    return;
  }
  sink.write(cx.qw.contig_names + ctg.name_off, ctg.name_len);
  sink.put('\t');
  put_i64(sink, s_k - ctg.offset + 1);
  sink.put('\t');
  // ID: union of the ';'-separated tokens of the live calls (merge_ID_field, broad_combined_gvcf.cc:730-763): sorted in the
  // reference's DEBUG build - the one the goldens come from - and in the order of a std::unordered_set<std::string> in every other
  // build (pl.id_order_unordered_set; gdb_id_union_add).  An empty union string (no token, or only the empty token of an ID value
  // ";") leaves the record's ID alone: '.' (:801-802)
  {
    const char* tp[GDB_MAX_ID_TOKENS]; int tn[GDB_MAX_ID_TOKENS]; int nt = 0;
    GdbUSetHashOrder us;
    gdb_useth_init(us);
    if (pl.f_ID >= 0 && !(hsite && !hsite->any_id))
      for (int64_t t = hb; t < he; ++t) {
        const int64_t c = cx.hl.cell[t];
        if (!field_valid(cx.cm, c, pl.f_ID)) continue;
        int n;
        const char* p = cell_field<char>(cx.fr, pl, pl.f_ID, c, n);
        gdb_id_union_value(p, n, tp, tn, nt, us, pl.id_order_unordered_set, err);
      }
    if (nt == 0 || (nt == 1 && tn[0] == 0)) sink.put('.');
    for (int i = 0; i < nt; ++i) { const int w = gdb_id_union_at(us, pl.id_order_unordered_set, i); if (i) sink.put(';'); sink.write(tp[w], tn[w]); }
  }
  sink.put('\t');
  sink.write(mref, mref_len);
  sink.put('\t');
  if (num_merged == 1) sink.put('.');
  for (int j = 1; j < nmerged; ++j) {
    if (j > 1) sink.put(',');
    sink.write(merged[j].p, merged[j].len);
    if (merged[j].suffix_from >= 0) sink.write(mref + merged[j].suffix_from, mref_len - merged[j].suffix_from);
  }
  if (non_ref_exists) { if (nmerged > 1) sink.put(','); const char nr[] = "<NON_REF>"; sink.write(nr, 9); }
  sink.put('\t');
  // QUAL
  {
    float q;
    if (pl.qual_combine_op != GDB_OP_UNKNOWN && pl.f_QUAL >= 0 && reduce_scalar<float>(cx, k, pl.f_QUAL, pl.qual_combine_op, true, q, err)) {
      put_float(sink, q);
    } else sink.put('.');
  }
  sink.put('\t');
  // FILTER (union over live calls, in the iteration order of the reference's std::unordered_set<int>)
  {
    int first_id = -1;
    bool multi = false;
    if (hsite) { first_id = hsite->filter_first_id; multi = hsite->filter_multi != 0; }
    else if (pl.produce_FILTER_field && pl.f_FILTER >= 0)
      for (int64_t t = hb; t < he; ++t) {
        int64_t c = cx.hl.cell[t];
        if (!field_valid(cx.cm, c, pl.f_FILTER)) continue;
        int n;
        const int32_t* p = cell_field<int32_t>(cx.fr, pl, pl.f_FILTER, c, n);
        for (int i = 0; i < n; ++i) { if (first_id < 0) first_id = p[i]; else if (p[i] != first_id) multi = true; }
      }
    if (multi) {     // several different ids: the order is the library's (gdb_uset_insert_range); rare, so a pass of its own
      GdbUSetOrder us;
      filter_union_order(cx, hb, he, us, err);
      bool any = false;
      for (int i = 0; i < us.n; ++i) {
        const int32_t id = us.key[i];
        if (!(id >= 0 && id < cx.names.n_filter_names && cx.names.filter_name_len[id] > 0)) continue;   // ids without a header line are dropped
        if (any) sink.put(';');
        sink.write(cx.names.text + cx.names.filter_name_off[id], cx.names.filter_name_len[id]);
        any = true;
      }
      if (!any) sink.put('.');
    } else if (first_id >= 0 && first_id < cx.names.n_filter_names && cx.names.filter_name_len[first_id] > 0)
      sink.write(cx.names.text + cx.names.filter_name_off[first_id], cx.names.filter_name_len[first_id]);
    else sink.put('.');
  }
  sink.put('\t');
  // INFO: END, reducers in query order, DP
  {
    bool any = false;
    if (e_k > s_k) { const char t[] = "END="; sink.write(t, 4); put_i64(sink, e_k - ctg.offset + 1); any = true; }
    for (int i = 0; i < pl.n_info; ++i) {
      int f = pl.info_field[i];
      const GdbFieldDesc& fd = pl.field[f];
      if (fd.length == GDB_VL_G && (num_merged - 1) > pl.max_diploid_alt_alleles) continue;
      if (fd.ndim == 2) { info_asa_sum(cx, k, f, num_merged, non_ref_exists, !ref_block_only, false, sink, any, err); continue; }
      if (fd.combine_op == GDB_OP_ELEMENT_WISE_SUM || fd.combine_op == GDB_OP_CONCATENATE) {
        if (fd.elem == GDB_ET_FLOAT) info_vector_combine<float>(cx, k, f, fd.combine_op, num_merged, non_ref_exists, !ref_block_only, sink, any, err);
        else info_vector_combine<int32_t>(cx, k, f, fd.combine_op, num_merged, non_ref_exists, !ref_block_only, sink, any, err);
        continue;
      }
      if (fd.elem == GDB_ET_FLOAT) {
        float v;
        if (!reduce_scalar<float>(cx, k, f, fd.combine_op, false, v, err, num_merged, non_ref_exists, !ref_block_only)) continue;
        if (any) sink.put(';');
        sink.write(cx.names.text + cx.names.field_name_off[f], cx.names.field_name_len[f]);
        sink.put('=');
        put_float(sink, v);
      } else {
        int32_t v;
        if (!reduce_scalar<int32_t>(cx, k, f, fd.combine_op, false, v, err, num_merged, non_ref_exists, !ref_block_only)) continue;
        if (any) sink.put(';');
        sink.write(cx.names.text + cx.names.field_name_off[f], cx.names.field_name_len[f]);
        sink.put('=');
        put_i32(sink, v);
      }
      any = true;
    }
    for (int i = 0; i < pl.n_histogram; ++i)
      info_asa_histogram(cx, k, pl.histogram_bin_field[i], pl.histogram_count_field[i], num_merged, non_ref_exists, !ref_block_only, false, sink, any, err);
    int32_t dp = cx.pc.dp_sum[k];
    if ((pl.f_DP >= 0 || pl.f_DP_FORMAT >= 0) && dp > 0 && !ref_block_only) {
      if (any) sink.put(';');
      const char t[] = "DP="; sink.write(t, 3); put_i32(sink, dp);
      any = true;
    }
    if (!any) sink.put('.');
  }
  // FORMAT keys
  if (fmt_mask) {
    sink.put('\t');
    bool first = true;
    for (int i = 0; i < pl.n_format; ++i) {
      if (!((fmt_mask >> i) & 1)) continue;
      if (!first) sink.put(':');
      first = false;
      int f = pl.format_field[i];
      if (f == pl.f_DP_FORMAT || f == pl.f_DP) { sink.put('D'); sink.put('P'); }
      else sink.write(cx.names.text + cx.names.field_name_off[f], cx.names.field_name_len[f]);
    }
  }
}

// ---- sample entries --------------------------------------------------------------------------------------
struct EntryCtx {
  FragmentView fr;
  CombinePlan pl;
  CellMeta cm;
  HeavyLists hl;
};
struct RecordInfo { int num_merged; uint32_t rflags; uint32_t fmt_mask; int64_t hbase, hend; };

template <class Sink> GDB_HD void put_int_vector(Sink& s, const int32_t* p, int n) {
  if (n == 0) { s.put('.'); return; }
  for (int j = 0; j < n; ++j) {
    if (p[j] == GDB_BCF_INT32_VECTOR_END) break;
    if (j) s.put(',');
    if (p[j] == GDB_BCF_INT32_MISSING) s.put('.'); else put_i32(s, p[j]);
  }
}
template <class Sink> GDB_HD void put_int_or_missing(Sink& s, bool has, int32_t v) {
  if (!has || v == GDB_BCF_INT32_MISSING) s.put('.'); else put_i32(s, v);
}

// Allele maps of one live call inside one record (built once per entry, handed to the field emitters by reference)
// (m2i is a pointer to an array of the CALLER, not a member: with the array inside, the variable-length fill of build_entry_maps kept the
//  whole struct in scratch memory on the device - every `em.remap` / `em.cf` in front of a field was a trip to memory, ~60 per entry)
struct EntryMaps {
  int8_t* m2i;                         // [GDB_MAX_MERGED_ALLELES] merged allele -> input allele of this call (-1: none); only filled for heavy calls
  const int8_t* i2m;                   // input -> merged (heavy calls), null for plain reference blocks
  int nr_in;                           // input idx of <NON_REF> in this call, -1 if it has none
  int n_in;                            // #input alleles incl. REF
  int64_t inc;                         // incidence idx of a heavy call, -1 otherwise
  uint32_t cf;                         // cell flags
  uint8_t iflag;
  bool remap, nr_exists;
  bool light;                          // plain reference block (REF,<NON_REF>): REF -> 0, every other merged allele <- <NON_REF>
  // input allele feeding merged allele j, with the <NON_REF> fallback of the reference's remap loops; -1 = missing
  GDB_HD int lookup(int j) const {
    if (light) return j == 0 ? 0 : nr_in;
    const int v = m2i[j];
    return v >= 0 ? v : nr_in;
  }
};

// The field emitters inline into the out-of-line wrappers of the kernels file (entry_store / entry_store_lds_capped), which
// read the EntryCtx from __constant__ memory: every plan / column-pointer access is then a scalar (SGPR) load.
#define GDB_FIELD_FN GDB_HD

// Each field kind has its own out-of-line emitter taking and returning the sink BY VALUE: keeps every function small
// and the cursor in registers (a char store through a by-reference sink may alias the sink itself).
// merged allele index of GT element out_i (input allele a) of a live call, -1 = no call ('.'):
// remap_GT_field (variant_operations.cc:233-263) with the min-PL override of spanning deletions
GDB_FIELD_FN int32_t gt_merged_allele(const EntryCtx& cx, const RecordInfo& ri, const EntryMaps& em, int32_t a, int out_i) {
  int32_t m = -1;
  if (em.iflag & GDB_IF_GT_OVERRIDE) {  // min-PL genotype over the reduced alleles: 0 REF, 1 '*', 2 <NON_REF>
    const int ra = cx.hl.gt_override[GDB_MAX_PLOIDY * em.inc + (out_i < GDB_MAX_PLOIDY ? out_i : GDB_MAX_PLOIDY - 1)];
    if (ra == 0) m = 0;
    else if (ra == 2) m = ri.num_merged - 1;
    else { for (int q = 1; q < em.n_in; ++q) if (em.i2m && em.i2m[q] >= 0 && em.i2m[q] != ri.num_merged - 1) { m = em.i2m[q]; break; } }
  } else if (a == GDB_TILEDB_NULL_INT32 || a == -1 || a == GDB_BCF_INT32_MISSING) {
    m = -1;
  } else if (!em.remap) {
    m = a;
  } else {
    int8_t mm = -1;
    if (a >= 0 && a < em.n_in) mm = em.i2m ? em.i2m[a] : (a == 0 ? (int8_t)0 : ((em.nr_exists && a == 1) ? (int8_t)(ri.num_merged - 1) : (int8_t)-1));
    if (mm >= 0) m = mm;
    else if ((em.iflag & GDB_IF_SPANNING) && (em.iflag & GDB_IF_NO_NR)) m = -1;
    else m = em.nr_exists ? ri.num_merged - 1 : -1;
  }
  return m;
}
template <class Sink> GDB_FIELD_FN Sink emit_GT(Sink s, const EntryCtx& cx, const RecordInfo& ri, const EntryMaps& em, int64_t c) {
  const CombinePlan& pl = cx.pl;
  const int f = pl.f_GT;
  int n;
  const int32_t* g = cell_field<int32_t>(cx.fr, pl, f, c, n);
  const bool pp = pl.field[f].length == GDB_VL_PP;
  const int step = pp ? 2 : 1;
  int out_i = 0;
  for (int j = 0; j < n; j += step, ++out_i) {
    if (out_i) s.put((pp && g[j - 1] > 0) ? '|' : '/');
    const int32_t m = pl.produce_GT_field ? gt_merged_allele(cx, ri, em, g[j], out_i) : -1;
    if (m < 0) s.put('.'); else put_i32(s, m);
  }
  if (out_i == 0) s.put('.');
  return s;
}

template <class Sink> GDB_FIELD_FN Sink emit_chars(Sink s, const EntryCtx& cx, int f, int64_t c) {
  int n;
  const char* p = cell_field<char>(cx.fr, cx.pl, f, c, n);
  // htslib bcf_fmt_array, char flavour: stop at the first NUL, 0x07 (bcf_str_missing) prints as '.'
  int len = 0;
  while (len < n && p[len]) ++len;
  if (n == 0) s.put('.');
  for (int j = 0; j < len; ++j) { const char ch = p[j]; s.put(ch == 0x07 ? '.' : ch); }
  return s;
}

template <class Sink, class T> GDB_FIELD_FN Sink emit_vector(Sink s, const T* p, int n, uint32_t* err) {
  if (n == 0) { s.put('.'); return s; }
  for (int j = 0; j < n; ++j) {
    if (elem_is_vector_end(p[j])) break;
    if (j) s.put(',');
    put_elem(s, p[j], err);
  }
  return s;
}

// remap_data_based_on_alleles: R- and A-length fields
template <class Sink, class T> GDB_FIELD_FN Sink emit_remap_alleles(Sink s, const T* p, int n, const EntryMaps& em, int num_merged, bool alt_only, uint32_t* err) {
  const int length = alt_only ? num_merged - 1 : num_merged;
  if (length == 0) { s.put('.'); return s; }
  for (int j = 0; j < length; ++j) {
    if (j) s.put(',');
    const int aj = alt_only ? j + 1 : j;
    const int in_j = em.lookup(aj);
    const int idx = alt_only ? in_j - 1 : in_j;
    const bool has = in_j >= 0 && idx >= 0 && idx < n;
    T v;
    if (has) v = p[idx]; else elem_set_missing(v);
    put_elem(s, v, err);
  }
  return s;
}

// remap_data_based_on_genotype_{haploid,diploid}: G-length fields (PL)
template <class Sink, class T> GDB_FIELD_FN Sink emit_remap_genotypes(Sink s, const T* p, int n, const EntryMaps& em, int num_merged, int ploidy, uint32_t* err) {
  if (ploidy == 1) {
    for (int j = 0; j < num_merged; ++j) {
      if (j) s.put(',');
      const int in_j = em.lookup(j);
      const bool has = in_j >= 0 && in_j < n;
      T v;
      if (has) v = p[in_j]; else elem_set_missing(v);
      put_elem(s, v, err);
    }
  } else if (ploidy == 2 && em.light) {
    // plain reference block: every merged ALT reads the call's <NON_REF>, so the whole vector repeats three values -
    // PL[REF/REF], PL[REF/NR], PL[NR/NR] - which are formatted once
    T v3[3];
    uint64_t w3[3];
    int n3[3];
    const int nr = em.nr_in;
    for (int q = 0; q < 3; ++q) {
      const int gi = q == 0 ? 0 : q == 1 ? (nr >= 0 ? gdb_alleles2gt(0, nr) : -1) : (nr >= 0 ? gdb_alleles2gt(nr, nr) : -1);
      if (gi >= 0 && gi < n) v3[q] = p[gi]; else elem_set_missing(v3[q]);
      n3[q] = elem_pack_text(v3[q], w3[q]);   // 0: not packable (float, negative, 9+ digits) -> generic formatter
    }
    for (int kk = 0; kk < num_merged; ++kk)
      for (int j = 0; j <= kk; ++j) {
        const int q = kk == 0 ? 0 : (j == 0 ? 1 : 2);
        if (kk | j) s.put(',');
        if (n3[q]) put_packed(s, w3[q], n3[q]); else put_elem(s, v3[q], err);
      }
  } else if (ploidy == 2) {
    // output order gt = k(k+1)/2 + j, j <= k
    for (int kk = 0; kk < num_merged; ++kk) {
      const int in_k = em.lookup(kk);
      for (int j = 0; j <= kk; ++j) {
        if (kk | j) s.put(',');
        const int in_j = em.lookup(j);
        const bool both = in_j >= 0 && in_k >= 0;
        const int gi = both ? gdb_alleles2gt(in_j, in_k) : 0;
        const bool has = both && gi < n;
        T v;
        if (has) v = p[gi]; else elem_set_missing(v);
        put_elem(s, v, err);
      }
    }
  } else if (ploidy >= 3 && ploidy <= GDB_MAX_PLOIDY) {
    // remap_data_based_on_genotype_general (variant_field_handler.cc:198-297): merged genotypes in VCF order
    int g[GDB_MAX_PLOIDY], in[GDB_MAX_PLOIDY];
    for (int q = 0; q < ploidy; ++q) g[q] = 0;
    bool first = true;
    do {
      if (!first) s.put(',');
      first = false;
      bool missing = false;
      for (int q = 0; q < ploidy; ++q) { in[q] = em.lookup(g[q]); if (in[q] < 0) missing = true; }
      T v;
      elem_set_missing(v);
      if (!missing) { const int64_t gi = gdb_genotype_index(in, ploidy); if (gi < n) v = p[gi]; }
      put_elem(s, v, err);
    } while (gdb_next_genotype(g, ploidy, num_merged));
  } else {
    *err |= GDB_ERR_UNSUPPORTED_PLOIDY;
    s.put('.');
  }
  return s;
}

GDB_FIELD_FN void build_entry_maps(const EntryCtx& cx, const RecordInfo& ri, int64_t c, EntryMaps& em, int8_t* m2i_store, uint32_t* err) {
  em.m2i = m2i_store;
  em.remap = (ri.rflags & GDB_RF_REMAPPING_NEEDED) != 0;
  em.nr_exists = (ri.rflags & GDB_RF_NON_REF_EXISTS) != 0;
  em.nr_in = -1; em.i2m = nullptr; em.n_in = 0; em.iflag = 0; em.inc = -1; em.cf = 0; em.light = false;
  if (c < 0) return;
  em.cf = cx.cm.cflags[c];
  em.n_in = (int)GDB_CF_NALT(em.cf) + 1;
  if (!em.remap) return;
  if (!(em.cf & GDB_CF_HEAVY)) { em.light = true; em.nr_in = em.nr_exists ? 1 : -1; return; }
  for (int j = 0; j < ri.num_merged; ++j) em.m2i[j] = -1;
  if (em.cf & GDB_CF_HEAVY) {
    const int32_t row = cx.fr.row[c];
    int64_t lo = ri.hbase, hi = ri.hend;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (cx.fr.row[cx.hl.cell[mid]] < row) lo = mid + 1; else hi = mid; }
    if (lo < ri.hend && cx.hl.cell[lo] == c) em.inc = lo; else *err |= GDB_ERR_INTERNAL;
  }
  if (em.inc >= 0) {
    em.i2m = cx.hl.i2m + cx.hl.i2m_off[em.inc];
    em.iflag = cx.hl.iflags[em.inc];
    for (int a = 0; a < em.n_in; ++a) { const int8_t t = em.i2m[a]; if (t >= 0) em.m2i[t] = (int8_t)a; }
    if (em.nr_exists && (em.cf & GDB_CF_HAS_NR) && em.m2i[ri.num_merged - 1] >= 0) em.nr_in = em.m2i[ri.num_merged - 1];
  }
}

// text of one FORMAT field of a live call
template <class Sink> GDB_HD Sink emit_field(Sink s, const EntryCtx& cx, const RecordInfo& ri, const EntryMaps& em, int i, int64_t c, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  const int f = pl.format_field[i];
  const GdbFieldDesc& fd = pl.field[f];
  if (f == pl.f_DP && pl.f_DP_FORMAT >= 0) {  // FORMAT DP := DP_FORMAT (broad_combined_gvcf.cc:689-719)
    int32_t v = GDB_BCF_INT32_MISSING;
    if (field_valid(cx.cm, c, pl.f_DP_FORMAT)) { int n; const int32_t* p = cell_field<int32_t>(cx.fr, pl, pl.f_DP_FORMAT, c, n); if (n > 0) v = p[0]; }
    if (gdb_int_valid(v)) put_i32(s, v); else s.put('.');
  } else if (!field_valid(cx.cm, c, f)) {
    s.put('.');
  } else if (f == pl.f_GT) {
    s = emit_GT(s, cx, ri, em, c);
  } else if (fd.elem == GDB_ET_CHAR || fd.elem == GDB_ET_FLAG) {
    s = emit_chars(s, cx, f, c);
  } else if (fd.elem == GDB_ET_FLOAT) {
    int n;
    const float* p = cell_field<float>(cx.fr, pl, f, c, n);
    const bool allele_dep = fd.length == GDB_VL_A || fd.length == GDB_VL_R || fd.length == GDB_VL_G;
    if (!em.remap || !allele_dep) s = emit_vector(s, p, n, err);
    else if (fd.length == GDB_VL_G) s = emit_remap_genotypes(s, p, n, em, ri.num_merged, (int)GDB_CF_PLOIDY(em.cf), err);
    else s = emit_remap_alleles(s, p, n, em, ri.num_merged, fd.length == GDB_VL_A, err);
  } else if (fd.elem != GDB_ET_INT) {
    *err |= GDB_ERR_INTERNAL;
    s.put('.');
  } else {
    int n;
    const int32_t* p = cell_field<int32_t>(cx.fr, pl, f, c, n);
    const bool allele_dep = fd.length == GDB_VL_A || fd.length == GDB_VL_R || fd.length == GDB_VL_G;
    if (!em.remap || !allele_dep) s = emit_vector(s, p, n, err);
    else if (fd.length == GDB_VL_G) s = emit_remap_genotypes(s, p, n, em, ri.num_merged, (int)GDB_CF_PLOIDY(em.cf), err);
    else s = emit_remap_alleles(s, p, n, em, ri.num_merged, fd.length == GDB_VL_A, err);
  }
  return s;
}

// text of one (record, sample) column.  c < 0: the sample has no live call.
template <class Sink> GDB_HD Sink entry_emit(const EntryCtx& cx, const RecordInfo& ri, int64_t c, Sink s, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  EntryMaps em;
  int8_t m2i_store[GDB_MAX_MERGED_ALLELES];
  build_entry_maps(cx, ri, c, em, m2i_store, err);
  bool first = true;
  for (int i = 0; i < pl.n_format; ++i) {
    if ((ri.fmt_mask >> i) & 1) {
      if (!first) s.put(':');
      first = false;
      if (c < 0) s.put('.');
      else s = emit_field(s, cx, ri, em, i, c, err);
    }
  }
  return s;
}
