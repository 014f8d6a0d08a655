// gdb_asa.hpp - allele-specific annotations: INFO fields of two dimensions (one inner vector per allele, written in the VCF as
// "1.0,2.0|3.0|4.0") and the two reducers the reference has for them.  Included by gdb_core.hpp (host simulation and device).
//
// The data of a 2-D field is a byte blob per element of its type tuple (reference genomicsdb_multid_vector_field.h:69-86):
//     <u64 size of data> <inner vector 0> <inner vector 1> ... <u64 #entries> <u64 offset of entry 0, 1, ..., #entries>
// GA4GHOperator re-indexes dimension 0 to the merged alleles before anything is reduced (remap_allele_specific_annotations,
// variant_operations.cc:482-549: an allele the call does not have takes the call's <NON_REF> entry; none: 0 bytes); the
// reducers then walk the remapped blobs of the valid calls in call order:
//     element_wise_sum  compute_valid_element_wise_sum_2D_vector + stringify_2D_vector (variant_field_handler.cc:666-740)
//     histogram_sum     compute_valid_histogram_sum_2D_vector_and_stringify (broad_combined_gvcf.cc:431-521), on the two
//                       flattened fields (bins, counts) of a composite field
// Here nothing is materialised: the entry of merged allele j is looked up through the call's allele LUT every time it is needed,
// sums are accumulated per output element in call order (the order the reference adds in, so float sums round the same way) and
// the histogram's ordered map becomes a selection of the next larger bin.  Values print with 3 decimals (std::fixed,
// std::setprecision(3)): put_fixed3 rounds the exact binary value to nearest, ties to even, as glibc's printf does.
#pragma once

GDB_HD uint64_t asa_rd_u64(const char* p) { uint64_t v = 0; for (int i = 0; i < 8; ++i) v |= (uint64_t)(uint8_t)p[i] << (8 * i); return v; }
GDB_HD uint32_t asa_rd_u32(const char* p) { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)(uint8_t)p[i] << (8 * i); return v; }
template <class T> GDB_HD T asa_rd_elem(const char* p);
template <> GDB_HD int32_t asa_rd_elem<int32_t>(const char* p) { return (int32_t)asa_rd_u32(p); }
template <> GDB_HD float asa_rd_elem<float>(const char* p) { union { uint32_t u; float f; } x; x.u = asa_rd_u32(p); return x.f; }

struct AsaBlob {
  const char* p;
  uint64_t data_size, n;
  GDB_HD bool open(const char* q, int nbytes) {
    p = q; data_size = 0; n = 0;
    if (nbytes < 24) return false;
    data_size = asa_rd_u64(q);
    if (data_size > (uint64_t)nbytes || 8u + data_size + 16u > (uint64_t)nbytes) return false;
    n = asa_rd_u64(q + 8 + data_size);
    if (n > (uint64_t)nbytes || 8u + data_size + 8u + 8u * (n + 1u) > (uint64_t)nbytes) { n = 0; return false; }
    return true;
  }
  GDB_HD uint64_t off(uint64_t i) const { return asa_rd_u64(p + 8 + data_size + 8 + 8 * i); }
};

// one valid call's view of a 2-D field for a record
struct AsaCall {
  AsaBlob blob;
  const int8_t* lut;      // input allele -> merged allele of this incidence
  int nal;                // input alleles (REF included)
  int nr_in;              // input index of <NON_REF> (-1: none)
  bool ok;
};
GDB_HD bool asa_call_open(const SiteCtx& cx, int64_t t, int64_t s_k, int f, bool allele_dep, bool non_ref_exists, int num_merged, AsaCall& a, uint32_t* err) {
  a.ok = false;
  if (inc_is_spanning(cx, t, s_k)) return false;          // INFO of spanning deletions is invalidated
  const int64_t c = cx.hl.cell[t];
  if (!field_valid(cx.cm, c, f)) return false;
  int nbytes;
  const char* p = cell_field<char>(cx.fr, cx.pl, f, c, nbytes);
  if (!a.blob.open(p, nbytes)) { *err |= GDB_ERR_CELL_STREAM; return false; }
  a.lut = cx.hl.i2m + cx.hl.i2m_off[t];
  a.nal = (int)GDB_CF_NALT(cx.cm.cflags[c]) + 1;
  a.nr_in = -1;
  if (allele_dep && non_ref_exists) for (int x = 0; x < a.nal; ++x) if (a.lut[x] == num_merged - 1) a.nr_in = x;
  a.ok = true;
  return true;
}
// the inner vector standing at outer index j (merged-allele order when allele_dep)
GDB_HD void asa_entry(const AsaCall& a, int j, bool allele_dep, bool alt_only, const char*& ptr, uint32_t& bytes) {
  ptr = a.blob.p; bytes = 0;
  int64_t idx = j;
  if (allele_dep) {
    const int aj = alt_only ? j + 1 : j;
    int in = -1;
    for (int x = 0; x < a.nal; ++x) if (a.lut[x] == aj) { in = x; break; }
    if (in < 0) in = a.nr_in;
    if (in < 0) return;
    idx = alt_only ? in - 1 : in;
  }
  if (idx < 0 || (uint64_t)idx >= a.blob.n) return;
  const uint64_t o0 = a.blob.off((uint64_t)idx), o1 = a.blob.off((uint64_t)idx + 1);
  if (o1 < o0 || o1 > a.blob.data_size) return;
  ptr = a.blob.p + 8 + o0;
  bytes = (uint32_t)(o1 - o0);
}

// "%.3f" of a float (converted to double by the stream: exact), and the plain decimal of an int
template <class Sink> GDB_HD void put_fixed3(Sink& s, float f, uint32_t* err) {
  const uint32_t u = gdb_f2u(f);
  const uint32_t ex = (u >> 23) & 0xFFu, mant = u & 0x7FFFFFu;
  if (ex == 0xFFu) {                             // what the reference's stream (std::fixed on a double) prints: [-]inf / [-]nan
    if (u >> 31) s.put('-');
    if (mant) { s.put('n'); s.put('a'); s.put('n'); } else { s.put('i'); s.put('n'); s.put('f'); }
    return;
  }
  const uint64_t m = ex ? (uint64_t)(mant | 0x800000u) : (uint64_t)mant;
  const int e = ex ? (int)ex - 150 : -149;
  const uint64_t M = m * 1000u;                  // < 2^34
  uint64_t N;
  if (e >= 0) {
    if (e > 29) {
      // an integer of up to 128 bits (m * 2^e, e <= 104): its exact decimal digits, then ".000".  Five 32-bit limbs, divided by
      // 10^9 until nothing is left (a float sum that overflowed the 53-bit range prints like this in the reference too)
      uint32_t limb[5] = {0, 0, 0, 0, 0};
      {
        const int word = e >> 5, bit = e & 31;
        const uint64_t lo = m << bit;            // m < 2^24: fits 56 bits
        limb[word] = (uint32_t)lo;
        if (word + 1 < 5) limb[word + 1] = (uint32_t)(lo >> 32);
      }
      uint32_t chunk[5];
      int nchunk = 0;
      for (;;) {
        uint64_t rem = 0;
        bool any = false;
        for (int i = 4; i >= 0; --i) { const uint64_t cur = (rem << 32) | limb[i]; limb[i] = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u; any |= limb[i] != 0; }
        chunk[nchunk++] = (uint32_t)rem;
        if (!any) break;
      }
      if (u >> 31) s.put('-');
      put_i64(s, (int64_t)chunk[nchunk - 1]);
      for (int c = nchunk - 2; c >= 0; --c) { uint32_t d = 100000000u, v = chunk[c]; for (int k = 0; k < 9; ++k) { s.put((char)('0' + v / d)); v %= d; d /= 10u; } }
      s.put('.'); s.put('0'); s.put('0'); s.put('0');
      return;
    }
    N = M << e;
  } else {
    const int sh = -e;
    if (sh >= 64) N = 0;
    else {
      N = M >> sh;
      const uint64_t rem = M & ((1ull << sh) - 1ull), half = 1ull << (sh - 1);
      if (rem > half || (rem == half && (N & 1ull))) ++N;
    }
  }
  if (u >> 31) s.put('-');
  put_i64(s, (int64_t)(N / 1000u));
  s.put('.');
  const uint32_t fr = (uint32_t)(N % 1000u);
  s.put((char)('0' + fr / 100u)); s.put((char)('0' + (fr / 10u) % 10u)); s.put((char)('0' + fr % 10u));
}
template <class Sink> GDB_HD void put_fixed3(Sink& s, int32_t v, uint32_t*) { put_i32(s, v); }
GDB_HD bool asa_elem_valid(int32_t v) { return v != GDB_BCF_INT32_MISSING && v != GDB_BCF_INT32_VECTOR_END; }
GDB_HD bool asa_elem_valid(float v) { const uint32_t u = gdb_f2u(v); return u != GDB_BCF_FLOAT_MISSING_BITS && u != GDB_BCF_FLOAT_VECTOR_END_BITS; }

struct AsaShape { int outer; uint64_t num_valid_elements; uint64_t num_calls_with_field; };
// dimension 0 of the result and the number of valid elements: the first walk over the calls
template <class T> GDB_HD AsaShape asa_shape(const SiteCtx& cx, int64_t k, int f, bool allele_dep, bool alt_only, int num_merged, bool non_ref_exists, uint32_t* err) {
  AsaShape sh; sh.outer = 0; sh.num_valid_elements = 0; sh.num_calls_with_field = 0;
  const int64_t b = cx.hl.base[k], e = cx.hl.base[k + 1], s_k = cx.rec.start[k];
  for (int64_t t = b; t < e; ++t) {
    AsaCall a;
    if (!asa_call_open(cx, t, s_k, f, allele_dep, non_ref_exists, num_merged, a, err)) continue;
    ++sh.num_calls_with_field;
    const int outer = allele_dep ? (alt_only ? num_merged - 1 : num_merged) : (int)a.blob.n;
    if (outer > sh.outer) sh.outer = outer;
    for (int j = 0; j < outer; ++j) {
      const char* p; uint32_t nb;
      asa_entry(a, j, allele_dep, alt_only, p, nb);
      for (uint32_t i = 0; i + sizeof(T) <= nb; i += (uint32_t)sizeof(T)) if (asa_elem_valid(asa_rd_elem<T>(p + i))) ++sh.num_valid_elements;
    }
  }
  return sh;
}

// element_wise_sum of a 2-D field: the value text ("8.000,10.000|28.000|..."); false = no valid element, nothing written
template <class T, class Sink> GDB_HD void asa_sum_text(const SiteCtx& cx, int64_t k, int f, int outer, bool allele_dep, bool alt_only, int num_merged, bool non_ref_exists,
                                                      Sink& sink, uint32_t* err) {
  const GdbFieldDesc& fd = cx.pl.field[f];
  const int64_t b = cx.hl.base[k], e = cx.hl.base[k + 1], s_k = cx.rec.start[k];
  for (int j = 0; j < outer; ++j) {
    if (j) sink.put((char)fd.delim0);
    uint32_t inner = 0;                            // the longest inner vector of any valid call (valid elements or not)
    for (int64_t t = b; t < e; ++t) {
      AsaCall a;
      if (!asa_call_open(cx, t, s_k, f, allele_dep, non_ref_exists, num_merged, a, err)) continue;
      const char* p; uint32_t nb;
      asa_entry(a, j, allele_dep, alt_only, p, nb);
      if (nb / (uint32_t)sizeof(T) > inner) inner = nb / (uint32_t)sizeof(T);
    }
    for (uint32_t i = 0; i < inner; ++i) {
      if (i) sink.put((char)fd.delim1);
      bool have = false;
      T acc = 0;
      for (int64_t t = b; t < e; ++t) {
        AsaCall a;
        if (!asa_call_open(cx, t, s_k, f, allele_dep, non_ref_exists, num_merged, a, err)) continue;
        const char* p; uint32_t nb;
        asa_entry(a, j, allele_dep, alt_only, p, nb);
        if ((i + 1u) * (uint32_t)sizeof(T) > nb) continue;
        const T v = asa_rd_elem<T>(p + i * sizeof(T));
        if (!asa_elem_valid(v)) continue;
        if (have) acc += v; else { acc = v; have = true; }
      }
      if (have) put_fixed3(sink, acc, err);
    }
  }
}

// histogram_sum of a (bins, counts) pair of 2-D fields: per outer index the bins in ascending order, equal bins merged
template <class T1, class T2, class Sink> GDB_HD void asa_histogram_text(const SiteCtx& cx, int64_t k, int f_bin, int f_count, int outer, bool allele_dep, bool alt_only,
                                                                       int num_merged, bool non_ref_exists, Sink& sink, uint32_t* err) {
  const GdbFieldDesc& fd = cx.pl.field[f_bin];
  const int64_t b = cx.hl.base[k], e = cx.hl.base[k + 1], s_k = cx.rec.start[k];
  for (int j = 0; j < outer; ++j) {
    if (j) sink.put((char)fd.delim0);
    bool have_prev = false, first = true;
    T1 prev = 0;
    for (;;) {
      bool found = false;
      T1 best = 0;
      for (int pass = 0; pass < 2; ++pass) {        // pass 0: the next bin; pass 1: its count
        bool have = false;
        T2 total = 0;
        for (int64_t t = b; t < e; ++t) {
          AsaCall ab, ac;
          if (!asa_call_open(cx, t, s_k, f_bin, allele_dep, non_ref_exists, num_merged, ab, err)) continue;
          if (!asa_call_open(cx, t, s_k, f_count, allele_dep, non_ref_exists, num_merged, ac, err)) continue;
          const char *pb, *pc; uint32_t nb, nc;
          asa_entry(ab, j, allele_dep, alt_only, pb, nb);
          asa_entry(ac, j, allele_dep, alt_only, pc, nc);
          const uint32_t ne = nb / (uint32_t)sizeof(T1) < nc / (uint32_t)sizeof(T2) ? nb / (uint32_t)sizeof(T1) : nc / (uint32_t)sizeof(T2);
          for (uint32_t i = 0; i < ne; ++i) {
            const T1 vb = asa_rd_elem<T1>(pb + i * sizeof(T1));
            const T2 vc = asa_rd_elem<T2>(pc + i * sizeof(T2));
            if (!asa_elem_valid(vb) || !asa_elem_valid(vc)) continue;
            if (pass == 0) {
              if ((!have_prev || prev < vb) && (!found || vb < best)) { best = vb; found = true; }
            } else if (!(vb < best) && !(best < vb)) {
              if (have) total += vc; else { total = vc; have = true; }
            }
          }
        }
        if (pass == 0 && !found) break;
        if (pass == 1) {
          if (!first) sink.put((char)fd.delim1);
          put_fixed3(sink, best, err);
          sink.put((char)fd.delim1);
          put_fixed3(sink, total, err);
          first = false;
        }
      }
      if (!found) break;
      prev = best; have_prev = true;
    }
  }
}

// ---- the two reducers as INFO entries of a record: VCF text (";NAME=value") or a BCF typed string (key, bcf_enc_vchar) ------------
struct AsaInfoOut { bool bcf; int32_t key; const char* name; int name_len; };
template <class Body, class Sink> GDB_HD bool asa_emit_info(const AsaInfoOut& o, Sink& sink, bool& any, const Body& body) {
  CountSink cs;
  body(cs);
  if (cs.n == 0) return false;                     // bcf_update_info with 0 values removes the tag (htslib vcf.c)
  if (o.bcf) {
    bcf_enc_int1(sink, o.key);
    bcf_enc_size(sink, (int)cs.n, GDB_BT_CHAR);
  } else {
    if (any) sink.put(';');
    sink.write(o.name, o.name_len);
    sink.put('=');
  }
  body(sink);
  any = true;
  return true;
}
template <class T, class Sink> struct AsaSumBody {
  const SiteCtx& cx; int64_t k; int f; int outer; bool allele_dep, alt_only; int num_merged; bool non_ref_exists; uint32_t* err;
  GDB_HD void operator()(Sink& s) const { asa_sum_text<T>(cx, k, f, outer, allele_dep, alt_only, num_merged, non_ref_exists, s, err); }
  GDB_HD void operator()(CountSink& s) const { asa_sum_text<T>(cx, k, f, outer, allele_dep, alt_only, num_merged, non_ref_exists, s, err); }
};
template <class T1, class T2, class Sink> struct AsaHistBody {
  const SiteCtx& cx; int64_t k; int fb, fc; int outer; bool allele_dep, alt_only; int num_merged; bool non_ref_exists; uint32_t* err;
  GDB_HD void operator()(Sink& s) const { asa_histogram_text<T1, T2>(cx, k, fb, fc, outer, allele_dep, alt_only, num_merged, non_ref_exists, s, err); }
  GDB_HD void operator()(CountSink& s) const { asa_histogram_text<T1, T2>(cx, k, fb, fc, outer, allele_dep, alt_only, num_merged, non_ref_exists, s, err); }
};
// (CountSink as the record sink: the two overloads above would collide)
template <class T> struct AsaSumBody<T, CountSink> {
  const SiteCtx& cx; int64_t k; int f; int outer; bool allele_dep, alt_only; int num_merged; bool non_ref_exists; uint32_t* err;
  GDB_HD void operator()(CountSink& s) const { asa_sum_text<T>(cx, k, f, outer, allele_dep, alt_only, num_merged, non_ref_exists, s, err); }
};
template <class T1, class T2> struct AsaHistBody<T1, T2, CountSink> {
  const SiteCtx& cx; int64_t k; int fb, fc; int outer; bool allele_dep, alt_only; int num_merged; bool non_ref_exists; uint32_t* err;
  GDB_HD void operator()(CountSink& s) const { asa_histogram_text<T1, T2>(cx, k, fb, fc, outer, allele_dep, alt_only, num_merged, non_ref_exists, s, err); }
};

template <class Sink> GDB_HD bool info_asa_sum(const SiteCtx& cx, int64_t k, int f, int num_merged, bool non_ref_exists, bool remapping_needed, bool bcf, Sink& sink, bool& any,
                                             uint32_t* err) {
  const GdbFieldDesc& fd = cx.pl.field[f];
  const bool allele_dep = remapping_needed && (fd.length == GDB_VL_A || fd.length == GDB_VL_R);
  const bool alt_only = fd.length == GDB_VL_A;
  const AsaInfoOut o{bcf, cx.pl.bcf_id[f], cx.names.text + cx.names.field_name_off[f], cx.names.field_name_len[f]};
  if (fd.elem2d == GDB_ET_FLOAT) {
    const AsaShape sh = asa_shape<float>(cx, k, f, allele_dep, alt_only, num_merged, non_ref_exists, err);
    if (sh.num_valid_elements == 0) return false;
    return asa_emit_info(o, sink, any, AsaSumBody<float, Sink>{cx, k, f, sh.outer, allele_dep, alt_only, num_merged, non_ref_exists, err});
  }
  const AsaShape sh = asa_shape<int32_t>(cx, k, f, allele_dep, alt_only, num_merged, non_ref_exists, err);
  if (sh.num_valid_elements == 0) return false;
  return asa_emit_info(o, sink, any, AsaSumBody<int32_t, Sink>{cx, k, f, sh.outer, allele_dep, alt_only, num_merged, non_ref_exists, err});
}
template <class Sink> GDB_HD bool info_asa_histogram(const SiteCtx& cx, int64_t k, int fb, int fc, int num_merged, bool non_ref_exists, bool remapping_needed, bool bcf, Sink& sink,
                                                   bool& any, uint32_t* err) {
  const GdbFieldDesc& db = cx.pl.field[fb];
  const GdbFieldDesc& dc = cx.pl.field[fc];
  const bool allele_dep = remapping_needed && (dc.length == GDB_VL_A || dc.length == GDB_VL_R);
  const bool alt_only = dc.length == GDB_VL_A;
  const AsaInfoOut o{bcf, cx.pl.bcf_id[fb], cx.names.text + cx.names.field_name_off[fb], cx.names.field_name_len[fb]};
  const AsaShape sh = asa_shape<int32_t>(cx, k, fb, allele_dep, alt_only, num_merged, non_ref_exists, err);   // (only the calls and dimension 0 matter here)
  if (sh.num_calls_with_field == 0) return false;
  const bool bf = db.elem2d == GDB_ET_FLOAT, cf = dc.elem2d == GDB_ET_FLOAT;
  if (bf && cf) return asa_emit_info(o, sink, any, AsaHistBody<float, float, Sink>{cx, k, fb, fc, sh.outer, allele_dep, alt_only, num_merged, non_ref_exists, err});
  if (bf) return asa_emit_info(o, sink, any, AsaHistBody<float, int32_t, Sink>{cx, k, fb, fc, sh.outer, allele_dep, alt_only, num_merged, non_ref_exists, err});
  if (cf) return asa_emit_info(o, sink, any, AsaHistBody<int32_t, float, Sink>{cx, k, fb, fc, sh.outer, allele_dep, alt_only, num_merged, non_ref_exists, err});
  return asa_emit_info(o, sink, any, AsaHistBody<int32_t, int32_t, Sink>{cx, k, fb, fc, sh.outer, allele_dep, alt_only, num_merged, non_ref_exists, err});
}
