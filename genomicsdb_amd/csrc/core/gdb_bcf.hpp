// gdb_bcf.hpp - the sample columns of a record as BCF2 values instead of text (output format "bu").
//
// Same roles as the text emitters of gdb_core.hpp (entry_emit / emit_field): what a live call contributes to a record of a given
// type, computed once per (cell, record type) / (record, variant call) / no-call slot - here as a small binary entry:
//     u16 summary[nf]            one per FORMAT field of the record type, in emission order (padded to a multiple of 4 bytes;
//                                up to 8 fields arrive with the entry's first 16-byte load):
//                                bits 0..13 element count n (0: the call has no value for this field), bits 14..15 integer
//                                class (0 int8, 1 int16, 2 int32: the narrowest BCF type that holds the call's values)
//     body                       the n elements of every field back to back: int32 / float as 4 bytes (on a 4-byte boundary of the
//                                entry: zero padding behind a string field), char as 1 byte
// The page assembly reduces the summaries of the N samples of a record to the per-(record, field) vector length and type
// (what htslib's bcf_update_format does in bcf_enc_vint, vcf.c) and then converts and pads every entry into its fixed-stride
// place: absent values become [missing, vector_end ...] (GT: vector_end; the htsjdk flags change both), as
// VariantFieldHandler::collect_and_extend_fields does (variant_field_handler.cc:804-871).  GT elements are BCF-encoded here:
// (allele + 1) << 1 | phased (encode_GT_vector, broad_combined_gvcf.cc:53-138).
#pragma once
#include "gdb_core.hpp"

struct BinTrack {            // running summary of the field being emitted
  uint32_t n;
  int32_t mn, mx;
  GDB_HD void reset() { n = 0; mn = INT32_MAX; mx = INT32_MIN + 1; }
  GDB_HD void see(int32_t v) { if (v != GDB_BCF_INT32_MISSING && v != GDB_BCF_INT32_VECTOR_END) { if (v < mn) mn = v; if (v > mx) mx = v; } }
  GDB_HD uint32_t word() const { const int t = bcf_int_type(mn, mx); return (n & 0x3FFFu) | ((uint32_t)(t - GDB_BT_INT8) << 14); }
};
template <class Sink> GDB_HD void bin_put_i32(Sink& s, BinTrack& tr, int32_t v) { bcf_put_u32(s, (uint32_t)v); tr.see(v); ++tr.n; }
template <class Sink> GDB_HD void bin_put_f32(Sink& s, BinTrack& tr, float v) { bcf_put_u32(s, gdb_f2u(v)); ++tr.n; }
template <class Sink> GDB_HD void bin_put_elem(Sink& s, BinTrack& tr, int32_t v) { bin_put_i32(s, tr, v); }
template <class Sink> GDB_HD void bin_put_elem(Sink& s, BinTrack& tr, float v) { bin_put_f32(s, tr, v); }

template <class Sink, class T> GDB_HD void bin_remap_alleles(Sink& s, BinTrack& tr, const T* p, int n, const EntryMaps& em, int num_merged, bool alt_only) {
  const int length = alt_only ? num_merged - 1 : num_merged;
  for (int j = 0; j < length; ++j) {
    const int aj = alt_only ? j + 1 : j;
    const int in_j = em.lookup(aj);
    const int idx = alt_only ? in_j - 1 : in_j;
    T v;
    if (in_j >= 0 && idx >= 0 && idx < n) v = p[idx]; else elem_set_missing(v);
    bin_put_elem(s, tr, v);
  }
}
template <class Sink, class T> GDB_HD void bin_remap_genotypes(Sink& s, BinTrack& tr, const T* p, int n, const EntryMaps& em, int num_merged, int ploidy, uint32_t* err) {
  if (ploidy == 1) {
    for (int j = 0; j < num_merged; ++j) {
      const int in_j = em.lookup(j);
      T v;
      if (in_j >= 0 && in_j < n) v = p[in_j]; else elem_set_missing(v);
      bin_put_elem(s, tr, v);
    }
  } else if (ploidy == 2) {
    for (int kk = 0; kk < num_merged; ++kk) {
      const int in_k = em.lookup(kk);
      for (int j = 0; j <= kk; ++j) {
        const int in_j = em.lookup(j);
        const bool both = in_j >= 0 && in_k >= 0;
        const int gi = both ? gdb_alleles2gt(in_j, in_k) : 0;
        T v;
        if (both && gi < n) v = p[gi]; else elem_set_missing(v);
        bin_put_elem(s, tr, v);
      }
    }
  } else if (ploidy >= 3 && ploidy <= GDB_MAX_PLOIDY) {
    int g[GDB_MAX_PLOIDY], in[GDB_MAX_PLOIDY];
    for (int q = 0; q < ploidy; ++q) g[q] = 0;
    do {
      bool missing = false;
      for (int q = 0; q < ploidy; ++q) { in[q] = em.lookup(g[q]); if (in[q] < 0) missing = true; }
      T v;
      elem_set_missing(v);
      if (!missing) { const int64_t gi = gdb_genotype_index(in, ploidy); if (gi < n) v = p[gi]; }
      bin_put_elem(s, tr, v);
    } while (gdb_next_genotype(g, ploidy, num_merged));
  } else {
    *err |= GDB_ERR_UNSUPPORTED_PLOIDY;
  }
}

// values of FORMAT field i of a live call (mirrors emit_field)
template <class Sink> GDB_HD void bin_field(Sink& s, BinTrack& tr, const EntryCtx& cx, const RecordInfo& ri, const EntryMaps& em, int i, int64_t c, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  const int f = pl.format_field[i];
  const GdbFieldDesc& fd = pl.field[f];
  if (f == pl.f_DP && pl.f_DP_FORMAT >= 0) {  // FORMAT DP := DP_FORMAT (broad_combined_gvcf.cc:689-719)
    if (field_valid(cx.cm, c, pl.f_DP_FORMAT)) { int n; const int32_t* p = cell_field<int32_t>(cx.fr, pl, pl.f_DP_FORMAT, c, n); if (n > 0 && gdb_int_valid(p[0])) bin_put_i32(s, tr, p[0]); }
  } else if (!field_valid(cx.cm, c, f)) {
    // no value: the page assembly writes [missing, vector_end ...]
  } else if (f == pl.f_GT) {
    int n;
    const int32_t* g = cell_field<int32_t>(cx.fr, pl, f, c, n);
    const bool pp = fd.length == GDB_VL_PP;
    const int step = pp ? 2 : 1;
    int out_i = 0;
    for (int j = 0; j < n; j += step, ++out_i) {
      const int32_t phased = (pp && j > 0 && g[j - 1] > 0) ? 1 : 0;
      const int32_t m = pl.produce_GT_field ? gt_merged_allele(cx, ri, em, g[j], out_i) : -1;
      bin_put_i32(s, tr, ((m < 0 ? 0 : m + 1) << 1) | phased);
    }
  } else if (fd.elem == GDB_ET_CHAR || fd.elem == GDB_ET_FLAG) {
    int n;
    const char* p = cell_field<char>(cx.fr, pl, f, c, n);
    for (int j = 0; j < n; ++j) { s.put(p[j]); ++tr.n; }
  } else if (fd.elem == GDB_ET_FLOAT) {
    int n;
    const float* p = cell_field<float>(cx.fr, pl, f, c, n);
    const bool allele_dep = fd.length == GDB_VL_A || fd.length == GDB_VL_R || fd.length == GDB_VL_G;
    if (!em.remap || !allele_dep) { for (int j = 0; j < n; ++j) bin_put_f32(s, tr, p[j]); }
    else if (fd.length == GDB_VL_G) bin_remap_genotypes(s, tr, p, n, em, ri.num_merged, (int)GDB_CF_PLOIDY(em.cf), err);
    else bin_remap_alleles(s, tr, p, n, em, ri.num_merged, fd.length == GDB_VL_A);
  } else if (fd.elem != GDB_ET_INT) {
    *err |= GDB_ERR_INTERNAL;
  } else {
    int n;
    const int32_t* p = cell_field<int32_t>(cx.fr, pl, f, c, n);
    const bool allele_dep = fd.length == GDB_VL_A || fd.length == GDB_VL_R || fd.length == GDB_VL_G;
    if (!em.remap || !allele_dep) { for (int j = 0; j < n; ++j) bin_put_i32(s, tr, p[j]); }
    else if (fd.length == GDB_VL_G) bin_remap_genotypes(s, tr, p, n, em, ri.num_merged, (int)GDB_CF_PLOIDY(em.cf), err);
    else bin_remap_alleles(s, tr, p, n, em, ri.num_merged, fd.length == GDB_VL_A);
  }
}

GDB_HD int bcf_field_elem_size(const CombinePlan& pl, int fmt_i) {
  const int f = pl.format_field[fmt_i];
  const int e = (f == pl.f_DP && pl.f_DP_FORMAT >= 0) ? GDB_ET_INT : pl.field[f].elem;
  return (e == GDB_ET_CHAR || e == GDB_ET_FLAG) ? 1 : 4;
}
// binary entry of one (record, sample) column.  c < 0: the sample has no live call (every summary stays 0).
template <class Sink> GDB_HD Sink entry_emit_bin(const EntryCtx& cx, const RecordInfo& ri, int64_t c, Sink s, uint32_t* err) {
  const CombinePlan& pl = cx.pl;
  EntryMaps em;
  int8_t m2i_store[GDB_MAX_MERGED_ALLELES];
  build_entry_maps(cx, ri, c, em, m2i_store, err);
  int nf = 0;
  for (uint32_t m = ri.fmt_mask; m; m &= m - 1) ++nf;
  const uint32_t hdr_at = s.pos();
  for (int q = 0; q < nf; q += 2) bcf_put_u32(s, 0u);
  int q = 0;
  uint32_t pair = 0;          // the summaries of fields q & ~1 and q | 1
  for (int i = 0; i < pl.n_format; ++i) {
    if (!((ri.fmt_mask >> i) & 1)) continue;
    if (c >= 0) {
      BinTrack tr;
      tr.reset();
      if (bcf_field_elem_size(pl, i) == 4) while ((s.pos() - hdr_at) & 3u) s.put((char)0);   // 4-byte values start on a 4-byte boundary of the entry
      bin_field(s, tr, cx, ri, em, i, c, err);
      if (tr.n > 0x3FFFu) *err |= GDB_ERR_INTERNAL;
      if (tr.n) pair |= tr.word() << (16 * (q & 1));
    }
    if ((q & 1) || q == nf - 1) {
      if (pair) s.patch_u32(hdr_at + 4u * (uint32_t)(q >> 1), pair);
      pair = 0;
    }
    ++q;
  }
  return s;
}
GDB_HD uint32_t bcf_summary_bytes(int nf) { return ((uint32_t)nf * 2u + 3u) & ~3u; }
GDB_HD uint32_t bcf_summary_n(uint32_t s) { return s & 0x3FFFu; }
GDB_HD uint32_t bcf_summary_class(uint32_t s) { return (s & 0x8000u) ? 2u : ((s >> 14) & 1u); }   // (a maximum kept as an OR of the two bits)

// ---- per-(record, field) layout of the FORMAT block ------------------------------------------------------------------------
// what the N summaries of a record's samples reduce to, per FORMAT field: bits 0..13 = longest vector, 14..15 = OR of the classes
GDB_HD uint32_t bcf_summary_max(uint32_t a, uint32_t b) {
  const uint32_t n = (a & 0x3FFFu) > (b & 0x3FFFu) ? (a & 0x3FFFu) : (b & 0x3FFFu);
  return n | ((a | b) & 0xC000u);
}
GDB_HD int bcf_field_type(const CombinePlan& pl, int fmt_i, uint32_t summary) {   // BCF type code of the field in this record
  const int f = pl.format_field[fmt_i];
  const int e = (f == pl.f_DP && pl.f_DP_FORMAT >= 0) ? GDB_ET_INT : pl.field[f].elem;
  if (e == GDB_ET_CHAR || e == GDB_ET_FLAG) return GDB_BT_CHAR;
  if (e == GDB_ET_FLOAT) return GDB_BT_FLOAT;
  return GDB_BT_INT8 + (int)bcf_summary_class(summary);
}
