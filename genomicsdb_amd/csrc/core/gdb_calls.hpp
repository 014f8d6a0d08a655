// gdb_calls.hpp - one cell of the array as the JSON object `gt_mpi_gather --print-calls` prints for it.
//
// The reference hands every cell of the queried intervals to a SingleCellOperatorBase (VariantQueryProcessor::iterate_over_cells,
// src/main/cpp/src/genomicsdb/query_variants.cc:557-576); VariantCallPrintOperator::operate_on_columnar_cell
// (src/query_operations/variant_operations.cc:810-833) prints it with GenomicsDBColumnarCell::print (src/genomicsdb/variant_cell.cc:119-165)
// and the per-type printers of genomicsdb_columnar_field.cc:103-199, 386-417.  Here a cell is a thread: the emitter below runs once
// with a CountSink (length), once with a ByteSink (text), like every other emitter of this library; it compiles under hipcc and g++.
#pragma once
#include "gdb_core.hpp"

namespace genomicsdb_amd {

struct CallsNames {            // attribute names of the plan fields (the JSON keys), in device memory
  const char* text;
  const int32_t* off;          // [nfields + 1]
  const int64_t* array_row;    // query row (what a staged cell carries) -> row of the array (what the reference prints); null: the same
};
GDB_HD int64_t calls_array_row(const FragmentView& fr, const CallsNames& names, int64_t c) { return names.array_row ? names.array_row[fr.row[c]] : (int64_t)fr.row[c]; }

template <class Sink> GDB_HD void put_lit(Sink& s, const char* z) { int n = 0; while (z[n]) ++n; s.write(z, n); }
template <class Sink> GDB_HD void put_spaces(Sink& s, int n) { for (int i = 0; i < n; ++i) s.put(' '); }

// what std::ostream << float prints with the stream's defaults (= printf("%g")): sign, nan / inf, 0, six significant digits
template <class Sink> GDB_HD void put_float_ostream(Sink& s, float f) {
  const uint32_t bits = gdb_f2u(f);
  if (bits >> 31) s.put('-');
  if ((bits & 0x7F800000u) == 0x7F800000u) { put_lit(s, (bits & 0x7FFFFFu) ? "nan" : "inf"); return; }
  if ((bits & 0x7FFFFFFFu) == 0u) { s.put('0'); return; }
  put_float_g(s, f);
}

// columnar validity (genomicsdb_columnar_field.cc:359-377): a fixed-length field by its elements, a variable-length one by its size
GDB_HD bool calls_field_valid(const FragmentView& fr, const CombinePlan& pl, int f, int64_t c) {
  const GdbFieldDesc& fd = pl.field[f];
  int n;
  if (fr.col[f].off) { (void)cell_field<char>(fr, pl, f, c, n); return n > 0; }
  if (fd.elem == GDB_ET_INT) {
    const int32_t* p = cell_field<int32_t>(fr, pl, f, c, n);
    for (int i = 0; i < n; ++i) if (p[i] != GDB_TILEDB_NULL_INT32) return true;
  } else if (fd.elem == GDB_ET_FLOAT) {
    const float* p = cell_field<float>(fr, pl, f, c, n);
    for (int i = 0; i < n; ++i) if (gdb_f2u(p[i]) != GDB_TILEDB_NULL_FLOAT_BITS) return true;
  } else {
    const char* p = cell_field<char>(fr, pl, f, c, n);
    for (int i = 0; i < n; ++i) if (p[i] != GDB_TILEDB_NULL_CHAR) return true;
  }
  return false;
}

// GenomicsDBColumnarFieldPrintOperator<T, print_as_list>::print; a fixed-length field of one element prints bare, a multi-char field is a string
template <class Sink> GDB_HD void calls_put_field(Sink& s, const FragmentView& fr, const CombinePlan& pl, int f, int64_t c) {
  const GdbFieldDesc& fd = pl.field[f];
  const bool singleton = fd.length == GDB_VL_FIXED && fd.fixed_num == 1;
  int n;
  if (f == pl.f_ALT) {         // print_ALT_data_in_buffer_at_index: '|' separated, "&" is <NON_REF>
    const char* p = cell_field<char>(fr, pl, f, c, n);
    put_lit(s, "[ ");
    int b = 0;
    for (bool first = true;; first = false) {
      int e = b;
      while (e < n && p[e] != '|') ++e;
      if (!first) put_lit(s, ", ");
      s.put('"');
      if (e - b == 1 && p[b] == '&') put_lit(s, "<NON_REF>"); else s.write(p + b, e - b);
      s.put('"');
      if (e >= n) break;
      b = e + 1;
    }
    put_lit(s, " ]");
    return;
  }
  if (fd.elem == GDB_ET_CHAR && !singleton) { const char* p = cell_field<char>(fr, pl, f, c, n); s.put('"'); s.write(p, n); s.put('"'); return; }
  if (!singleton) put_lit(s, "[ ");
  if (fd.elem == GDB_ET_INT) {
    const int32_t* p = cell_field<int32_t>(fr, pl, f, c, n);
    for (int i = 0; i < (singleton ? 1 : n); ++i) { if (i) put_lit(s, ", "); put_i32(s, p[i]); }
  } else if (fd.elem == GDB_ET_FLOAT) {     // std::ostream << float: %g, precision 6
    const float* p = cell_field<float>(fr, pl, f, c, n);
    for (int i = 0; i < (singleton ? 1 : n); ++i) { if (i) put_lit(s, ", "); put_float_ostream(s, p[i]); }
  } else {                                   // char: the character; flag: std::ostream << bool
    const char* p = cell_field<char>(fr, pl, f, c, n);
    for (int i = 0; i < (singleton ? 1 : n); ++i) { if (i) put_lit(s, ", "); if (fd.elem == GDB_ET_FLAG) s.put(p[i] ? '1' : '0'); else s.put(p[i]); }
  }
  if (!singleton) put_lit(s, " ]");
}

// Is cell c one of the query interval [qb, qe]'s, and with which END?  (SingleCellTileDBIterator: genomicsdb_iterators.cc:181-301, 425-510, see
// kernels/gdb_pipeline.hip k_calls.)  with_intersecting: the intervals that began in front of qb and reach it count too.
GDB_HD bool calls_select(const FragmentView& fr, const int64_t* eff_end, int64_t c, int64_t qb, int64_t qe, bool with_intersecting, int64_t& end) {
  const int64_t b = fr.begin[c];
  if (b >= qb && b <= qe) { end = fr.end[c]; return true; }
  if (with_intersecting && b < qb && eff_end[c] >= qb) { end = eff_end[c]; return true; }
  return false;
}

// GenomicsDBColumnarCell::print.  [begin, end]: the cell's interval as the iterator hands it out (for a cell found by the search for
// intervals that intersect the query's begin: its END copy's coordinate, i.e. the END the loader truncated at the sample's next cell)
template <class Sink> GDB_HD void calls_emit_cell(Sink& s, const FragmentView& fr, const CombinePlan& pl, const QueryWindow& qw, const CallsNames& names, int64_t c,
                                                  int64_t end, int indent) {
  const int64_t begin = fr.begin[c];
  put_spaces(s, indent); put_lit(s, "{\n");
  put_spaces(s, indent + 4); put_lit(s, "\"row\": "); put_i64(s, calls_array_row(fr, names, c)); put_lit(s, ",\n");
  put_spaces(s, indent + 4); put_lit(s, "\"interval\": [ "); put_i64(s, begin); put_lit(s, ", "); put_i64(s, end); put_lit(s, " ],\n");
  const int ci = find_contig(qw, begin);
  if (ci >= 0) {
    const GdbContig& g = qw.contigs[ci];
    const int64_t pos = begin - g.offset;
    put_spaces(s, indent + 4); put_lit(s, "\"genomic_interval\": { \""); s.write(qw.contig_names + g.name_off, g.name_len);
    put_lit(s, "\" : [ "); put_i64(s, pos + 1); put_lit(s, ", "); put_i64(s, pos + 1 + (end - begin)); put_lit(s, " ] },\n");
  }
  put_spaces(s, indent + 4); put_lit(s, "\"fields\": {\n");
  bool first = true;
  for (int f = 0; f < pl.nfields; ++f) {
    if (!calls_field_valid(fr, pl, f, c)) continue;
    if (!first) put_lit(s, ",\n");
    put_spaces(s, indent + 8); s.put('"'); s.write(names.text + names.off[f], names.off[f + 1] - names.off[f]); put_lit(s, "\": ");
    calls_put_field(s, fr, pl, f, c);
    first = false;
  }
  s.put('\n'); put_spaces(s, indent + 4); put_lit(s, "}\n"); put_spaces(s, indent); s.put('}');
}


// VariantCallPrintCSVOperator::operate_on_columnar_cell -> GenomicsDBColumnarCell::print_csv (variant_cell.cc:167-184): row, begin, end, then every
// queried attribute behind END through the print_csv of its type (genomicsdb_columnar_field.cc:116-199, 419-424): a list type of variable length
// starts with its element count (0 and nothing else when the field is not valid), a fixed-length list that is not valid leaves its commas, a single
// value or a string that is not valid leaves nothing.  (ALT has no special treatment here: the stored string, "|" and "&" included.)
template <class Sink> GDB_HD void calls_emit_csv(Sink& s, const FragmentView& fr, const CombinePlan& pl, const CallsNames& names, int64_t c, int64_t end) {
  put_i64(s, calls_array_row(fr, names, c)); s.put(','); put_i64(s, fr.begin[c]); s.put(','); put_i64(s, end);
  for (int f = 0; f < pl.nfields; ++f) {
    s.put(',');
    const GdbFieldDesc& fd = pl.field[f];
    const bool singleton = fd.length == GDB_VL_FIXED && fd.fixed_num == 1;
    const bool is_var = fd.length != GDB_VL_FIXED;
    const bool valid = calls_field_valid(fr, pl, f, c);
    int n;
    if (fd.elem == GDB_ET_CHAR && !singleton) { const char* p = cell_field<char>(fr, pl, f, c, n); if (valid) s.write(p, n); continue; }
    if (singleton) {
      if (!valid) continue;
      if (fd.elem == GDB_ET_INT) put_i32(s, *cell_field<int32_t>(fr, pl, f, c, n));
      else if (fd.elem == GDB_ET_FLOAT) put_float_ostream(s, *cell_field<float>(fr, pl, f, c, n));
      else { const char ch = *cell_field<char>(fr, pl, f, c, n); if (fd.elem == GDB_ET_FLAG) s.put(ch ? '1' : '0'); else s.put(ch); }
      continue;
    }
    const void* p = fd.elem == GDB_ET_INT ? (const void*)cell_field<int32_t>(fr, pl, f, c, n) : fd.elem == GDB_ET_FLOAT ? (const void*)cell_field<float>(fr, pl, f, c, n)
                                                                                                                          : (const void*)cell_field<char>(fr, pl, f, c, n);
    if (is_var) put_i64(s, (int64_t)n);
    if (valid) {
      if (is_var) s.put(',');
      for (int i = 0; i < n; ++i) {
        if (i) s.put(',');
        if (fd.elem == GDB_ET_INT) put_i32(s, ((const int32_t*)p)[i]);
        else if (fd.elem == GDB_ET_FLOAT) put_float_ostream(s, ((const float*)p)[i]);
        else s.put(fd.elem == GDB_ET_FLAG ? (((const char*)p)[i] ? '1' : '0') : ((const char*)p)[i]);
      }
    } else if (!is_var) for (int i = 1; i < n; ++i) s.put(',');
  }
  s.put('\n');
}

// AlleleCountOperator::operate_on_columnar_cell (variant_operations.cc:951-1008) + normalize_REF_ALT_pair (:1012-1056): one line
// "column<TAB>REF<TAB>ALT" per GT element that names an ALT allele; counting equal lines (per query interval, ordered by column, REF, ALT) is
// print_allele_counts (:1069-1089).  gt_step: 2 when GT carries phase elements (BCF_VL_Phased_Ploidy), else 1.
template <class Sink> GDB_HD void calls_emit_allele_lines(Sink& s, const FragmentView& fr, const CombinePlan& pl, int64_t c, int gt_step, uint32_t* err) {
  if (pl.f_GT < 0 || !calls_field_valid(fr, pl, pl.f_REF, c) || !calls_field_valid(fr, pl, pl.f_ALT, c) || !calls_field_valid(fr, pl, pl.f_GT, c)) return;
  int nref, nalt, ngt;
  const char* ref = cell_field<char>(fr, pl, pl.f_REF, c, nref);
  const char* alt = cell_field<char>(fr, pl, pl.f_ALT, c, nalt);
  const int32_t* gt = cell_field<int32_t>(fr, pl, pl.f_GT, c, ngt);
  for (int i = 0; i < ngt; i += gt_step) {
    const int32_t g = gt[i];
    if (!gdb_int_valid(g) || g <= 0) continue;
    int b = 0, e = 0, idx = 0;                       // the (g - 1)-th '|' separated piece of ALT (empty pieces count: memchr)
    for (;;) { e = b; while (e < nalt && alt[e] != '|') ++e; if (idx == g - 1 || e >= nalt) break; b = e + 1; ++idx; }
    if (idx != g - 1) { *err |= GDB_ERR_INTERNAL; continue; }
    const char* a = alt + b;
    int alen = e - b, rlen = nref;
    if (rlen > 1 && alen > 0) {                      // the cell contains a deletion: bring this allele's pair to its own normal form
      if (allele_is_symbolic(a, alen)) rlen = 1;
      else {
        int suffix = 0;
        if (alen >= rlen) suffix = rlen - 1;         // SNV / insertion next to a deletion: the last REF length - 1 bases are shared
        else if (alen > 1) suffix = alen - 1;        // a shorter deletion next to a longer one
        rlen -= suffix; alen -= suffix;
      }
    }
    put_i64(s, fr.begin[c]); s.put('\t'); s.write(ref, rlen); s.put('\t'); s.write(a, alen); s.put('\n');
  }
}

}  // namespace genomicsdb_amd
