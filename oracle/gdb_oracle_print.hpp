// gdb_oracle_print.hpp - TEST ORACLE, NOT PRODUCT CODE: `gt_mpi_gather --print-calls` restated.
//
// tools/src/gt_mpi_gather.cc:369-383 (print_calls, COMMAND_PRINT_CALLS) -> VariantQueryProcessor::iterate_over_cells
// (query_variants.cc:557-576) over a SingleCellTileDBIterator (genomicsdb_iterators.cc:181-301, 425-510) with a
// VariantCallPrintOperator (variant_operations.cc:803-843); a cell is printed by GenomicsDBColumnarCell::print
// (variant_cell.cc:119-165) with the per-type printers of genomicsdb_columnar_field.cc:103-199, 386-417.
//
// The iterator, per query column interval [b, e]:
//  * "find intersecting intervals" (:255-283, :425-471): the array is read from column b on; of every queried row the FIRST cell
//    counts - if it is the END copy of an interval that began before b (coordinates = its END, END attribute = its begin) the
//    interval intersects b: coordinates and END are swapped and the cell waits in a priority queue ordered column-major; the
//    search ends when every queried row has been seen or the array is exhausted;
//  * operator++ (:474-510) first hands out the queued cells (by begin column, then row), then traverses [b, e] and skips END copies
//    and rows that are not queried (:535-637);
//  * an interval that yields no cell at all prints nothing, not even its header (at_new_query_column_interval() is a cell's property).
#pragma once
#include "gdb_oracle_scan.hpp"

#include <cstdio>
#include <set>

namespace gdb_oracle {

// what `std::ostream << float` prints with the stream's defaults: %g, precision 6
inline void print_float_like_ostream(std::string& o, float v) { char b[64]; snprintf(b, sizeof b, "%g", (double)v); o += b; }

// GenomicsDBColumnarFieldPrintOperator<T, print_as_list>::print (genomicsdb_columnar_field.cc:103-199); singleton = fixed length 1 (:272-279)
inline void print_columnar_field(std::string& o, ElementType et, bool singleton, const uint8_t* p, unsigned n) {
  auto one = [&](unsigned i) {
    char b[32];
    switch (et) {
      case ET_INT: { int32_t v; memcpy(&v, p + 4u * i, 4); snprintf(b, sizeof b, "%d", v); o += b; break; }
      case ET_FLOAT: { float v; memcpy(&v, p + 4u * i, 4); print_float_like_ostream(o, v); break; }
      case ET_FLAG: o += p[i] ? "1" : "0"; break;           // ostream << bool
      default: o.push_back((char)p[i]); break;              // ostream << char
    }
  };
  if (et == ET_CHAR && !singleton) { o += "\""; o.append((const char*)p, n); o += "\""; return; }   // a multi-char field is a string
  if (singleton) { one(0); return; }
  o += "[ ";
  one(0);
  for (unsigned i = 1; i < n; ++i) { o += ", "; one(i); }
  o += " ]";
}

// GenomicsDBColumnarField::print_ALT_data_in_buffer_at_index (:386-417)
inline void print_columnar_ALT(std::string& o, const uint8_t* p, unsigned n) {
  o += "[ ";
  size_t s = 0;
  bool first = true;
  for (;;) {
    size_t e = s;
    while (e < n && p[e] != '|') ++e;
    if (!first) o += ", ";
    o += "\"";
    if (e - s == 1 && p[s] == '&') o += g_vcf_NON_REF; else o.append((const char*)p + s, e - s);
    o += "\"";
    first = false;
    if (e >= n) break;
    s = e + 1;
  }
  o += " ]";
}

// columnar validity (genomicsdb_columnar_field.cc:359-377): fixed-length fields by their elements, variable-length ones by their size
inline bool columnar_field_valid(const SchemaAttr& sa, const CellAttrView& v) {
  if (sa.var) return v.num > 0;
  for (unsigned i = 0; i < v.num; ++i) {
    bool missing;
    if (sa.et == ET_INT) { int32_t x; memcpy(&x, v.ptr + 4u * i, 4); missing = is_tiledb_missing_value(x); }
    else if (sa.et == ET_FLOAT) { float x; memcpy(&x, v.ptr + 4u * i, 4); missing = is_tiledb_missing_value(x); }
    else missing = is_tiledb_missing_value((char)v.ptr[i]);
    if (!missing) return true;
  }
  return false;
}

// GenomicsDBColumnarCell::print (variant_cell.cc:119-165); begin / end: the cell's interval after the iterator's swap
inline void print_columnar_cell(std::string& o, const VariantArray& array, const QueryConfig& qc, const VidMapper& vid, const DiskCell& d, int64_t begin, int64_t end,
                                const std::string& indent) {
  std::vector<CellAttrView> attr;
  parse_cell_attributes(array.schema, d.raw, attr);
  const std::string in1 = indent + "    ", in2 = in1 + "    ";
  char b[160];
  o += indent + "{\n";
  snprintf(b, sizeof b, "\"row\": %lld,\n", (long long)d.row); o += in1 + b;
  snprintf(b, sizeof b, "\"interval\": [ %lld, %lld ],\n", (long long)begin, (long long)end); o += in1 + b;
  std::string contig; int64_t pos = 0;
  if (vid.get_contig_location(begin, contig, pos)) {
    snprintf(b, sizeof b, "\" : [ %lld, %lld ] },\n", (long long)(pos + 1), (long long)(pos + 1 + (end - begin)));
    o += in1 + "\"genomic_interval\": { \"" + contig + b;
  }
  o += in1 + "\"fields\": {\n";
  bool first = true;
  for (unsigned i = 1; i < qc.num_queried_attributes(); ++i) {          // (the first queried attribute is always END)
    const QueryAttr& qa = qc.attrs[i];
    const SchemaAttr& sa = array.schema.attrs[qa.schema_idx];
    const CellAttrView& v = attr[qa.schema_idx];
    if (!columnar_field_valid(sa, v)) continue;
    if (!first) o += ",\n";
    o += in2 + "\"" + qa.name + "\": ";
    if (qc.get_known_field_enum_for_query_idx(i) == GVCF_ALT_IDX) print_columnar_ALT(o, v.ptr, v.num);
    else {
      ElementType et = sa.et;
      if (qa.info && qa.info->et == ET_FLAG) et = ET_FLAG;
      const bool singleton = qa.info ? (qa.info->ld == VL_FIXED && qa.info->num_elements == 1u) : (!sa.var && sa.num == 1u);
      print_columnar_field(o, et, singleton, v.ptr, v.num);
    }
    first = false;
  }
  o += "\n" + in1 + "}\n" + indent + "}";
}

// the iterator: the cells of one query interval in the order operator++ hands them out (begin / end after the swap of an END copy)
struct CellHit { int64_t begin, end; const DiskCell* d; };
inline std::vector<CellHit> cells_of_interval(const VariantArray& array, const QueryConfig& qc, int64_t b, int64_t e, bool whole_array) {
  std::vector<CellHit> hits;
  const size_t from = std::lower_bound(array.cells.begin(), array.cells.end(), b, [](const DiskCell& c, int64_t v) { return c.col < v; }) - array.cells.begin();
  if (!whole_array) {      // find intersecting intervals: the first cell of every queried row at or behind column b
    std::set<int64_t> seen;
    for (size_t i = from; i < array.cells.size() && seen.size() < qc.get_num_rows_to_query(); ++i) {
      const DiskCell& d = array.cells[i];
      if (!qc.is_queried_array_row_idx(d.row) || !seen.insert(d.row).second) continue;
      if (d.END < d.col && d.END < b) hits.push_back({d.END, d.col, &d});      // an END copy of an interval that began before b: swapped
    }
    std::sort(hits.begin(), hits.end(), [](const CellHit& x, const CellHit& y) { return x.begin < y.begin || (x.begin == y.begin && x.d->row < y.d->row); });
  }
  for (size_t i = from; i < array.cells.size() && array.cells[i].col <= e; ++i) {   // simple traversal
    const DiskCell& d = array.cells[i];
    if (d.END < d.col || !qc.is_queried_array_row_idx(d.row)) continue;
    hits.push_back({d.col, d.END, &d});
  }
  return hits;
}
inline std::vector<std::pair<int64_t, int64_t>> intervals_of(const QueryConfig& qc, bool& whole_array) {
  std::vector<std::pair<int64_t, int64_t>> ivs = qc.column_intervals;
  whole_array = ivs.empty();
  if (whole_array) ivs.emplace_back(0, INT64_MAX - 1);
  return ivs;
}

// print_calls + VariantCallPrintOperator
inline std::string print_calls(const VariantArray& array, const QueryConfig& qc, const VidMapper& vid) {
  const std::string ip = "    ";
  std::string o = "{\n" + ip + "\"variant_calls\": [\n";
  const std::string p0 = ip + ip, p1 = p0 + ip, p2 = p1 + ip;
  unsigned intervals_printed = 0;
  bool whole_array;
  for (const auto& iv : intervals_of(qc, whole_array)) {
    const std::vector<CellHit> hits = cells_of_interval(array, qc, iv.first, iv.second, whole_array);
    if (hits.empty()) continue;
    if (intervals_printed) o += "\n" + p1 + "]\n" + p0 + "},\n";
    char bb[96];
    snprintf(bb, sizeof bb, "\"query_interval\": [ %lld, %lld ],\n", (long long)iv.first, (long long)iv.second);
    o += p0 + "{\n" + p1 + bb + p1 + "\"variant_calls\": [\n";
    for (size_t h = 0; h < hits.size(); ++h) {
      if (h) o += ",\n";
      print_columnar_cell(o, array, qc, vid, *hits[h].d, hits[h].begin, hits[h].end, p2);
    }
    ++intervals_printed;
  }
  if (intervals_printed) o += "\n" + p1 + "]\n" + p0 + "}";
  o += "\n" + ip + "]\n}\n";
  return o;
}

// --print-csv: VariantCallPrintCSVOperator::operate_on_columnar_cell (variant_operations.cc:899-903) -> GenomicsDBColumnarCell::print_csv
// (variant_cell.cc:167-184) -> GenomicsDBColumnarFieldPrintOperator<...>::print_csv (genomicsdb_columnar_field.cc:116-199, 419-424).
// (No golden in the reference's tests: parity unpinned.)
inline std::string print_csv(const VariantArray& array, const QueryConfig& qc) {
  std::string o;
  bool whole_array;
  for (const auto& iv : intervals_of(qc, whole_array))
    for (const CellHit& h : cells_of_interval(array, qc, iv.first, iv.second, whole_array)) {
      std::vector<CellAttrView> attr;
      parse_cell_attributes(array.schema, h.d->raw, attr);
      char b[96];
      snprintf(b, sizeof b, "%lld,%lld,%lld", (long long)h.d->row, (long long)h.begin, (long long)h.end);
      o += b;
      for (unsigned i = 1; i < qc.num_queried_attributes(); ++i) {
        o += ",";
        const QueryAttr& qa = qc.attrs[i];
        const SchemaAttr& sa = array.schema.attrs[qa.schema_idx];
        const CellAttrView& v = attr[qa.schema_idx];
        const bool valid = columnar_field_valid(sa, v);
        ElementType et = sa.et;
        if (qa.info && qa.info->et == ET_FLAG) et = ET_FLAG;
        const bool singleton = qa.info ? (qa.info->ld == VL_FIXED && qa.info->num_elements == 1u) : (!sa.var && sa.num == 1u);
        const bool is_var = qa.info ? qa.info->ld != VL_FIXED : sa.var;
        auto one = [&](unsigned k) {
          char t[32];
          if (et == ET_INT) { int32_t x; memcpy(&x, v.ptr + 4u * k, 4); snprintf(t, sizeof t, "%d", x); o += t; }
          else if (et == ET_FLOAT) { float x; memcpy(&x, v.ptr + 4u * k, 4); print_float_like_ostream(o, x); }
          else if (et == ET_FLAG) o += v.ptr[k] ? "1" : "0";
          else o.push_back((char)v.ptr[k]);
        };
        if (et == ET_CHAR && !singleton) { if (valid) o.append((const char*)v.ptr, v.num); continue; }   // string: the bytes
        if (singleton) { if (valid) one(0); continue; }
        if (is_var) { snprintf(b, sizeof b, "%u", v.num); o += b; }
        if (valid) { if (is_var) o += ","; one(0); for (unsigned k = 1; k < v.num; ++k) { o += ","; one(k); } }
        else if (!is_var) for (unsigned k = 1; k < v.num; ++k) o += ",";
      }
      o += "\n";
    }
  return o;
}

// --print-AC: AlleleCountOperator (variant_operations.cc:905-1089) - operate_on_columnar_cell (:951-1008), normalize_REF_ALT_pair (:1012-1056),
// print_allele_counts (:1069-1089); one map per query interval that has cells.  (No golden in the reference's tests: parity unpinned.)
inline std::string print_allele_counts(const VariantArray& array, const QueryConfig& qc) {
  if (!qc.is_defined_query_idx_for_known_field_enum(GVCF_GT_IDX)) throw OracleException("GT field must be queried for AlleleCountOperator");
  const unsigned gi = qc.get_query_idx_for_known_field_enum(GVCF_GT_IDX), ri = qc.get_query_idx_for_known_field_enum(GVCF_REF_IDX), ai = qc.get_query_idx_for_known_field_enum(GVCF_ALT_IDX);
  const unsigned step = (qc.attrs[gi].info && qc.attrs[gi].info->ld == VL_PP) ? 2u : 1u;
  std::string o;
  bool whole_array;
  for (const auto& iv : intervals_of(qc, whole_array)) {
    std::map<int64_t, std::map<std::pair<std::string, std::string>, uint64_t>> counts;
    for (const CellHit& h : cells_of_interval(array, qc, iv.first, iv.second, whole_array)) {
      std::vector<CellAttrView> attr;
      parse_cell_attributes(array.schema, h.d->raw, attr);
      const CellAttrView& R = attr[qc.attrs[ri].schema_idx]; const CellAttrView& A = attr[qc.attrs[ai].schema_idx]; const CellAttrView& G = attr[qc.attrs[gi].schema_idx];
      if (!columnar_field_valid(array.schema.attrs[qc.attrs[ri].schema_idx], R) || !columnar_field_valid(array.schema.attrs[qc.attrs[ai].schema_idx], A) ||
          !columnar_field_valid(array.schema.attrs[qc.attrs[gi].schema_idx], G)) continue;
      std::vector<std::string> alts;          // memchr on '|': empty pieces count
      { std::string a((const char*)A.ptr, A.num); size_t s0 = 0; for (;;) { size_t e0 = a.find('|', s0); if (e0 == std::string::npos) { alts.push_back(a.substr(s0)); break; } alts.push_back(a.substr(s0, e0 - s0)); s0 = e0 + 1; } }
      for (unsigned i = 0; i < G.num; i += step) {
        int32_t g; memcpy(&g, G.ptr + 4u * i, 4);
        if (g == bcf_int32_missing || g == bcf_int32_vector_end || g <= 0) continue;
        ORACLE_VERIFY((size_t)(g - 1) < alts.size());
        std::pair<std::string, std::string> ra(std::string((const char*)R.ptr, R.num), alts[(size_t)g - 1]);
        const size_t rl = ra.first.size(), al = ra.second.size();
        if (rl > 1u && al) {
          if (VariantUtils::is_symbolic_allele(ra.second)) ra.first.resize(1u);
          else {
            size_t suffix = 0;
            if (al == rl) suffix = rl - 1u; else if (al > rl) suffix = rl - 1u; else if (al > 1u) suffix = al - 1u;
            ra.first.resize(rl - suffix); ra.second.resize(al - suffix);
          }
        }
        ++counts[h.begin][ra];
      }
    }
    for (const auto& col : counts)
      for (const auto& e : col.second) o += std::to_string(col.first) + " " + e.first.first + " " + e.first.second + " " + std::to_string(e.second) + "\n";
  }
  return o;
}

}  // namespace gdb_oracle
