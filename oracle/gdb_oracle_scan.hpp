// gdb_oracle_scan.hpp - TEST ORACLE (cell model + sweep).  NOT PRODUCT CODE.
//
// CPU restatement, one cell at a time, of the reference's scan:
//   cell layout / parser      src/genomicsdb/variant_cell.cc:79-117, src/vcf/vcf2binary.cc:991-1196
//   END-copy duplication      src/loader/load_operators.cc:33-79, :161-298  (LoaderArrayWriter)
//   field model + validity    include/genomicsdb/variant_field_data.h:99-105,:221-246,:365-384,:511-537
//   sweep                     src/genomicsdb/query_variants.cc:296-555 (handle_gvcf_ranges, scan_and_operate,
//                             scan_handle_cell), :845-941 (gt_get_column), :1014-1117 (gt_fill_row)
//   deletion / ref-block flags include/vcf/known_field_info.h:209-237, src/utils/known_field_info.cc:310-319
#pragma once
#include <queue>

#include "gdb_oracle_meta.hpp"

namespace gdb_oracle {

static const std::string g_vcf_NON_REF = "<NON_REF>";
static const std::string g_vcf_SPANNING_DELETION = "*";
inline bool IS_NON_REF_ALLELE(const std::string& a) { return a.length() > 0 && a[0] == '&'; }

struct VariantUtils {
  static bool is_symbolic_allele(const std::string& a) {
    return IS_NON_REF_ALLELE(a) || a == g_vcf_SPANNING_DELETION ||
           (a.length() > 0u && ((a[0] == '<' && a[a.length() - 1u] == '>') ||
                                (a.find_first_of('[') != std::string::npos || a.find_first_of(']') != std::string::npos)));
  }
  static bool is_deletion(const std::string& REF, const std::string& alt) {
    return REF.length() > 1u && ((alt.length() == 1u && alt[0] == '*') || (!is_symbolic_allele(alt) && alt.length() < REF.length()));
  }
  static bool contains_deletion(const std::string& REF, const std::vector<std::string>& ALT) {
    if (REF.length() <= 1u) return false;
    for (auto& a : ALT) if (!is_symbolic_allele(a) && a.length() < REF.length()) return true;
    return false;
  }
  static bool is_reference_block(const std::string& REF, const std::vector<std::string>& ALT) {
    return REF.length() == 1u && ALT.size() == 1u && IS_NON_REF_ALLELE(ALT[0]);
  }
};

// ---- one cell of the sparse array as the scan sees it ---------------------------------------------
struct CellAttrView { const uint8_t* ptr = nullptr; unsigned num = 0; };  // num = #elements

struct ParsedCell {
  int64_t row = 0, col = 0;       // coordinates (col may be an END-copy coordinate)
  int64_t END = 0;                // value of the END attribute as stored (begin for END copies)
  const uint8_t* raw = nullptr;   // begin-cell bytes
  std::vector<CellAttrView> attr; // per schema attribute
};

// BufferVariantCell::set_cell (variant_cell.cc:79-117)
inline void parse_cell_attributes(const ArraySchema& schema, const uint8_t* cell, std::vector<CellAttrView>& out, size_t* cell_size_out = nullptr) {
  const uint8_t* p = cell + 16;
  uint64_t cell_size;
  memcpy(&cell_size, p, 8);
  p += 8;
  out.resize(schema.attrs.size());
  for (size_t i = 0; i < schema.attrs.size(); ++i) {
    const auto& a = schema.attrs[i];
    unsigned n = a.num;
    if (a.var) { int32_t len; memcpy(&len, p, 4); p += 4; n = (unsigned)len; }
    out[i].ptr = p;
    out[i].num = n;
    p += (size_t)n * ArraySchema::elem_size(a.et);
  }
  if ((uint64_t)(p - cell) != cell_size) throw OracleException("cell size mismatch while parsing cell");
  if (cell_size_out) *cell_size_out = cell_size;
}

// ---- the "array": what LoaderArrayWriter puts on disk, in TileDB column-major order -----------------
struct DiskCell { int64_t row, col, END; const uint8_t* raw; };

class VariantArray {
 public:
  ArraySchema schema;
  std::vector<DiskCell> cells;  // (col,row) order, END copies included
  int64_t num_rows = 0;         // #valid rows (max row + 1 from callsets)

  // LoaderArrayWriter::operate / write_top_element_to_disk / finish (load_operators.cc:161-310), with
  // LoaderOperatorBase::handle_intervals_spanning_partition_begin (:33-79)
  void load(const uint8_t* buf, size_t nbytes, int64_t partition_begin = 0, int64_t partition_end = INT64_MAX - 1) {
    struct Wrapper { int64_t row, begin, end; const uint8_t* raw; };
    auto gt = [](const Wrapper& a, const Wrapper& b) { return a.begin > b.begin || (a.begin == b.begin && a.row > b.row); };
    std::priority_queue<Wrapper, std::vector<Wrapper>, decltype(gt)> pq(gt);
    std::vector<int64_t> last_end(std::max<int64_t>(num_rows, 1), -1);
    std::vector<const uint8_t*> spanning_copy(last_end.size(), nullptr);
    bool crossed = false;
    auto write_top = [&]() {
      Wrapper t = pq.top();
      pq.pop();
      cells.push_back({t.row, t.begin, t.end, t.raw});
      if (t.end > t.begin) { std::swap(t.begin, t.end); pq.push(t); }
    };
    std::function<void(const uint8_t*)> operate = [&](const uint8_t* cell) {
      int64_t row, cb, ce;
      memcpy(&row, cell, 8); memcpy(&cb, cell + 8, 8); memcpy(&ce, cell + 24, 8);
      if ((size_t)row >= last_end.size()) { last_end.resize(row + 1, -1); spanning_copy.resize(row + 1, nullptr); }
      if (!crossed) {
        if (cb > partition_begin) {
          crossed = true;
          std::vector<const uint8_t*> copies;
          for (size_t i = 0; i < last_end.size(); ++i) {
            if (last_end[i] >= 0) copies.push_back(spanning_copy[i]);
            spanning_copy[i] = nullptr;
            last_end[i] = -1;
          }
          std::sort(copies.begin(), copies.end(), [](const uint8_t* a, const uint8_t* b) {
            int64_t ra, ca, rb, cb2;
            memcpy(&ra, a, 8); memcpy(&ca, a + 8, 8); memcpy(&rb, b, 8); memcpy(&cb2, b + 8, 8);
            return ca < cb2 || (ca == cb2 && ra < rb);
          });
          for (auto c : copies) operate(c);
        } else {
          if (ce >= partition_begin) { last_end[row] = ce; spanning_copy[row] = cell; }
          else last_end[row] = -1;
          return;
        }
      }
      Wrapper cur{row + 1, cb - 1, -1, nullptr};
      while (!pq.empty() && gt(cur, pq.top())) write_top();
      if (last_end[row] >= cb) {
        std::vector<Wrapper> tmp;
        bool found = false;
        while (!pq.empty() && !found) {
          if (pq.top().row == row) found = true;
          tmp.push_back(pq.top());
          pq.pop();
        }
        ORACLE_VERIFY(found);
        for (size_t i = 0; i + 1 < tmp.size(); ++i) pq.push(tmp[i]);
        Wrapper last = tmp.back();
        if (last.end < last.begin) {  // END copy: truncate to cb-1
          last.begin = cb - 1;
          if (last.begin != last.end) cells.push_back({last.row, last.begin, last.end, last.raw});
        } else {
          throw OracleException("ERROR: two cells in incorrect order found");
        }
      }
      pq.push({row, cb, ce, cell});
      last_end[row] = ce;
    };
    size_t off = 0;
    while (off < nbytes) {
      uint64_t sz;
      memcpy(&sz, buf + off + 16, 8);
      int64_t cb;
      memcpy(&cb, buf + off + 8, 8);
      if (cb <= partition_end) operate(buf + off);
      off += sz;
    }
    if (!crossed) {  // LoaderOperatorBase::finish: force cross (load_operators.cc:108-112)
      crossed = true;
      std::vector<const uint8_t*> copies;
      for (size_t i = 0; i < last_end.size(); ++i) { if (last_end[i] >= 0) copies.push_back(spanning_copy[i]); last_end[i] = -1; }
      std::sort(copies.begin(), copies.end(), [](const uint8_t* a, const uint8_t* b) {
        int64_t ra, ca, rb, cb2;
        memcpy(&ra, a, 8); memcpy(&ca, a + 8, 8); memcpy(&rb, b, 8); memcpy(&cb2, b + 8, 8);
        return ca < cb2 || (ca == cb2 && ra < rb);
      });
      for (auto c : copies) operate(c);
    }
    while (!pq.empty()) write_top();
    // write_cell_sorted contract: the stream must already be in column-major order
    for (size_t i = 1; i < cells.size(); ++i)
      ORACLE_VERIFY(cells[i - 1].col < cells[i].col || (cells[i - 1].col == cells[i].col && cells[i - 1].row <= cells[i].row));
  }
};

// VariantArrayCellIterator over the subarray [row_lo,row_hi] x [col_lo, INT64_MAX]
class CellIterator {
 public:
  CellIterator(const VariantArray* a, int64_t row_lo, int64_t row_hi, int64_t col_lo) : a_(a) { reset_subarray(row_lo, row_hi, col_lo); }
  void reset_subarray(int64_t row_lo, int64_t row_hi, int64_t col_lo) {
    row_lo_ = row_lo; row_hi_ = row_hi;
    pos_ = std::lower_bound(a_->cells.begin(), a_->cells.end(), col_lo,
                            [](const DiskCell& c, int64_t v) { return c.col < v; }) - a_->cells.begin();
    skip();
  }
  bool end() const { return pos_ >= a_->cells.size(); }
  void operator++() { ++pos_; skip(); }
  const ParsedCell& operator*() {
    const DiskCell& d = a_->cells[pos_];
    cur_.row = d.row; cur_.col = d.col; cur_.END = d.END; cur_.raw = d.raw;
    parse_cell_attributes(a_->schema, d.raw, cur_.attr);
    return cur_;
  }
 private:
  void skip() { while (pos_ < a_->cells.size() && (a_->cells[pos_].row < row_lo_ || a_->cells[pos_].row > row_hi_)) ++pos_; }
  const VariantArray* a_;
  int64_t row_lo_ = 0, row_hi_ = 0;
  size_t pos_ = 0;
  ParsedCell cur_;
};

// ---- in-memory Variant / VariantCall / fields (include/genomicsdb/variant.h, variant_field_data.h) ---
enum FieldKind { FK_NONE = 0, FK_INT, FK_FLOAT, FK_STRING, FK_ALT, FK_INT8 };
struct Field {
  bool non_null = false;  // unique_ptr != nullptr
  bool valid = false;
  FieldKind kind = FK_NONE;
  std::vector<int32_t> iv;
  std::vector<float> fv;
  std::string sv;
  std::vector<std::string> alt;
  std::vector<int8_t> bv;
  size_t length() const {
    switch (kind) { case FK_INT: return iv.size(); case FK_FLOAT: return fv.size(); case FK_STRING: return sv.size();
                    case FK_ALT: return alt.size(); case FK_INT8: return bv.size(); default: return 0; }
  }
  void resize(unsigned n) {
    switch (kind) { case FK_INT: iv.resize(n); break; case FK_FLOAT: fv.resize(n); break; case FK_ALT: alt.resize(n); break;
                    case FK_INT8: bv.resize(n); break; default: break; }
  }
};
// copy_field (variant.cc:194-215)
inline void copy_field(Field& dst, const Field& src) {
  if (!dst.non_null && !src.non_null) return;
  if (dst.non_null && !src.non_null) { dst.valid = false; return; }
  dst = src;
}

struct VariantCall {
  bool is_valid = false, is_initialized = false, contains_deletion = false, is_reference_block = false;
  int64_t row_idx = 0, col_begin = 0, col_end = 0;
  std::vector<Field> fields;
  void reset_for_new_interval() { is_initialized = false; is_valid = false; contains_deletion = false; is_reference_block = false; }
};

struct Variant {
  int64_t col_begin = 0, col_end = 0;
  std::vector<VariantCall> calls;
  std::vector<Field> common_fields;  // GA4GHOperator: [0]=REF, [1]=ALT
  void set_column_interval(int64_t b, int64_t e) { col_begin = b; col_end = e; }
  void reset_for_new_interval() { for (auto& c : calls) c.reset_for_new_interval(); }
  void resize_based_on_query(const QueryConfig& qc) {
    uint64_t n = qc.get_num_rows_to_query();
    calls.resize(n);
    for (uint64_t i = 0; i < n; ++i) {
      calls[i].row_idx = qc.get_array_row_idx_for_query_row_idx(i);
      calls[i].fields.resize(qc.num_queried_attributes());
    }
  }
  // deep_copy_simple_members (variant.cc:851-859): flags + intervals, not field contents
  void deep_copy_simple_members(const Variant& o) {
    col_begin = o.col_begin; col_end = o.col_end;
    calls.resize(o.calls.size());
    for (size_t i = 0; i < o.calls.size(); ++i) {
      auto& d = calls[i]; const auto& s = o.calls[i];
      d.is_valid = s.is_valid; d.is_initialized = s.is_initialized; d.contains_deletion = s.contains_deletion;
      d.is_reference_block = s.is_reference_block; d.row_idx = s.row_idx; d.col_begin = s.col_begin; d.col_end = s.col_end;
      if (d.fields.size() < s.fields.size()) d.fields.resize(s.fields.size());
    }
  }
};

// Operator interface (include/query_operations/variant_operations.h:349-388)
class SingleVariantOperatorBase {
 public:
  virtual ~SingleVariantOperatorBase() {}
  virtual void operate(Variant& variant, const QueryConfig& qc) = 0;
  virtual bool overflow() const { return false; }
};

struct EndCmp { bool operator()(const VariantCall* x, const VariantCall* y) const { return x->col_end > y->col_end; } };
typedef std::priority_queue<VariantCall*, std::vector<VariantCall*>, EndCmp> VariantCallEndPQ;

struct ScanState {  // VariantQueryProcessorScanState (query_variants.h:126-191)
  bool done = false;
  CellIterator* iter = nullptr;
  int64_t current_start_position = -1;
  uint64_t num_calls_with_deletions = 0;
  VariantCallEndPQ end_pq;
  Variant variant;
  ~ScanState() { delete iter; }
  bool end() const { return done; }
  void invalidate() { current_start_position = -1; }
  void reset() { invalidate(); done = false; num_calls_with_deletions = 0; while (!end_pq.empty()) end_pq.pop(); }
};

struct ScanStats { uint64_t num_cells = 0, num_cells_in_left_sweep = 0, num_valid_cells = 0, num_operator_invocations = 0; };

class QueryProcessor {
 public:
  const VariantArray* array;
  ScanStats stats;
  explicit QueryProcessor(const VariantArray* a) : array(a) {}

  // fill_field + copy_data_from_tile (query_variants.cc:943-961; variant_field_data.h:99-105, 221-246, 365-384, 511-537)
  void fill_field(Field& f, const ParsedCell& cell, const QueryConfig& qc, unsigned qidx) const {
    const QueryAttr& qa = qc.attrs[qidx];
    const SchemaAttr& sa = array->schema.attrs[qa.schema_idx];
    const CellAttrView& v = cell.attr[qa.schema_idx];
    f.non_null = true;
    f.valid = true;
    unsigned ke = qc.get_known_field_enum_for_query_idx(qidx);
    if (ke == GVCF_ALT_IDX) {
      f.kind = FK_ALT;
      f.alt.clear();
      std::string tmp((const char*)v.ptr, v.num);
      size_t s = 0;
      while (s <= tmp.size()) {  // strtok_r on '|': empty tokens are skipped
        size_t e = tmp.find('|', s);
        if (e == std::string::npos) e = tmp.size();
        if (e > s) f.alt.emplace_back(tmp.substr(s, e - s));
        s = e + 1;
      }
      return;
    }
    ElementType et = sa.et;
    if (qa.info && qa.info->et == ET_FLAG) et = ET_FLAG;
    bool all_missing = true;
    switch (et) {
      case ET_INT:
        f.kind = FK_INT; f.iv.resize(v.num);
        if (v.num) memcpy(f.iv.data(), v.ptr, 4u * v.num);
        for (auto x : f.iv) if (!is_tiledb_missing_value(x)) { all_missing = false; break; }
        if (all_missing) { f.valid = false; f.iv.clear(); }
        break;
      case ET_FLOAT:
        f.kind = FK_FLOAT; f.fv.resize(v.num);
        if (v.num) memcpy(f.fv.data(), v.ptr, 4u * v.num);
        for (auto x : f.fv) if (!is_tiledb_missing_value(x)) { all_missing = false; break; }
        if (all_missing) { f.valid = false; f.fv.clear(); }
        break;
      case ET_FLAG:
        f.kind = FK_INT8; f.bv.resize(v.num);
        if (v.num) memcpy(f.bv.data(), v.ptr, v.num);
        for (auto x : f.bv) if (!is_tiledb_missing_value(x)) { all_missing = false; break; }
        if (all_missing) { f.valid = false; f.bv.clear(); }
        break;
      case ET_CHAR:
        f.kind = FK_STRING; f.sv.assign((const char*)v.ptr, v.num);
        for (auto x : f.sv) if (!is_tiledb_missing_value(x)) { all_missing = false; break; }
        if (all_missing) { f.valid = false; f.sv.clear(); }
        break;
      default:
        throw OracleException("unsupported attribute type in fill_field");
    }
  }

  // gt_fill_row (query_variants.cc:1014-1117), DUPLICATE_CELL_AT_END build
  void gt_fill_row(Variant& variant, int64_t row, int64_t column, const QueryConfig& qc, const ParsedCell& cell, bool traverse_end_copies = false) {
    VariantCall& call = variant.calls[qc.get_query_row_idx_for_array_row_idx(row)];
    call.is_initialized = true;
    call.contains_deletion = false;
    call.is_reference_block = false;
    int64_t query_column_value = variant.col_begin;
    int64_t cell_begin_value = column;
    int64_t END_v = cell.END;
    if ((traverse_end_copies && cell_begin_value <= END_v && cell_begin_value > query_column_value) ||
        (!traverse_end_copies && cell_begin_value > END_v)) {
      call.is_valid = false;
      return;
    }
    call.is_valid = true;
    ++stats.num_valid_cells;
    if (column > END_v) std::swap(column, END_v);
    call.col_begin = column;
    call.col_end = END_v;
    for (unsigned i = 1; i < qc.num_queried_attributes(); ++i) fill_field(call.fields[i], cell, qc, i);
    const Field* REF = qc.is_defined_query_idx_for_known_field_enum(GVCF_REF_IDX) ? &call.fields[qc.get_query_idx_for_known_field_enum(GVCF_REF_IDX)] : nullptr;
    const Field* ALT = qc.is_defined_query_idx_for_known_field_enum(GVCF_ALT_IDX) ? &call.fields[qc.get_query_idx_for_known_field_enum(GVCF_ALT_IDX)] : nullptr;
    if (REF && REF->valid && ALT && ALT->valid) {
      call.contains_deletion = VariantUtils::contains_deletion(REF->sv, ALT->alt);
      call.is_reference_block = VariantUtils::is_reference_block(REF->sv, ALT->alt);
    }
  }

  // gt_get_column (query_variants.cc:845-941), DUPLICATE_CELL_AT_END build
  void gt_get_column(const QueryConfig& qc, unsigned interval_idx, Variant& variant, CellIterator*& iter) {
    variant.reset_for_new_interval();
    variant.set_column_interval(qc.column_intervals[interval_idx].first, qc.column_intervals[interval_idx].second);
    int64_t col = qc.column_intervals[interval_idx].first;
    gt_initialize_forward_iter(qc, col, iter);
    uint64_t filled_rows = 0;
    while (!iter->end() && filled_rows < qc.get_num_rows_to_query()) {
      ++stats.num_cells; ++stats.num_cells_in_left_sweep;
      const ParsedCell& cell = **iter;
      if (cell.col >= col && qc.is_queried_array_row_idx(cell.row)) {
        auto& call = variant.calls[qc.get_query_row_idx_for_array_row_idx(cell.row)];
        if (!call.is_initialized) {
          gt_fill_row(variant, cell.row, cell.col, qc, cell, true);
          ++filled_rows;
        }
      }
      ++(*iter);
    }
  }

  void gt_initialize_forward_iter(const QueryConfig& qc, int64_t column, CellIterator*& iter) {  // :1119-1132
    int64_t lo = qc.smallest_row_idx, hi = qc.num_rows_in_array + qc.smallest_row_idx - 1;
    if (iter) iter->reset_subarray(lo, hi, column);
    else iter = new CellIterator(array, lo, hi, column);
  }

  // handle_gvcf_ranges (query_variants.cc:296-332)
  void handle_gvcf_ranges(VariantCallEndPQ& end_pq, const QueryConfig& qc, Variant& variant, SingleVariantOperatorBase& op,
                          int64_t& current_start_position, int64_t next_start_position, bool is_last_call, uint64_t& num_calls_with_deletions) {
    while (!end_pq.empty() && (current_start_position < next_start_position || is_last_call) && !op.overflow()) {
      int64_t top_end_pq = end_pq.top()->col_end;
      int64_t min_end_point = (is_last_call || (top_end_pq < (next_start_position - 1))) ? top_end_pq : (next_start_position - 1);
      min_end_point = num_calls_with_deletions ? current_start_position : min_end_point;
      variant.set_column_interval(current_start_position, min_end_point);
      ++stats.num_operator_invocations;
      op.operate(variant, qc);
      while (!end_pq.empty() && end_pq.top()->col_end == min_end_point) {
        auto top = end_pq.top();
        if (top->contains_deletion) --num_calls_with_deletions;
        top->is_valid = false;
        end_pq.pop();
      }
      current_start_position = min_end_point + 1;
    }
  }

  // scan_handle_cell (query_variants.cc:478-555)
  bool scan_handle_cell(const QueryConfig& qc, unsigned interval_idx, Variant& variant, SingleVariantOperatorBase& op, const ParsedCell& cell,
                        VariantCallEndPQ& end_pq, std::vector<VariantCall*>& tmp_pq_buffer, int64_t& current_start_position,
                        int64_t& next_start_position, uint64_t& num_calls_with_deletions, bool handle_spanning_deletions) {
    if (!qc.column_intervals.empty() && cell.col > qc.column_intervals[interval_idx].second) return true;
    if (cell.col != current_start_position) {
      next_start_position = cell.col;
      handle_gvcf_ranges(end_pq, qc, variant, op, current_start_position, next_start_position, false, num_calls_with_deletions);
      if (op.overflow()) return false;
      current_start_position = next_start_position;
      variant.set_column_interval(current_start_position, current_start_position);
    }
    if (qc.is_queried_array_row_idx(cell.row)) {
      auto& call = variant.calls[qc.get_query_row_idx_for_array_row_idx(cell.row)];
      if (call.is_valid && call.col_end >= cell.col) {
        bool found = false;
        size_t n_tmp = 0;
        while (!end_pq.empty() && !found) {
          auto top = end_pq.top();
          if (top == &call) found = true; else tmp_pq_buffer[n_tmp++] = top;
          end_pq.pop();
        }
        ORACLE_VERIFY(found);
        for (size_t i = 0; i < n_tmp; ++i) end_pq.push(tmp_pq_buffer[i]);
        if (!call.contains_deletion && !call.is_reference_block)
          throw OracleException("Unhandled overlapping variants at columns " + std::to_string(call.col_begin) + " and " +
                                std::to_string(cell.col) + " for row " + std::to_string(cell.row));
        if (call.contains_deletion) { ORACLE_VERIFY(num_calls_with_deletions > 0u); --num_calls_with_deletions; }
      }
      call.reset_for_new_interval();
      gt_fill_row(variant, cell.row, cell.col, qc, cell, false);
      if (call.is_valid) {
        end_pq.push(&call);
        if (handle_spanning_deletions && call.contains_deletion) ++num_calls_with_deletions;
      }
    }
    return false;
  }

  // scan_and_operate (query_variants.cc:334-476)
  void scan_and_operate(const QueryConfig& qc, SingleVariantOperatorBase& op, unsigned interval_idx, bool handle_spanning_deletions, ScanState* ss) {
    ORACLE_VERIFY(ss != nullptr);
    VariantCallEndPQ& end_pq = ss->end_pq;
    int64_t start_column = 0;
    int64_t current_start_position = ss->current_start_position;
    Variant& variant = ss->variant;
    variant.resize_based_on_query(qc);
    uint64_t num_calls_with_deletions = ss->num_calls_with_deletions;
    std::vector<VariantCall*> tmp_pq_buffer(qc.get_num_rows_to_query());
    CellIterator* forward_iter = ss->iter;
    if (!(ss->iter && ss->current_start_position >= 0)) {
      if (!qc.column_intervals.empty()) {
        gt_get_column(qc, interval_idx, variant, forward_iter);
        for (auto& c : variant.calls)
          if (c.is_valid) {
            end_pq.push(&c);
            if (handle_spanning_deletions && c.contains_deletion) ++num_calls_with_deletions;
          }
        if (end_pq.size() > 0) current_start_position = qc.column_intervals[interval_idx].first;
        start_column = qc.column_intervals[interval_idx].first + 1;
      }
      gt_initialize_forward_iter(qc, start_column, forward_iter);
    }
    if (current_start_position < 0 && !forward_iter->end()) current_start_position = (**forward_iter).col;
    variant.set_column_interval(current_start_position, current_start_position);
    int64_t next_start_position = -1;
    bool end_loop = false;
    for (; !forward_iter->end() && !end_loop && !op.overflow(); ++(*forward_iter)) {
      const ParsedCell& cell = **forward_iter;
      ++stats.num_cells;
      if (cell.col > cell.END) continue;  // END copy
      end_loop = scan_handle_cell(qc, interval_idx, variant, op, cell, end_pq, tmp_pq_buffer, current_start_position, next_start_position,
                                  num_calls_with_deletions, handle_spanning_deletions);
      if (op.overflow()) break;
    }
    if (end_loop || forward_iter->end()) {
      bool is_last_call = false;
      if (!qc.column_intervals.empty()) {
        next_start_position = qc.column_intervals[interval_idx].second;
        if (next_start_position != INT64_MAX) ++next_start_position;
      } else {
        next_start_position = 0;
        is_last_call = true;
      }
      handle_gvcf_ranges(end_pq, qc, variant, op, current_start_position, next_start_position, is_last_call, num_calls_with_deletions);
      ss->iter = forward_iter; ss->current_start_position = current_start_position; ss->num_calls_with_deletions = num_calls_with_deletions;
      if (!op.overflow()) { ss->invalidate(); ss->done = true; }
    } else {
      ss->iter = forward_iter; ss->current_start_position = current_start_position; ss->num_calls_with_deletions = num_calls_with_deletions;
    }
  }
};

}  // namespace gdb_oracle
