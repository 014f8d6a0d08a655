// gdb_oracle_combine.hpp - TEST ORACLE (combine operator + VCF text).  NOT PRODUCT CODE.
//
// CPU restatement of the reference's per-interval combine:
//   allele merge + LUT        src/query_operations/variant_operations.cc:73-263, :362-378; include/utils/lut.h
//   remap kernels             src/genomicsdb/variant_field_handler.cc:41-398 (alleles / genotype haploid,
//                             diploid, general ploidy), :402-494 (min-PL genotype), :529-871 (INFO reducers,
//                             FORMAT collect_and_extend_fields)
//   GA4GHOperator::operate    src/query_operations/variant_operations.cc:572-695
//   BroadCombinedGVCFOperator src/query_operations/broad_combined_gvcf.cc:53-138 (GT encode), :140-356 (ctor,
//                             header), :374-429, :523-727 (INFO/FORMAT), :730-763 (ID), :765-901 (operate),
//                             :903-909 (contig switch), :912-1118 (spanning deletions)
//   header assembly           src/vcf/vcf_adapter.cc:59-199
// The VCF *text* serialisation itself lives in the Intel htslib fork, which is not in the reference tree
// (dependencies/htslib is an empty submodule).  vcf_format()/format_float() below restate the published
// htslib vcf_format()/bcf_fmt_array()/bcf_format_gt() behaviour and are pinned by the reference's
// tests/golden_outputs (see tests/test_oracle_golden.py); float formatting outside the value classes that
// appear in those goldens is PARITY-UNPINNED (documented in DESIGN.md).
#pragma once
#include <cmath>
#include <functional>
#include <iomanip>
#include <map>
#include <set>
#include <sstream>
#include <unordered_map>
#include <unordered_set>

#include "gdb_oracle_scan.hpp"
#include "oracle_json.hpp"

namespace gdb_oracle {

static const int64_t lut_missing_value = -1;

class CombineAllelesLUT {  // include/utils/lut.h:65-343 (input-ordered layout)
 public:
  void resize_luts_if_needed(size_t ncalls, size_t nalleles) {
    const bool more_rows = i2m_.size() < ncalls, more_cols = nalleles > ncols_;
    if (!more_rows && !more_cols && sized_) return;      // (the rows are only walked when something grows: the call is made per allele pair)
    if (more_rows) { i2m_.resize(ncalls); m2i_.resize(ncalls); }
    ncols_ = std::max(ncols_, nalleles);
    for (auto& v : i2m_) if (v.size() < ncols_) v.resize(ncols_, lut_missing_value);
    for (auto& v : m2i_) if (v.size() < ncols_) v.resize(ncols_, lut_missing_value);
    sized_ = true;
  }
  void resize_luts_if_needed(size_t nalleles) { resize_luts_if_needed(i2m_.size(), nalleles); }
  void reset_luts() {
    for (auto& v : i2m_) std::fill(v.begin(), v.end(), lut_missing_value);
    for (auto& v : m2i_) std::fill(v.begin(), v.end(), lut_missing_value);
  }
  void add_input_merged_idx_pair(size_t call, int64_t in, int64_t merged) {
    resize_luts_if_needed(std::max(i2m_.size(), call + 1), (size_t)std::max(in, merged) + 1);
    i2m_[call][in] = merged;
    m2i_[call][merged] = in;
  }
  int64_t get_input_idx_for_merged(size_t call, int64_t merged) const {
    if (call >= m2i_.size() || merged < 0 || (size_t)merged >= m2i_[call].size()) return lut_missing_value;
    return m2i_[call][merged];
  }
  int64_t get_merged_idx_for_input(size_t call, int64_t in) const {
    if (call >= i2m_.size() || in < 0 || (size_t)in >= i2m_[call].size()) return lut_missing_value;
    return i2m_[call][in];
  }
  static bool is_missing_value(int64_t v) { return v == lut_missing_value; }
 private:
  std::vector<std::vector<int64_t>> i2m_, m2i_;
  size_t ncols_ = 10;
  bool sized_ = false;
};

inline int bcf_alleles2gt(int a, int b) { return a > b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; }

// VariantOperations::get_genotype_index (variant_field_handler.cc:299-321)
inline uint64_t get_genotype_index(std::vector<int>& v, bool is_sorted) {
  switch (v.size()) {
    case 0u: return 0u;
    case 1u: return (uint64_t)v[0];
    case 2u: return (uint64_t)bcf_alleles2gt(v[0], v[1]);
    default: {
      if (!is_sorted) std::sort(v.begin(), v.end());
      uint64_t gt = 0;
      for (uint64_t i = 0; i < v.size(); ++i) gt += nCr(i + v[i], v[i] - 1);
      return gt;
    }
  }
}

// ---- 2-D fields: <u64 size of data><inner vectors back to back><u64 #entries><u64 offsets x (#entries + 1)> -----------------------
// (genomicsdb_multid_vector_field.h:69-86; the walk of GenomicsDBMultiDVectorIdx over dimension 0, multid_vector_field.cc:118-205)
struct MultiD2View {
  const uint8_t* p;
  uint64_t data_size = 0, n = 0;
  explicit MultiD2View(const std::string& blob) : p((const uint8_t*)blob.data()) {
    ORACLE_VERIFY(blob.size() >= 24u);
    memcpy(&data_size, p, 8);
    memcpy(&n, p + 8 + data_size, 8);
  }
  uint64_t off(uint64_t i) const { uint64_t v; memcpy(&v, p + 8 + data_size + 8 + 8 * i, 8); return v; }
  uint64_t num_entries() const { return n; }
  const uint8_t* ptr(uint64_t i) const { return p + 8 + off(i); }
  uint64_t bytes(uint64_t i) const { return off(i + 1) - off(i); }
};
// remap_allele_specific_annotations (variant_operations.cc:482-549): dimension 0 of the input re-indexed to the merged alleles,
// an allele the input does not have takes the input's <NON_REF> entry (none: 0 bytes)
inline std::string remap_allele_specific_annotations(const std::string& orig, uint64_t call, const class CombineAllelesLUT& lut, unsigned num_merged,
                                                     bool NON_REF_exists, bool alt_only);

// ---- typed access helpers over Field -----------------------------------------------------------
template <class T> struct FieldVec;
template <> struct FieldVec<int32_t> {
  static std::vector<int32_t>& get(Field& f) { return f.iv; }
  static const std::vector<int32_t>& get(const Field& f) { return f.iv; }
  static int32_t missing() { return bcf_int32_missing; }
  static int32_t vector_end() { return bcf_int32_vector_end; }
};
template <> struct FieldVec<float> {
  static std::vector<float>& get(Field& f) { return f.fv; }
  static const std::vector<float>& get(const Field& f) { return f.fv; }
  static float missing() { return u2f(bcf_float_missing_bits); }
  static float vector_end() { return u2f(bcf_float_vector_end_bits); }
};

inline std::string remap_allele_specific_annotations(const std::string& orig, uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged,
                                                     bool NON_REF_exists, bool alt_only) {
  MultiD2View in(orig);
  const int64_t merged_nr = NON_REF_exists ? (int64_t)(int)(num_merged - 1) : lut_missing_value;
  const int64_t input_nr = NON_REF_exists ? lut.get_input_idx_for_merged(call, merged_nr) : lut_missing_value;
  const unsigned length = alt_only ? num_merged - 1u : num_merged;
  std::vector<uint64_t> offsets(length + 1u, 0u);
  std::string data;
  for (unsigned j = 0; j < length; ++j) {
    const unsigned allele_j = alt_only ? j + 1u : j;
    int64_t in_j = lut.get_input_idx_for_merged(call, allele_j);
    if (CombineAllelesLUT::is_missing_value(in_j)) {
      if (CombineAllelesLUT::is_missing_value(input_nr)) { offsets[j + 1u] = offsets[j]; continue; }
      in_j = input_nr;
    }
    const int64_t input_j = alt_only ? in_j - 1 : in_j;
    if (input_j >= 0 && (uint64_t)input_j < in.num_entries()) data.append((const char*)in.ptr((uint64_t)input_j), in.bytes((uint64_t)input_j));
    offsets[j + 1u] = data.size();
  }
  std::string out;
  const uint64_t size = data.size(), n = length;
  out.append((const char*)&size, 8);
  out += data;
  out.append((const char*)&n, 8);
  out.append((const char*)offsets.data(), offsets.size() * 8u);
  return out;
}

typedef std::function<void(uint64_t out_idx, bool has_value, uint64_t in_idx)> RemapSink;

// remap_data_based_on_alleles (variant_field_handler.cc:41-81): sink(out, has, in) with has=false -> missing
inline void remap_based_on_alleles(size_t input_size, uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged, bool NON_REF_exists,
                                   bool alt_only, const RemapSink& sink) {
  const int64_t merged_nr = NON_REF_exists ? (int64_t)(int)(num_merged - 1) : lut_missing_value;
  const int64_t input_nr = NON_REF_exists ? lut.get_input_idx_for_merged(call, merged_nr) : lut_missing_value;
  unsigned length = alt_only ? num_merged - 1u : num_merged;
  for (unsigned j = 0; j < length; ++j) {
    unsigned allele_j = alt_only ? j + 1u : j;
    int64_t in_j = lut.get_input_idx_for_merged(call, allele_j);
    if (CombineAllelesLUT::is_missing_value(in_j)) {
      if (CombineAllelesLUT::is_missing_value(input_nr)) { sink(j, false, 0); continue; }
      in_j = input_nr;
    }
    int64_t idx = alt_only ? in_j - 1 : in_j;
    if ((size_t)idx >= input_size) sink(j, false, 0); else sink(j, true, (uint64_t)idx);
  }
}
// remap_data_based_on_genotype_haploid (:83-122)
inline void remap_genotype_haploid(size_t input_size, uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged, bool NON_REF_exists, const RemapSink& sink) {
  const int64_t merged_nr = NON_REF_exists ? (int64_t)(int)(num_merged - 1) : lut_missing_value;
  const int64_t input_nr = NON_REF_exists ? lut.get_input_idx_for_merged(call, merged_nr) : lut_missing_value;
  for (unsigned j = 0; j < num_merged; ++j) {
    int64_t in_j = lut.get_input_idx_for_merged(call, j);
    if (CombineAllelesLUT::is_missing_value(in_j)) {
      if (CombineAllelesLUT::is_missing_value(input_nr)) { sink(j, false, 0); continue; }
      in_j = input_nr;
    }
    if ((size_t)in_j >= input_size) sink(j, false, 0); else sink(j, true, (uint64_t)in_j);
  }
}
// remap_data_based_on_genotype_diploid (:134-191)
inline void remap_genotype_diploid(size_t input_size, uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged, bool NON_REF_exists, const RemapSink& sink) {
  const int64_t merged_nr = NON_REF_exists ? (int64_t)(int)(num_merged - 1) : lut_missing_value;
  const int64_t input_nr = NON_REF_exists ? lut.get_input_idx_for_merged(call, merged_nr) : lut_missing_value;
  for (unsigned j = 0; j < num_merged; ++j) {
    int64_t in_j = lut.get_input_idx_for_merged(call, j);
    if (CombineAllelesLUT::is_missing_value(in_j)) {
      if (CombineAllelesLUT::is_missing_value(input_nr)) {
        for (unsigned k = j; k < num_merged; ++k) sink((uint64_t)bcf_alleles2gt(j, k), false, 0);
        continue;
      }
      in_j = input_nr;
    }
    for (unsigned k = j; k < num_merged; ++k) {
      uint64_t gt_idx = (uint64_t)bcf_alleles2gt(j, k);
      int64_t in_k = lut.get_input_idx_for_merged(call, k);
      if (CombineAllelesLUT::is_missing_value(in_k)) {
        if (CombineAllelesLUT::is_missing_value(input_nr)) { sink(gt_idx, false, 0); continue; }
        in_k = input_nr;
      }
      uint64_t in_gt = (uint64_t)bcf_alleles2gt((int)in_j, (int)in_k);
      if (in_gt >= input_size) sink(gt_idx, false, 0); else sink(gt_idx, true, in_gt);
    }
  }
}
// remap_data_based_on_genotype_general (:198-297): enumerates merged genotypes in VCF order; op gets
// (remapped_gt_idx, contains_missing_allele, input allele idx vector)
typedef std::function<void(uint64_t remapped_gt_idx, bool contains_missing, std::vector<int>& input_alleles)> GeneralOp;
inline void remap_genotype_general(uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged, bool NON_REF_exists, unsigned ploidy, const GeneralOp& op) {
  if (ploidy == 0u) return;
  const int64_t merged_nr = NON_REF_exists ? (int64_t)(int)(num_merged - 1) : lut_missing_value;
  const int64_t input_nr = NON_REF_exists ? lut.get_input_idx_for_merged(call, merged_nr) : lut_missing_value;
  std::vector<int> remapped(ploidy + 1u), input(ploidy);
  std::vector<std::pair<int, int>> stack(get_number_of_genotypes(num_merged - 1u, ploidy) + ploidy + 2u);
  stack[0] = std::make_pair((int)ploidy, (int)num_merged - 1);
  unsigned n = 1u;
  uint64_t remapped_gt_idx = 0;
  while (n > 0u) {
    int allele_idx = stack[n - 1u].second, ploidy_idx = stack[n - 1u].first;
    --n;
    remapped[ploidy_idx] = allele_idx;
    if (ploidy_idx == 0) {
      bool missing = false;
      for (unsigned i = 0; i < ploidy; ++i) {
        int64_t in = lut.get_input_idx_for_merged(call, remapped[i]);
        if (CombineAllelesLUT::is_missing_value(in)) {
          input[i] = (int)input_nr;
          missing = missing || CombineAllelesLUT::is_missing_value(input_nr);
        } else input[i] = (int)in;
      }
      op(remapped_gt_idx, missing, input);
      ++remapped_gt_idx;
    } else {
      --ploidy_idx;
      for (int i = allele_idx; i >= 0; --i) {
        if (n >= stack.size()) stack.resize(2 * stack.size());
        stack[n++] = std::make_pair(ploidy_idx, i);
      }
    }
  }
}
// remap_data_based_on_genotype dispatch (:358-398) + reorder_field_based_on_genotype_index (:323-355)
inline void remap_based_on_genotype(size_t input_size, uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged, bool NON_REF_exists,
                                    unsigned ploidy, const RemapSink& sink) {
  switch (ploidy) {
    case 1u: remap_genotype_haploid(input_size, call, lut, num_merged, NON_REF_exists, sink); break;
    case 2u: remap_genotype_diploid(input_size, call, lut, num_merged, NON_REF_exists, sink); break;
    default:
      remap_genotype_general(call, lut, num_merged, NON_REF_exists, ploidy, [&](uint64_t gt, bool missing, std::vector<int>& in) {
        if (missing) { sink(gt, false, 0); return; }
        uint64_t in_gt = get_genotype_index(in, false);
        if (in_gt >= input_size) sink(gt, false, 0); else sink(gt, true, in_gt);
      });
  }
}

// ---- float text (htslib-fork rule pinned by goldens; see file header) ---------------------------------
inline void format_float(std::string& s, float f) {
  double d = f;
  if (d == 0) { s += std::signbit(d) ? "-0" : "0"; return; }
  if (d < 0) { s += '-'; d = -d; }
  if (!(d >= 0.0001 && d <= 999999)) { char b[64]; snprintf(b, sizeof(b), "%g", d); s += b; return; }
  uint64_t i = (uint64_t)(d * 10000000000LL);
  if (d < .0001) i += 0; else if (d < 0.001) i += 5; else if (d < 0.01) i += 50; else if (d < 0.1) i += 500;
  else if (d < 1) i += 5000; else if (d < 10) i += 50000; else if (d < 100) i += 500000; else if (d < 1000) i += 5000000;
  else if (d < 10000) i += 50000000; else if (d < 100000) i += 500000000; else i += 5000000000LL;
  char digits[32];
  int p = 0;
  { char tmp[32]; int n = 0; do { tmp[n++] = (char)('0' + i % 10); i /= 10; } while (i >= 1); while (n) digits[p++] = tmp[--n]; }
  std::string out;
  if (p <= 10) {                       // d < 1: "0." + leading zeros + first 6 digits of i
    out = "0.";
    out.append((size_t)(10 - p), '0');
    out.append(digits, digits + std::min(p, 6));
  } else {                             // integer part = first p-10 digits, 6 significant digits in total
    int ip = p - 10;
    out.assign(digits, digits + ip);
    if (ip < 6) { out += '.'; out.append(digits + ip, digits + 6); }
  }
  size_t dot = out.find('.');
  if (dot != std::string::npos) {      // cull trailing zeros but keep one digit after the point ("8.0")
    size_t e = out.size();
    while (e > dot + 2 && out[e - 1] == '0') --e;
    out.resize(e);
  }
  s += out;
}

// ---- an in-memory "bcf1_t" and its text form (htslib vcf_format / bcf_fmt_array / bcf_format_gt) ----
struct RecInfo { std::string key; int type; std::vector<int32_t> iv; std::vector<float> fv; std::string sv; };  // type 0 int,1 float,2 str
struct RecFmt { std::string key; int type; unsigned n; std::vector<int32_t> iv; std::vector<float> fv; std::string cv; bool is_gt = false; };
struct BcfRecord {
  std::string chrom, id;
  int64_t pos = 0;
  std::vector<std::string> alleles;
  bool qual_missing = true; float qual = 0;
  std::vector<std::string> filters;
  std::vector<RecInfo> info;
  std::vector<RecFmt> fmt;
  unsigned n_sample = 0;
  void clear() { id.clear(); alleles.clear(); qual_missing = true; filters.clear(); info.clear(); fmt.clear(); }
  RecInfo& info_slot(const std::string& k) {  // bcf_update_info: replace in place, else append
    for (auto& x : info) if (x.key == k) return x;
    info.emplace_back(); info.back().key = k; return info.back();
  }
  RecFmt& fmt_slot(const std::string& k) {    // bcf_update_format: replace in place, else append; GT goes first
    for (auto& x : fmt) if (x.key == k) return x;
    if (k == "GT") { fmt.emplace(fmt.begin()); fmt.front().key = k; fmt.front().is_gt = true; return fmt.front(); }
    fmt.emplace_back(); fmt.back().key = k; return fmt.back();
  }
};
inline void fmt_int_array(std::string& s, const int32_t* p, unsigned n) {
  if (n == 0) { s += '.'; return; }
  for (unsigned j = 0; j < n; ++j) {
    if (p[j] == bcf_int32_vector_end) break;
    if (j) s += ',';
    if (p[j] == bcf_int32_missing) s += '.'; else s += std::to_string(p[j]);
  }
}
inline void fmt_float_array(std::string& s, const float* p, unsigned n) {
  if (n == 0) { s += '.'; return; }
  for (unsigned j = 0; j < n; ++j) {
    if (bcf_float_is_vector_end(p[j])) break;
    if (j) s += ',';
    if (bcf_float_is_missing(p[j])) s += '.'; else format_float(s, p[j]);
  }
}
inline void fmt_char_array(std::string& s, const char* p, unsigned n) {
  if (n == 0) { s += '.'; return; }
  for (unsigned j = 0; j < n && p[j]; ++j) s += (p[j] == bcf_str_missing) ? '.' : p[j];
}
inline void vcf_format(const BcfRecord& r, std::string& s) {
  s += r.chrom; s += '\t'; s += std::to_string(r.pos + 1); s += '\t';
  s += r.id.empty() ? "." : r.id; s += '\t';
  s += r.alleles.empty() ? "." : r.alleles[0]; s += '\t';
  if (r.alleles.size() > 1) { for (size_t i = 1; i < r.alleles.size(); ++i) { if (i > 1) s += ','; s += r.alleles[i]; } } else s += '.';
  s += '\t';
  if (r.qual_missing) s += '.'; else format_float(s, r.qual);
  s += '\t';
  if (r.filters.empty()) s += '.'; else for (size_t i = 0; i < r.filters.size(); ++i) { if (i) s += ';'; s += r.filters[i]; }
  s += '\t';
  if (r.info.empty()) s += '.';
  else {
    bool first = true;
    for (auto& x : r.info) {
      if (!first) s += ';';
      first = false;
      s += x.key; s += '=';
      if (x.type == 0) fmt_int_array(s, x.iv.data(), (unsigned)x.iv.size());
      else if (x.type == 1) fmt_float_array(s, x.fv.data(), (unsigned)x.fv.size());
      else s += x.sv;
    }
  }
  if (r.n_sample && !r.fmt.empty()) {
    s += '\t';
    for (size_t i = 0; i < r.fmt.size(); ++i) { if (i) s += ':'; s += r.fmt[i].key; }
    for (unsigned smp = 0; smp < r.n_sample; ++smp) {
      s += '\t';
      for (size_t i = 0; i < r.fmt.size(); ++i) {
        if (i) s += ':';
        const RecFmt& f = r.fmt[i];
        if (f.is_gt) {  // bcf_format_gt
          const int32_t* p = f.iv.data() + (size_t)smp * f.n;
          unsigned k;
          for (k = 0; k < f.n && p[k] != bcf_int32_vector_end; ++k) {
            if (k) s += "/|"[p[k] & 1];
            if (!(p[k] >> 1)) s += '.'; else s += std::to_string((p[k] >> 1) - 1);
          }
          if (k == 0) s += '.';
        } else if (f.type == 0) fmt_int_array(s, f.iv.data() + (size_t)smp * f.n, f.n);
        else if (f.type == 1) fmt_float_array(s, f.fv.data() + (size_t)smp * f.n, f.n);
        else fmt_char_array(s, f.cv.data() + (size_t)smp * f.n, f.n);
      }
    }
  }
  s += '\n';
}

// ---- reference genome (vcf_adapter.cc:30-56) ---------------------------------------------------------
class ReferenceGenome {
 public:
  void load_fasta(const std::string& path) {
    std::string txt = oracle_json::gz_read_all(path);
    size_t p = 0;
    std::string* cur = nullptr;
    while (p < txt.size()) {
      size_t e = txt.find('\n', p);
      if (e == std::string::npos) e = txt.size();
      if (txt[p] == '>') {
        size_t ne = p + 1;
        while (ne < e && txt[ne] != ' ' && txt[ne] != '\t') ++ne;
        cur = &seqs_[txt.substr(p + 1, ne - p - 1)];
      } else if (cur) cur->append(txt, p, e - p);
      p = e + 1;
    }
  }
  std::function<char(const std::string&, int64_t)> custom;  // synthetic reference for the bench generator
  char base_at(const std::string& contig, int64_t pos) const {
    if (custom) return custom(contig, pos);
    auto it = seqs_.find(contig);
    if (it == seqs_.end() || pos < 0 || (size_t)pos >= it->second.size()) return 'N';
    return it->second[(size_t)pos];
  }
 private:
  std::map<std::string, std::string> seqs_;
};

// ---- the operator -----------------------------------------------------------------------------------
struct FieldTuple { unsigned known_enum; unsigned query_idx; const FieldInfo* info; };

class BroadCombinedGVCFOperator : public SingleVariantOperatorBase {
 public:
  std::string header_text;       // "##..." lines + #CHROM line (print_header)
  std::string out;               // body produced so far (drained by the caller)
  size_t buffer_limit = 0;       // 0 = never overflow (plain VCFAdapter); else VCFSerializedBufferAdapter rule
  size_t bytes_in_buffer = 0;
  uint64_t num_records = 0;

  BroadCombinedGVCFOperator(const VidMapper& vid, const QueryConfig& qc, const std::string& template_header_text, const ReferenceGenome* ref,
                            unsigned max_alt = 50)
      : vid_(&vid), qc_(&qc), ref_(ref), max_alt_(max_alt) {
    GT_query_idx_ = UNDEFINED_IDX;
    for (unsigned q = 0; q < qc.num_queried_attributes(); ++q) {
      if (qc.attrs[q].info->is_allele_dependent()) remapped_fields_query_idxs_.push_back(q);
      if (qc.get_known_field_enum_for_query_idx(q) == GVCF_GT_IDX) GT_query_idx_ = q;
    }
    ploidy_.resize((size_t)std::max<int64_t>(qc.num_rows_in_array, 1));
    remapped_variant_.common_fields.resize(2);
    build_field_lists_and_header(template_header_text);
    std::string nm; int64_t off;
    vid.get_next_contig_location(-1, nm, off);
    next_contig_name_ = nm; next_contig_begin_ = off;
    switch_contig();
    spanning_deletions_remapped_fields_.resize(remapped_fields_query_idxs_.size());
  }
  bool overflow() const override { return buffer_limit && bytes_in_buffer >= buffer_limit; }

  // BroadCombinedGVCFOperator::operate (broad_combined_gvcf.cc:765-901)
  void operate(Variant& variant, const QueryConfig& qc) override {
    handle_deletions(variant, qc);
    ga4gh_operate(variant, qc);
    if (remapped_variant_.col_begin >= next_contig_begin_) {
      std::string cn; int64_t cp;
      if (!vid_->get_contig_location(remapped_variant_.col_begin, cn, cp))
        throw OracleException("Unknown contig for position " + std::to_string(remapped_variant_.col_begin));
      int64_t cb = remapped_variant_.col_begin - cp;
      if (cb != next_contig_begin_) { next_contig_name_ = cn; next_contig_begin_ = cb; }
      switch_contig();
    }
    rec_.clear();
    rec_.n_sample = qc.sites_only_query ? 0u : (unsigned)qc.get_num_rows_to_query();
    rec_.chrom = curr_contig_name_;
    rec_.pos = remapped_variant_.col_begin - curr_contig_begin_;
    if (qc.is_defined_query_idx_for_known_field_enum(GVCF_ID_IDX)) {
      merge_ID_field(variant, qc.get_query_idx_for_known_field_enum(GVCF_ID_IDX));
      rec_.id = ID_value_;
    }
    rec_.qual_missing = true;
    if (qual_tuple_.info && qual_tuple_.info->combine_op != OP_UNKNOWN && qual_tuple_.query_idx != UNDEFINED_IDX) {
      CombineResult res;
      if (handle_VCF_field_combine_operation(variant, qual_tuple_, res) && !res.fv.empty()) { rec_.qual_missing = false; rec_.qual = res.fv[0]; }
    }
    std::string& ref_allele = remapped_variant_.common_fields[0].sv;
    if (ref_allele.length() == 1u && ref_allele[0] == 'N') {
      char b = ref_ ? ref_->base_at(curr_contig_name_, rec_.pos) : 'N';
      ref_allele[0] = (b == 'A' || b == 'T' || b == 'G' || b == 'C') ? b : 'N';
    }
    rec_.alleles.push_back(ref_allele);
    for (auto& a : remapped_variant_.common_fields[1].alt) rec_.alleles.push_back(IS_NON_REF_ALLELE(a) ? g_vcf_NON_REF : a);
    if (qc.produce_FILTER_field && qc.is_defined_query_idx_for_known_field_enum(GVCF_FILTER_IDX)) {
      unsigned fq = qc.get_query_idx_for_known_field_enum(GVCF_FILTER_IDX);
      std::unordered_set<int> filter_idx_set;  // iteration order = libstdc++'s (the product restates it: gdb_core.hpp gdb_uset_insert_range)
      for (auto& call : variant.calls) {
        if (!call.is_valid) continue;
        const Field& f = call.fields[fq];
        if (f.non_null && f.valid) filter_idx_set.insert(f.iv.begin(), f.iv.end());
      }
      for (int g : filter_idx_set)
        if (g >= 0 && (size_t)g < vid_->fields.size() && hdr_ids_.count(vid_->fields[g].vcf_name)) rec_.filters.push_back(vid_->fields[g].vcf_name);
    }
    handle_INFO_fields(variant);
    handle_FORMAT_fields(variant);
    size_t before = out.size();
    vcf_format(rec_, out);
    bytes_in_buffer += out.size() - before;
    ++num_records;
  }

 private:
  const VidMapper* vid_;
  const QueryConfig* qc_;
  const ReferenceGenome* ref_;
  unsigned max_alt_;
  // SingleVariantOperatorBase state
  CombineAllelesLUT alleles_LUT_, reduced_alleles_LUT_;
  bool NON_REF_exists_ = false, remapping_needed_ = false, is_reference_block_only_ = false;
  std::string merged_reference_allele_;
  std::vector<std::string> merged_alt_alleles_;
  // GA4GHOperator state
  Variant remapped_variant_;
  unsigned GT_query_idx_;
  std::vector<unsigned> remapped_fields_query_idxs_;
  std::vector<unsigned> ploidy_;
  // BroadCombinedGVCFOperator state
  std::vector<FieldTuple> INFO_fields_vec_, FORMAT_fields_vec_;
  // composite vid field idx -> (query idx of the bins, of the counts).  The reference's container (broad_combined_gvcf.h:123) and
  // with it the reference's emission order of several histogram_sum fields: the iteration order of libstdc++'s unordered_map
  std::unordered_map<unsigned, std::pair<unsigned, unsigned>> INFO_histogram_field_map_;
  FieldTuple qual_tuple_{UNDEFINED_IDX, UNDEFINED_IDX, nullptr};
  std::set<std::string> hdr_ids_;
  std::string curr_contig_name_, next_contig_name_;
  int64_t curr_contig_begin_ = 0, next_contig_begin_ = 0;
  BcfRecord rec_;
  std::string ID_value_;
  std::vector<Field> spanning_deletions_remapped_fields_;
  std::vector<int> MIN_DP_vector_, DP_FORMAT_vector_;

  bool too_many_alt_alleles(unsigned num_alt) const { return num_alt > max_alt_; }
  void switch_contig() {  // :903-909
    curr_contig_name_ = next_contig_name_;
    curr_contig_begin_ = next_contig_begin_;
    vid_->get_next_contig_location(next_contig_begin_, next_contig_name_, next_contig_begin_);
  }

  // ctor part of broad_combined_gvcf.cc:140-356 + VCFAdapter::add_field_to_hdr_if_missing (vcf_adapter.cc:59-199)
  void build_field_lists_and_header(const std::string& tmpl) {
    std::vector<std::string> lines;
    {
      size_t p = 0;
      while (p < tmpl.size()) {
        size_t e = tmpl.find('\n', p);
        if (e == std::string::npos) e = tmpl.size();
        std::string l = tmpl.substr(p, e - p);
        if (l.size() >= 2 && l[0] == '#' && l[1] == '#') lines.push_back(l);
        p = e + 1;
      }
    }
    if (lines.empty()) {  // initialize_default_header (vcf_adapter.cc:374-381)
      lines.push_back("##fileformat=VCFv4.2");
      lines.push_back("##FILTER=<ID=PASS,Description=\"All filters passed\">");
      lines.push_back("##ALT=<ID=NON_REF,Description=\"Represents any possible alternative allele at this location\">");
      lines.push_back("##INFO=<ID=END,Number=1,Type=Integer,Description=\"Stop position of the interval\">");
    }
    std::set<std::string> have[3];  // FILTER, INFO, FORMAT ids
    std::set<std::string> contigs_in_hdr;
    for (auto& l : lines) {
      int cls = l.compare(0, 9, "##FILTER=") == 0 ? 0 : l.compare(0, 7, "##INFO=") == 0 ? 1 : l.compare(0, 9, "##FORMAT=") == 0 ? 2 : -1;
      size_t idp = l.find("<ID=");
      if (idp == std::string::npos) continue;
      size_t ide = l.find_first_of(",>", idp + 4);
      std::string id = l.substr(idp + 4, ide - idp - 4);
      if (cls >= 0) { have[cls].insert(id); hdr_ids_.insert(id); }
      if (l.compare(0, 9, "##contig=") == 0) contigs_in_hdr.insert(id);
    }
    auto add_field_to_hdr_if_missing = [&](const std::string& field_name, int cls) {
      const FieldInfo* fi = vid_->get_field_info(field_name);
      // a multi-D or tuple field must be a String with Number=1 in the header: a line that says otherwise is removed and added
      // again with its description (vcf_adapter.cc:62-95)
      const bool multid = fi && (fi->num_elements_in_tuple() > 1u || fi->ndim > 1u);
      std::string kept_description;
      if (multid && have[cls].count(field_name)) {
        const std::string prefix = std::string("##") + (cls == 1 ? "INFO" : "FORMAT") + "=<ID=" + field_name + ",";
        for (size_t li = 0; li < lines.size(); ++li)
          if (lines[li].compare(0, prefix.size(), prefix) == 0) {
            size_t dp = lines[li].find("Description=");
            if (dp != std::string::npos) { size_t de = lines[li].rfind('>'); kept_description = lines[li].substr(dp + 12, de - dp - 12); }
            lines.erase(lines.begin() + (long)li);
            break;
          }
        have[cls].erase(field_name);
      }
      if (have[cls].count(field_name)) return;
      std::string h = std::string("##") + (cls == 0 ? "FILTER" : cls == 1 ? "INFO" : "FORMAT") + "=<ID=" + field_name;
      if (cls != 0 && multid) {
        h += ",Number=1,Type=String";
      } else if (cls != 0) {
        if (cls == 2 && field_name == "GT") h += ",Number=1,Type=String,Description=\"Genotype\"";
        else {
          ORACLE_VERIFY(fi != nullptr);
          h += ",Number=";
          if (fi->et == ET_FLAG) h += '0';
          else switch (fi->ld) {
            case VL_FIXED: h += std::to_string(fi->num_elements); break;
            case VL_VAR: h += "."; break;
            case VL_A: h += "A"; break;
            case VL_R: h += "R"; break;
            case VL_G: h += "G"; break;
            default: throw OracleException("Unhandled field length descriptor");
          }
          h += ",Type=";
          h += fi->et == ET_FLAG ? "Flag" : fi->et == ET_INT ? "Integer" : fi->et == ET_FLOAT ? "Float" : "String";
        }
      }
      if (!kept_description.empty()) h += ",Description=" + kept_description;
      else if (!(cls == 2 && field_name == "GT")) h += ",Description=\"" + field_name + "\"";
      h += ">";
      lines.push_back(h);
      have[cls].insert(field_name);
      hdr_ids_.insert(field_name);
    };
    const QueryConfig& qc = *qc_;
    FieldTuple DP_INFO_as_FORMAT{UNDEFINED_IDX, UNDEFINED_IDX, nullptr};
    bool is_DP_INFO_queried = false;
    for (unsigned i = 0; i < qc.num_queried_attributes(); ++i) {
      const FieldInfo* fi = vid_->get_field_info(qc.attrs[i].name);
      if (!fi) continue;
      unsigned ke = qc.get_known_field_enum_for_query_idx(i);
      CombineOp op = fi->combine_op;
      bool sites_only = qc.sites_only_query;
      bool add_INFO = fi->is_INFO && ke != GVCF_END_IDX && (ke != GVCF_DP_IDX || op != OP_DP) && op != OP_MOVE_TO_FORMAT;
      bool add_FORMAT = (fi->is_FORMAT && (!sites_only || ke == GVCF_DP_FORMAT_IDX || ke == GVCF_MIN_DP_IDX)) ||
                        (fi->is_INFO && ((ke == GVCF_DP_IDX && op == OP_DP) || (op == OP_MOVE_TO_FORMAT && !sites_only)));
      if (add_INFO) {
        if (op == OP_UNKNOWN) { /* WARNING: field will NOT be part of INFO */ }
        else if (op == OP_HISTOGRAM_SUM) {   // broad_combined_gvcf.cc:194-221: the pair (bin field, count field) of the composite parent
          INFO_fields_vec_.push_back({ke, i, fi});
          add_field_to_hdr_if_missing(fi->vcf_name, 1);
          ORACLE_VERIFY(fi->is_flattened);
          const FieldInfo& parent = vid_->fields[(size_t)fi->parent_composite_field_idx];
          if (parent.num_elements_in_tuple() != 2u)
            throw OracleException("Operation histogram_sum is only supported for fields whose elements are tuple with 2 constituent elements; field " + parent.name);
          if (fi->et != ET_INT && fi->et != ET_FLOAT) throw OracleException("histogram_sum needs int or float tuple elements; field " + parent.name);
          auto& pr = INFO_histogram_field_map_.insert(std::make_pair((unsigned)fi->parent_composite_field_idx, std::make_pair(0u, 0u))).first->second;
          if (fi->element_index_in_tuple == 0u) pr.first = i; else pr.second = i;
        }
        else { INFO_fields_vec_.push_back({ke, i, fi}); add_field_to_hdr_if_missing(fi->vcf_name, 1); }
      }
      if (add_FORMAT) {
        FieldTuple t{ke, i, fi};
        if (fi->is_FORMAT || op == OP_MOVE_TO_FORMAT) { FORMAT_fields_vec_.push_back(t); add_field_to_hdr_if_missing(fi->vcf_name, 2); }
        else { DP_INFO_as_FORMAT = t; is_DP_INFO_queried = true; add_field_to_hdr_if_missing("DP", 1); }
      }
    }
    if (qc.is_defined_query_idx_for_known_field_enum(GVCF_FILTER_IDX))
      for (auto& f : vid_->fields) if (f.is_FILTER) add_field_to_hdr_if_missing(f.vcf_name, 0);
    if (is_DP_INFO_queried) FORMAT_fields_vec_.push_back(DP_INFO_as_FORMAT);
    qual_tuple_ = {GVCF_QUAL_IDX, qc.get_query_idx_for_known_field_enum(GVCF_QUAL_IDX), vid_->get_field_info("QUAL")};
    for (auto& c : vid_->contigs)
      if (!contigs_in_hdr.count(c.name)) lines.push_back("##contig=<ID=" + c.name + ",length=" + std::to_string(c.length) + ">");
    header_text.clear();
    for (auto& l : lines) { header_text += l; header_text += '\n'; }
    header_text += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO";
    if (!qc.sites_only_query) {
      header_text += "\tFORMAT";
      for (uint64_t i = 0; i < qc.get_num_rows_to_query(); ++i) {
        int64_t row = qc.get_array_row_idx_for_query_row_idx(i);
        if ((size_t)row >= vid_->row_to_callset.size() || vid_->row_to_callset[row].empty())
          throw OracleException("No sample/CallSet name specified for TileDB row " + std::to_string(row));
        header_text += '\t';
        header_text += vid_->row_to_callset[row];
      }
    }
    header_text += '\n';
  }

  // merge_reference_allele (variant_operations.cc:73-122)
  void merge_reference_allele(const Variant& variant, const QueryConfig& qc, std::string& merged) {
    size_t merged_len = merged.length();
    if (merged_len == 0u) { merged = "N"; merged_len = 1u; }
    unsigned rq = qc.get_query_idx_for_known_field_enum(GVCF_REF_IDX);
    for (const auto& call : variant.calls) {
      if (!call.is_valid) continue;
      if (call.col_begin < variant.col_begin) continue;
      const std::string& cur = call.fields[rq].sv;
      size_t cur_len = cur.length();
      if (cur_len > merged_len) {
        if (merged_len > 0 && merged[0] == 'N') merged = cur;
        else merged.append(cur, merged_len, cur_len - merged_len);
        merged_len = cur_len;
      } else if (merged[0] == 'N' && !(cur[0] == 'N')) merged = cur;
    }
  }
  // merge_alt_alleles (variant_operations.cc:134-228)
  void merge_alt_alleles(const Variant& variant, const QueryConfig& qc, const std::string& merged_ref, CombineAllelesLUT& lut,
                         std::vector<std::string>& merged_alts, bool& NON_REF_exists) {
    std::unordered_map<std::string, int> seen{{g_vcf_NON_REF, -1}};
    merged_alts.clear();
    size_t merged_ref_len = merged_ref.length();
    lut.reset_luts();
    std::vector<int> input_nr_idx(variant.calls.size(), -1);
    unsigned merged_allele_idx = 1u;
    NON_REF_exists = false;
    unsigned rq = qc.get_query_idx_for_known_field_enum(GVCF_REF_IDX), aq = qc.get_query_idx_for_known_field_enum(GVCF_ALT_IDX);
    for (size_t ci = 0; ci < variant.calls.size(); ++ci) {
      const auto& call = variant.calls[ci];
      if (!call.is_valid) continue;
      const std::string& cur_ref = call.fields[rq].sv;
      size_t cur_ref_len = cur_ref.length();
      const auto& alts = call.fields[aq].alt;
      bool suffix_needed = false;
      size_t suffix_len = 0;
      if (cur_ref_len < merged_ref_len) { suffix_needed = true; suffix_len = merged_ref_len - cur_ref_len; }
      lut.add_input_merged_idx_pair(ci, 0, 0);
      unsigned input_allele_idx = 1u;
      for (const auto& allele : alts) {
        if (IS_NON_REF_ALLELE(allele)) { input_nr_idx[ci] = (int)input_allele_idx; NON_REF_exists = true; }
        else {
          std::string a = allele;
          if (suffix_needed && !VariantUtils::is_symbolic_allele(allele)) a.append(merged_ref, cur_ref_len, suffix_len);
          auto it = seen.find(a);
          if (it == seen.end()) {
            seen[a] = (int)merged_allele_idx;
            lut.resize_luts_if_needed(merged_allele_idx + 1);
            lut.add_input_merged_idx_pair(ci, input_allele_idx, merged_allele_idx);
            merged_alts.push_back(a);
            ++merged_allele_idx;
          } else lut.add_input_merged_idx_pair(ci, input_allele_idx, it->second);
        }
        ++input_allele_idx;
      }
    }
    if (NON_REF_exists) {
      merged_alts.push_back("&");  // kept in TileDB spelling; printed as <NON_REF>
      size_t nr = merged_alts.size();
      lut.resize_luts_if_needed(nr + 1);
      for (size_t ci = 0; ci < variant.calls.size(); ++ci)
        if (variant.calls[ci].is_valid && input_nr_idx[ci] >= 0) lut.add_input_merged_idx_pair(ci, input_nr_idx[ci], (int64_t)nr);
    }
  }
  // remap_GT_field (variant_operations.cc:233-263)
  static void remap_GT_field(const std::vector<int>& in, std::vector<int>& outv, const CombineAllelesLUT& lut, uint64_t call, unsigned num_merged,
                             bool NON_REF_exists, const FieldInfo& gt_info) {
    unsigned step = gt_info.contains_phase_information() ? 2u : 1u;
    for (unsigned i = 0; i < in.size(); i += step) {
      if (is_tiledb_missing_value(in[i]) || in[i] == -1 || in[i] == bcf_int32_missing) outv[i] = in[i];
      else {
        int64_t o = lut.get_merged_idx_for_input(call, in[i]);
        if (CombineAllelesLUT::is_missing_value(o)) outv[i] = NON_REF_exists ? (int)(num_merged - 1u) : -1;
        else outv[i] = (int)o;
      }
      if (step == 2u && i + 1u < in.size()) outv[i + 1u] = in[i + 1u];
    }
  }
  // VariantFieldHandler<T>::remap_vector_data (variant_field_handler.cc:497-527): orig -> dst (already resized)
  void remap_vector_data(const Field& orig, Field& dst, uint64_t call, const CombineAllelesLUT& lut, unsigned num_merged, bool NON_REF_exists,
                         unsigned ploidy, const FieldInfo& fi) {
    if (!orig.non_null) return;
    auto run = [&](size_t input_size, const RemapSink& sink) {
      if (fi.is_genotype_dependent()) remap_based_on_genotype(input_size, call, lut, num_merged, NON_REF_exists, ploidy, sink);
      else remap_based_on_alleles(input_size, call, lut, num_merged, NON_REF_exists, fi.is_only_ALT_dependent(), sink);
    };
    if (orig.kind == FK_INT) {
      run(orig.iv.size(), [&](uint64_t o, bool has, uint64_t i) { if (o < dst.iv.size()) dst.iv[o] = has ? orig.iv[i] : bcf_int32_missing; });
    } else if (orig.kind == FK_FLOAT) {
      run(orig.fv.size(), [&](uint64_t o, bool has, uint64_t i) { if (o < dst.fv.size()) dst.fv[o] = has ? orig.fv[i] : u2f(bcf_float_missing_bits); });
    } else if (orig.kind == FK_STRING && fi.ndim == 2u) {   // remap_allele_specific_annotations (variant_operations.cc:551-570)
      dst.sv = remap_allele_specific_annotations(orig.sv, call, lut, num_merged, NON_REF_exists, fi.is_only_ALT_dependent());
    } else throw OracleException("remap of non-numeric allele-dependent field is not supported");
  }

  // SingleVariantOperatorBase::operate + GA4GHOperator::operate (variant_operations.cc:362-378, 572-695)
  void ga4gh_operate(Variant& variant, const QueryConfig& qc) {
    merged_reference_allele_.resize(0u);
    merged_alt_alleles_.clear();
    merge_reference_allele(variant, qc, merged_reference_allele_);
    alleles_LUT_.resize_luts_if_needed(variant.calls.size(), 10u);
    merge_alt_alleles(variant, qc, merged_reference_allele_, alleles_LUT_, merged_alt_alleles_, NON_REF_exists_);
    is_reference_block_only_ = (merged_reference_allele_.length() == 1u && merged_alt_alleles_.size() == 1u && IS_NON_REF_ALLELE(merged_alt_alleles_[0]));
    remapping_needed_ = !is_reference_block_only_;
    remapped_variant_.deep_copy_simple_members(variant);
    for (auto& c : remapped_variant_.calls) if (c.fields.size() < qc.num_queried_attributes()) c.fields.resize(qc.num_queried_attributes());
    unsigned num_merged = (unsigned)merged_alt_alleles_.size() + 1u;
    if (remapping_needed_) {
      if (GT_query_idx_ != UNDEFINED_IDX) {
        const FieldInfo& gti = *qc.attrs[GT_query_idx_].info;
        for (size_t ci = 0; ci < remapped_variant_.calls.size(); ++ci) {
          auto& rc = remapped_variant_.calls[ci];
          if (!rc.is_valid) continue;
          if (ploidy_.size() <= ci) ploidy_.resize(ci + 1);
          ploidy_[ci] = 0u;
          Field& rf = rc.fields[GT_query_idx_];
          const Field& of = variant.calls[ci].fields[GT_query_idx_];
          copy_field(rf, of);
          if (rf.non_null && rf.valid) {
            remap_GT_field(of.iv, rf.iv, alleles_LUT_, ci, num_merged, NON_REF_exists_, gti);
            ploidy_[ci] = gti.get_ploidy((unsigned)of.iv.size());
          }
        }
      }
      for (unsigned q : remapped_fields_query_idxs_) {
        const FieldInfo& fi = *qc.attrs[q].info;
        if (fi.is_genotype_dependent() && too_many_alt_alleles(num_merged - 1u)) continue;  // warning on stderr in the reference
        for (size_t ci = 0; ci < remapped_variant_.calls.size(); ++ci) {
          auto& rc = remapped_variant_.calls[ci];
          if (!rc.is_valid) continue;
          Field& rf = rc.fields[q];
          const Field& of = variant.calls[ci].fields[q];
          copy_field(rf, of);
          if (rf.non_null && rf.valid) {
            unsigned cur_ploidy = ci < ploidy_.size() ? ploidy_[ci] : 0u;
            unsigned num_merged_elements = num_elements_for_length(fi, num_merged - 1u, cur_ploidy, 0u);
            rf.resize(num_merged_elements);
            remap_vector_data(of, rf, ci, alleles_LUT_, num_merged, NON_REF_exists_, cur_ploidy, fi);
          }
        }
      }
    }
    Field& REF = remapped_variant_.common_fields[0];
    REF.non_null = REF.valid = true; REF.kind = FK_STRING; REF.sv = merged_reference_allele_;
    Field& ALT = remapped_variant_.common_fields[1];
    ALT.non_null = ALT.valid = true; ALT.kind = FK_ALT; ALT.alt = merged_alt_alleles_;
  }

  // handle_deletions (broad_combined_gvcf.cc:912-1078)
  void handle_deletions(Variant& variant, const QueryConfig& qc) {
    reduced_alleles_LUT_.resize_luts_if_needed(variant.calls.size(), 3u);
    reduced_alleles_LUT_.reset_luts();
    const FieldInfo* gti = qc.is_defined_query_idx_for_known_field_enum(GVCF_GT_IDX) ? qc.attrs[GT_query_idx_].info : nullptr;
    unsigned rq = qc.get_query_idx_for_known_field_enum(GVCF_REF_IDX), aq = qc.get_query_idx_for_known_field_enum(GVCF_ALT_IDX);
    for (size_t ci = 0; ci < variant.calls.size(); ++ci) {
      auto& call = variant.calls[ci];
      if (!call.is_valid) continue;
      if (!(call.contains_deletion && variant.col_begin > call.col_begin)) continue;
      std::string& ref_allele = call.fields[rq].sv;
      auto& alt_alleles = call.fields[aq].alt;
      ORACLE_VERIFY(alt_alleles.size() > 0u);
      if (alt_alleles[0u] == g_vcf_SPANNING_DELETION && (alt_alleles.size() == 1u || (alt_alleles.size() == 2u && IS_NON_REF_ALLELE(alt_alleles[1u])))) continue;
      reduced_alleles_LUT_.resize_luts_if_needed(variant.calls.size(), alt_alleles.size() + 1u);
      reduced_alleles_LUT_.add_input_merged_idx_pair(ci, 0, 0);
      unsigned ploidy = 0u;
      Field* orig_GT = (GT_query_idx_ != UNDEFINED_IDX) ? &call.fields[GT_query_idx_] : nullptr;
      if (orig_GT && orig_GT->non_null && orig_GT->valid) ploidy = gti->get_ploidy((unsigned)orig_GT->iv.size());
      else orig_GT = nullptr;
      int lowest_deletion_allele_idx = -1;
      int lowest_PL_value = INT_MAX;
      const Field* PL = qc.is_defined_query_idx_for_known_field_enum(GVCF_PL_IDX) ? &call.fields[qc.get_query_idx_for_known_field_enum(GVCF_PL_IDX)] : nullptr;
      bool has_NON_REF = false;
      bool PL_exists = PL && PL->non_null && PL->valid;
      static const std::vector<int> empty;
      const std::vector<int>& PL_vector = PL_exists ? PL->iv : empty;
      std::vector<int> cur_gt(ploidy);
      for (unsigned i = 0; i < alt_alleles.size(); ++i) {
        int allele_idx = (int)i + 1;
        if (VariantUtils::is_deletion(ref_allele, alt_alleles[i])) {
          if (lowest_deletion_allele_idx < 0) lowest_deletion_allele_idx = allele_idx;
          if (PL_exists) {
            cur_gt.assign(ploidy, allele_idx);
            uint64_t gt_idx = get_genotype_index(cur_gt, true);
            if (gt_idx < PL_vector.size() && PL_vector[gt_idx] < lowest_PL_value) { lowest_PL_value = PL_vector[gt_idx]; lowest_deletion_allele_idx = allele_idx; }
          }
        } else if (IS_NON_REF_ALLELE(alt_alleles[i])) { reduced_alleles_LUT_.add_input_merged_idx_pair(ci, allele_idx, 2); has_NON_REF = true; }
      }
      ORACLE_VERIFY(lowest_deletion_allele_idx >= 1);
      reduced_alleles_LUT_.add_input_merged_idx_pair(ci, lowest_deletion_allele_idx, 1);
      if (has_NON_REF) { alt_alleles.resize(2u); alt_alleles[1u] = "&"; } else alt_alleles.resize(1u);
      ref_allele = "N";
      alt_alleles[0u] = g_vcf_SPANNING_DELETION;
      unsigned num_reduced = (unsigned)alt_alleles.size() + 1u;
      for (unsigned i = 0; i < remapped_fields_query_idxs_.size(); ++i) {
        unsigned q = remapped_fields_query_idxs_[i];
        const FieldInfo& fi = *qc.attrs[q].info;
        Field& cur = call.fields[q];
        if (cur.non_null && cur.valid) {
          unsigned num_reduced_elements = num_elements_for_length(fi, num_reduced - 1u, ploidy, 0u);
          copy_field(spanning_deletions_remapped_fields_[i], cur);
          cur.resize(num_reduced_elements);
          remap_vector_data(spanning_deletions_remapped_fields_[i], cur, ci, reduced_alleles_LUT_, num_reduced, has_NON_REF, ploidy, fi);
        }
      }
      if (orig_GT) {
        std::vector<int>& input_GT = orig_GT->iv;
        bool remap_on_input = update_GT_to_correspond_to_min_PL_value(qc, PL, input_GT, *gti, num_reduced, has_NON_REF);
        if (remap_on_input) {
          std::vector<int> tmp(input_GT.size());
          remap_GT_field(input_GT, tmp, reduced_alleles_LUT_, ci, num_reduced, has_NON_REF, *gti);
          input_GT = tmp;
        }
      }
      for (const auto& t : INFO_fields_vec_) { Field& f = call.fields[t.query_idx]; if (f.non_null) f.valid = false; }
    }
  }
  // update_GT_to_correspond_to_min_PL_value (:1080-1118) + determine_allele_combination_and_genotype_index_for_min_value
  // (variant_field_handler.cc:402-494).  NB: PL here is the *already reduced* PL of the call.
  bool update_GT_to_correspond_to_min_PL_value(const QueryConfig& qc, const Field* PL, std::vector<int>& input_GT, const FieldInfo& gti,
                                               unsigned num_alleles, bool has_NON_REF) {
    bool remap_on_input = true;
    if (PL && PL->non_null && PL->valid && qc.produce_GT_with_min_PL_value_for_spanning_deletions) {
      remap_on_input = false;
      unsigned ploidy = gti.get_ploidy((unsigned)input_GT.size());
      const std::vector<int>& data = PL->iv;
      int cur_min = INT_MAX;
      bool found = false;
      std::vector<int> best;
      auto track = [&](std::vector<int>& alleles) {  // GenotypeForMinValueTracker::track_minimum
        std::vector<int> keep = alleles;
        uint64_t gt_idx = get_genotype_index(alleles, false);
        if (gt_idx < data.size() && is_bcf_valid_value(data[gt_idx]) && data[gt_idx] < cur_min) {
          cur_min = data[gt_idx]; best = alleles; found = true;
        }
        (void)keep;
      };
      std::vector<int> v(ploidy);
      unsigned num_genotypes = get_number_of_genotypes(num_alleles - 1u, ploidy);
      switch (ploidy) {
        case 1u:
          for (unsigned i = 0; i < std::min<unsigned>(num_genotypes, (unsigned)data.size()); ++i) { v[0] = (int)i; track(v); }
          break;
        case 2u:
          for (unsigned i = 0; i < num_alleles; ++i) { v[0] = (int)i; for (unsigned j = i; j < num_alleles; ++j) { v[1] = (int)j; track(v); } }
          break;
        default: {
          CombineAllelesLUT id;
          id.resize_luts_if_needed(1u, num_alleles);
          for (unsigned i = 0; i < num_alleles; ++i) id.add_input_merged_idx_pair(0u, i, i);
          remap_genotype_general(0, id, num_alleles, has_NON_REF, ploidy, [&](uint64_t, bool, std::vector<int>& in) { track(in); });
        }
      }
      if (found) {
        unsigned step = gti.contains_phase_information() ? 2u : 1u;
        for (unsigned i = 0, j = 0; i < input_GT.size(); i += step, ++j) input_GT[i] = best[j];
      } else remap_on_input = true;
    }
    return remap_on_input;
  }

  // ---- INFO ---------------------------------------------------------------------------------------
  struct CombineResult { std::vector<int32_t> iv; std::vector<float> fv; std::string sv; int type = 0; };
  const Variant& source_variant(const Variant& variant, const FieldInfo& fi, unsigned q) const {
    return (remapping_needed_ && (fi.is_allele_dependent() || q == GT_query_idx_)) ? remapped_variant_ : variant;
  }
  // handle_VCF_field_combine_operation (:374-429) with the reducers of variant_field_handler.cc:529-802
  bool handle_VCF_field_combine_operation(const Variant& variant, const FieldTuple& t, CombineResult& res) {
    const FieldInfo& fi = *t.info;
    unsigned q = t.query_idx;
    if (fi.is_genotype_dependent() && too_many_alt_alleles((unsigned)merged_alt_alleles_.size())) return false;
    const Variant& src = source_variant(variant, fi, q);
    bool is_float = fi.et == ET_FLOAT, is_int = fi.et == ET_INT;
    res.type = is_float ? 1 : is_int ? 0 : 2;
    auto for_each_valid = [&](const std::function<void(const Field&)>& fn) {
      for (const auto& c : src.calls) { if (!c.is_valid) continue; const Field& f = c.fields[q]; if (f.non_null && f.valid) fn(f); }
    };
    switch (fi.combine_op) {
      case OP_SUM: case OP_MEAN: {
        if (is_float) {
          float sum = 0; unsigned n = 0;
          for_each_valid([&](const Field& f) { float v = f.fv[0]; if (is_bcf_valid_value(v)) { sum += v; ++n; } });
          if (!n) return false;
          if (fi.combine_op == OP_MEAN) sum = sum / n;
          res.fv.assign(1, sum); return true;
        } else if (is_int) {
          int sum = 0; unsigned n = 0;
          for_each_valid([&](const Field& f) { int v = f.iv[0]; if (is_bcf_valid_value(v)) { sum += v; ++n; } });
          if (!n) return false;
          if (fi.combine_op == OP_MEAN) sum = sum / n;   // int / unsigned, as written in get_valid_mean (:596-607): the sum is converted to unsigned
          res.iv.assign(1, sum); return true;
        }
        throw OracleException("sum/mean on a string field");
      }
      case OP_MEDIAN: {
        if (is_float) {
          std::vector<float> v;
          for_each_valid([&](const Field& f) { if (is_bcf_valid_value(f.fv[0])) v.push_back(f.fv[0]); });
          if (v.empty()) return false;
          size_t mid = v.size() / 2u;
          std::nth_element(v.begin(), v.begin() + mid, v.end());
          res.fv.assign(1, v[mid]); return true;
        } else if (is_int) {
          std::vector<int> v;
          for_each_valid([&](const Field& f) { if (is_bcf_valid_value(f.iv[0])) v.push_back(f.iv[0]); });
          if (v.empty()) return false;
          size_t mid = v.size() / 2u;
          std::nth_element(v.begin(), v.begin() + mid, v.end());
          res.iv.assign(1, v[mid]); return true;
        }
        throw OracleException("median on a string field");
      }
      case OP_ELEMENT_WISE_SUM: {  // compute_valid_element_wise_sum (:618-664)
        if (fi.ndim == 2u) {       // compute_valid_element_wise_sum_2D_vector + stringify_2D_vector (:666-740)
          res.type = 2;
          if (is_float) return element_wise_sum_2D<float>(src, q, fi, res.sv);
          if (is_int) return element_wise_sum_2D<int32_t>(src, q, fi, res.sv);
          throw OracleException("2-D element_wise_sum on a string field");
        }
        unsigned num_valid = 0;
        if (is_float) {
          std::vector<float>& r = res.fv;
          for_each_valid([&](const Field& f) {
            if (f.fv.size() > r.size()) r.resize(f.fv.size());
            for (size_t i = 0; i < f.fv.size(); ++i) {
              float v = f.fv[i];
              if (!is_bcf_valid_value(v)) continue;
              if (i < num_valid && is_bcf_valid_value(r[i])) r[i] += v;
              else { r[i] = v; if (i >= num_valid) { for (size_t j = num_valid; j < i; ++j) r[j] = u2f(bcf_float_missing_bits); num_valid = (unsigned)i + 1u; } }
            }
          });
          if (num_valid > 0u) r.resize(num_valid);
        } else if (is_int) {
          std::vector<int>& r = res.iv;
          for_each_valid([&](const Field& f) {
            if (f.iv.size() > r.size()) r.resize(f.iv.size());
            for (size_t i = 0; i < f.iv.size(); ++i) {
              int v = f.iv[i];
              if (!is_bcf_valid_value(v)) continue;
              if (i < num_valid && is_bcf_valid_value(r[i])) r[i] += v;
              else { r[i] = v; if (i >= num_valid) { for (size_t j = num_valid; j < i; ++j) r[j] = bcf_int32_missing; num_valid = (unsigned)i + 1u; } }
            }
          });
          if (num_valid > 0u) r.resize(num_valid);
        } else throw OracleException("element_wise_sum on a string field");
        return num_valid > 0u;
      }
      case OP_CONCATENATE: {
        size_t n = 0;
        for_each_valid([&](const Field& f) {
          if (is_float) { res.fv.insert(res.fv.end(), f.fv.begin(), f.fv.end()); n += f.fv.size(); }
          else if (is_int) { res.iv.insert(res.iv.end(), f.iv.begin(), f.iv.end()); n += f.iv.size(); }
          else { res.sv += f.sv; n += f.sv.size(); }
        });
        return n > 0u;
      }
      case OP_HISTOGRAM_SUM: return false;
      default: throw OracleException("Unknown VCF field combine operation");
    }
  }
  // compute_valid_element_wise_sum_2D_vector (variant_field_handler.cc:666-714) + stringify_2D_vector (:716-740)
  template <class T> bool element_wise_sum_2D(const Variant& src, unsigned q, const FieldInfo& fi, std::string& text) {
    uint64_t num_valid_elements = 0;
    std::vector<std::vector<T>> result;
    for (const auto& c : src.calls) {
      if (!c.is_valid) continue;
      const Field& f = c.fields[q];
      if (!(f.non_null && f.valid)) continue;
      MultiD2View idx(f.sv);
      if (idx.num_entries() > result.size()) result.resize(idx.num_entries());
      for (uint64_t d0 = 0; d0 < idx.num_entries(); ++d0) {
        const uint64_t num_elements = idx.bytes(d0) / sizeof(T);
        if (num_elements > result[d0].size()) result[d0].resize(num_elements, FieldVec<T>::missing());
        for (uint64_t i = 0; i < num_elements; ++i) {
          T val; memcpy(&val, idx.ptr(d0) + i * sizeof(T), sizeof(T));
          if (is_bcf_valid_value(val)) {
            if (is_bcf_valid_value(result[d0][i])) result[d0][i] += val; else result[d0][i] = val;
            ++num_valid_elements;
          }
        }
      }
    }
    if (num_valid_elements == 0u) return false;
    std::stringstream s;
    for (size_t i = 0; i < result.size(); ++i) {
      if (i) s << fi.vcf_delimiter[0];
      for (size_t j = 0; j < result[i].size(); ++j) {
        if (j) s << fi.vcf_delimiter[1];
        if (is_bcf_valid_value(result[i][j])) s << std::fixed << std::setprecision(3) << result[i][j];
      }
    }
    text = s.str();
    return true;
  }
  // compute_valid_histogram_sum_2D_vector_and_stringify (broad_combined_gvcf.cc:431-521)
  template <class T1, class T2> bool histogram_sum_2D(const Variant& src, unsigned q_bin, unsigned q_count, const FieldInfo& fi_bin, std::string& text) {
    uint64_t num_calls_with_field = 0;
    std::vector<std::map<T1, T2>> histogram_map_vec;
    for (const auto& c : src.calls) {
      if (!c.is_valid) continue;
      const Field& fb = c.fields[q_bin];
      const Field& fc = c.fields[q_count];
      if (!(fb.non_null && fb.valid)) continue;
      ORACLE_VERIFY(fc.non_null && fc.valid);
      MultiD2View ib(fb.sv), ic(fc.sv);
      ORACLE_VERIFY(ib.num_entries() == ic.num_entries());
      if (ib.num_entries() > histogram_map_vec.size()) histogram_map_vec.resize(ib.num_entries());
      for (uint64_t d0 = 0; d0 < ib.num_entries(); ++d0) {
        const uint64_t num_elements = ib.bytes(d0) / sizeof(T1);
        ORACLE_VERIFY(num_elements == ic.bytes(d0) / sizeof(T2));
        auto& m = histogram_map_vec[d0];
        for (uint64_t i = 0; i < num_elements; ++i) {
          T1 vb; T2 vc;
          memcpy(&vb, ib.ptr(d0) + i * sizeof(T1), sizeof(T1));
          memcpy(&vc, ic.ptr(d0) + i * sizeof(T2), sizeof(T2));
          if (is_bcf_valid_value(vb) && is_bcf_valid_value(vc)) {
            auto ins = m.insert(std::pair<T1, T2>(vb, vc));
            if (!ins.second) ins.first->second += vc;
          }
        }
      }
      ++num_calls_with_field;
    }
    if (num_calls_with_field == 0u) return false;
    std::stringstream s;
    for (size_t i = 0; i < histogram_map_vec.size(); ++i) {
      if (i) s << fi_bin.vcf_delimiter[0];
      bool first = true;
      for (auto& pr : histogram_map_vec[i]) {
        if (!first) s << fi_bin.vcf_delimiter[1];
        s << std::fixed << std::setprecision(3) << pr.first << fi_bin.vcf_delimiter[1] << pr.second;
        first = false;
      }
    }
    text = s.str();
    return true;
  }
  // handle_INFO_fields (:523-601)
  void handle_INFO_fields(const Variant& variant) {
    if (remapped_variant_.col_end > remapped_variant_.col_begin) {
      RecInfo& e = rec_.info_slot("END");
      e.type = 0; e.iv.assign(1, (int)(remapped_variant_.col_end - curr_contig_begin_ + 1));
    }
    for (const auto& t : INFO_fields_vec_) {
      CombineResult res;
      if (handle_VCF_field_combine_operation(variant, t, res)) {
        RecInfo& e = rec_.info_slot(t.info->vcf_name);
        e.type = res.type; e.iv = res.iv; e.fv = res.fv; e.sv = res.sv;
      }
    }
    for (auto& kv : INFO_histogram_field_map_) {   // (:559-600) after the other INFO fields, in the unordered_map's iteration order
      const unsigned q_bin = kv.second.first, q_count = kv.second.second;
      const FieldInfo& fb = *qc_->attrs[q_bin].info;
      const FieldInfo& fc = *qc_->attrs[q_count].info;
      const Variant& src = (remapping_needed_ && fc.is_allele_dependent()) ? remapped_variant_ : variant;
      std::string text;
      bool found = false;
      if (fb.et == ET_FLOAT && fc.et == ET_FLOAT) found = histogram_sum_2D<float, float>(src, q_bin, q_count, fb, text);
      else if (fb.et == ET_FLOAT) found = histogram_sum_2D<float, int32_t>(src, q_bin, q_count, fb, text);
      else if (fc.et == ET_FLOAT) found = histogram_sum_2D<int32_t, float>(src, q_bin, q_count, fb, text);
      else found = histogram_sum_2D<int32_t, int32_t>(src, q_bin, q_count, fb, text);
      if (found && !text.empty()) { RecInfo& e = rec_.info_slot(fb.vcf_name); e.type = 2; e.sv = text; }   // (bcf_update_info with 0 values adds nothing)
    }
  }

  // ---- FORMAT -------------------------------------------------------------------------------------
  // collect_and_extend_fields (variant_field_handler.cc:804-871), text VCF flavour (no htsjdk quirk flags)
  template <class T, class GetVec>
  bool collect_and_extend(const Variant& src, unsigned q, bool is_GT, T missing, T vend, GetVec getv, std::vector<T>& outv, unsigned& per_call) {
    unsigned max_per_call = 0, valid = 0;
    for (const auto& c : src.calls) { if (!c.is_valid) continue; const Field& f = c.fields[q]; if (f.non_null && f.valid) { max_per_call = std::max<unsigned>(max_per_call, (unsigned)f.length()); ++valid; } }
    if (!valid) return false;
    outv.clear();
    for (const auto& c : src.calls) {
      const Field& f = c.fields[q];
      unsigned inserted = 0;
      if (c.is_valid && f.non_null && f.valid) { const auto& v = getv(f); outv.insert(outv.end(), v.begin(), v.end()); inserted = (unsigned)v.size(); }
      if (inserted == 0u) { outv.push_back(is_GT ? vend : missing); ++inserted; }
      for (; inserted < max_per_call; ++inserted) outv.push_back(vend);
    }
    per_call = std::max(max_per_call, 1u);
    return true;
  }
  // encode_GT_vector (broad_combined_gvcf.cc:53-138)
  void encode_GT(std::vector<int>& v, unsigned per_sample, unsigned num_calls, bool phase_in_tiledb, bool produce_GT, unsigned& out_per_sample) {
    auto enc = [&](int value, bool phased, bool with_phase) -> int {
      if (!is_bcf_valid_value(value)) return value;
      if (produce_GT) return with_phase && phased ? (((value + 1) << 1) | 1) : ((value + 1) << 1);
      return with_phase && phased ? (bcf_gt_missing | 1) : bcf_gt_missing;
    };
    unsigned max_ploidy = phase_in_tiledb ? ((per_sample + 1u) >> 1u) : per_sample;
    std::vector<int> o;
    o.reserve((size_t)max_ploidy * num_calls);
    for (unsigned s = 0; s < num_calls; ++s) {
      size_t base = (size_t)s * per_sample;
      if (per_sample > 0u) o.push_back(enc(v[base], false, false));
      if (phase_in_tiledb) for (unsigned k = 2u; k < per_sample; k += 2u) o.push_back(enc(v[base + k], v[base + k - 1u] > 0, true));
      else for (unsigned k = 1u; k < per_sample; ++k) o.push_back(enc(v[base + k], false, false));
    }
    v.swap(o);
    out_per_sample = max_ploidy;
  }
  // handle_FORMAT_fields (:603-727)
  void handle_FORMAT_fields(const Variant& variant) {
    const QueryConfig& qc = *qc_;
    bool valid_DP_found = false, valid_MIN_DP_found = false, valid_DP_FORMAT_found = false;
    size_t ncalls = remapped_variant_.calls.size();
    MIN_DP_vector_.resize(ncalls); DP_FORMAT_vector_.resize(ncalls);
    std::vector<int> DP_INFO_vec;
    for (const auto& t : FORMAT_fields_vec_) {
      const FieldInfo& fi = *t.info;
      unsigned q = t.query_idx;
      if (fi.is_genotype_dependent() && too_many_alt_alleles((unsigned)merged_alt_alleles_.size())) continue;
      const Variant& src = source_variant(variant, fi, q);
      bool do_insert = !qc.sites_only_query;
      unsigned per_call = 0;
      if (fi.et == ET_INT) {
        std::vector<int> v;
        if (!collect_and_extend<int>(src, q, t.known_enum == GVCF_GT_IDX, bcf_int32_missing, bcf_int32_vector_end,
                                     [](const Field& f) -> const std::vector<int>& { return f.iv; }, v, per_call)) continue;
        switch (t.known_enum) {
          case GVCF_GT_IDX: {
            bool phase_in_tiledb = qc.is_defined_query_idx_for_known_field_enum(GVCF_GT_IDX) ? fi.contains_phase_information() : true;
            bool produce_GT = qc.is_defined_query_idx_for_known_field_enum(GVCF_GT_IDX) ? qc.produce_GT_field : false;
            unsigned o = 0;
            encode_GT(v, per_call, (unsigned)variant.calls.size(), phase_in_tiledb, produce_GT, o);
            per_call = o;
            break;
          }
          case GVCF_MIN_DP_IDX: for (size_t i = 0; i < ncalls; ++i) MIN_DP_vector_[i] = v[i * per_call]; valid_MIN_DP_found = true; break;
          case GVCF_DP_FORMAT_IDX: for (size_t i = 0; i < ncalls; ++i) DP_FORMAT_vector_[i] = v[i * per_call]; valid_DP_FORMAT_found = true; do_insert = false; break;
          case GVCF_DP_IDX: DP_INFO_vec = v; valid_DP_found = true; do_insert = false; break;
          default: break;
        }
        if (do_insert) { RecFmt& e = rec_.fmt_slot(fi.vcf_name); e.type = 0; e.n = per_call; e.iv = v; }
      } else if (fi.et == ET_FLOAT) {
        std::vector<float> v;
        if (!collect_and_extend<float>(src, q, false, u2f(bcf_float_missing_bits), u2f(bcf_float_vector_end_bits),
                                       [](const Field& f) -> const std::vector<float>& { return f.fv; }, v, per_call)) continue;
        if (do_insert) { RecFmt& e = rec_.fmt_slot(fi.vcf_name); e.type = 1; e.n = per_call; e.fv = v; }
      } else {
        std::vector<char> v;
        struct G { std::vector<char> tmp; } g;
        unsigned max_per_call = 0, valid = 0;
        for (const auto& c : src.calls) { if (!c.is_valid) continue; const Field& f = c.fields[q]; if (f.non_null && f.valid) { max_per_call = std::max<unsigned>(max_per_call, (unsigned)f.sv.size()); ++valid; } }
        if (!valid) continue;
        for (const auto& c : src.calls) {
          const Field& f = c.fields[q];
          unsigned inserted = 0;
          if (c.is_valid && f.non_null && f.valid) { v.insert(v.end(), f.sv.begin(), f.sv.end()); inserted = (unsigned)f.sv.size(); }
          if (inserted == 0u) { v.push_back(bcf_str_missing); ++inserted; }
          for (; inserted < max_per_call; ++inserted) v.push_back(bcf_str_vector_end);
        }
        per_call = std::max(max_per_call, 1u);
        if (do_insert) { RecFmt& e = rec_.fmt_slot(fi.vcf_name); e.type = 2; e.n = per_call; e.cv.assign(v.begin(), v.end()); }
      }
    }
    if (valid_DP_found || valid_DP_FORMAT_found) {
      int sum_INFO_DP = 0;
      bool found_one_valid_DP_FORMAT = false;
      size_t dp_stride = valid_DP_found ? DP_INFO_vec.size() / std::max<size_t>(ncalls, 1) : 1;
      for (size_t j = 0; j < ncalls; ++j) {
        int dp_info_val = valid_DP_found ? DP_INFO_vec[j * dp_stride] : bcf_int32_missing;
        int dp_format_val = valid_DP_FORMAT_found ? DP_FORMAT_vector_[j] : bcf_int32_missing;
        if (!is_bcf_valid_value(dp_info_val)) {
          if (valid_MIN_DP_found && is_bcf_valid_value(MIN_DP_vector_[j])) dp_info_val = MIN_DP_vector_[j];
          else dp_info_val = dp_format_val;
        }
        DP_FORMAT_vector_[j] = dp_format_val;
        found_one_valid_DP_FORMAT = is_bcf_valid_value(dp_format_val) || found_one_valid_DP_FORMAT;
        sum_INFO_DP += (is_bcf_valid_value(dp_info_val) ? dp_info_val : 0);
      }
      if (found_one_valid_DP_FORMAT && !qc.sites_only_query) { RecFmt& e = rec_.fmt_slot("DP"); e.type = 0; e.n = 1; e.iv = DP_FORMAT_vector_; }
      if (sum_INFO_DP > 0 && !is_reference_block_only_) { RecInfo& e = rec_.info_slot("DP"); e.type = 0; e.iv.assign(1, sum_INFO_DP); }
    }
  }
  // merge_ID_field (:730-763): a std::set<std::string> #ifdef DEBUG (the build the goldens come from), a
  // std::unordered_set<std::string> otherwise - here the library's own container, so the order is libstdc++'s by construction
  template <class Set> void merge_ID_tokens(const Variant& variant, unsigned q) {
    Set ids;
    for (const auto& c : variant.calls) {
      if (!c.is_valid) continue;
      const Field& f = c.fields[q];
      if (!(f.non_null && f.valid)) continue;
      size_t last = 0;
      for (size_t i = 0; i < f.sv.length(); ++i) if (f.sv[i] == ';') { ids.insert(f.sv.substr(last, i - last)); last = i + 1; }
      if (f.sv.length() > last) ids.insert(f.sv.substr(last));
    }
    ID_value_.clear();
    for (auto& s : ids) { ID_value_ += s; ID_value_ += ';'; }
    if (!ID_value_.empty()) ID_value_.pop_back();
  }
  void merge_ID_field(const Variant& variant, unsigned q) {
    if (qc_->id_union_order_unordered_set) merge_ID_tokens<std::unordered_set<std::string>>(variant, q);
    else merge_ID_tokens<std::set<std::string>>(variant, q);
  }
};

}  // namespace gdb_oracle
