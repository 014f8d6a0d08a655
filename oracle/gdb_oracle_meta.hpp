// gdb_oracle_meta.hpp - TEST ORACLE (metadata layer).  NOT PRODUCT CODE.
//
// CPU restatement of the metadata the reference's scan/combine path consults: vid mapping,
// callset mapping, array schema and the query configuration.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Reference files restated here (paths relative to /root/reference/src/main/cpp):
//   src/utils/vid_mapper.cc:33-104 (name tables), :240-304 (contig lookup), :354-442 (schema),
//     :611-684 (mandatory fields), :727-748 (INFO+FORMAT split), :802-830, :1202-1535 (JSON)
//   src/utils/known_field_info.cc:42-71, :239-308 (known fields, default lengths / combine ops)
//   src/config/json_config.cc:195-658 (query JSON keys), src/config/variant_query_config.cc:37-278
//   src/genomicsdb/query_variants.cc:243-294, :578-685 (do_query_bookkeeping)
#pragma once
#include <algorithm>
#include <cassert>
#include <climits>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "oracle_json.hpp"

namespace gdb_oracle {

// ---- htslib / TileDB sentinels (include/vcf/vcf.h:59-218; htslib vcf.h) -------------------------
static const int32_t bcf_int32_missing = INT32_MIN;
static const int32_t bcf_int32_vector_end = INT32_MIN + 1;
static const uint32_t bcf_float_missing_bits = 0x7F800001u;
static const uint32_t bcf_float_vector_end_bits = 0x7F800002u;
static const char bcf_str_missing = 0x07;
static const char bcf_str_vector_end = 0;
static const int32_t bcf_gt_missing = 0;
static const int32_t TILEDB_EMPTY_INT32 = INT32_MAX;
static const int64_t TILEDB_EMPTY_INT64 = INT64_MAX;
static const uint32_t TILEDB_EMPTY_FLOAT32_BITS = 0x7F7FFFFFu;  // FLT_MAX
static const char TILEDB_EMPTY_CHAR = CHAR_MAX;

inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
inline bool bcf_float_is_missing(float f) { return f2u(f) == bcf_float_missing_bits; }
inline bool bcf_float_is_vector_end(float f) { return f2u(f) == bcf_float_vector_end_bits; }
inline bool is_bcf_valid_value(int32_t v) { return v != bcf_int32_missing && v != bcf_int32_vector_end; }
inline bool is_bcf_valid_value(float v) { return !bcf_float_is_missing(v) && !bcf_float_is_vector_end(v); }
inline bool is_bcf_valid_value(char v) { return v != bcf_str_missing && v != bcf_str_vector_end; }
inline bool is_tiledb_missing_value(int32_t v) { return v == TILEDB_EMPTY_INT32; }
inline bool is_tiledb_missing_value(float v) { return f2u(v) == TILEDB_EMPTY_FLOAT32_BITS; }
inline bool is_tiledb_missing_value(char v) { return v == TILEDB_EMPTY_CHAR; }
inline bool is_tiledb_missing_value(int8_t v) { return v == (int8_t)TILEDB_EMPTY_CHAR; }

class OracleException : public std::runtime_error {
 public:
  explicit OracleException(const std::string& m) : std::runtime_error(m) {}
};
#define ORACLE_VERIFY(X) do { if (!(X)) throw OracleException(std::string("oracle check failed: ") + #X); } while (0)

// ---- enums -------------------------------------------------------------------------------------
enum LengthDescriptor { VL_FIXED = 0, VL_VAR, VL_A, VL_G, VL_R, VL_P, VL_PP };
enum ElementType { ET_INT = 0, ET_FLOAT, ET_CHAR, ET_FLAG, ET_INT64 };
enum CombineOp {
  OP_SUM = 0, OP_MEAN, OP_MEDIAN, OP_DP, OP_MOVE_TO_FORMAT, OP_ELEMENT_WISE_SUM, OP_CONCATENATE,
  OP_HISTOGRAM_SUM, OP_UNKNOWN
};
enum KnownField {  // include/vcf/known_field_info.h:30-61
  GVCF_END_IDX = 0, GVCF_REF_IDX, GVCF_ALT_IDX, GVCF_QUAL_IDX, GVCF_FILTER_IDX, GVCF_BASEQRANKSUM_IDX,
  GVCF_CLIPPINGRANKSUM_IDX, GVCF_MQRANKSUM_IDX, GVCF_READPOSRANKSUM_IDX, GVCF_DP_IDX, GVCF_MQ_IDX,
  GVCF_RAW_MQ_IDX, GVCF_MQ0_IDX, GVCF_DP_FORMAT_IDX, GVCF_MIN_DP_IDX, GVCF_GQ_IDX, GVCF_SB_IDX,
  GVCF_AD_IDX, GVCF_PL_IDX, GVCF_AF_IDX, GVCF_AN_IDX, GVCF_AC_IDX, GVCF_GT_IDX, GVCF_PS_IDX,
  GVCF_PGT_IDX, GVCF_PID_IDX, GVCF_EXCESS_HET, GVCF_ID_IDX, GVCF_NUM_KNOWN_FIELDS
};
static const char* const g_known_field_names[GVCF_NUM_KNOWN_FIELDS] = {
    "END", "REF", "ALT", "QUAL", "FILTER", "BaseQRankSum", "ClippingRankSum", "MQRankSum",
    "ReadPosRankSum", "DP", "MQ", "RAW_MQ", "MQ0", "DP_FORMAT", "MIN_DP", "GQ", "SB", "AD", "PL",
    "AF", "AN", "AC", "GT", "PS", "PGT", "PID", "ExcessHet", "ID"};
static const unsigned UNDEFINED_IDX = 0xFFFFFFFFu;

inline unsigned known_field_enum_for_name(const std::string& n) {
  for (unsigned i = 0; i < GVCF_NUM_KNOWN_FIELDS; ++i)
    if (n == g_known_field_names[i]) return i;
  return UNDEFINED_IDX;
}
// known_field_info.cc:239-283
inline void known_field_default_length(unsigned e, LengthDescriptor& ld, unsigned& n) {
  n = 1;
  switch (e) {
    case GVCF_REF_IDX: case GVCF_ALT_IDX: case GVCF_FILTER_IDX: case GVCF_PGT_IDX: case GVCF_PID_IDX:
      ld = VL_VAR; break;
    case GVCF_AF_IDX: case GVCF_AC_IDX: ld = VL_A; break;
    case GVCF_AD_IDX: ld = VL_R; break;
    case GVCF_PL_IDX: ld = VL_G; break;
    case GVCF_GT_IDX: ld = VL_PP; break;
    case GVCF_SB_IDX: ld = VL_FIXED; n = 4; break;
    default: ld = VL_FIXED; n = 1; break;
  }
}
// known_field_info.cc:285-308
inline CombineOp known_field_default_combine_op(unsigned e) {
  switch (e) {
    case GVCF_BASEQRANKSUM_IDX: case GVCF_CLIPPINGRANKSUM_IDX: case GVCF_MQRANKSUM_IDX:
    case GVCF_READPOSRANKSUM_IDX: case GVCF_MQ_IDX: case GVCF_MQ0_IDX: case GVCF_EXCESS_HET:
      return OP_MEDIAN;
    case GVCF_RAW_MQ_IDX: return OP_SUM;
    case GVCF_DP_IDX: return OP_DP;
    default: return OP_UNKNOWN;
  }
}

struct FieldInfo {
  std::string name, vcf_name;
  bool is_INFO = false, is_FORMAT = false, is_FILTER = false;
  int idx = -1;
  LengthDescriptor ld = VL_FIXED;
  unsigned num_elements = 1;  // fixed-length fields
  ElementType et = ET_INT;
  CombineOp combine_op = OP_UNKNOWN;
  // multi-dimensional fields and fields whose elements are tuples (vid_mapper.h:151-280 FieldLengthDescriptor /
  // FieldElementTypeDescriptor): `ld` is the descriptor of dimension 0, the data of a 2-D field is one byte blob per tuple
  // element (genomicsdb_multid_vector_field.h:69-86), every tuple element is a flattened field of its own
  unsigned ndim = 1;
  char vcf_delimiter[2] = {'|', ','};
  std::vector<ElementType> tuple_et;     // element types of the tuple (size 1 for plain fields)
  bool is_flattened = false;
  unsigned element_index_in_tuple = 0;
  int parent_composite_field_idx = -1;
  unsigned num_elements_in_tuple() const { return tuple_et.empty() ? 1u : (unsigned)tuple_et.size(); }
  bool is_fixed() const { return ld == VL_FIXED; }
  bool is_allele_dependent() const { return ld == VL_A || ld == VL_R || ld == VL_G; }
  bool is_genotype_dependent() const { return ld == VL_G; }
  bool is_only_ALT_dependent() const { return ld == VL_A; }
  bool is_ploidy_dependent() const { return ld == VL_P || ld == VL_PP; }
  bool contains_phase_information() const { return ld == VL_PP; }
  unsigned get_ploidy(unsigned n) const {  // known_field_info.h:142-155
    if (ld == VL_P) return n;
    if (ld == VL_PP) return (n + 1u) >> 1u;
    throw OracleException("Unknown length descriptor for GT field");
  }
};

inline uint64_t nCr(uint64_t n, uint64_t r) {  // variant_operations.h nCr
  if (r > n) return 0;
  if (r > n - r) r = n - r;
  uint64_t v = 1;
  for (uint64_t i = 1; i <= r; ++i) v = (v * (n - r + i)) / i;
  return v;
}
// known_field_info.cc:130-143
inline unsigned get_number_of_genotypes(unsigned num_ALT, unsigned ploidy) {
  switch (ploidy) {
    case 1u: return num_ALT + 1u;
    case 2u: return ((num_ALT + 1u) * (num_ALT + 2u)) >> 1u;
    default: return (unsigned)nCr(ploidy + num_ALT, num_ALT);
  }
}
// known_field_info.cc:145-162
inline unsigned num_elements_for_length(const FieldInfo& f, unsigned num_ALT, unsigned ploidy, unsigned n) {
  switch (f.ld) {
    case VL_A: return num_ALT;
    case VL_R: return num_ALT + 1u;
    case VL_G: return get_number_of_genotypes(num_ALT, ploidy);
    case VL_P: case VL_PP: return ploidy;
    default: return n;
  }
}

struct ContigInfo { std::string name; int64_t offset = 0, length = 0; };

class VidMapper {
 public:
  std::vector<FieldInfo> fields;
  std::unordered_map<std::string, int> field_name_to_idx;
  std::vector<ContigInfo> contigs;                      // vid order
  std::vector<std::pair<int64_t, int>> contig_begin_2_idx;  // sorted by offset
  std::vector<std::string> row_to_callset;              // row idx -> name

  const FieldInfo* get_field_info(const std::string& n) const {
    auto it = field_name_to_idx.find(n);
    return it == field_name_to_idx.end() ? nullptr : &fields[it->second];
  }
  // vid_mapper.cc:240-280
  bool get_contig_location(int64_t q, std::string& name, int64_t& pos) const {
    int idx = -1;
    auto it = std::lower_bound(contig_begin_2_idx.begin(), contig_begin_2_idx.end(), std::make_pair(q, 0),
                               [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) { return a.first < b.first; });
    if (it == contig_begin_2_idx.end()) idx = contig_begin_2_idx.back().second;
    else if (it->first == q) idx = it->second;
    else { if (it == contig_begin_2_idx.begin()) return false; idx = (it - 1)->second; }
    if (idx < 0) return false;
    const auto& c = contigs[idx];
    if (q >= c.offset && q < c.offset + c.length) { name = c.name; pos = q - c.offset; return true; }
    return false;
  }
  // vid_mapper.cc:281-304
  bool get_next_contig_location(int64_t q, std::string& name, int64_t& off) const {
    auto it = std::upper_bound(contig_begin_2_idx.begin(), contig_begin_2_idx.end(), std::make_pair(q, 0),
                               [](const std::pair<int64_t, int>& a, const std::pair<int64_t, int>& b) { return a.first < b.first; });
    if (it == contig_begin_2_idx.end()) { name = ""; off = INT64_MAX; return false; }
    name = contigs[it->second].name;
    off = contigs[it->second].offset;
    return true;
  }
  bool get_contig_info(const std::string& n, ContigInfo& out) const {
    for (auto& c : contigs) if (c.name == n) { out = c; return true; }
    return false;
  }

  void load_vid(const oracle_json::Value& doc) {
    ORACLE_VERIFY(doc.HasMember("contigs"));
    const auto& cc = doc["contigs"];
    ORACLE_VERIFY(cc.IsObject() || cc.IsArray());
    size_t nc = cc.Size();
    for (size_t i = 0; i < nc; ++i) {
      const oracle_json::Value& d = cc.IsArray() ? cc[i] : cc.obj[i].second;
      ContigInfo ci;
      if (cc.IsArray()) {
        for (const char* k : {"name", "contig_name", "chromosome_name"})
          if (d.HasMember(k)) ci.name = d[k].GetString();
      } else {
        ci.name = cc.obj[i].first;
      }
      ci.offset = d["tiledb_column_offset"].GetInt64();
      ci.length = d["length"].GetInt64();
      contigs.push_back(ci);
      contig_begin_2_idx.emplace_back(ci.offset, (int)i);
    }
    std::sort(contig_begin_2_idx.begin(), contig_begin_2_idx.end());
    ORACLE_VERIFY(doc.HasMember("fields"));
    const auto& fc = doc["fields"];
    for (size_t i = 0; i < fc.Size(); ++i) {
      const oracle_json::Value& d = fc.IsArray() ? fc[i] : fc.obj[i].second;
      std::string name = fc.IsArray() ? (d.HasMember("name") ? d["name"].GetString() : d["field_name"].GetString())
                                      : fc.obj[i].first;
      if (field_name_to_idx.count(name)) throw OracleException("Duplicate field name " + name);
      FieldInfo f;
      f.name = f.vcf_name = name;
      f.idx = (int)fields.size();
      unsigned ke = known_field_enum_for_name(name);
      if (d.HasMember("vcf_field_class")) {
        const auto& a = d["vcf_field_class"];
        for (size_t j = 0; j < a.Size(); ++j) {
          const std::string& c = a[j].GetString();
          if (c == "INFO") f.is_INFO = true; else if (c == "FORMAT") f.is_FORMAT = true; else if (c == "FILTER") f.is_FILTER = true;
        }
      }
      if (d.HasMember("length")) parse_length(name, d["length"], f);
      else if (ke != UNDEFINED_IDX) known_field_default_length(ke, f.ld, f.num_elements);
      ORACLE_VERIFY(d.HasMember("type"));
      if (d["type"].IsString()) f.tuple_et.assign(1, parse_type(d["type"].GetString()));
      else for (size_t j = 0; j < d["type"].Size(); ++j) f.tuple_et.push_back(parse_type(d["type"][j].GetString()));   // vid_mapper.cc:1456-1482
      ORACLE_VERIFY(!f.tuple_et.empty());
      f.et = f.tuple_et[0];
      if (d.HasMember("vcf_delimiter")) {  // vid_mapper.cc:1412-1428
        const auto& vd = d["vcf_delimiter"];
        if (vd.IsString()) f.vcf_delimiter[0] = vd.GetString()[0];
        else for (size_t j = 0; j < vd.Size() && j < 2; ++j) f.vcf_delimiter[j] = vd[j].GetString()[0];
      }
      if (d.HasMember("VCF_field_combine_operation")) {
        f.combine_op = parse_combine_op(d["VCF_field_combine_operation"].GetString(), name);
        if (f.combine_op == OP_CONCATENATE && f.ld != VL_VAR)
          throw OracleException("'concatenate' needs a VAR length field: " + name);
      } else if (ke != UNDEFINED_IDX) {
        f.combine_op = known_field_default_combine_op(ke);
      }
      field_name_to_idx[name] = f.idx;
      fields.push_back(f);
      // flatten_field: INFO + FORMAT -> extra <name>_FORMAT entry right after (vid_mapper.cc:727-748)
      if (f.is_INFO && f.is_FORMAT) {
        FieldInfo g = f;
        g.name = name + "_FORMAT";
        g.is_INFO = false;
        g.idx = (int)fields.size();
        g.combine_op = OP_UNKNOWN;
        fields[f.idx].is_FORMAT = false;
        field_name_to_idx[g.name] = g.idx;
        fields.push_back(g);
      }
      // every element of a type tuple becomes a field <name>_tuple_element_<i> (vid_mapper.cc:751-787)
      if (f.num_elements_in_tuple() > 1u) {
        const int original = f.idx, format_idx = (f.is_INFO && f.is_FORMAT) ? f.idx + 1 : f.idx;
        for (unsigned j = 0; j < ((f.is_INFO && f.is_FORMAT) ? 2u : 1u); ++j)
          for (unsigned t = 0; t < f.num_elements_in_tuple(); ++t) {
            FieldInfo g = fields[j == 0u ? original : format_idx];
            g.name += "_tuple_element_" + std::to_string(t);
            g.idx = (int)fields.size();
            g.tuple_et.assign(1, f.tuple_et[t]);
            g.et = f.tuple_et[t];
            g.element_index_in_tuple = t;
            g.is_flattened = true;
            g.parent_composite_field_idx = j == 0u ? original : format_idx;
            field_name_to_idx[g.name] = g.idx;
            fields.push_back(g);
          }
      }
    }
    add_mandatory_fields();
  }
  const FieldInfo* get_flattened_field_info(const FieldInfo* fi, unsigned tuple_element_index) const {  // vid_mapper.cc:790-800
    return get_field_info(fi->name + "_tuple_element_" + std::to_string(tuple_element_index));
  }

  void load_callsets(const oracle_json::Value& doc) {
    const oracle_json::Value& cs = doc.HasMember("callsets") ? doc["callsets"] : doc["callset_mapping"]["callsets"];
    for (size_t i = 0; i < cs.Size(); ++i) {
      const oracle_json::Value& d = cs.IsArray() ? cs[i] : cs.obj[i].second;
      std::string name;
      if (cs.IsArray()) {
        for (const char* k : {"sample_name", "name", "callset_name"}) if (d.HasMember(k)) name = d[k].GetString();
      } else {
        name = cs.obj[i].first;
      }
      int64_t row = d["row_idx"].GetInt64();
      if ((size_t)row >= row_to_callset.size()) row_to_callset.resize(row + 1);
      row_to_callset[row] = name;
    }
  }

 private:
  static ElementType parse_type(const std::string& t) {  // vid_mapper.cc:51-89
    if (t == "int" || t == "Int" || t == "integer" || t == "Integer") return ET_INT;
    if (t == "float" || t == "Float") return ET_FLOAT;
    if (t == "bool" || t == "Bool" || t == "boolean" || t == "Boolean" || t == "flag" || t == "Flag") return ET_FLAG;
    if (t == "string" || t == "String" || t == "char" || t == "Char") return ET_CHAR;
    throw OracleException("Unhandled field type " + t);
  }
  static CombineOp parse_combine_op(const std::string& s, const std::string& field) {  // vid_mapper.cc:91-101
    if (s == "sum") return OP_SUM;
    if (s == "mean") return OP_MEAN;
    if (s == "median") return OP_MEDIAN;
    if (s == "move_to_FORMAT") return OP_MOVE_TO_FORMAT;
    if (s == "element_wise_sum" || s == "elementwise_sum") return OP_ELEMENT_WISE_SUM;
    if (s == "concatenate") return OP_CONCATENATE;
    if (s == "histogram_sum") return OP_HISTOGRAM_SUM;
    throw OracleException("Unknown VCF field combine operation " + s + " specified for field " + field);
  }
  static void parse_length(const std::string& name, const oracle_json::Value& v, FieldInfo& f) {
    if (v.IsInt64()) { f.ld = VL_FIXED; f.num_elements = (unsigned)v.GetInt64(); return; }
    if (v.IsString()) {  // vid_mapper.cc:802-830
      std::string up = v.GetString();
      for (auto& c : up) c = (char)toupper(c);
      static const std::map<std::string, LengthDescriptor> tbl = {
          {"BCF_VL_FIXED", VL_FIXED}, {"BCF_VL_A", VL_A}, {"A", VL_A}, {"BCF_VL_R", VL_R}, {"R", VL_R},
          {"BCF_VL_G", VL_G}, {"G", VL_G}, {"BCF_VL_P", VL_P}, {"P", VL_P}, {"BCF_VL_VAR", VL_VAR},
          {"VAR", VL_VAR}, {"PP", VL_PP}, {"PHASED_PLOIDY", VL_PP}};
      auto it = tbl.find(up);
      if (it != tbl.end()) { f.ld = it->second; return; }
      char* endp = nullptr;
      const std::string& raw = v.GetString();
      unsigned long long n = strtoull(raw.c_str(), &endp, 0);
      if (!raw.empty() && (size_t)(endp - raw.c_str()) == raw.size()) { f.ld = VL_FIXED; f.num_elements = (unsigned)n; }
      else f.ld = VL_VAR;
      return;
    }
    if (v.IsObject()) {
      if (v.HasMember("variable_length_descriptor")) { parse_length(name, v["variable_length_descriptor"], f); return; }
      f.ld = VL_FIXED; f.num_elements = (unsigned)v["fixed_length"].GetInt64(); return;
    }
    if (v.IsArray() && v.Size() == 1) { parse_length(name, v[0], f); return; }
    if (v.IsArray() && v.Size() == 2) {  // [ dimension 0, dimension 1 ]: the descriptor of dimension 0 decides the allele dependence
      parse_length(name, v[0], f);
      f.ndim = 2;
      return;
    }
    throw OracleException("field " + name + " has more than 2 dimensions: not supported by the oracle");
  }
  void add_one(const char* n, ElementType et, LengthDescriptor ld, bool info) {
    if (field_name_to_idx.count(n)) return;
    FieldInfo f;
    f.name = f.vcf_name = n; f.idx = (int)fields.size(); f.et = et; f.ld = ld; f.is_INFO = info;
    field_name_to_idx[n] = f.idx;
    fields.push_back(f);
  }
  void add_mandatory_fields() {  // vid_mapper.cc:611-684
    add_one("END", ET_INT, VL_FIXED, true);
    add_one("REF", ET_CHAR, VL_VAR, false);
    add_one("ALT", ET_CHAR, VL_VAR, false);
    add_one("QUAL", ET_FLOAT, VL_FIXED, false);
    add_one("FILTER", ET_INT, VL_VAR, false);
  }
};

// ---- array schema (vid_mapper.cc:354-442) --------------------------------------------------------
struct SchemaAttr { std::string name; ElementType et; bool var; unsigned num; };
struct ArraySchema {
  std::vector<SchemaAttr> attrs;
  int find(const std::string& n) const {
    for (size_t i = 0; i < attrs.size(); ++i) if (attrs[i].name == n) return (int)i;
    return -1;
  }
  static unsigned elem_size(ElementType et) { return et == ET_INT || et == ET_FLOAT ? 4u : (et == ET_INT64 ? 8u : 1u); }
};
inline ArraySchema build_array_schema(const VidMapper& vid) {
  ArraySchema s;
  s.attrs.push_back({"END", ET_INT64, false, 1});
  s.attrs.push_back({"REF", ET_CHAR, true, 0});
  s.attrs.push_back({"ALT", ET_CHAR, true, 0});
  if (vid.field_name_to_idx.count("ID")) s.attrs.push_back({"ID", ET_CHAR, true, 0});
  s.attrs.push_back({"QUAL", ET_FLOAT, false, 1});
  s.attrs.push_back({"FILTER", ET_INT, true, 0});
  // (a composite field - tuple of several elements - is not an attribute, its flattened elements are; the TileDB type of a
  // multi-D field is a variable number of bytes)
  for (const auto& f : vid.fields) {
    if (f.name == "END" || f.num_elements_in_tuple() > 1u) continue;
    if (!f.is_INFO) continue;
    if (f.ndim > 1u) s.attrs.push_back({f.name, ET_CHAR, true, 0u});
    else s.attrs.push_back({f.name, f.et, !f.is_fixed(), f.is_fixed() ? f.num_elements : 0u});
  }
  for (const auto& f : vid.fields) {
    if (f.name == "END" || f.num_elements_in_tuple() > 1u) continue;
    if (!f.is_FORMAT) continue;
    const std::string n = f.is_INFO ? f.name + "_FORMAT" : f.name;
    if (f.ndim > 1u) s.attrs.push_back({n, ET_CHAR, true, 0u});
    else s.attrs.push_back({n, f.et, !f.is_fixed(), f.is_fixed() ? f.num_elements : 0u});
  }
  return s;
}

// ---- query configuration -----------------------------------------------------------------------
struct QueryAttr { std::string name; unsigned schema_idx = UNDEFINED_IDX; const FieldInfo* info = nullptr; };

class QueryConfig {
 public:
  std::vector<QueryAttr> attrs;
  std::unordered_map<std::string, unsigned> name_to_qidx;
  std::vector<std::pair<int64_t, int64_t>> column_intervals;
  bool scan_full = false;
  bool query_all_rows = true;
  std::vector<int64_t> query_rows;
  int64_t num_rows_in_array = 0, smallest_row_idx = 0;
  unsigned first_normal_field_query_idx = 0;
  unsigned known_to_qidx[GVCF_NUM_KNOWN_FIELDS];
  std::vector<unsigned> qidx_to_known;
  bool produce_GT_field = false, produce_FILTER_field = false, sites_only_query = false;
  bool id_union_order_unordered_set = false;   // merge_ID_field without DEBUG (broad_combined_gvcf.cc:732-737); same knob as the product's
  bool produce_GT_with_min_PL_value_for_spanning_deletions = false;
  unsigned max_diploid_alt_alleles_that_can_be_genotyped = 50;
  size_t combined_vcf_records_buffer_size_limit = 1048576u;
  std::string vcf_header_filename, reference_genome;

  QueryConfig() { for (auto& x : known_to_qidx) x = UNDEFINED_IDX; }
  unsigned num_queried_attributes() const { return (unsigned)attrs.size(); }
  bool get_query_idx_for_name(const std::string& n, unsigned& q) const {
    auto it = name_to_qidx.find(n);
    if (it == name_to_qidx.end()) return false;
    q = it->second;
    return true;
  }
  void add_attribute_to_query(const std::string& n, unsigned schema_idx) {  // variant_query_config.cc:37-45
    if (name_to_qidx.find(n) == name_to_qidx.end()) {
      name_to_qidx[n] = (unsigned)attrs.size();
      QueryAttr a; a.name = n; a.schema_idx = schema_idx;
      attrs.push_back(a);
    }
  }
  void clear_attributes() { attrs.clear(); name_to_qidx.clear(); }
  bool is_defined_query_idx_for_known_field_enum(unsigned e) const { return known_to_qidx[e] != UNDEFINED_IDX; }
  unsigned get_query_idx_for_known_field_enum(unsigned e) const { return known_to_qidx[e]; }
  unsigned get_known_field_enum_for_query_idx(unsigned q) const { return qidx_to_known[q]; }
  uint64_t get_num_rows_to_query() const { return query_all_rows ? (uint64_t)num_rows_in_array : query_rows.size(); }
  int64_t get_array_row_idx_for_query_row_idx(uint64_t q) const { return query_all_rows ? (int64_t)q + smallest_row_idx : query_rows[q]; }
  bool is_queried_array_row_idx(int64_t r) const {
    if (query_all_rows) return true;
    return std::binary_search(query_rows.begin(), query_rows.end(), r);
  }
  uint64_t get_query_row_idx_for_array_row_idx(int64_t r) const {
    if (query_all_rows) return (uint64_t)(r - smallest_row_idx);
    return (uint64_t)(std::lower_bound(query_rows.begin(), query_rows.end(), r) - query_rows.begin());
  }
  void reorder_query_fields() {  // variant_query_config.cc:161-185
    const char* special[] = {"END", "REF", "ALT"};
    first_normal_field_query_idx = 0;
    for (const char* sn : special) {
      unsigned q = 0;
      if (get_query_idx_for_name(sn, q)) {
        if (q > first_normal_field_query_idx) {
          std::string other = attrs[first_normal_field_query_idx].name;
          name_to_qidx[sn] = first_normal_field_query_idx;
          name_to_qidx[other] = q;
          std::swap(attrs[q], attrs[first_normal_field_query_idx]);
        }
        ++first_normal_field_query_idx;
      }
    }
  }

  // JSONConfigBase::read_from_file (json_config.cc:195-658), the keys the scan/combine path reads
  void read_query_json(const oracle_json::Value& j, const VidMapper& vid, int rank = 0) {
    if (j.HasMember("scan_full")) {
      scan_full = true;
    } else if (j.HasMember("query_column_ranges")) {
      const auto& q1 = j["query_column_ranges"];
      ORACLE_VERIFY(q1.IsArray());
      size_t idx = q1.Size() == 1 ? 0 : (size_t)rank;
      ORACLE_VERIFY(idx < q1.Size());
      const auto& e = q1[idx];
      const oracle_json::Value* q2 = &e;
      if (e.IsObject()) {
        if (e.MemberCount() == 0) q2 = nullptr;
        else q2 = e.HasMember("range_list") ? &e["range_list"] : &e["column_or_interval_list"];
      }
      if (q2) for (size_t k = 0; k < q2->Size(); ++k) {
        const auto& q3 = (*q2)[k];
        int64_t a = 0, b = 0;
        if (q3.IsArray()) { ORACLE_VERIFY(q3.Size() == 2); a = q3[0].GetInt64(); b = q3[1].GetInt64(); }
        else if (q3.IsInt64()) { a = b = q3.GetInt64(); }
        else if (q3.IsString()) {
          ContigInfo ci; if (!vid.get_contig_info(q3.GetString(), ci)) throw OracleException("Invalid contig name");
          a = ci.offset; b = ci.offset + ci.length - 1;
        } else if (q3.IsObject() && q3.HasMember("low") && q3.HasMember("high")) { a = q3["low"].GetInt64(); b = q3["high"].GetInt64(); }
        else if (q3.IsObject() && q3.HasMember("column_interval")) {
          const auto& io = q3["column_interval"];
          if (io.HasMember("column_interval")) { a = io["column_interval"]["begin"].GetInt64(); b = io["column_interval"]["end"].GetInt64(); }
          else {
            const auto& ci_ = io["contig_interval"]; ContigInfo ci;
            if (!vid.get_contig_info(ci_["contig"].GetString(), ci)) throw OracleException("Invalid contig name");
            a = ci.offset + ci_["begin"].GetInt64() - 1; b = ci.offset + ci_["end"].GetInt64() - 1;
          }
        } else if (q3.IsObject() && q3.HasMember("column")) {
          const auto& io = q3["column"];
          if (io.HasMember("tiledb_column")) a = b = io["tiledb_column"].GetInt64();
          else {
            const auto& cp = io["contig_position"]; ContigInfo ci;
            if (!vid.get_contig_info(cp["contig"].GetString(), ci)) throw OracleException("Invalid contig name");
            a = b = ci.offset + cp["position"].GetInt64() - 1;
          }
        } else if (q3.IsObject() && q3.MemberCount() == 1) {  // { "chr" : [b, e] } or { "chr" : p }, 1-based
          ContigInfo ci; if (!vid.get_contig_info(q3.obj[0].first, ci)) throw OracleException("Invalid contig name");
          const auto& p = q3.obj[0].second;
          if (p.IsArray()) { a = ci.offset + p[0].GetInt64() - 1; b = ci.offset + p[1].GetInt64() - 1; }
          else { a = b = ci.offset + p.GetInt64() - 1; }
        } else throw OracleException("unsupported query_column_ranges entry");
        if (a > b) std::swap(a, b);
        column_intervals.emplace_back(a, b);
      }
      std::stable_sort(column_intervals.begin(), column_intervals.end(),
                       [](const std::pair<int64_t, int64_t>& x, const std::pair<int64_t, int64_t>& y) { return x.first < y.first; });
    }
    if (!scan_full && j.HasMember("query_row_ranges")) {
      const auto& q1 = j["query_row_ranges"];
      size_t idx = q1.Size() == 1 ? 0 : (size_t)rank;
      const auto& e = q1[idx];
      const oracle_json::Value& q2 = e.IsArray() ? e : e["range_list"];
      std::vector<int64_t> rows;
      for (size_t k = 0; k < q2.Size(); ++k) {
        const auto& q3 = q2[k];
        int64_t a, b;
        if (q3.IsArray()) { a = q3[0].GetInt64(); b = q3[1].GetInt64(); }
        else if (q3.IsInt64()) { a = b = q3.GetInt64(); }
        else { a = q3["low"].GetInt64(); b = q3["high"].GetInt64(); }
        if (a > b) std::swap(a, b);
        for (int64_t r = a; r <= b; ++r) rows.push_back(r);
      }
      std::sort(rows.begin(), rows.end());
      query_rows = rows;
      query_all_rows = false;
    }
    const char* ak = j.HasMember("query_attributes") ? "query_attributes" : (j.HasMember("attributes") ? "attributes" : nullptr);
    if (ak) for (size_t i = 0; i < j[ak].Size(); ++i) add_attribute_to_query(j[ak][i].GetString(), UNDEFINED_IDX);
    auto str_or_rank = [&](const char* k, std::string& out) {
      if (!j.HasMember(k)) return;
      const auto& v = j[k];
      out = v.IsArray() ? v[(size_t)rank].GetString() : v.GetString();
    };
    str_or_rank("vcf_header_filename", vcf_header_filename);
    str_or_rank("reference_genome", reference_genome);
    if (j.HasMember("max_diploid_alt_alleles_that_can_be_genotyped"))
      max_diploid_alt_alleles_that_can_be_genotyped = (unsigned)j["max_diploid_alt_alleles_that_can_be_genotyped"].GetInt64();
    if (j.HasMember("combined_vcf_records_buffer_size_limit"))
      combined_vcf_records_buffer_size_limit = std::max<size_t>(1, (size_t)j["combined_vcf_records_buffer_size_limit"].GetInt64());
    auto flag = [&](const char* k) { return j.HasMember(k) && j[k].GetBool(); };
    produce_GT_field = flag("produce_GT_field");
    produce_FILTER_field = flag("produce_FILTER_field");
    {
      std::string order;
      if (j.HasMember("id_union_order")) order = j["id_union_order"].GetString();
      id_union_order_unordered_set = order == "unordered_set";
    }
    sites_only_query = flag("sites_only_query");
    produce_GT_with_min_PL_value_for_spanning_deletions = flag("produce_GT_with_min_PL_value_for_spanning_deletions");
  }

  // VariantQueryProcessor::do_query_bookkeeping (query_variants.cc:578-685) with
  // finalize_queried_attributes (:243-279) and obtain_TileDB_attribute_idxs (:281-294)
  // VariantQueryConfig::flatten_composite_fields (variant_query_config.cc:187-229): the elements of a queried composite field
  // join the end of the list, the composite itself leaves it
  void flatten_composite_fields(const VidMapper& vid) {
    std::vector<std::string> names;
    for (auto& a : attrs) names.push_back(a.name);
    std::vector<std::string> keep, extra;
    for (auto& n : names) {
      const FieldInfo* fi = vid.get_field_info(n);
      if (!fi) throw OracleException("Field " + n + " not found in vid mapping");
      if (fi->num_elements_in_tuple() > 1u) { for (unsigned j = 0; j < fi->num_elements_in_tuple(); ++j) extra.push_back(vid.get_flattened_field_info(fi, j)->name); }
      else keep.push_back(n);
    }
    if (extra.empty()) return;
    clear_attributes();
    for (auto& n : keep) add_attribute_to_query(n, UNDEFINED_IDX);
    for (auto& n : extra) add_attribute_to_query(n, UNDEFINED_IDX);
  }
  void do_query_bookkeeping(const ArraySchema& schema, const VidMapper& vid, int64_t num_rows, int64_t lb_row) {
    flatten_composite_fields(vid);
    if (attrs.empty() || sites_only_query) {
      std::vector<std::string> names;
      if (attrs.empty()) for (auto& a : schema.attrs) names.push_back(a.name);
      else for (auto& a : attrs) names.push_back(a.name);
      std::vector<bool> valid(names.size(), true);
      if (sites_only_query)
        for (size_t i = 0; i < names.size(); ++i) {
          const FieldInfo* fi = vid.get_field_info(names[i]);
          if (fi && fi->is_FORMAT && names[i] != "DP_FORMAT" && names[i] != "MIN_DP") valid[i] = false;
        }
      clear_attributes();
      for (size_t i = 0; i < names.size(); ++i) if (valid[i]) add_attribute_to_query(names[i], 0u);
    }
    for (auto& a : attrs) {
      int si = schema.find(a.name);
      if (si < 0) throw OracleException("Invalid query attribute : " + a.name);
      a.schema_idx = (unsigned)si;
    }
    add_attribute_to_query("END", (unsigned)schema.find("END"));
    bool added_ALT_REF = true, added_GT = false;  // alleles_required == true on this path
    add_attribute_to_query("ALT", (unsigned)schema.find("ALT"));
    add_attribute_to_query("REF", (unsigned)schema.find("REF"));
    for (unsigned i = 0; i < num_queried_attributes(); ++i) {
      const FieldInfo* fi = vid.get_field_info(attrs[i].name);
      if (!fi) throw OracleException("No vid info for attribute " + attrs[i].name);
      attrs[i].info = fi;
      if (!added_GT && fi->is_genotype_dependent()) {
        add_attribute_to_query("GT", (unsigned)schema.find("GT"));
        added_GT = true;
      }
    }
    (void)added_ALT_REF;
    reorder_query_fields();
    for (auto& a : attrs) a.info = vid.get_field_info(a.name);  // re-resolve after the swaps
    qidx_to_known.assign(attrs.size(), UNDEFINED_IDX);
    for (auto& x : known_to_qidx) x = UNDEFINED_IDX;
    for (unsigned i = 0; i < attrs.size(); ++i) {
      unsigned e = known_field_enum_for_name(attrs[i].name);
      if (e != UNDEFINED_IDX) { known_to_qidx[e] = i; qidx_to_known[i] = e; }
    }
    num_rows_in_array = num_rows;
    smallest_row_idx = lb_row;
    if (!query_all_rows) {  // setup_array_row_idx_to_query_row_idx_map: drop out-of-bounds rows
      std::vector<int64_t> keep;
      for (auto r : query_rows) if (r >= smallest_row_idx && r < smallest_row_idx + num_rows_in_array) keep.push_back(r);
      query_rows = keep;
    }
  }
};

}  // namespace gdb_oracle
