#include "vid_mapper.h"

#include <algorithm>
#include <cctype>

namespace genomicsdb_amd {

static const char* const kKnownNames[GVCF_NUM_KNOWN_FIELDS] = {
    "END", "REF", "ALT", "QUAL", "FILTER", "BaseQRankSum", "ClippingRankSum", "MQRankSum", "ReadPosRankSum", "DP",
    "MQ", "RAW_MQ", "MQ0", "DP_FORMAT", "MIN_DP", "GQ", "SB", "AD", "PL", "AF", "AN", "AC", "GT", "PS", "PGT", "PID",
    "ExcessHet", "ID"};

int known_field_enum_for_name(const std::string& name) {
  for (int i = 0; i < GVCF_NUM_KNOWN_FIELDS; ++i)
    if (name == kKnownNames[i]) return i;
  return -1;
}

namespace {

// defaults for known fields without "length" / "VCF_field_combine_operation" in the vid (reference known_field_info.cc:239-308)
void default_length(int ke, GdbLength& ld, unsigned& n) {
  n = 1;
  switch (ke) {
    case GVCF_REF_IDX: case GVCF_ALT_IDX: case GVCF_FILTER_IDX: case GVCF_PGT_IDX: case GVCF_PID_IDX: ld = GDB_VL_VAR; break;
    case GVCF_AF_IDX: case GVCF_AC_IDX: ld = GDB_VL_A; break;
    case GVCF_AD_IDX: ld = GDB_VL_R; break;
    case GVCF_PL_IDX: ld = GDB_VL_G; break;
    case GVCF_GT_IDX: ld = GDB_VL_PP; break;
    case GVCF_SB_IDX: ld = GDB_VL_FIXED; n = 4; break;
    default: ld = GDB_VL_FIXED; break;
  }
}
GdbCombineOp default_combine_op(int ke) {
  switch (ke) {
    case GVCF_BASEQRANKSUM_IDX: case GVCF_CLIPPINGRANKSUM_IDX: case GVCF_MQRANKSUM_IDX: case GVCF_READPOSRANKSUM_IDX:
    case GVCF_MQ_IDX: case GVCF_MQ0_IDX: case GVCF_EXCESS_HET: return GDB_OP_MEDIAN;
    case GVCF_RAW_MQ_IDX: return GDB_OP_SUM;
    case GVCF_DP_IDX: return GDB_OP_DP;
    default: return GDB_OP_UNKNOWN;
  }
}
GdbElem parse_type_name(const std::string& t) {
  std::string l = t;
  for (auto& c : l) c = (char)tolower(c);
  if (l == "int" || l == "integer") return GDB_ET_INT;
  if (l == "float") return GDB_ET_FLOAT;
  if (l == "bool" || l == "boolean" || l == "flag") return GDB_ET_FLAG;
  if (l == "string" || l == "char") return GDB_ET_CHAR;
  throw VidMapperException("Unhandled field type " + t);
}
GdbCombineOp parse_combine_op(const std::string& s, const std::string& field) {
  if (s == "sum") return GDB_OP_SUM;
  if (s == "mean") return GDB_OP_MEAN;
  if (s == "median") return GDB_OP_MEDIAN;
  if (s == "move_to_FORMAT") return GDB_OP_MOVE_TO_FORMAT;
  if (s == "element_wise_sum" || s == "elementwise_sum") return GDB_OP_ELEMENT_WISE_SUM;
  if (s == "concatenate") return GDB_OP_CONCATENATE;
  if (s == "histogram_sum") return GDB_OP_HISTOGRAM_SUM;
  throw VidMapperException("Unknown VCF field combine operation " + s + " specified for field " + field);
}
// "length": int | "A"/"R"/"G"/"P"/"PP"/"VAR"/"BCF_VL_*" | numeric string | {variable_length_descriptor|fixed_length} | [ ... ]
bool parse_length(const mini_json::Value& v, GdbLength& ld, unsigned& n) {
  if (v.IsInt64()) { ld = GDB_VL_FIXED; n = (unsigned)v.GetInt64(); return true; }
  if (v.IsString()) {
    std::string up = v.GetString();
    for (auto& c : up) c = (char)toupper(c);
    if (up.rfind("BCF_VL_", 0) == 0) up = up.substr(7);
    if (up == "FIXED") { ld = GDB_VL_FIXED; return true; }
    if (up == "A") { ld = GDB_VL_A; return true; }
    if (up == "R") { ld = GDB_VL_R; return true; }
    if (up == "G") { ld = GDB_VL_G; return true; }
    if (up == "P") { ld = GDB_VL_P; return true; }
    if (up == "PP" || up == "PHASED_PLOIDY") { ld = GDB_VL_PP; return true; }
    if (up == "VAR") { ld = GDB_VL_VAR; return true; }
    char* e = nullptr;
    unsigned long long x = strtoull(v.GetString().c_str(), &e, 0);
    if (!v.GetString().empty() && *e == 0) { ld = GDB_VL_FIXED; n = (unsigned)x; }
    else ld = GDB_VL_VAR;  // "unknown length descriptor ... setting to 'VAR'"
    return true;
  }
  if (v.IsObject()) {
    if (v.HasMember("variable_length_descriptor")) return parse_length(v["variable_length_descriptor"], ld, n);
    ld = GDB_VL_FIXED; n = (unsigned)v["fixed_length"].GetInt64();
    return true;
  }
  if (v.IsArray() && v.Size() == 1) return parse_length(v[0], ld, n);
  return false;  // multi-dimensional: the caller handles 2 dimensions
}

}  // namespace

const FieldInfo* VidMapper::get_field_info(const std::string& name) const {
  auto it = m_field_name_to_idx.find(name);
  return it == m_field_name_to_idx.end() ? nullptr : &m_field_idx_to_info[it->second];
}

const FieldInfo* VidMapper::get_flattened_field_info(const FieldInfo* field_info, unsigned tuple_element_index) const {
  return get_field_info(field_info->m_name + "_tuple_element_" + std::to_string(tuple_element_index));
}

bool VidMapper::get_contig_info(const std::string& name, ContigInfo& out) const {
  for (auto& c : m_contig_idx_to_info) if (c.m_name == name) { out = c; return true; }
  return false;
}

bool VidMapper::get_contig_location(int64_t position, std::string& contig_name, int64_t& contig_position) const {
  // last contig whose offset <= position, then a range check (reference vid_mapper.cc:240-280)
  auto it = std::upper_bound(m_contig_begin_2_idx.begin(), m_contig_begin_2_idx.end(), position,
                             [](int64_t p, const std::pair<int64_t, int>& e) { return p < e.first; });
  if (it == m_contig_begin_2_idx.begin()) return false;
  const ContigInfo& c = m_contig_idx_to_info[(it - 1)->second];
  if (position < c.m_tiledb_column_offset + c.m_length) {
    contig_name = c.m_name;
    contig_position = position - c.m_tiledb_column_offset;
    return true;
  }
  return false;
}

bool VidMapper::get_callset_name(int64_t row_idx, std::string& name) const {
  if (row_idx < 0 || (size_t)row_idx >= m_row_idx_to_name.size()) return false;
  name = m_row_idx_to_name[(size_t)row_idx];
  return true;
}

void VidMapper::parse_vid_json(const mini_json::Value& doc) {
  if (!doc.HasMember("contigs") || !doc.HasMember("fields")) throw VidMapperException("vid mapping needs \"contigs\" and \"fields\"");
  const auto& cc = doc["contigs"];
  for (size_t i = 0; i < cc.Size(); ++i) {
    const mini_json::Value& d = cc.IsArray() ? cc[i] : cc.obj[i].second;
    ContigInfo ci;
    if (cc.IsArray()) {
      int found = 0;
      for (const char* k : {"name", "contig_name", "chromosome_name"}) if (d.HasMember(k)) { ci.m_name = d[k].GetString(); ++found; }
      if (found != 1) throw VidMapperException("Contig info dict needs exactly one of \"name\", \"contig_name\", \"chromosome_name\"");
    } else ci.m_name = cc.obj[i].first;
    for (auto& e : m_contig_idx_to_info) if (e.m_name == ci.m_name) throw VidMapperException("Duplicate contig/chromosome name " + ci.m_name);
    ci.m_tiledb_column_offset = d["tiledb_column_offset"].GetInt64();
    ci.m_length = d["length"].GetInt64();
    m_contig_begin_2_idx.emplace_back(ci.m_tiledb_column_offset, (int)m_contig_idx_to_info.size());
    m_contig_idx_to_info.push_back(ci);
  }
  std::sort(m_contig_begin_2_idx.begin(), m_contig_begin_2_idx.end());
  for (size_t i = 1; i < m_contig_begin_2_idx.size(); ++i) {
    const auto& prev = m_contig_idx_to_info[m_contig_begin_2_idx[i - 1].second];
    if (m_contig_begin_2_idx[i].first <= prev.m_tiledb_column_offset + prev.m_length - 1)
      throw VidMapperException("Overlapping contigs exist in vid file");
  }
  const auto& fc = doc["fields"];
  for (size_t i = 0; i < fc.Size(); ++i) {
    const mini_json::Value& d = fc.IsArray() ? fc[i] : fc.obj[i].second;
    std::string name;
    if (fc.IsArray()) {
      if (d.HasMember("name") == d.HasMember("field_name")) throw VidMapperException("Field dict needs exactly one of \"name\" / \"field_name\"");
      name = d.HasMember("name") ? d["name"].GetString() : d["field_name"].GetString();
    } else name = fc.obj[i].first;
    if (m_field_name_to_idx.count(name)) throw VidMapperException("Duplicate field name " + name);
    FieldInfo f;
    f.m_name = f.m_vcf_name = name;
    f.m_field_idx = (int)m_field_idx_to_info.size();
    int ke = known_field_enum_for_name(name);
    if (d.HasMember("vcf_field_class"))
      for (size_t j = 0; j < d["vcf_field_class"].Size(); ++j) {
        const std::string& c = d["vcf_field_class"][j].GetString();
        if (c == "INFO") f.m_is_vcf_INFO_field = true;
        else if (c == "FORMAT") f.m_is_vcf_FORMAT_field = true;
        else if (c == "FILTER") f.m_is_vcf_FILTER_field = true;
      }
    if (d.HasMember("length")) {
      if (!parse_length(d["length"], f.m_length_descriptor, f.m_num_elements)) {
        const auto& la = d["length"];
        if (la.IsArray() && la.Size() == 2 && parse_length(la[0], f.m_length_descriptor, f.m_num_elements)) f.m_num_dimensions = 2;
        else f.m_unsupported_on_device = true;
      }
    } else if (ke >= 0) default_length(ke, f.m_length_descriptor, f.m_num_elements);
    if (!d.HasMember("type")) throw VidMapperException("Field " + name + " has no \"type\"");
    if (d["type"].IsString()) f.m_tuple_element_types.assign(1, parse_type_name(d["type"].GetString()));
    else for (size_t j = 0; j < d["type"].Size(); ++j) f.m_tuple_element_types.push_back(parse_type_name(d["type"][j].GetString()));
    if (f.m_tuple_element_types.empty()) throw VidMapperException("Field " + name + " has an empty \"type\" list");
    f.m_element_type = f.m_tuple_element_types[0];
    if (d.HasMember("vcf_delimiter")) {
      const auto& vd = d["vcf_delimiter"];
      if (vd.IsString()) { if (!vd.GetString().empty()) f.m_vcf_delimiter[0] = vd.GetString()[0]; }
      else for (size_t j = 0; j < vd.Size() && j < 2; ++j) if (!vd[j].GetString().empty()) f.m_vcf_delimiter[j] = vd[j].GetString()[0];
    }
    if (d.HasMember("VCF_field_combine_operation")) {
      f.m_VCF_field_combine_operation = parse_combine_op(d["VCF_field_combine_operation"].GetString(), name);
      if (f.m_VCF_field_combine_operation == GDB_OP_CONCATENATE && f.m_length_descriptor != GDB_VL_VAR)
        throw VidMapperException("VCF field combined operation 'concatenate' can only be used with 'VAR' length fields; field " + name);
    } else if (ke >= 0) f.m_VCF_field_combine_operation = default_combine_op(ke);
    m_field_name_to_idx[name] = f.m_field_idx;
    m_field_idx_to_info.push_back(f);
    if (f.m_is_vcf_INFO_field && f.m_is_vcf_FORMAT_field) {  // INFO+FORMAT: second entry <name>_FORMAT (vid_mapper.cc:727-748)
      FieldInfo g = f;
      g.m_name = name + "_FORMAT";
      g.m_is_vcf_INFO_field = false;
      g.m_field_idx = (int)m_field_idx_to_info.size();
      g.m_VCF_field_combine_operation = GDB_OP_UNKNOWN;
      m_field_idx_to_info[f.m_field_idx].m_is_vcf_FORMAT_field = false;
      m_field_name_to_idx[g.m_name] = g.m_field_idx;
      m_field_idx_to_info.push_back(g);
    }
    if (f.get_num_elements_in_tuple() > 1u) {  // the elements of the tuple as fields of their own (vid_mapper.cc:751-787)
      const bool both = f.m_is_vcf_INFO_field && f.m_is_vcf_FORMAT_field;
      const int original = f.m_field_idx, format_idx = both ? f.m_field_idx + 1 : f.m_field_idx;
      for (unsigned j = 0; j < (both ? 2u : 1u); ++j)
        for (unsigned t = 0; t < f.get_num_elements_in_tuple(); ++t) {
          FieldInfo g = m_field_idx_to_info[(size_t)(j == 0u ? original : format_idx)];
          g.m_name += "_tuple_element_" + std::to_string(t);
          g.m_field_idx = (int)m_field_idx_to_info.size();
          g.m_tuple_element_types.assign(1, f.m_tuple_element_types[t]);
          g.m_element_type = f.m_tuple_element_types[t];
          g.m_element_index_in_tuple = t;
          g.m_is_flattened_field = true;
          g.m_parent_composite_field_idx = j == 0u ? original : format_idx;
          m_field_name_to_idx[g.m_name] = g.m_field_idx;
          m_field_idx_to_info.push_back(g);
        }
    }
  }
  add_mandatory_fields();
  m_is_initialized = true;
}

void VidMapper::add_mandatory_fields() {  // reference vid_mapper.cc:611-684
  struct M { const char* n; GdbElem t; GdbLength l; bool info; };
  const M m[] = {{"END", GDB_ET_INT, GDB_VL_FIXED, true}, {"REF", GDB_ET_CHAR, GDB_VL_VAR, false}, {"ALT", GDB_ET_CHAR, GDB_VL_VAR, false},
                 {"QUAL", GDB_ET_FLOAT, GDB_VL_FIXED, false}, {"FILTER", GDB_ET_INT, GDB_VL_VAR, false}};
  for (const auto& x : m) {
    if (m_field_name_to_idx.count(x.n)) continue;
    FieldInfo f;
    f.m_name = f.m_vcf_name = x.n;
    f.m_field_idx = (int)m_field_idx_to_info.size();
    f.m_element_type = x.t;
    f.m_length_descriptor = x.l;
    f.m_is_vcf_INFO_field = x.info;
    m_field_name_to_idx[x.n] = f.m_field_idx;
    m_field_idx_to_info.push_back(f);
  }
}

void VidMapper::parse_callsets_json(const mini_json::Value& doc) {
  const mini_json::Value* cs = nullptr;
  if (doc.HasMember("callsets")) cs = &doc["callsets"];
  else if (doc.HasMember("callset_mapping") && doc["callset_mapping"].HasMember("callsets")) cs = &doc["callset_mapping"]["callsets"];
  if (!cs) throw VidMapperException("callset mapping needs \"callsets\"");
  for (size_t i = 0; i < cs->Size(); ++i) {
    const mini_json::Value& d = cs->IsArray() ? (*cs)[i] : cs->obj[i].second;
    std::string name;
    if (cs->IsArray()) { for (const char* k : {"sample_name", "name", "callset_name"}) if (d.HasMember(k)) name = d[k].GetString(); }
    else name = cs->obj[i].first;
    int64_t row = d["row_idx"].GetInt64();
    if (row < 0) throw VidMapperException("negative row_idx for callset " + name);
    if ((size_t)row >= m_row_idx_to_name.size()) m_row_idx_to_name.resize((size_t)row + 1);
    m_row_idx_to_name[(size_t)row] = name;
    CallSetInfo ci;
    ci.m_name = name; ci.m_row_idx = row;
    if (d.HasMember("idx_in_file")) ci.m_idx_in_file = d["idx_in_file"].GetInt64();
    if (d.HasMember("filename")) ci.m_filename = d["filename"].GetString();
    m_callsets.push_back(ci);
  }
  m_is_callset_mapping_initialized = true;
}

std::vector<std::string> VidMapper::schema_attribute_names() const {
  std::vector<std::string> a = {"END", "REF", "ALT"};
  if (m_field_name_to_idx.count("ID")) a.push_back("ID");
  a.push_back("QUAL");
  a.push_back("FILTER");
  // (a composite field is not an attribute, its flattened tuple elements are: vid_mapper.cc:399-404)
  for (auto& f : m_field_idx_to_info) if (f.m_name != "END" && f.get_num_elements_in_tuple() == 1u && f.m_is_vcf_INFO_field) a.push_back(f.m_name);
  for (auto& f : m_field_idx_to_info)
    if (f.m_name != "END" && f.get_num_elements_in_tuple() == 1u && f.m_is_vcf_FORMAT_field) a.push_back(f.m_is_vcf_INFO_field ? f.m_name + "_FORMAT" : f.m_name);
  return a;
}

}  // namespace genomicsdb_amd
