#include "variant_query_config.h"
#include <cstdlib>
#include <cstdio>
#include <set>
#include <mutex>

#include <algorithm>
#include <climits>

namespace genomicsdb_amd {

namespace {

std::string join_path(const std::string& base, const std::string& p) {
  if (p.empty() || p[0] == '/' || base.empty()) return p;
  return base + "/" + p;
}
std::string dir_of(const std::string& f) {
  size_t s = f.find_last_of('/');
  return s == std::string::npos ? std::string() : f.substr(0, s);
}
const mini_json::Value& pick_rank(const mini_json::Value& v, int rank) {
  if (v.IsArray()) {
    if ((size_t)rank >= v.Size()) throw GenomicsDBConfigException("rank >= array size in configuration");
    return v[(size_t)rank];
  }
  return v;
}
// { "chr" : [b, e] } / { "chr" : p } (1-based) or the protobuf-generated forms (json_config.cc:54-193)
ColumnRange interval_from_object(const mini_json::Value& o, const VidMapper& vid) {
  auto contig_iv = [&](const std::string& name, int64_t b, int64_t e) {
    ContigInfo ci;
    if (!vid.get_contig_info(name, ci)) throw VidMapperException("Invalid contig name : " + name);
    if (b > ci.m_length || e > ci.m_length) throw GenomicsDBConfigException("Position in contig " + name + " is beyond its length");
    return ColumnRange(ci.m_tiledb_column_offset + b - 1, ci.m_tiledb_column_offset + e - 1);
  };
  if (o.MemberCount() == 2 && o.HasMember("low") && o.HasMember("high")) return ColumnRange(o["low"].GetInt64(), o["high"].GetInt64());
  if (o.MemberCount() == 1 && o.HasMember("column_interval")) {
    const auto& io = o["column_interval"];
    if (io.HasMember("column_interval")) return ColumnRange(io["column_interval"]["begin"].GetInt64(), io["column_interval"]["end"].GetInt64());
    const auto& c = io["contig_interval"];
    return contig_iv(c["contig"].GetString(), c["begin"].GetInt64(), c["end"].GetInt64());
  }
  if (o.MemberCount() == 1 && o.HasMember("column")) {
    const auto& io = o["column"];
    if (io.HasMember("tiledb_column")) return ColumnRange(io["tiledb_column"].GetInt64(), io["tiledb_column"].GetInt64());
    const auto& c = io["contig_position"];
    return contig_iv(c["contig"].GetString(), c["position"].GetInt64(), c["position"].GetInt64());
  }
  if (o.MemberCount() != 1) throw GenomicsDBConfigException("Malformed column interval object");
  std::string name = o.obj[0].first;
  const mini_json::Value* pos = &o.obj[0].second;
  if (name == "contig_position") { name = (*pos)["contig"].GetString(); pos = &(*pos)["position"]; }
  if (pos->IsArray()) return contig_iv(name, (*pos)[0].GetInt64(), (*pos)[1].GetInt64());
  return contig_iv(name, pos->GetInt64(), pos->GetInt64());
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
void GenomicsDBImportConfig::read_from_file(const std::string& filename, int rank) {
  read_from_json(mini_json::parse_file(filename), rank);
}

void GenomicsDBImportConfig::read_from_json(const mini_json::Value& j, int rank) {
  if (j.HasMember("vid_mapping_file")) m_vid_mapping_file = pick_rank(j["vid_mapping_file"], rank).GetString();
  if (j.HasMember("callset_mapping_file")) m_callset_mapping_file = pick_rank(j["callset_mapping_file"], rank).GetString();
  if (j.HasMember("vcf_header_filename")) m_vcf_header_filename = pick_rank(j["vcf_header_filename"], rank).GetString();
  if (j.HasMember("reference_genome")) m_reference_genome = pick_rank(j["reference_genome"], rank).GetString();
  m_treat_deletions_as_intervals = j.HasMember("treat_deletions_as_intervals") && j["treat_deletions_as_intervals"].GetBool();
  m_produce_combined_vcf = j.HasMember("produce_combined_vcf") && j["produce_combined_vcf"].GetBool();
  if (j.HasMember("workspace")) { const auto& w = j["workspace"]; if (w.IsArray()) for (size_t i = 0; i < w.Size(); ++i) m_workspaces.push_back(w[i].GetString()); else m_workspaces.push_back(w.GetString()); }
  const char* ak = j.HasMember("array") ? "array" : (j.HasMember("array_name") ? "array_name" : nullptr);
  if (ak) { const auto& a = j[ak]; if (a.IsArray()) for (size_t i = 0; i < a.Size(); ++i) m_array_names.push_back(a[i].GetString()); else m_array_names.push_back(a.GetString()); }
  if (j.HasMember("column_partitions")) {
    const auto& cp = j["column_partitions"];
    std::string ws = m_workspaces.size() == 1 ? m_workspaces[0] : "", an = m_array_names.size() == 1 ? m_array_names[0] : "";
    bool per_part_ws = false, per_part_an = false;
    std::vector<std::string> wsv(cp.Size(), ws), anv(cp.Size(), an);
    // a TileDB column, or a contig position { "chr1" : 5 } / { "chr1" : [5, 6] } (1-based; json_config.cc:359-375), for which
    // the vid mapping the loader JSON names is read (once)
    VidMapper vid;
    auto contig_column = [&](const mini_json::Value& o) {
      if (!o.IsObject()) throw GenomicsDBConfigException("column partition bound must be a TileDB column or { contig : position }");
      if (!vid.is_initialized()) {
        if (!m_vid_mapping_file.empty()) vid.parse_vid_json(mini_json::parse_file(m_vid_mapping_file));
        else if (j.HasMember("vid_mapping")) vid.parse_vid_json(j["vid_mapping"]);
        else throw GenomicsDBConfigException("contig-style column partition bounds need \"vid_mapping_file\" (or \"vid_mapping\") in the loader JSON");
      }
      return interval_from_object(o, vid).first;
    };
    for (size_t i = 0; i < cp.Size(); ++i) {
      const auto& d = cp[i];
      ColumnRange r(0, INT64_MAX - 1);
      if (!d.HasMember("begin")) throw GenomicsDBConfigException("column partition without \"begin\"");
      if (d["begin"].IsInt64()) r.first = d["begin"].GetInt64();
      else r.first = contig_column(d["begin"]);
      if (d.HasMember("end")) r.second = d["end"].IsInt64() ? d["end"].GetInt64() : contig_column(d["end"]);
      if (r.first > r.second) std::swap(r.first, r.second);
      m_column_partitions.push_back(r);
      if (d.HasMember("workspace")) { wsv[i] = d["workspace"].GetString(); per_part_ws = true; }
      if (d.HasMember("array") || d.HasMember("array_name")) { anv[i] = d.HasMember("array") ? d["array"].GetString() : d["array_name"].GetString(); per_part_an = true; }
    }
    if (per_part_ws) m_workspaces = wsv;
    if (per_part_an) m_array_names = anv;
    // ends are derived from the sorted begins (json_config.cc:405-416)
    std::vector<size_t> order(m_column_partitions.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return m_column_partitions[a].first < m_column_partitions[b].first; });
    for (size_t i = 0; i + 1 < order.size(); ++i) {
      auto& cur = m_column_partitions[order[i]];
      const auto& nxt = m_column_partitions[order[i + 1]];
      if (cur.first == nxt.first) throw GenomicsDBConfigException("Cannot have two column partitions with the same begin value");
      if (cur.second >= nxt.first) cur.second = nxt.first - 1;
    }
    m_sorted_column_partitions = m_column_partitions;
    std::sort(m_sorted_column_partitions.begin(), m_sorted_column_partitions.end());
  }
  m_loaded = true;
}

ColumnRange GenomicsDBImportConfig::get_column_partition(int rank) const {
  if (m_column_partitions.empty()) return ColumnRange(0, INT64_MAX - 1);
  return m_column_partitions.at((size_t)rank);
}

// ---------------------------------------------------------------------------------------------------
void VariantQueryConfig::set_vcf_output_format(const std::string& f) {
  if (f != "" && f != "z" && f != "b" && f != "bu") throw GenomicsDBConfigException("Unknown VCF output format " + f);
  m_vcf_output_format = f;
}

void VariantQueryConfig::read_from_file(const std::string& filename, int rank) {
  read_from_json(mini_json::parse_file(filename), rank, "");
}

void VariantQueryConfig::update_from_loader(const GenomicsDBImportConfig& l, int rank) {
  // defaults taken from the loader when the query JSON is silent (config_base.cc:182-203)
  if (m_vid_mapping_file.empty()) m_vid_mapping_file = l.m_vid_mapping_file;
  if (m_callset_mapping_file.empty()) m_callset_mapping_file = l.m_callset_mapping_file;
  if (m_vcf_header_filename.empty()) m_vcf_header_filename = l.m_vcf_header_filename;
  if (m_reference_genome.empty()) m_reference_genome = l.m_reference_genome;
  if (m_workspace.empty() && !l.m_workspaces.empty()) m_workspace = l.get_workspace(rank);
  if (m_array_name.empty() && !l.m_array_names.empty()) m_array_name = l.get_array_name(rank);
}

void VariantQueryConfig::subset_query_column_ranges_based_on_partition(const GenomicsDBImportConfig& loader, int rank) {
  if (!loader.is_partitioned_by_column()) return;
  const ColumnRange mine = loader.get_column_partition(rank);
  std::vector<ColumnRange> kept;
  for (const ColumnRange& r : m_query_column_intervals)
    if (r.second >= mine.first && r.first <= mine.second) kept.push_back(r);
  m_query_column_intervals = kept;
}

void VariantQueryConfig::read_from_json(const mini_json::Value& j, int rank, const std::string& base_dir) {
  (void)dir_of;
  if (j.HasMember("vid_mapping_file")) m_vid_mapping_file = pick_rank(j["vid_mapping_file"], rank).GetString();
  if (j.HasMember("callset_mapping_file")) m_callset_mapping_file = pick_rank(j["callset_mapping_file"], rank).GetString();
  if (!m_vid_mapper.is_initialized()) {
    if (!m_vid_mapping_file.empty()) m_vid_mapper.parse_vid_json(mini_json::parse_file(join_path(base_dir, m_vid_mapping_file)));
    else if (j.HasMember("vid_mapping")) m_vid_mapper.parse_vid_json(j["vid_mapping"]);
  }
  if (!m_vid_mapper.is_callset_mapping_initialized()) {
    if (!m_callset_mapping_file.empty()) m_vid_mapper.parse_callsets_json(mini_json::parse_file(join_path(base_dir, m_callset_mapping_file)));
    else if (j.HasMember("callset_mapping") || j.HasMember("callsets")) m_vid_mapper.parse_callsets_json(j);
  }
  if (!(m_vid_mapper.is_initialized() && m_vid_mapper.is_callset_mapping_initialized()))
    throw GenomicsDBConfigException("m_vid_mapper.is_initialized() && m_vid_mapper.is_callset_mapping_initialized()");
  if (j.HasMember("workspace")) m_workspace = pick_rank(j["workspace"], rank).GetString();
  if (j.HasMember("array") && j.HasMember("array_name")) throw GenomicsDBConfigException("both \"array\" and \"array_name\" specified");
  if (j.HasMember("array")) m_array_name = pick_rank(j["array"], rank).GetString();
  if (j.HasMember("array_name")) m_array_name = pick_rank(j["array_name"], rank).GetString();
  if (j.HasMember("scan_full")) {
    m_scan_whole_array = true;
  } else if (j.HasMember("query_column_ranges")) {
    const auto& q1 = j["query_column_ranges"];
    if (!q1.IsArray() || q1.Size() == 0) throw GenomicsDBConfigException("\"query_column_ranges\" must be a non-empty array");
    const auto& e = q1.Size() == 1 ? q1[0] : pick_rank(q1, rank);
    const mini_json::Value* q2 = &e;
    if (e.IsObject()) {
      if (e.MemberCount() == 0) q2 = nullptr;
      else if (e.HasMember("range_list")) q2 = &e["range_list"];
      else if (e.HasMember("column_or_interval_list")) q2 = &e["column_or_interval_list"];
      else throw GenomicsDBConfigException("Malformed \"query_column_ranges\" entry");
    }
    if (q2)
      for (size_t k = 0; k < q2->Size(); ++k) {
        const auto& q3 = (*q2)[k];
        ColumnRange r;
        if (q3.IsArray()) { if (q3.Size() != 2) throw GenomicsDBConfigException("query interval needs 2 elements"); r = ColumnRange(q3[0].GetInt64(), q3[1].GetInt64()); }
        else if (q3.IsInt64()) r = ColumnRange(q3.GetInt64(), q3.GetInt64());
        else if (q3.IsString()) {
          ContigInfo ci;
          if (!m_vid_mapper.get_contig_info(q3.GetString(), ci)) throw VidMapperException("Invalid contig name : " + q3.GetString());
          r = ColumnRange(ci.m_tiledb_column_offset, ci.m_tiledb_column_offset + ci.m_length - 1);
        } else if (q3.IsObject()) r = interval_from_object(q3, m_vid_mapper);
        else throw GenomicsDBConfigException("Malformed query interval");
        if (r.first > r.second) std::swap(r.first, r.second);
        m_query_column_intervals.push_back(r);
      }
    std::stable_sort(m_query_column_intervals.begin(), m_query_column_intervals.end(), [](const ColumnRange& a, const ColumnRange& b) { return a.first < b.first; });
  }
  if (!m_scan_whole_array && j.HasMember("query_row_ranges")) {
    const auto& q1 = j["query_row_ranges"];
    const auto& e = q1.Size() == 1 ? q1[0] : pick_rank(q1, rank);
    const mini_json::Value& q2 = e.IsArray() ? e : e["range_list"];
    std::vector<int64_t> rows;
    for (size_t k = 0; k < q2.Size(); ++k) {
      const auto& q3 = q2[k];
      int64_t a, b;
      if (q3.IsArray()) { a = q3[0].GetInt64(); b = q3[1].GetInt64(); }
      else if (q3.IsInt64()) a = b = q3.GetInt64();
      else { a = q3["low"].GetInt64(); b = q3["high"].GetInt64(); }
      if (a > b) std::swap(a, b);
      for (int64_t r = a; r <= b; ++r) rows.push_back(r);
    }
    std::sort(rows.begin(), rows.end());
    rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
    m_query_rows = rows;
    m_query_all_rows = false;
  }
  if (j.HasMember("query_attributes") && j.HasMember("attributes")) throw GenomicsDBConfigException("Query configuration cannot have both \"query_attributes\" and \"attributes\"");
  const char* ak = j.HasMember("query_attributes") ? "query_attributes" : (j.HasMember("attributes") ? "attributes" : nullptr);
  if (ak) { m_attributes.clear(); for (size_t i = 0; i < j[ak].Size(); ++i) m_attributes.push_back(j[ak][i].GetString()); }
  if (j.HasMember("vcf_header_filename")) m_vcf_header_filename = join_path(base_dir, pick_rank(j["vcf_header_filename"], rank).GetString());
  if (j.HasMember("vcf_output_filename")) m_vcf_output_filename = pick_rank(j["vcf_output_filename"], rank).GetString();
  if (j.HasMember("vcf_output_format")) set_vcf_output_format(j["vcf_output_format"].GetString());
  if (j.HasMember("reference_genome")) m_reference_genome = join_path(base_dir, pick_rank(j["reference_genome"], rank).GetString());
  if (j.HasMember("max_diploid_alt_alleles_that_can_be_genotyped")) m_max_diploid_alt_alleles_that_can_be_genotyped = (unsigned)j["max_diploid_alt_alleles_that_can_be_genotyped"].GetInt64();
  if (j.HasMember("combined_vcf_records_buffer_size_limit")) set_combined_vcf_records_buffer_size_limit((size_t)j["combined_vcf_records_buffer_size_limit"].GetInt64());
  {
    std::string order;
    if (j.HasMember("id_union_order")) order = j["id_union_order"].GetString();   // (the query key only: no environment fallback - an ambient variable must not change output bytes)
    if (!order.empty() && order != "sorted" && order != "unordered_set")
      throw GenomicsDBConfigException("id_union_order must be \"sorted\" or \"unordered_set\", not \"" + order + "\"");
    m_id_union_order_unordered_set = order == "unordered_set";
  }
  auto flag = [&](const char* k) { return j.HasMember(k) && j[k].GetBool(); };
  m_produce_GT_field = flag("produce_GT_field");
  m_produce_FILTER_field = flag("produce_FILTER_field");
  m_sites_only_query = flag("sites_only_query");
  m_index_output_VCF = flag("index_output_VCF");   // json_config.cc:648: a .tbi ("z") / .csi ("b") next to the output file
  m_produce_GT_with_min_PL_value_for_spanning_deletions = flag("produce_GT_with_min_PL_value_for_spanning_deletions");
  // Keys of the reference's query JSON (json_config.cc) that are accepted here and change nothing: said once per key and process, on stderr,
  // instead of being swallowed (GDBAMD_QUIET_JSON=1 silences it).  segment_size sizes TileDB's read buffers (this build stages column windows of
  // GDBAMD_STAGE_BUDGET_MB instead).
  if (j.IsObject() && !getenv("GDBAMD_QUIET_JSON")) {
    static std::mutex mu;
    static std::set<std::string> told;
    for (const char* k : {"segment_size", "query_filter", "num_parallel_vcf_files", "size_per_column_partition", "lb_callset_row_idx", "ub_callset_row_idx"}) {
      if (!j.HasMember(k)) continue;
      std::lock_guard<std::mutex> g(mu);
      if (told.insert(k).second) fprintf(stderr, "[genomicsdb_amd] warning: query JSON key \"%s\" is accepted and ignored by this build\n", k);
    }
  }
}

void VariantQueryConfig::add_attribute_to_query(const std::string& name) {
  if (m_query_attribute_name_to_query_idx.count(name)) return;
  m_query_attribute_name_to_query_idx[name] = (unsigned)m_query_attributes.size();
  QueryAttributeInfo a;
  a.m_name = name;
  m_query_attributes.push_back(a);
}

void VariantQueryConfig::reorder_query_fields() {
  // END, REF, ALT first, by pair-wise swaps (reference variant_query_config.cc:161-185: a displaced field lands where the
  // special field was - this permutes the INFO/FORMAT emission order exactly like the reference does)
  unsigned first_normal = 0;
  for (const char* sn : {"END", "REF", "ALT"}) {
    auto it = m_query_attribute_name_to_query_idx.find(sn);
    if (it == m_query_attribute_name_to_query_idx.end()) continue;
    unsigned q = it->second;
    if (q > first_normal) {
      std::string other = m_query_attributes[first_normal].m_name;
      m_query_attribute_name_to_query_idx[sn] = first_normal;
      m_query_attribute_name_to_query_idx[other] = q;
      std::swap(m_query_attributes[q], m_query_attributes[first_normal]);
    }
    ++first_normal;
  }
}

void VariantQueryConfig::do_query_bookkeeping(int64_t num_rows_in_array, int64_t lb_row_idx) {
  m_query_attributes.clear();
  m_query_attribute_name_to_query_idx.clear();
  std::vector<std::string> schema = m_vid_mapper.schema_attribute_names();
  std::vector<std::string> names = m_attributes.empty() ? schema : m_attributes;  // no attributes = all (query_variants.cc:245-251)
  {  // flatten_composite_fields (variant_query_config.cc:187-229): the tuple elements of a queried composite join the end of the
     // list, the composite itself leaves it
    std::vector<std::string> keep, extra;
    for (auto& n : names) {
      const FieldInfo* fi = m_vid_mapper.get_field_info(n);
      if (!fi) throw UnknownQueryAttributeException("Field " + n + " not found in vid mapping");
      if (fi->get_num_elements_in_tuple() > 1u) for (unsigned j = 0; j < fi->get_num_elements_in_tuple(); ++j) extra.push_back(m_vid_mapper.get_flattened_field_info(fi, j)->m_name);
      else keep.push_back(n);
    }
    keep.insert(keep.end(), extra.begin(), extra.end());
    names = keep;
  }
  for (auto& n : names) {
    const FieldInfo* fi = m_vid_mapper.get_field_info(n);
    if (!fi) throw UnknownQueryAttributeException("Field " + n + " not found in vid mapping");
    if (m_sites_only_query && fi->m_is_vcf_FORMAT_field && n != "DP_FORMAT" && n != "MIN_DP") continue;  // :257-273
    if (std::find(schema.begin(), schema.end(), n) == schema.end()) throw UnknownQueryAttributeException("Invalid query attribute : " + n);
    add_attribute_to_query(n);
  }
  add_attribute_to_query("END");
  add_attribute_to_query("ALT");
  add_attribute_to_query("REF");
  bool added_GT = false;
  for (unsigned i = 0; i < m_query_attributes.size(); ++i) {
    const FieldInfo* fi = m_vid_mapper.get_field_info(m_query_attributes[i].m_name);
    if (!added_GT && fi->is_length_genotype_dependent()) {
      if (std::find(schema.begin(), schema.end(), "GT") == schema.end()) throw UnknownQueryAttributeException("GT needed but not in schema");
      add_attribute_to_query("GT");
      added_GT = true;
    }
  }
  reorder_query_fields();
  for (auto& k : m_known_to_query) k = -1;
  for (unsigned i = 0; i < m_query_attributes.size(); ++i) {
    auto& a = m_query_attributes[i];
    a.m_field_info = m_vid_mapper.get_field_info(a.m_name);
    a.m_known_enum = known_field_enum_for_name(a.m_name);
    if (a.m_known_enum >= 0) m_known_to_query[a.m_known_enum] = (int)i;
  }
  m_num_rows_in_array = num_rows_in_array;
  m_smallest_row_idx = lb_row_idx;
  if (!m_query_all_rows) {  // out-of-bounds rows are ignored (variant_query_config.cc:96-119)
    std::vector<int64_t> keep;
    for (auto r : m_query_rows) if (r >= lb_row_idx && r < lb_row_idx + num_rows_in_array) keep.push_back(r);
    m_query_rows = keep;
  }
  m_done_bookkeeping = true;
}

}  // namespace genomicsdb_amd
