// variant_query_config.h - product host layer: loader / query JSON -> query configuration.
// Same JSON keys, defaults and accessor names as the reference's GenomicsDBConfigBase / JSONConfigBase /
// VariantQueryConfig / GenomicsDBImportConfig for the scan/combine path
// (reference src/main/cpp/src/config/json_config.cc:195-658,709-820; include/config/variant_query_config.h:54-378;
//  src/genomicsdb/query_variants.cc:243-294,578-685 for the attribute bookkeeping).
#pragma once
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "vid_mapper.h"

namespace genomicsdb_amd {

class GenomicsDBConfigException : public std::runtime_error {
 public:
  explicit GenomicsDBConfigException(const std::string& m) : std::runtime_error("GenomicsDBConfigException : " + m) {}
};
class UnknownQueryAttributeException : public std::runtime_error {
 public:
  explicit UnknownQueryAttributeException(const std::string& m) : std::runtime_error("UnknownQueryAttributeException : " + m) {}
};

typedef std::pair<int64_t, int64_t> ColumnRange;

// Loader JSON: only what the query side consumes (column partitions, default vid/callset/header/reference paths).
class GenomicsDBImportConfig {
 public:
  void read_from_file(const std::string& filename, int rank = 0);
  void read_from_json(const mini_json::Value& doc, int rank = 0);
  ColumnRange get_column_partition(int rank) const;
  bool is_partitioned_by_column() const { return !m_column_partitions.empty(); }
  const std::string& get_workspace(int rank) const { return m_workspaces.size() == 1 ? m_workspaces[0] : m_workspaces.at((size_t)rank); }
  const std::string& get_array_name(int rank) const { return m_array_names.size() == 1 ? m_array_names[0] : m_array_names.at((size_t)rank); }
  std::string m_vid_mapping_file, m_callset_mapping_file, m_vcf_header_filename, m_reference_genome;
  std::vector<ColumnRange> m_sorted_column_partitions, m_column_partitions;
  std::vector<std::string> m_workspaces, m_array_names;
  bool m_treat_deletions_as_intervals = false, m_produce_combined_vcf = false;
  bool m_loaded = false;
};

struct QueryAttributeInfo { std::string m_name; const FieldInfo* m_field_info = nullptr; int m_known_enum = -1; };

class VariantQueryConfig {
 public:
  // --- reading (reference VariantQueryConfig::read_from_file / update_from_loader) ---
  void read_from_file(const std::string& filename, int rank = 0);
  void read_from_json(const mini_json::Value& doc, int rank = 0, const std::string& base_dir = "");
  void update_from_loader(const GenomicsDBImportConfig& loader, int rank = 0);
  // keeps the queried column ranges that overlap the loader's column partition of this rank (reference
  // GenomicsDBConfigBase::subset_query_column_ranges_based_on_partition, genomicsdb_config_base.cc:205-223; the query stream
  // calls it right after reading the query JSON, genomicsdb_bcf_generator.cc:52-53)
  void subset_query_column_ranges_based_on_partition(const GenomicsDBImportConfig& loader, int rank = 0);
  // do_query_bookkeeping for the produce-Broad-GVCF path (alleles always required)
  void do_query_bookkeeping(int64_t num_rows_in_array, int64_t lb_row_idx = 0);
  bool is_bookkeeping_done() const { return m_done_bookkeeping; }
  // --- accessors with the reference's names ---
  const VidMapper& get_vid_mapper() const { return m_vid_mapper; }
  VidMapper& get_vid_mapper() { return m_vid_mapper; }
  unsigned get_num_column_intervals() const { return (unsigned)m_query_column_intervals.size(); }
  int64_t get_column_begin(unsigned i) const { return m_query_column_intervals[i].first; }
  int64_t get_column_end(unsigned i) const { return m_query_column_intervals[i].second; }
  void set_column_interval_to_query(int64_t b, int64_t e) { m_query_column_intervals.assign(1, ColumnRange(b, e)); }
  uint64_t get_num_rows_to_query() const { return m_query_all_rows ? (uint64_t)m_num_rows_in_array : m_query_rows.size(); }
  int64_t get_array_row_idx_for_query_row_idx(uint64_t q) const { return m_query_all_rows ? (int64_t)q + m_smallest_row_idx : m_query_rows[q]; }
  int64_t get_num_rows_in_array() const { return m_num_rows_in_array; }
  unsigned get_num_queried_attributes() const { return (unsigned)m_query_attributes.size(); }
  const std::string& get_query_attribute_name(unsigned q) const { return m_query_attributes[q].m_name; }
  const FieldInfo* get_field_info_for_query_attribute_idx(unsigned q) const { return m_query_attributes[q].m_field_info; }
  bool is_defined_query_idx_for_known_field_enum(unsigned e) const { return m_known_to_query[e] >= 0; }
  unsigned get_query_idx_for_known_field_enum(unsigned e) const { return (unsigned)m_known_to_query[e]; }
  int get_known_field_enum_for_query_idx(unsigned q) const { return m_query_attributes[q].m_known_enum; }
  bool produce_GT_field() const { return m_produce_GT_field; }
  bool produce_FILTER_field() const { return m_produce_FILTER_field; }
  bool sites_only_query() const { return m_sites_only_query; }
  bool index_output_VCF() const { return m_index_output_VCF; }
  bool produce_GT_with_min_PL_value_for_spanning_deletions() const { return m_produce_GT_with_min_PL_value_for_spanning_deletions; }
  unsigned get_max_diploid_alt_alleles_that_can_be_genotyped() const { return m_max_diploid_alt_alleles_that_can_be_genotyped; }
  // ID union order: false = sorted (the reference's DEBUG build, the goldens), true = std::unordered_set<std::string> (any other
  // build of the reference, broad_combined_gvcf.cc:732-737: what a drop-in for a Release libtiledbgenomicsdb.so sets).  Query JSON
  // "id_union_order": "sorted" | "unordered_set", or the setter below; default sorted.  The library reads no environment variable for it.
  bool id_union_order_unordered_set() const { return m_id_union_order_unordered_set; }
  void set_id_union_order_unordered_set(bool v) { m_id_union_order_unordered_set = v; }
  void set_max_diploid_alt_alleles_that_can_be_genotyped(unsigned v) { m_max_diploid_alt_alleles_that_can_be_genotyped = v; }
  size_t get_combined_vcf_records_buffer_size_limit() const { return m_combined_vcf_records_buffer_size_limit; }
  void set_combined_vcf_records_buffer_size_limit(size_t v) { m_combined_vcf_records_buffer_size_limit = v ? v : 1; }
  const std::string& get_vcf_header_filename() const { return m_vcf_header_filename; }
  const std::string& get_vcf_output_filename() const { return m_vcf_output_filename; }
  const std::string& get_vcf_output_format() const { return m_vcf_output_format; }
  void set_vcf_output_format(const std::string& f);
  const std::string& get_reference_genome() const { return m_reference_genome; }
  const std::string& get_workspace(int) const { return m_workspace; }
  const std::string& get_array_name(int) const { return m_array_name; }
  bool scan_whole_array() const { return m_scan_whole_array; }
  void set_attributes_to_query(const std::vector<std::string>& names) { m_attributes = names; }

  std::string m_workspace, m_array_name, m_vcf_header_filename, m_vcf_output_filename = "-", m_vcf_output_format, m_reference_genome;
  std::string m_vid_mapping_file, m_callset_mapping_file;

 private:
  void add_attribute_to_query(const std::string& name);
  void reorder_query_fields();
  VidMapper m_vid_mapper;
  std::vector<std::string> m_attributes;
  std::vector<QueryAttributeInfo> m_query_attributes;
  std::unordered_map<std::string, unsigned> m_query_attribute_name_to_query_idx;
  int m_known_to_query[GVCF_NUM_KNOWN_FIELDS];
  std::vector<ColumnRange> m_query_column_intervals;
  bool m_scan_whole_array = false, m_query_all_rows = true;
  std::vector<int64_t> m_query_rows;
  int64_t m_num_rows_in_array = 0, m_smallest_row_idx = 0;
  bool m_produce_GT_field = false, m_produce_FILTER_field = false, m_sites_only_query = false, m_index_output_VCF = false;
  bool m_produce_GT_with_min_PL_value_for_spanning_deletions = false;
  unsigned m_max_diploid_alt_alleles_that_can_be_genotyped = 50;
  bool m_id_union_order_unordered_set = false;
  size_t m_combined_vcf_records_buffer_size_limit = 1048576u;
  bool m_done_bookkeeping = false;
};

}  // namespace genomicsdb_amd
