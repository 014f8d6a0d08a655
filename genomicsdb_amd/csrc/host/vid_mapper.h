// vid_mapper.h - product host layer: vid / callset JSON -> field, contig and callset tables.
// Source-compatible subset of the reference's VidMapper / FileBasedVidMapper as used by the
// scan/combine path (reference src/main/cpp/include/utils/vid_mapper.h:151-433, :439-).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../common/mini_json.hpp"
#include "../core/gdb_types.h"

namespace genomicsdb_amd {

class VidMapperException : public std::runtime_error {
 public:
  explicit VidMapperException(const std::string& m) : std::runtime_error("VidMapperException : " + m) {}
};

enum KnownVariantFieldsEnum {  // reference include/vcf/known_field_info.h:30-61 (same order)
  GVCF_END_IDX = 0, GVCF_REF_IDX, GVCF_ALT_IDX, GVCF_QUAL_IDX, GVCF_FILTER_IDX, GVCF_BASEQRANKSUM_IDX,
  GVCF_CLIPPINGRANKSUM_IDX, GVCF_MQRANKSUM_IDX, GVCF_READPOSRANKSUM_IDX, GVCF_DP_IDX, GVCF_MQ_IDX,
  GVCF_RAW_MQ_IDX, GVCF_MQ0_IDX, GVCF_DP_FORMAT_IDX, GVCF_MIN_DP_IDX, GVCF_GQ_IDX, GVCF_SB_IDX,
  GVCF_AD_IDX, GVCF_PL_IDX, GVCF_AF_IDX, GVCF_AN_IDX, GVCF_AC_IDX, GVCF_GT_IDX, GVCF_PS_IDX,
  GVCF_PGT_IDX, GVCF_PID_IDX, GVCF_EXCESS_HET, GVCF_ID_IDX, GVCF_NUM_KNOWN_FIELDS
};
int known_field_enum_for_name(const std::string& name);  // -1 when not a known field

struct FieldInfo {
  std::string m_name, m_vcf_name;
  bool m_is_vcf_INFO_field = false, m_is_vcf_FORMAT_field = false, m_is_vcf_FILTER_field = false;
  int m_field_idx = -1;
  GdbLength m_length_descriptor = GDB_VL_FIXED;
  unsigned m_num_elements = 1;
  GdbElem m_element_type = GDB_ET_INT;
  GdbCombineOp m_VCF_field_combine_operation = GDB_OP_UNKNOWN;
  bool m_unsupported_on_device = false;  // fields of more than 2 dimensions
  // 2-D fields (allele-specific annotations, "length": [ "R", "var" ]) and fields whose elements are tuples ("type": [ "float",
  // "int" ]): m_length_descriptor describes dimension 0, the data is one byte blob per tuple element
  // (genomicsdb_multid_vector_field.h:69-86), every tuple element is a flattened field <name>_tuple_element_<i> of its own
  // with the vcf name of the composite (reference vid_mapper.cc:727-800)
  unsigned m_num_dimensions = 1;
  char m_vcf_delimiter[2] = {'|', ','};
  std::vector<GdbElem> m_tuple_element_types;
  bool m_is_flattened_field = false;
  unsigned m_element_index_in_tuple = 0;
  int m_parent_composite_field_idx = -1;
  unsigned get_num_elements_in_tuple() const { return m_tuple_element_types.empty() ? 1u : (unsigned)m_tuple_element_types.size(); }
  bool is_flattened_field() const { return m_is_flattened_field; }
  bool is_fixed_length_field() const { return m_length_descriptor == GDB_VL_FIXED; }
  bool is_length_allele_dependent() const { return m_length_descriptor == GDB_VL_A || m_length_descriptor == GDB_VL_R || m_length_descriptor == GDB_VL_G; }
  bool is_length_genotype_dependent() const { return m_length_descriptor == GDB_VL_G; }
};

struct ContigInfo { std::string m_name; int64_t m_tiledb_column_offset = 0, m_length = 0; };
// one entry of the callset mapping (reference vid_mapper.cc:88-146 CallSetInfo: row, file, index of the sample in the file)
struct CallSetInfo { std::string m_name; int64_t m_row_idx = -1, m_idx_in_file = 0; std::string m_filename; };

class VidMapper {
 public:
  void parse_vid_json(const mini_json::Value& doc);
  void parse_callsets_json(const mini_json::Value& doc);
  bool is_initialized() const { return m_is_initialized; }
  bool is_callset_mapping_initialized() const { return m_is_callset_mapping_initialized; }
  unsigned get_num_fields() const { return (unsigned)m_field_idx_to_info.size(); }
  const FieldInfo& get_field_info(unsigned idx) const { return m_field_idx_to_info[idx]; }
  const FieldInfo* get_field_info(const std::string& name) const;
  const FieldInfo* get_flattened_field_info(const FieldInfo* field_info, unsigned tuple_element_index) const;
  unsigned get_num_contigs() const { return (unsigned)m_contig_idx_to_info.size(); }
  const ContigInfo& get_contig_info(unsigned idx) const { return m_contig_idx_to_info[idx]; }
  bool get_contig_info(const std::string& name, ContigInfo& out) const;
  bool get_contig_location(int64_t position, std::string& contig_name, int64_t& contig_position) const;
  bool get_callset_name(int64_t row_idx, std::string& name) const;
  int64_t get_num_callsets() const { return (int64_t)m_row_idx_to_name.size(); }
  const std::vector<CallSetInfo>& get_callsets() const { return m_callsets; }   // mapping order
  // attribute order of the array schema (reference vid_mapper.cc:354-442): END, REF, ALT, [ID], QUAL, FILTER, INFO.., FORMAT..
  std::vector<std::string> schema_attribute_names() const;
 private:
  void add_mandatory_fields();
  std::vector<FieldInfo> m_field_idx_to_info;
  std::unordered_map<std::string, int> m_field_name_to_idx;
  std::vector<ContigInfo> m_contig_idx_to_info;
  std::vector<std::pair<int64_t, int>> m_contig_begin_2_idx;
  std::vector<std::string> m_row_idx_to_name;
  std::vector<CallSetInfo> m_callsets;
  bool m_is_initialized = false, m_is_callset_mapping_initialized = false;
};

}  // namespace genomicsdb_amd
