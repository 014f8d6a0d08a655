// combine_plan.h - product host layer: (vid mapping + query configuration) -> device CombinePlan, the text tables
// the kernels index, and the VCF header.  This is the host half of the reference's BroadCombinedGVCFOperator
// constructor (reference src/main/cpp/src/query_operations/broad_combined_gvcf.cc:140-356) and of
// VCFAdapter::add_field_to_hdr_if_missing (src/vcf/vcf_adapter.cc:59-199).
#pragma once
#include <string>
#include <vector>

#include "variant_query_config.h"

namespace genomicsdb_amd {

class UnsupportedOnDeviceException : public std::runtime_error {
 public:
  explicit UnsupportedOnDeviceException(const std::string& m) : std::runtime_error("UnsupportedOnDeviceException : " + m) {}
};
class BroadCombinedGVCFException : public std::runtime_error {
 public:
  explicit BroadCombinedGVCFException(const std::string& m) : std::runtime_error("BroadCombinedGVCFException : " + m) {}
};

struct HostPlan {
  CombinePlan plan;
  std::vector<std::string> field_names;  // plan field idx -> array attribute name
  std::string header_text;               // template "##" lines + added lines + #CHROM line
  // name tables (NameTables on the device)
  std::string names_text;
  std::vector<int32_t> field_name_off, field_name_len, filter_name_off, filter_name_len;
  // contig table sorted by offset
  std::vector<GdbContig> contigs;
  std::string contig_names;
};

HostPlan build_combine_plan(const VariantQueryConfig& qc, const std::string& template_header_text);

}  // namespace genomicsdb_amd
