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
  bool bgzf = false;                     // output formats "z" / "b": the stream is BGZF-compressed (on the device, kernels/gdb_bgzf.hip)
  std::vector<std::string> field_names;  // plan field idx -> array attribute name
  std::string header_text;               // template "##" lines + added lines + #CHROM line
  // BCF2 ("bu") flavour of the stream: "BCF\2\2", l_text, the same text with an IDX key on every FILTER / INFO / FORMAT / contig
  // line (or without, keep_idx_fields_in_bcf_header = false), NUL  (htslib bcf_hdr_write; fork: bcf_hdr_serialize, vcf_adapter.cc:475-488)
  std::string bcf_header_bytes(bool keep_idx_fields) const;
  std::vector<std::string> header_lines;  // "##" lines of header_text in order
  std::vector<int32_t> header_line_idx;   // dictionary index of each such line (-1: the line is no dictionary entry)
  std::vector<int32_t> filter_bcf_id;     // per vid field idx: dictionary index of the FILTER of that name (-1: not in the header)
  // name tables (NameTables on the device)
  std::string names_text;
  std::vector<int32_t> field_name_off, field_name_len, filter_name_off, filter_name_len;
  // contig table sorted by offset
  std::vector<GdbContig> contigs;
  std::string contig_names;
};

// output_format "": VCF text; "bu": BCF2 records; "z" / "b": the same two as BGZF blocks (genomicsdb_config_base.cc:34,156-165).  use_missing_values_not_vector_end: the JNI flag for htsjdk.
HostPlan build_combine_plan(const VariantQueryConfig& qc, const std::string& template_header_text, const std::string& output_format = "",
                            bool use_missing_values_not_vector_end = false);

}  // namespace genomicsdb_amd
