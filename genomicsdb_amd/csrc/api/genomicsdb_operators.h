// genomicsdb_operators.h - the reference's C++ operator / scan surface, source-compatible, over the device engine.
//
// A caller written against the reference - tools/src/gt_mpi_gather.cc:322-366 (scan_and_produce_Broad_GVCF) and :531-612 (main) -
// keeps compiling: the same class names, constructors and scan_and_operate signature
//   SingleVariantOperatorBase        src/main/cpp/include/query_operations/variant_operations.h:349-388
//   GA4GHOperator                    variant_operations.h:648-673
//   BroadCombinedGVCFOperator        include/query_operations/broad_combined_gvcf.h:59-61
//   VariantQueryProcessor::scan_and_operate, VariantQueryProcessorScanState   include/genomicsdb/query_variants.h:126-191, 241-243
//   VCFAdapter / VCFSerializedBufferAdapter / RWBuffer                        include/vcf/vcf_adapter.h:60-260
//   VariantStorageManager (constructor, close_array)                          include/genomicsdb/variant_storage_manager.h
// What differs is who does the work.  The reference calls operate(Variant&) once per output interval on the host; here the
// scan, the combine operator and the serialisation of a whole column interval run on the GPU, so scan_and_operate RECOGNISES
// the built-in BroadCombinedGVCFOperator and hands its adapter whole pages of finished VCF / BCF.  A user-defined per-record
// operate() cannot be run at device speed: it is refused loudly (VariantOperationException), and the batched hook
// BatchedVariantOperatorBase::operate_on_page() is the extension point instead (pages of combined records where they lie in HBM).
// Classes live in namespace genomicsdb_amd; define GENOMICSDB_AMD_GLOBAL_NAMES before the include to get the reference's
// global names.
#pragma once
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "genomicsdb_bcf_generator.h"

namespace genomicsdb_amd {

class VariantOperationException : public std::runtime_error {
 public:
  explicit VariantOperationException(const std::string& m) : std::runtime_error("VariantOperationException : " + m) {}
};
class VCFAdapterException : public std::runtime_error {
 public:
  explicit VCFAdapterException(const std::string& m) : std::runtime_error("VCFAdapterException : " + m) {}
};

constexpr unsigned MAX_DIPLOID_ALT_ALLELES_THAT_CAN_BE_GENOTYPED = 50u;

// the reference's read/write byte buffer (include/vcf/vcf_adapter.h: RWBuffer)
struct RWBuffer {
  explicit RWBuffer(size_t capacity = 1048576u) : m_buffer(capacity) {}
  std::vector<uint8_t> m_buffer;
  size_t m_num_valid_bytes = 0, m_next_read_idx = 0;
  size_t get_num_remaining_bytes() const { return m_num_valid_bytes - m_next_read_idx; }
  const uint8_t* get_pointer_at_read_position() const { return &m_buffer[m_next_read_idx]; }
};

// Output side.  VCFAdapter writes straight to the file named by the query ("-" / empty: stdout); VCFSerializedBufferAdapter
// collects a batch in an RWBuffer of the caller and writes it when the caller says so (the paged "-p" mode of gt_mpi_gather).
class VCFAdapter {
 public:
  explicit VCFAdapter(bool open_output = true) : m_open_output(open_output) {}
  virtual ~VCFAdapter();
  virtual void initialize(const VariantQueryConfig& query_config);
  virtual bool overflow() const { return false; }
  const std::string& get_output_format() const { return m_output_format; }
  // engine side: bytes of finished records (or the header)
  virtual void handoff(const uint8_t* bytes, size_t n);
 protected:
  bool m_open_output;
  FILE* m_out = nullptr;
  bool m_owns_out = false, m_wrote_bytes = false, m_index_output = false;   // m_index_output: "index_output_VCF" (.tbi / .csi when the file closes)
  std::string m_output_format, m_output_filename;
  size_t m_buffer_limit = 1048576u;
};
class VCFSerializedBufferAdapter : public VCFAdapter {
 public:
  VCFSerializedBufferAdapter(bool keep_idx_fields_in_bcf_header = true, bool do_output = false) : VCFAdapter(do_output), m_keep_idx(keep_idx_fields_in_bcf_header) {}
  void set_buffer(RWBuffer& buffer) { m_rw_buffer = &buffer; }
  bool keep_idx_fields_in_bcf_header() const { return m_keep_idx; }
  bool overflow() const override { return m_rw_buffer && m_rw_buffer->m_num_valid_bytes >= m_buffer_limit; }
  void do_output();                                  // writes the valid bytes of the buffer to the output file
  void handoff(const uint8_t* bytes, size_t n) override;
 private:
  bool m_keep_idx;
  RWBuffer* m_rw_buffer = nullptr;
};

class Variant;                       // per-record objects are not materialised by this build
class VariantArraySchema {};
class CombineAllelesLUT {};          // (member of the operator base in the reference; the LUTs live in HBM here)

class VariantStorageManager {        // names a workspace; arrays are opened by the query processor
 public:
  explicit VariantStorageManager(const std::string& workspace, size_t segment_size = 10u * 1024u * 1024u) : m_workspace(workspace), m_segment_size(segment_size) {}
  const std::string& get_workspace() const { return m_workspace; }
  size_t get_segment_size() const { return m_segment_size; }
  void close_array(int) {}
 private:
  std::string m_workspace;
  size_t m_segment_size;
};

class SingleVariantOperatorBase {
 public:
  explicit SingleVariantOperatorBase(const VidMapper* vid_mapper) : m_vid_mapper(vid_mapper && vid_mapper->is_initialized() ? vid_mapper : nullptr) { clear(); }
  virtual ~SingleVariantOperatorBase() {}
  void clear() { m_merged_reference_allele.clear(); m_merged_alt_alleles.clear(); }
  // per-record host call of the reference; never reached on the device path (scan_and_operate refuses operators it does not recognise)
  virtual void operate(Variant& variant, const VariantQueryConfig& query_config);
  virtual bool overflow() const { return false; }
 protected:
  CombineAllelesLUT m_alleles_LUT;
  bool m_NON_REF_exists = false;
  std::string m_merged_reference_allele;
  std::vector<std::string> m_merged_alt_alleles;
  bool m_remapping_needed = true, m_is_reference_block_only = false;
  const VidMapper* m_vid_mapper;
};

class GA4GHOperator : public SingleVariantOperatorBase {
 public:
  GA4GHOperator(const VariantQueryConfig& query_config, const VidMapper& vid_mapper,
                const unsigned max_diploid_alt_alleles_that_can_be_genotyped = MAX_DIPLOID_ALT_ALLELES_THAT_CAN_BE_GENOTYPED)
      : SingleVariantOperatorBase(&vid_mapper), m_max_diploid_alt_alleles_that_can_be_genotyped(max_diploid_alt_alleles_that_can_be_genotyped) { (void)query_config; }
  const Variant& get_remapped_variant() const;     // throws: records stay on the device
  bool too_many_alt_alleles_for_genotype_length_fields(unsigned num_alt_alleles) const { return num_alt_alleles > m_max_diploid_alt_alleles_that_can_be_genotyped; }
  unsigned get_max_diploid_alt_alleles_that_can_be_genotyped() const { return m_max_diploid_alt_alleles_that_can_be_genotyped; }
 protected:
  unsigned m_max_diploid_alt_alleles_that_can_be_genotyped;
};

// the recognised built-in: its semantics are compiled into the device CombinePlan (csrc/host/combine_plan.cc)
class BroadCombinedGVCFOperator : public GA4GHOperator {
 public:
  BroadCombinedGVCFOperator(VCFAdapter& vcf_adapter, const VidMapper& id_mapper, const VariantQueryConfig& query_config,
                            const unsigned max_diploid_alt_alleles_that_can_be_genotyped = MAX_DIPLOID_ALT_ALLELES_THAT_CAN_BE_GENOTYPED,
                            const bool use_missing_values_only_not_vector_end = false);
  bool overflow() const override { return m_vcf_adapter->overflow(); }
  VCFAdapter& get_vcf_adapter() const { return *m_vcf_adapter; }
  bool use_missing_values_only_not_vector_end() const { return m_use_missing_values_not_vector_end; }
 private:
  VCFAdapter* m_vcf_adapter;
  bool m_use_missing_values_not_vector_end;
};

// Batched hook for operators of the caller's own: called once per page of combined records (VCF text or BCF2 records, whole
// records only), with the page where the device assembled it.
class BatchedVariantOperatorBase : public SingleVariantOperatorBase {
 public:
  explicit BatchedVariantOperatorBase(const VidMapper* vid_mapper) : SingleVariantOperatorBase(vid_mapper) {}
  virtual void operate_on_page(const char* device_ptr, uint64_t nbytes, int64_t column_begin, int64_t column_end) = 0;
};

// The reference's profiling counters (include/genomicsdb/query_variants.h:67-124, printed by scan_and_operate under -DDO_PROFILING),
// same enum names; filled from the device's per-interval counters (gdbamd_interval_stats.gt_profile_stats, include/genomicsdb_amd.h)
// after every piece of a scan, always on.
class GTProfileStats {
 public:
  enum GTStatIdx {
    GT_NUM_CELLS = 0, GT_NUM_CELLS_IN_LEFT_SWEEP, GT_NUM_VALID_CELLS_IN_QUERY, GT_NUM_ATTR_CELLS_ACCESSED,
    GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS, GT_NUM_OPERATOR_INVOCATIONS, GT_NUM_STATS
  };
  void update_stat(unsigned stat_idx, uint64_t value) {
    if (stat_idx >= GT_NUM_STATS) return;
    m_sum[stat_idx] += value; m_sum_sq[stat_idx] += (double)value * (double)value;
  }
  void increment_num_queries() { ++m_num_queries; }
  uint64_t get_stat(unsigned stat_idx) const { return stat_idx < GT_NUM_STATS ? m_sum[stat_idx] : 0; }
  uint64_t get_num_queries() const { return m_num_queries; }
  static const char* stat_name(unsigned stat_idx);
  void print_stats(FILE* fptr = stderr) const;      // "stat_name,sum,sum_sq,mean,std-dev" lines as the reference prints them
 private:
  uint64_t m_sum[GT_NUM_STATS] = {0, 0, 0, 0, 0, 0};
  double m_sum_sq[GT_NUM_STATS] = {0, 0, 0, 0, 0, 0};
  uint64_t m_num_queries = 0;
};

class VariantQueryProcessorScanState {           // where a scan stands between two scan_and_operate calls of one column interval
 public:
  bool end() const { return m_done; }
  void reset() { m_done = false; m_started = false; }
 private:
  friend class VariantQueryProcessor;
  bool m_done = false, m_started = false;
  int64_t m_piece_begin = 0, m_interval_end = 0;
  bool m_piece_active = false;
};

class VariantQueryProcessor {
 public:
  VariantQueryProcessor(VariantStorageManager* storage_manager, const std::string& array_name, const VidMapper& vid_mapper);
  ~VariantQueryProcessor();
  int get_array_descriptor() const { return 0; }
  const VariantArraySchema& get_array_schema() const { return m_schema; }
  void do_query_bookkeeping(const VariantArraySchema& schema, VariantQueryConfig& query_config, const VidMapper& vid_mapper, const bool alleles_required) const;
  // One call = the next batch of the column interval `column_interval_idx`: with a serialized-buffer adapter it returns when the
  // operator's buffer has overflowed (scan state says whether the interval is done), otherwise it runs the interval to its end.
  void scan_and_operate(const int ad, const VariantQueryConfig& query_config, SingleVariantOperatorBase& variant_operator, unsigned column_interval_idx = 0u,
                        bool handle_spanning_deletions = false, VariantQueryProcessorScanState* scan_state = 0) const;
  const GTProfileStats& get_profile_stats() const { return m_stats; }   // summed over every piece of every scan of this processor
 private:
  mutable GTProfileStats m_stats;
  VariantStorageManager* m_storage_manager;
  std::string m_array_name;
  VariantArraySchema m_schema;
  struct Engine;
  mutable std::unique_ptr<Engine> m_engine;
};

}  // namespace genomicsdb_amd

#ifdef GENOMICSDB_AMD_GLOBAL_NAMES
using genomicsdb_amd::BatchedVariantOperatorBase;
using genomicsdb_amd::BroadCombinedGVCFOperator;
using genomicsdb_amd::GA4GHOperator;
using genomicsdb_amd::GenomicsDBBCFGenerator;
using genomicsdb_amd::GenomicsDBImportConfig;
using genomicsdb_amd::GTProfileStats;
using genomicsdb_amd::RWBuffer;
using genomicsdb_amd::SingleVariantOperatorBase;
using genomicsdb_amd::Variant;
using genomicsdb_amd::VariantArraySchema;
using genomicsdb_amd::VariantOperationException;
using genomicsdb_amd::VariantQueryConfig;
using genomicsdb_amd::VariantQueryProcessor;
using genomicsdb_amd::VariantQueryProcessorScanState;
using genomicsdb_amd::VariantStorageManager;
using genomicsdb_amd::VCFAdapter;
using genomicsdb_amd::VCFSerializedBufferAdapter;
using genomicsdb_amd::VidMapper;
#endif
