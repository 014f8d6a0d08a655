// gt_mpi_gather_shaped.cc - TEST: a caller written the way the reference's own tool is (tools/src/gt_mpi_gather.cc:322-366
// scan_and_produce_Broad_GVCF and :531-612 main), with the reference's global class names, compiled against this build's
// headers and linked to libgenomicsdb_amd.so.  It must compile unchanged in shape and, on a GPU, print the reference's golden.
//   gt_mpi_gather_shaped <query.json> [page_size]
#define GENOMICSDB_AMD_GLOBAL_NAMES
#include "../../genomicsdb_amd/csrc/api/genomicsdb_operators.h"

#include <algorithm>
#include <iostream>

void scan_and_produce_Broad_GVCF(const VariantQueryProcessor& qp, const VariantQueryConfig& query_config, VCFAdapter& vcf_adapter, const VidMapper& id_mapper,
                                 int my_world_mpi_rank) {
  // Read output in batches if required.  Must initialize buffer before constructing gvcf_op
  RWBuffer rw_buffer;
  auto serialized_vcf_adapter_ptr = dynamic_cast<VCFSerializedBufferAdapter*>(&vcf_adapter);
  if (serialized_vcf_adapter_ptr) serialized_vcf_adapter_ptr->set_buffer(rw_buffer);
  SingleVariantOperatorBase* op_ptr = new BroadCombinedGVCFOperator(vcf_adapter, id_mapper, query_config, query_config.get_max_diploid_alt_alleles_that_can_be_genotyped());
  // At least 1 iteration
  VariantQueryProcessorScanState scan_state;
  for (auto i = 0u; i < std::max(1u, query_config.get_num_column_intervals()); ++i) {
    while (!scan_state.end()) {
      qp.scan_and_operate(qp.get_array_descriptor(), query_config, *op_ptr, i, true, &scan_state);
      if (serialized_vcf_adapter_ptr) {
        serialized_vcf_adapter_ptr->do_output();
        rw_buffer.m_num_valid_bytes = 0u;
      }
    }
    scan_state.reset();
  }
  (void)my_world_mpi_rank;
  delete op_ptr;
}

// an operator of the caller's own: per-record operate() is refused by the device scan, loudly
class MyPerRecordOperator : public SingleVariantOperatorBase {
 public:
  explicit MyPerRecordOperator(const VidMapper* m) : SingleVariantOperatorBase(m) {}
};
// ... while the batched hook receives the pages
class CountingOperator : public BatchedVariantOperatorBase {
 public:
  explicit CountingOperator(const VidMapper* m) : BatchedVariantOperatorBase(m) {}
  void operate_on_page(const char*, uint64_t nbytes, int64_t, int64_t) override { bytes += nbytes; ++pages; }
  uint64_t bytes = 0; int pages = 0;
};

// a class derived from the built-in with its own per-record operate(): must be refused too (nothing would ever call it)
class MyDerivedGVCFOperator : public BroadCombinedGVCFOperator {
 public:
  using BroadCombinedGVCFOperator::BroadCombinedGVCFOperator;
  void operate(Variant&, const VariantQueryConfig&) override { ++calls; }
  int calls = 0;
};

int main(int argc, char** argv) {
  if (argc >= 3 && std::string(argv[1]) == "--operator-selftest") {
    // host only (the refusals come before any device work): which operators does the device scan accept?
    try {
      VariantQueryConfig query_config;
      query_config.read_from_file(argv[2], 0);
      VCFAdapter vcf_adapter;
      VariantStorageManager sm("/nonexistent-workspace", 10u * 1024u * 1024u);
      VariantQueryProcessor qp(&sm, "nonexistent-array", query_config.get_vid_mapper());
      qp.do_query_bookkeeping(qp.get_array_schema(), query_config, query_config.get_vid_mapper(), true);
      MyPerRecordOperator mine(&query_config.get_vid_mapper());
      MyDerivedGVCFOperator derived(vcf_adapter, query_config.get_vid_mapper(), query_config);
      int refused = 0;
      try { qp.scan_and_operate(qp.get_array_descriptor(), query_config, mine, 0, true, 0); } catch (const VariantOperationException&) { refused |= 1; }
      try { qp.scan_and_operate(qp.get_array_descriptor(), query_config, derived, 0, true, 0); } catch (const VariantOperationException&) { refused |= 2; }
      std::cout << "refused " << refused << " derived_calls " << derived.calls << "\n";
      return refused == 3 ? 0 : 1;
    } catch (const std::exception& e) { std::cerr << e.what() << "\n"; return -2; }
  }
  if (argc < 2) { std::cerr << "usage: gt_mpi_gather_shaped <query.json> [page_size]\n"; return -1; }
  if (std::string(argv[1]) == "--config-selftest" && argc >= 5) {
    // host only: loader + query JSON the way GenomicsDBBCFGenerator reads them (genomicsdb_bcf_generator.cc:44-53), then the ranges left
    try {
      const int rank = atoi(argv[4]);
      GenomicsDBImportConfig loader_config;
      loader_config.read_from_file(argv[3], rank);
      VariantQueryConfig query_config;
      query_config.update_from_loader(loader_config, rank);
      query_config.read_from_file(argv[2], rank);
      query_config.subset_query_column_ranges_based_on_partition(loader_config, rank);
      for (unsigned i = 0; i < query_config.get_num_column_intervals(); ++i) std::cout << query_config.get_column_begin(i) << "-" << query_config.get_column_end(i) << "\n";
      return 0;
    } catch (const std::exception& e) { std::cerr << e.what() << "\n"; return -2; }
  }
  const std::string json_config_file = argv[1];
  const size_t page_size = argc > 2 ? strtoull(argv[2], 0, 10) : 0u;
  const int my_world_mpi_rank = 0;
  try {
    VariantQueryConfig query_config;
    VCFAdapter vcf_adapter_base;
    VCFSerializedBufferAdapter serialized_vcf_adapter(true, true);
    auto& vcf_adapter = (page_size > 0u) ? dynamic_cast<VCFAdapter&>(serialized_vcf_adapter) : vcf_adapter_base;
    query_config.read_from_file(json_config_file, my_world_mpi_rank);
    if (page_size > 0u) query_config.set_combined_vcf_records_buffer_size_limit(page_size);
    vcf_adapter.initialize(query_config);
    const std::string workspace = query_config.get_workspace(my_world_mpi_rank);
    const std::string array_name = query_config.get_array_name(my_world_mpi_rank);
    VariantStorageManager sm(workspace, 10u * 1024u * 1024u);
    VariantQueryProcessor qp(&sm, array_name, query_config.get_vid_mapper());
    qp.do_query_bookkeeping(qp.get_array_schema(), query_config, query_config.get_vid_mapper(), true);
    scan_and_produce_Broad_GVCF(qp, query_config, vcf_adapter, query_config.get_vid_mapper(), my_world_mpi_rank);
    if (argc > 3) {   // the other two kinds of operator
      MyPerRecordOperator mine(&query_config.get_vid_mapper());
      bool refused = false;
      try { qp.scan_and_operate(qp.get_array_descriptor(), query_config, mine, 0, true, 0); } catch (const VariantOperationException&) { refused = true; }
      CountingOperator counter(&query_config.get_vid_mapper());
      qp.scan_and_operate(qp.get_array_descriptor(), query_config, counter, 0, true, 0);
      std::cerr << "per-record operator refused: " << (refused ? "yes" : "no") << ", batched hook: " << counter.pages << " pages, " << counter.bytes << " bytes\n";
    }
    if (getenv("GDBAMD_PRINT_PROFILE")) qp.get_profile_stats().print_stats(stderr);   // (the reference prints them under -DDO_PROFILING, query_variants.cc:455-459)
    sm.close_array(qp.get_array_descriptor());
  } catch (const std::exception& e) {
    std::cerr << "gt_mpi_gather_shaped: " << e.what() << "\n";
    return -1;
  }
  return 0;
}
