#include "genomicsdb_operators.h"
#include "../host/vcf_index.h"
#include "../kernels/gdb_bgzf.h"

#include <typeinfo>

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace genomicsdb_amd {

const char* GTProfileStats::stat_name(unsigned i) {
  static const char* const names[GT_NUM_STATS] = {"GT_NUM_CELLS", "GT_NUM_CELLS_IN_LEFT_SWEEP", "GT_NUM_VALID_CELLS_IN_QUERY", "GT_NUM_ATTR_CELLS_ACCESSED",
                                                  "GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS", "GT_NUM_OPERATOR_INVOCATIONS"};
  return i < GT_NUM_STATS ? names[i] : "";
}
void GTProfileStats::print_stats(FILE* f) const {    // (reference: GTProfileStats::print_stats, query_variants.cc:90-110)
  fprintf(f, "stat_name,sum,sum_sq,mean,std-dev\n");
  for (unsigned i = 0; i < GT_NUM_STATS; ++i) {
    fprintf(f, "%s,%llu,%.6g", stat_name(i), (unsigned long long)m_sum[i], m_sum_sq[i]);
    if (m_num_queries == 0) fprintf(f, ",*,*\n");
    else {
      const double mean = (double)m_sum[i] / (double)m_num_queries;
      double var = m_sum_sq[i] / (double)m_num_queries - mean * mean;
      if (var < 0) var = -var;
      fprintf(f, ",%.6g,%.6g\n", mean, std::sqrt(var));
    }
  }
}

// ---- adapters -----------------------------------------------------------------------------------------------------------------
VCFAdapter::~VCFAdapter() {
  // BGZF output ("z" / "b"): the file ends with the empty EOF block, as htslib's bgzf_close leaves it
  if (m_out && m_wrote_bytes && (m_output_format == "z" || m_output_format == "b")) (void)fwrite(kBgzfEofBlock, 1, sizeof(kBgzfEofBlock), m_out);
  if (m_out && m_owns_out) fclose(m_out); else if (m_out) fflush(m_out);
  // "index_output_VCF": the reference builds the index here, from the finished file (vcf_adapter.cc:275-295: tbx_index_build for "z", bcf_index_build(.., 14) for "b")
  if (m_out && m_owns_out && m_index_output && m_wrote_bytes && (m_output_format == "z" || m_output_format == "b")) {
    try { if (m_output_format == "z") build_tbi_index(m_output_filename); else build_csi_index(m_output_filename); }
    catch (const std::exception& e) { fprintf(stderr, "WARNING: error in creating index for output file %s: %s\n", m_output_filename.c_str(), e.what()); }
  }
}

void VCFAdapter::initialize(const VariantQueryConfig& qc) {
  m_output_format = qc.get_vcf_output_format();
  m_buffer_limit = qc.get_combined_vcf_records_buffer_size_limit();
  if (!m_open_output) return;
  const std::string& fn = qc.get_vcf_output_filename();
  if (fn.empty() || fn == "-") { m_out = stdout; m_owns_out = false; }
  else { m_out = fopen(fn.c_str(), "wb"); m_owns_out = true; if (!m_out) throw VCFAdapterException("cannot open " + fn); }
  m_output_filename = fn; m_index_output = qc.index_output_VCF();
}
void VCFAdapter::handoff(const uint8_t* bytes, size_t n) {
  if (!m_out) throw VCFAdapterException("VCFAdapter::initialize() has not opened an output");
  if (n && fwrite(bytes, 1, n, m_out) != n) throw VCFAdapterException("short write");
  if (n) m_wrote_bytes = true;
}
void VCFSerializedBufferAdapter::handoff(const uint8_t* bytes, size_t n) {
  if (!m_rw_buffer) throw VCFAdapterException("VCFSerializedBufferAdapter: set_buffer() first");
  RWBuffer& b = *m_rw_buffer;
  if (b.m_buffer.size() < b.m_num_valid_bytes + n) b.m_buffer.resize(std::max(2 * b.m_buffer.size() + 1, b.m_num_valid_bytes + n));   // (the reference grows it the same way)
  if (n) memcpy(&b.m_buffer[b.m_num_valid_bytes], bytes, n);
  b.m_num_valid_bytes += n;
}
void VCFSerializedBufferAdapter::do_output() {
  if (!m_rw_buffer || !m_out) return;
  const RWBuffer& b = *m_rw_buffer;
  if (b.m_num_valid_bytes && fwrite(&b.m_buffer[0], 1, b.m_num_valid_bytes, m_out) != b.m_num_valid_bytes) throw VCFAdapterException("short write");
}

// ---- operators ----------------------------------------------------------------------------------------------------------------
void SingleVariantOperatorBase::operate(Variant&, const VariantQueryConfig&) {
  throw VariantOperationException("per-record operate() is not called on the device path: the scan hands whole pages to the recognised BroadCombinedGVCFOperator, "
                                  "or to BatchedVariantOperatorBase::operate_on_page()");
}
const Variant& GA4GHOperator::get_remapped_variant() const {
  throw VariantOperationException("remapped Variant objects are not materialised on the host (the records are combined in HBM)");
}
BroadCombinedGVCFOperator::BroadCombinedGVCFOperator(VCFAdapter& vcf_adapter, const VidMapper& id_mapper, const VariantQueryConfig& query_config,
                                                     const unsigned max_alt, const bool use_missing_values_only_not_vector_end)
    : GA4GHOperator(query_config, id_mapper, max_alt), m_vcf_adapter(&vcf_adapter), m_use_missing_values_not_vector_end(use_missing_values_only_not_vector_end) {}

// ---- the scan -------------------------------------------------------------------------------------------------------------------
struct VariantQueryProcessor::Engine {
  std::unique_ptr<CombineEngine> eng;
  std::string format;
  bool use_missing = false, header_done = false;
  unsigned max_alt = 0;
  // drain: pages are assembled alternately in the pipeline's two arenas; the finished one leaves in chunks over a copy stream
  // into two pinned buffers while the page behind it is being assembled (the C ABI's ring, at the size this caller needs)
  static constexpr size_t kChunk = (size_t)64 << 20;
  hipStream_t copy = nullptr;
  hipEvent_t chunk_done[2] = {nullptr, nullptr}, arena_read[2] = {nullptr, nullptr};
  uint8_t* pin[2] = {nullptr, nullptr};
  size_t pin_cap[2] = {0, 0};
  int toggle = 0;
  bool next_valid = false;
  DevicePipeline::PageTicket next;
  ~Engine() {
    if (copy) (void)hipStreamSynchronize(copy);
    eng.reset();
    for (int i = 0; i < 2; ++i) { if (chunk_done[i]) (void)hipEventDestroy(chunk_done[i]); if (arena_read[i]) (void)hipEventDestroy(arena_read[i]); if (pin[i]) (void)hipHostFree(pin[i]); }
    if (copy) (void)hipStreamDestroy(copy);
  }
};

#define OPS_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw GenomicsDBDeviceException(std::string(#expr) + " failed: " + hipGetErrorString(_e)); } while (0)

VariantQueryProcessor::VariantQueryProcessor(VariantStorageManager* sm, const std::string& array_name, const VidMapper&) : m_storage_manager(sm), m_array_name(array_name) {
  if (!sm) throw VariantOperationException("VariantQueryProcessor needs a VariantStorageManager");
}
VariantQueryProcessor::~VariantQueryProcessor() {}

void VariantQueryProcessor::do_query_bookkeeping(const VariantArraySchema&, VariantQueryConfig& query_config, const VidMapper& vid_mapper, const bool) const {
  if (!query_config.is_bookkeeping_done()) query_config.do_query_bookkeeping(vid_mapper.get_num_callsets(), 0);
}

void VariantQueryProcessor::scan_and_operate(const int, const VariantQueryConfig& query_config, SingleVariantOperatorBase& variant_operator, unsigned column_interval_idx,
                                             bool, VariantQueryProcessorScanState* scan_state) const {
  // The built-in is recognised by its EXACT type: a class derived from BroadCombinedGVCFOperator may override operate() (or any
  // of the protected hooks the reference's operator calls per record), and nothing here would ever call it - it is refused like
  // any other per-record operator instead of silently getting the built-in's semantics.
  auto* gvcf = typeid(variant_operator) == typeid(BroadCombinedGVCFOperator) ? static_cast<BroadCombinedGVCFOperator*>(&variant_operator) : nullptr;
  auto* batched = dynamic_cast<BatchedVariantOperatorBase*>(&variant_operator);
  if (!gvcf && !batched)
    throw VariantOperationException("this operator's per-record operate() cannot run on the device path; the built-in BroadCombinedGVCFOperator (the class itself, not a "
                                    "class derived from it) is recognised, other operators take pages through BatchedVariantOperatorBase::operate_on_page()");
  if (batched) gvcf = nullptr;
  std::string format = gvcf ? gvcf->get_vcf_adapter().get_output_format() : std::string();
  // the buffer adapter of the reference never compresses: bcf_hdr_serialize / bcf_serialize only look at m_is_bcf (vcf_adapter.cc:475-505),
  // so "z" is plain VCF text and "b" plain BCF2 in an RWBuffer; BGZF is for the file-writing VCFAdapter (which also writes the EOF block)
  if (gvcf && dynamic_cast<VCFSerializedBufferAdapter*>(&gvcf->get_vcf_adapter())) { if (format == "z") format = ""; else if (format == "b") format = "bu"; }
  const bool use_missing = gvcf && gvcf->use_missing_values_only_not_vector_end();
  if (!m_engine || m_engine->format != format || m_engine->use_missing != use_missing) {
    m_engine.reset(new Engine);
    m_engine->format = format; m_engine->use_missing = use_missing;
    int device = 0;
    if (const char* e = getenv("GDBAMD_DEVICE")) device = atoi(e); else if (const char* e2 = getenv("LOCAL_RANK")) device = atoi(e2);
    // the engine is configured from the caller's query configuration (its JSON form): attributes, rows, intervals, switches
    m_engine->eng.reset(new CombineEngine(query_config, device, format, use_missing, gvcf ? gvcf->get_max_diploid_alt_alleles_that_can_be_genotyped() : 0u));
    m_engine->eng->open_array(m_storage_manager->get_workspace() + "/" + m_array_name);
  }
  Engine& E = *m_engine;
  CombineEngine& eng = *E.eng;
  VariantQueryProcessorScanState local_state;
  VariantQueryProcessorScanState& st = scan_state ? *scan_state : local_state;
  if (gvcf && !E.header_done) {       // the reference's operator constructor writes the header through its adapter
    const HostPlan& hp = eng.plan();
    const auto* ser = dynamic_cast<VCFSerializedBufferAdapter*>(&gvcf->get_vcf_adapter());
    std::string h = hp.plan.bcf_mode ? hp.bcf_header_bytes(ser ? ser->keep_idx_fields_in_bcf_header() : true) : hp.header_text;
    if (hp.bgzf) h = bgzf_compress_host(h);      // "z" / "b": the header is a BGZF block of its own; the adapter writes the EOF block when it closes
    gvcf->get_vcf_adapter().handoff((const uint8_t*)h.data(), h.size());
    E.header_done = true;
  }
  const VariantQueryConfig& qc = eng.query_config();
  const unsigned nint = std::max(1u, qc.get_num_column_intervals());
  if (column_interval_idx >= nint) { st.m_done = true; return; }
  if (!st.m_started) {
    st.m_started = true;
    st.m_piece_begin = qc.get_num_column_intervals() ? qc.get_column_begin(column_interval_idx) : 0;
    st.m_interval_end = qc.get_num_column_intervals() ? qc.get_column_end(column_interval_idx) : INT64_MAX - 1;
    st.m_piece_active = false;
    E.next_valid = false;
  }
  const uint64_t page_bytes = gvcf && dynamic_cast<VCFSerializedBufferAdapter*>(&gvcf->get_vcf_adapter()) ? std::max<uint64_t>(1, query_config.get_combined_vcf_records_buffer_size_limit())
                                                                                                          : (uint64_t)256 << 20;
  // (eng.pipeline() is looked up at every use: cover() may swap the engine's two pipelines - overlapped staging)
  if (!E.copy) {
    OPS_HIP(hipStreamCreateWithFlags(&E.copy, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) { OPS_HIP(hipEventCreateWithFlags(&E.chunk_done[i], hipEventDisableTiming)); OPS_HIP(hipEventCreateWithFlags(&E.arena_read[i], hipEventDisableTiming)); }
  }
  for (;;) {
    if (!st.m_piece_active) {
      if (st.m_piece_begin > st.m_interval_end) { st.m_done = true; return; }
      const CombineEngine::Coverage cov = eng.cover(st.m_piece_begin);
      const int64_t n = std::max<int64_t>(1, (int64_t)eng.plan().plan.num_query_rows);
      const int64_t pe = eng.pipeline().split_point(st.m_piece_begin, std::min(st.m_interval_end, cov.hi), std::max<int64_t>(1000, (int64_t)(48ll << 30) / (n * 64)));
      eng.stage_reference_for(st.m_piece_begin, pe);
      eng.pipeline().prepare_interval(st.m_piece_begin, pe);
      {
        const IntervalStats& is = eng.pipeline().interval_stats();
        for (unsigned i = 0; i < GTProfileStats::GT_NUM_STATS; ++i) m_stats.update_stat(i, is.gt_profile[i]);
        m_stats.increment_num_queries();
      }
      st.m_piece_begin = pe + 1;             // (the piece in flight is remembered by the pipeline)
      st.m_piece_active = true;
      E.next_valid = false;
    }
    DevicePipeline::PageTicket page;
    if (E.next_valid) { page = E.next; E.next_valid = false; }
    else if (eng.pipeline().begin_page(page_bytes, E.toggle, &page)) E.toggle ^= 1;
    else { st.m_piece_active = false; continue; }
    eng.pipeline().finish_page(page);
    // the page behind it is assembled in the other arena while this one is handed over
    if (eng.pipeline().begin_page(page_bytes, E.toggle, &E.next)) { E.toggle ^= 1; E.next_valid = true; }
    else st.m_piece_active = false;          // the piece is exhausted; the page in hand is its last
    if (batched) batched->operate_on_page(page.dev, page.nbytes, qc.get_num_column_intervals() ? qc.get_column_begin(column_interval_idx) : 0, st.m_interval_end);
    else {
      // device -> pinned host in chunks on the copy stream, two buffers: chunk k + 1 is in flight while chunk k is handed off
      const uint64_t nb = page.nbytes;
      int k = 0;
      uint64_t prev_len = 0;
      for (uint64_t off = 0; off < nb; off += Engine::kChunk, ++k) {
        const int slot = k & 1;
        const size_t len = (size_t)std::min<uint64_t>(Engine::kChunk, nb - off);
        if (E.pin_cap[slot] < len) {
          if (E.pin[slot]) OPS_HIP(hipHostFree(E.pin[slot]));
          E.pin[slot] = nullptr; E.pin_cap[slot] = 0;
          const size_t want = std::min(Engine::kChunk, std::max<size_t>(len, (size_t)1 << 20));
          OPS_HIP(hipHostMalloc((void**)&E.pin[slot], want, hipHostMallocDefault));
          E.pin_cap[slot] = want;
        }
        OPS_HIP(hipMemcpyAsync(E.pin[slot], page.dev + off, len, hipMemcpyDeviceToHost, E.copy));
        OPS_HIP(hipEventRecord(E.chunk_done[slot], E.copy));
        if (k > 0) { OPS_HIP(hipEventSynchronize(E.chunk_done[slot ^ 1])); gvcf->get_vcf_adapter().handoff(E.pin[slot ^ 1], (size_t)prev_len); }
        prev_len = len;
      }
      if (k > 0) {
        OPS_HIP(hipEventRecord(E.arena_read[page.arena & 1], E.copy));      // the arena may be written again once the last chunk has left
        eng.pipeline().set_arena_release_event(page.arena, E.arena_read[page.arena & 1]);
        OPS_HIP(hipEventSynchronize(E.chunk_done[(k - 1) & 1]));
        gvcf->get_vcf_adapter().handoff(E.pin[(k - 1) & 1], (size_t)prev_len);
      }
      // The reference's scan pauses on overflow() only when the caller passed a scan state to come back with (query_variants.cc:
      // 453-468); without one there is nowhere to resume from, so the interval runs to its end.
      if (scan_state && gvcf->overflow()) return;
    }
  }
}

}  // namespace genomicsdb_amd
