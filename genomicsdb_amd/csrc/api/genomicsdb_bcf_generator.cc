#include "genomicsdb_bcf_generator.h"

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <climits>
#include <thread>
#include <cstdlib>
#include <cstring>

namespace genomicsdb_amd {

CombineEngine::CombineEngine(const mini_json::Value& query_json, int device, const GenomicsDBImportConfig* loader, int rank) {
  if (loader) m_qc.update_from_loader(*loader, rank);
  m_qc.read_from_json(query_json, rank, "");
  m_qc.do_query_bookkeeping(m_qc.get_vid_mapper().get_num_callsets(), 0);
  std::string tmpl;
  if (!m_qc.get_vcf_header_filename().empty()) tmpl = mini_json::read_text_file(m_qc.get_vcf_header_filename());
  m_hp = build_combine_plan(m_qc, tmpl);
  m_pipe.reset(new DevicePipeline(m_hp, device));
  if (!m_qc.get_reference_genome().empty()) m_ref.initialize(m_qc.get_reference_genome());
}

void CombineEngine::stage_cells(const uint8_t* cells, uint64_t nbytes) {
  stage_cells_begin();
  stage_cells_append(cells, nbytes);
  stage_cells_end();
}

void CombineEngine::stage_cells_begin() {
  reference_cell_bytes = 0; num_cells = 0; has_cells = false; min_begin = INT64_MAX; max_end = 0;
  m_pipe->begin_staging();
}
void CombineEngine::stage_cells_append(const uint8_t* cells, uint64_t nbytes) {
  if (!m_layout) m_layout.reset(new CellStreamLayout(m_qc, m_hp));
  // the stream goes to HBM as it is and is taken apart there (DevicePipeline::append_cells)
  const DevicePipeline::CellStreamInfo info = m_pipe->append_cells(cells, nbytes, m_layout->schema, m_layout->attr_to_field, m_layout->row_map);
  reference_cell_bytes += info.reference_cell_bytes;
  num_cells += info.ncells;
  if (info.ncells > 0) { min_begin = std::min(min_begin, info.min_begin); max_end = std::max(max_end, info.max_end); }
}
void CombineEngine::stage_cells_end() {
  has_cells = num_cells > 0;
  m_pipe->finish_staging();
}

void CombineEngine::save_fragment(const std::string& path) {
  FragmentFileMeta meta;
  meta.reference_cell_bytes = reference_cell_bytes; meta.min_begin = min_begin; meta.max_end = max_end; meta.ncells = num_cells;
  m_pipe->save_fragment(path, meta);
}
void CombineEngine::load_fragment(const std::string& path) {
  // the file holds QUERY row indices: it only fits a query over all rows of the array in callset order
  const VariantQueryConfig& qc = m_qc;
  for (uint64_t q = 0; q < qc.get_num_rows_to_query(); ++q)
    if (qc.get_array_row_idx_for_query_row_idx(q) != (int64_t)q) throw GenomicsDBConfigException("a columnar fragment file serves queries over all rows only");
  const FragmentFileMeta meta = m_pipe->load_fragment(path);
  reference_cell_bytes = meta.reference_cell_bytes; min_begin = meta.min_begin; max_end = meta.max_end; num_cells = meta.ncells; has_cells = num_cells > 0;
}

void CombineEngine::stage_reference_for(int64_t qb, int64_t qe) {
  if (!m_ref.is_initialized() || !has_cells) return;
  int64_t b = std::max(qb, min_begin), e = std::min(qe, max_end);
  if (e < b) return;
  m_pipe->set_reference_window(b, m_ref.window(m_qc.get_vid_mapper(), b, e - b + 1));
}

GenomicsDBBCFGenerator::GenomicsDBBCFGenerator(const std::string& loader_config_file, const std::string& query_config_file, const char* chr,
                                               const int start, const int end, int my_rank, size_t buffer_capacity, size_t, const char* output_format,
                                               const bool produce_header_only, const bool, const bool)
    : m_buffer_capacity(buffer_capacity) {
  if (output_format && strlen(output_format) > 0)
    throw UnsupportedOnDeviceException(std::string("VCF output format \"") + output_format + "\": only text VCF (\"\") is produced by this build (SURVEY 8f-2)");
  GenomicsDBImportConfig loader;
  if (!loader_config_file.empty()) loader.read_from_file(loader_config_file, my_rank);
  // one process per GPU: the device is the launcher's LOCAL_RANK (torchrun / mpirun wrappers export it), else GDBAMD_DEVICE, else 0
  int device = 0;
  if (const char* e = getenv("GDBAMD_DEVICE")) device = atoi(e);
  else if (const char* e2 = getenv("LOCAL_RANK")) device = atoi(e2);
  m_engine.reset(new CombineEngine(mini_json::parse_file(query_config_file), device, loader_config_file.empty() ? nullptr : &loader, my_rank));
  VariantQueryConfig& qc = m_engine->query_config();
  if (chr && strlen(chr) > 0u) {
    ContigInfo ci;
    if (!qc.get_vid_mapper().get_contig_info(chr, ci)) throw GenomicsDBJNIException(std::string("Could not find TileDB column interval for contig: ") + chr);
    qc.set_column_interval_to_query(ci.m_tiledb_column_offset + (int64_t)start - 1, ci.m_tiledb_column_offset + (int64_t)end - 1);
  }
  // array storage of this build: <workspace>/<array>/cells.bin = begin-cells in the reference binary-cell layout
  // (the Intel TileDB fork's on-disk format is not available: SURVEY 8(f) rank 1)
  const std::string dir = qc.get_workspace(my_rank) + "/" + qc.get_array_name(my_rank);
  bool loaded = false;
  {  // columnar fragment first: file -> HBM copies, no parsing
    const std::string frag = dir + "/fragment.gdbamd";
    if (FILE* fp = fopen(frag.c_str(), "rb")) {
      fclose(fp);
      try { m_engine->load_fragment(frag); loaded = true; } catch (const GenomicsDBConfigException&) { loaded = false; }   // row subset: take the cells
    }
  }
  if (!loaded) {
    std::vector<uint8_t> cells = read_binary_file(dir + "/cells.bin");
    m_engine->stage_cells(cells.data(), cells.size());
  }
  common_init(produce_header_only);
}

GenomicsDBBCFGenerator::GenomicsDBBCFGenerator(const std::string& query_json_text, const uint8_t* cells, uint64_t nbytes, size_t buffer_capacity, bool produce_header_only)
    : m_buffer_capacity(buffer_capacity) {
  m_engine.reset(new CombineEngine(mini_json::parse(query_json_text), 0));
  m_engine->stage_cells(cells, nbytes);
  common_init(produce_header_only);
}

#define GEN_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw GenomicsDBDeviceException(std::string(#expr) + " failed: " + hipGetErrorString(_e)); } while (0)

void GenomicsDBBCFGenerator::common_init(bool produce_header_only) {
  m_produce_header_only = produce_header_only;
  const std::string& h = m_engine->plan().header_text;  // first bytes = header (vcf_adapter.cc:475-488)
  m_header.assign(h.begin(), h.end());
  m_next_read_idx = 0;
  if (produce_header_only) m_done = true;
  // ring geometry: GDBAMD_RING_SLOT_MB x GDBAMD_RING_SLOTS of pinned memory, allocated when the first page is drained
  size_t slot_mb = 64, nslots = 4;
  if (const char* e = getenv("GDBAMD_RING_SLOT_MB")) slot_mb = (size_t)std::max(1, atoi(e));
  if (const char* e = getenv("GDBAMD_RING_SLOTS")) nslots = (size_t)std::max(2, atoi(e));
  m_slot_bytes = slot_mb << 20;
  m_ring.resize(nslots);
}

GenomicsDBBCFGenerator::~GenomicsDBBCFGenerator() {
  if (m_copy_stream) (void)hipStreamSynchronize((hipStream_t)m_copy_stream);
  for (auto& s : m_ring) { if (s.done) (void)hipEventDestroy((hipEvent_t)s.done); if (s.host) (void)hipHostFree(s.host); }
  m_engine.reset();   // (the pipeline synchronises its stream before it lets go of the arenas the events refer to)
  for (auto& e : m_arena_read) if (e) (void)hipEventDestroy((hipEvent_t)e);
  if (m_copy_stream) (void)hipStreamDestroy((hipStream_t)m_copy_stream);
}

int64_t GenomicsDBBCFGenerator::max_window_columns() const {
  if (const char* e = getenv("GDBAMD_MAX_WINDOW_COLUMNS")) return std::max<int64_t>(1, atoll(e));
  // ~64 bytes of HBM per (column, sample) for text, resolved matrix and tables: keep a piece under ~48 GB
  const int64_t n = std::max<int64_t>(1, (int64_t)m_engine->plan().plan.num_query_rows);
  return std::max<int64_t>(1000, (int64_t)(48ll << 30) / (n * 64));
}

// Bytes of VCF text the device assembles per page.  Deliberately NOT the caller's buffer_capacity (1 MiB by default): a page
// costs a handful of kernel launches and one event wait, which a 1 MiB page cannot amortise.  GDBAMD_DEVICE_PAGE_BYTES /
// GDBAMD_DEVICE_PAGE_MB override (the tests page with a few hundred bytes).
uint64_t GenomicsDBBCFGenerator::device_page_bytes() const {
  if (const char* e = getenv("GDBAMD_DEVICE_PAGE_BYTES")) return (uint64_t)std::max<long long>(1, atoll(e));
  if (const char* e = getenv("GDBAMD_DEVICE_PAGE_MB")) return (uint64_t)std::max<long long>(1, atoll(e)) << 20;
  return 2048ull << 20;
}

// Makes m_page the next page of the stream (kernels complete) and starts the assembly of the one behind it.
bool GenomicsDBBCFGenerator::advance_page() {
  if (m_done) return false;
  DevicePipeline& pipe = m_engine->pipeline();
  VariantQueryConfig& qc = m_engine->query_config();
  const unsigned nint = std::max(1u, qc.get_num_column_intervals());
  const uint64_t page_cap = device_page_bytes();
  bool have = false;
  if (m_next_valid) { m_page = m_next_page; m_next_valid = false; have = true; }
  while (!have) {
    if (!m_interval_active) {
      if (m_query_column_interval_idx >= nint) { m_done = true; return false; }
      const int64_t qb = qc.get_num_column_intervals() ? qc.get_column_begin(m_query_column_interval_idx) : 0;
      const int64_t qe = qc.get_num_column_intervals() ? qc.get_column_end(m_query_column_interval_idx) : INT64_MAX - 1;
      // a wide interval (a whole chromosome) is worked off in pieces whose buffers fit HBM; the cuts sit right before cell
      // begins, where the sweep closes its interval anyway, so the stream is byte-identical to the unsplit one
      if (m_piece_begin < qb || m_piece_begin > qe) m_piece_begin = qb;
      const int64_t pe = pipe.split_point(m_piece_begin, qe, max_window_columns());
      m_engine->stage_reference_for(m_piece_begin, pe);
      pipe.prepare_interval(m_piece_begin, pe);
      m_piece_end = pe;
      m_interval_end = qe;
      m_interval_active = true;
    }
    if (pipe.begin_page(page_cap, m_arena_toggle, &m_page)) { m_arena_toggle ^= 1; have = true; }
    else {
      m_interval_active = false;
      if (m_piece_end >= m_interval_end) { ++m_query_column_interval_idx; m_piece_begin = INT64_MIN; }
      else m_piece_begin = m_piece_end + 1;
    }
  }
  pipe.finish_page(m_page);
  // the page behind it goes into the other arena while this one drains (pages of the next piece follow once this piece is done:
  // prepare_interval needs the host)
  if (m_interval_active) {
    if (pipe.begin_page(page_cap, m_arena_toggle, &m_next_page)) { m_arena_toggle ^= 1; m_next_valid = true; }
    else {
      m_interval_active = false;
      if (m_piece_end >= m_interval_end) { ++m_query_column_interval_idx; m_piece_begin = INT64_MIN; }
      else m_piece_begin = m_piece_end + 1;
    }
  }
  m_page_valid = true;
  m_page_off = 0;
  ++m_drain.pages;
  return true;
}

void GenomicsDBBCFGenerator::fill_ring() {
  if (m_produce_header_only) return;
  while (m_ring_count < m_ring.size()) {
    if (!m_page_valid) {
      const auto t0 = std::chrono::steady_clock::now();
      const bool more = advance_page();
      m_drain.seconds_producing += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (!more) return;
      if (m_page.nbytes == 0) { m_page_valid = false; continue; }
    }
    if (!m_copy_stream) {
      hipStream_t cs;
      GEN_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
      m_copy_stream = cs;
      for (auto& e : m_arena_read) { hipEvent_t ev; GEN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); e = ev; }
    }
    RingSlot& slot = m_ring[(m_ring_head + m_ring_count) % m_ring.size()];
    const uint64_t n = std::min<uint64_t>(m_slot_bytes, m_page.nbytes - m_page_off);
    if (slot.cap < n) {   // grow-only; small queries never pin the full slot size
      if (slot.host) GEN_HIP(hipHostFree(slot.host));
      slot.host = nullptr; slot.cap = 0;
      const size_t want = (size_t)std::min<uint64_t>(m_slot_bytes, std::max<uint64_t>(n, 1u << 20));
      GEN_HIP(hipHostMalloc((void**)&slot.host, want, hipHostMallocDefault));
      slot.cap = want;
    }
    if (!slot.done) { hipEvent_t ev; GEN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); slot.done = ev; }
    GEN_HIP(hipMemcpyAsync(slot.host, m_page.dev + m_page_off, n, hipMemcpyDeviceToHost, (hipStream_t)m_copy_stream));
    GEN_HIP(hipEventRecord((hipEvent_t)slot.done, (hipStream_t)m_copy_stream));
    slot.len = n; slot.waited = false;
    ++m_ring_count; ++m_drain.chunks; m_drain.bytes += n;
    m_page_off += n;
    if (m_page_off >= m_page.nbytes) {   // last chunk of the page: the arena may be overwritten once this copy is through
      GEN_HIP(hipEventRecord((hipEvent_t)m_arena_read[m_page.arena], (hipStream_t)m_copy_stream));
      m_engine->pipeline().set_arena_release_event(m_page.arena, m_arena_read[m_page.arena]);
      m_page_valid = false;
    }
  }
}

void GenomicsDBBCFGenerator::pop_slot() {
  m_ring_head = (m_ring_head + 1) % m_ring.size();
  --m_ring_count;
  m_next_read_idx = 0;
}

namespace {
// host copy out of the pinned ring: one core moves ~10-20 GB/s, less than the link delivers, so large reads are split over a few threads
void ring_copy(uint8_t* dst, const uint8_t* src, size_t n) {
  constexpr size_t kParallelFrom = 8u << 20;
  if (n < kParallelFrom) { memcpy(dst, src, n); return; }
  const unsigned nt = 4;
  std::thread th[nt - 1];
  const size_t part = ((n / nt) + 4095) & ~(size_t)4095;
  for (unsigned i = 1; i < nt; ++i) {
    const size_t b = std::min(n, i * part), e = std::min(n, (i + 1) * part);
    th[i - 1] = std::thread([=]() { if (e > b) memcpy(dst + b, src + b, e - b); });
  }
  memcpy(dst, src, std::min(n, part));
  for (auto& t : th) t.join();
}
}  // namespace

size_t GenomicsDBBCFGenerator::read_and_advance(uint8_t* dst, size_t offset, size_t n) {
  if (n == SIZE_MAX) { fill_ring(); return 0; }
  size_t total = 0;
  while (total < n) {
    if (m_next_read_idx < m_header.size()) {          // the header comes first
      const size_t k = std::min(n - total, m_header.size() - m_next_read_idx);
      if (dst) memcpy(dst + offset + total, m_header.data() + m_next_read_idx, k);
      m_next_read_idx += k; total += k;
      if (m_next_read_idx >= m_header.size()) fill_ring();   // start draining while the caller digests the header
      continue;
    }
    if (!m_header.empty()) { m_header.clear(); m_next_read_idx = 0; }   // header consumed: the index now runs over the ring's head slot
    if (m_ring_count == 0) { fill_ring(); if (m_ring_count == 0) break; }
    RingSlot& slot = m_ring[m_ring_head];
    if (!slot.waited) {
      const auto t0 = std::chrono::steady_clock::now();
      GEN_HIP(hipEventSynchronize((hipEvent_t)slot.done));
      m_drain.seconds_waiting_for_copies += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      slot.waited = true;
    }
    const size_t k = std::min(n - total, slot.len - m_next_read_idx);
    if (dst && k) ring_copy(dst + offset + total, slot.host + m_next_read_idx, k);
    m_next_read_idx += k; total += k;
    if (m_next_read_idx >= slot.len) { pop_slot(); fill_ring(); }
  }
  return total;
}

GenomicsDBBCFGenerator::RWBuffer GenomicsDBBCFGenerator::get_read_batch() {
  if (m_next_read_idx < m_header.size()) return RWBuffer{m_header.data(), m_header.size(), m_next_read_idx};
  if (!m_header.empty()) { m_header.clear(); m_next_read_idx = 0; }
  if (m_ring_count == 0) fill_ring();
  if (m_ring_count == 0) return RWBuffer{nullptr, 0, 0};
  RingSlot& slot = m_ring[m_ring_head];
  if (!slot.waited) { GEN_HIP(hipEventSynchronize((hipEvent_t)slot.done)); slot.waited = true; }
  return RWBuffer{slot.host, slot.len, m_next_read_idx};
}

uint8_t GenomicsDBBCFGenerator::read_next_byte() {
  uint8_t b = 0xFF;
  if (read_and_advance(&b, 0, 1) != 1) return 0xFF;
  return b;
}

}  // namespace genomicsdb_amd
