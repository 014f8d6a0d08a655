#include "genomicsdb_bcf_generator.h"

#include <hip/hip_runtime_api.h>

#include <climits>
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

void GenomicsDBBCFGenerator::common_init(bool produce_header_only) {
  m_produce_header_only = produce_header_only;
  const std::string& h = m_engine->plan().header_text;  // first bytes = header (vcf_adapter.cc:475-488)
  m_buffer.assign(h.begin(), h.end());
  m_next_read_idx = 0;
  if (produce_header_only) m_done = true;
}

int64_t GenomicsDBBCFGenerator::max_window_columns() const {
  if (const char* e = getenv("GDBAMD_MAX_WINDOW_COLUMNS")) return std::max<int64_t>(1, atoll(e));
  // ~64 bytes of HBM per (column, sample) for text, resolved matrix and tables: keep a piece under ~48 GB
  const int64_t n = std::max<int64_t>(1, (int64_t)m_engine->plan().plan.num_query_rows);
  return std::max<int64_t>(1000, (int64_t)(48ll << 30) / (n * 64));
}

void GenomicsDBBCFGenerator::produce_next_batch() {
  m_buffer.clear();
  m_next_read_idx = 0;
  if (m_done) return;
  VariantQueryConfig& qc = m_engine->query_config();
  const unsigned nint = std::max(1u, qc.get_num_column_intervals());
  while (m_buffer.empty()) {
    if (!m_interval_active) {
      if (m_query_column_interval_idx >= nint) { m_done = true; return; }
      const int64_t qb = qc.get_num_column_intervals() ? qc.get_column_begin(m_query_column_interval_idx) : 0;
      const int64_t qe = qc.get_num_column_intervals() ? qc.get_column_end(m_query_column_interval_idx) : INT64_MAX - 1;
      // a wide interval (a whole chromosome) is worked off in pieces whose buffers fit HBM; the cuts sit right before cell
      // begins, where the sweep closes its interval anyway, so the stream is byte-identical to the unsplit one
      if (m_piece_begin < qb || m_piece_begin > qe) m_piece_begin = qb;
      const int64_t pe = m_engine->pipeline().split_point(m_piece_begin, qe, max_window_columns());
      m_engine->stage_reference_for(m_piece_begin, pe);
      m_engine->pipeline().prepare_interval(m_piece_begin, pe);
      m_piece_end = pe;
      m_interval_end = qe;
      m_interval_active = true;
    }
    const char* dev = nullptr;
    uint64_t n = 0;
    if (m_engine->pipeline().next_page(m_buffer_capacity, &dev, &n)) {
      m_buffer.resize(n);
      if (n && hipMemcpy(m_buffer.data(), dev, n, hipMemcpyDeviceToHost) != hipSuccess) throw GenomicsDBDeviceException("page copy to host failed");
    } else {
      m_interval_active = false;
      if (m_piece_end >= m_interval_end) { ++m_query_column_interval_idx; m_piece_begin = INT64_MIN; }
      else m_piece_begin = m_piece_end + 1;
    }
  }
}

size_t GenomicsDBBCFGenerator::read_and_advance(uint8_t* dst, size_t offset, size_t n) {
  size_t total = 0;
  if (n == SIZE_MAX) { produce_next_batch(); return 0; }
  while (total < n && !end()) {
    if (m_next_read_idx >= m_buffer.size()) { produce_next_batch(); continue; }
    size_t k = std::min(n - total, m_buffer.size() - m_next_read_idx);
    if (dst) memcpy(dst + offset + total, m_buffer.data() + m_next_read_idx, k);
    m_next_read_idx += k;
    total += k;
    if (m_next_read_idx >= m_buffer.size()) produce_next_batch();
  }
  return total;
}

uint8_t GenomicsDBBCFGenerator::read_next_byte() {
  uint8_t b = 0xFF;
  if (read_and_advance(&b, 0, 1) != 1) return 0xFF;
  return b;
}

}  // namespace genomicsdb_amd
