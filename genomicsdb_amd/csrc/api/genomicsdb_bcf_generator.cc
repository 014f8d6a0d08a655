#include "genomicsdb_bcf_generator.h"
#include <functional>
#include <map>
#include "../kernels/gdb_bgzf.h"

#include <hip/hip_runtime_api.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <climits>
#include <thread>
#include <cstdlib>
#include <cstring>

namespace genomicsdb_amd {

CombineEngine::CombineEngine(const mini_json::Value& query_json, int device, const GenomicsDBImportConfig* loader, int rank, const std::string& output_format,
                             bool use_missing_values_only_not_vector_end) {
  if (loader) m_qc.update_from_loader(*loader, rank);
  m_qc.read_from_json(query_json, rank, "");
  if (loader) m_qc.subset_query_column_ranges_based_on_partition(*loader, rank);
  m_qc.do_query_bookkeeping(m_qc.get_vid_mapper().get_num_callsets(), 0);
  std::string tmpl;
  if (!m_qc.get_vcf_header_filename().empty()) tmpl = mini_json::read_text_file(m_qc.get_vcf_header_filename());
  m_hp = build_combine_plan(m_qc, tmpl, output_format, use_missing_values_only_not_vector_end);
  m_device = device;
  m_pipe.reset(new DevicePipeline(m_hp, device));
  if (!m_qc.get_reference_genome().empty()) m_ref.initialize(m_qc.get_reference_genome());
}

CombineEngine::CombineEngine(const VariantQueryConfig& query_config, int device, const std::string& output_format, bool use_missing_values_only_not_vector_end,
                             unsigned max_diploid_alt_alleles) : m_qc(query_config) {
  if (max_diploid_alt_alleles) m_qc.set_max_diploid_alt_alleles_that_can_be_genotyped(max_diploid_alt_alleles);
  if (!m_qc.is_bookkeeping_done()) m_qc.do_query_bookkeeping(m_qc.get_vid_mapper().get_num_callsets(), 0);
  std::string tmpl;
  if (!m_qc.get_vcf_header_filename().empty()) tmpl = mini_json::read_text_file(m_qc.get_vcf_header_filename());
  m_hp = build_combine_plan(m_qc, tmpl, output_format, use_missing_values_only_not_vector_end);
  m_device = device;
  m_pipe.reset(new DevicePipeline(m_hp, device));
  if (!m_qc.get_reference_genome().empty()) m_ref.initialize(m_qc.get_reference_genome());
}

void CombineEngine::stage_cells(const uint8_t* cells, uint64_t nbytes) {
  stage_cells_begin();
  stage_cells_append(cells, nbytes);
  stage_cells_end();
}

void CombineEngine::stage_cells_begin() {
  (void)join_prefetch(false);
  reference_cell_bytes = 0; num_cells = 0; has_cells = false; min_begin = INT64_MAX; max_end = 0;
  if (m_src.fd >= 0) { ::close(m_src.fd); m_src.fd = -1; }
  m_src.kind = SRC_NONE;
  m_pipe->begin_staging();
}
void CombineEngine::stage_cells_append(const uint8_t* cells, uint64_t nbytes) {
  // the stream goes to HBM as it is and is taken apart there (DevicePipeline::append_cells)
  const CellStreamLayout& L = layout();
  const DevicePipeline::CellStreamInfo info = m_pipe->append_cells(cells, nbytes, L.schema, L.attr_to_field, L.row_map);
  reference_cell_bytes += info.reference_cell_bytes;
  num_cells += info.ncells;
  if (info.ncells > 0) { min_begin = std::min(min_begin, info.min_begin); max_end = std::max(max_end, info.max_end); }
}
void CombineEngine::stage_cells_end() {
  has_cells = num_cells > 0;
  m_pipe->finish_staging();
}

const CellStreamLayout& CombineEngine::layout() {
  if (!m_layout) m_layout.reset(new CellStreamLayout(m_qc, m_hp));
  return *m_layout;
}
std::vector<ColumnLayout> CombineEngine::expected_columns() {
  const CellStreamLayout& L = layout();
  std::vector<ColumnLayout> out((size_t)m_hp.plan.nfields);
  for (size_t ai = 0; ai < L.schema.attrs.size(); ++ai) {
    const int f = L.attr_to_field[ai];
    if (f < 0) continue;
    out[(size_t)f].var = L.schema.attrs[ai].var; out[(size_t)f].elem_size = L.schema.attrs[ai].elem_size; out[(size_t)f].fixed_num = L.schema.attrs[ai].num;
  }
  return out;
}
// what a fragment file depends on besides the cells: the array schema and the row <-> callset assignment
uint64_t CombineEngine::schema_hash() {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
  const CellStreamLayout& L = layout();
  for (const auto& a : L.schema.attrs) { mix(a.name.data(), a.name.size() + 1); const int32_t d[4] = {(int32_t)a.elem, a.var ? 1 : 0, a.num, a.elem_size}; mix(d, sizeof(d)); }
  const VidMapper& vid = m_qc.get_vid_mapper();
  for (int64_t r = 0; r < vid.get_num_callsets(); ++r) { std::string name; vid.get_callset_name(r, name); mix(name.data(), name.size() + 1); }
  for (const CallSetInfo& c : vid.get_callsets()) {   // which sample of which file feeds which row
    mix(c.m_name.data(), c.m_name.size() + 1); mix(c.m_filename.data(), c.m_filename.size() + 1);
    const int64_t d[2] = {c.m_row_idx, c.m_idx_in_file}; mix(d, sizeof(d));
  }
  return h ? h : 1;
}

void CombineEngine::save_fragment(const std::string& path, bool compress) {
  FragmentFileMeta meta;
  meta.reference_cell_bytes = reference_cell_bytes; meta.min_begin = min_begin; meta.max_end = max_end; meta.ncells = num_cells;
  meta.schema_hash = schema_hash();
  m_pipe->save_fragment(path, meta, compress);
}
void CombineEngine::load_fragment(const std::string& path) {
  (void)join_prefetch(false);
  // the file holds QUERY row indices: it only fits a query over all rows of the array in callset order
  const VariantQueryConfig& qc = m_qc;
  for (uint64_t q = 0; q < qc.get_num_rows_to_query(); ++q)
    if (qc.get_array_row_idx_for_query_row_idx(q) != (int64_t)q) throw GenomicsDBConfigException("a columnar fragment file serves queries over all rows only");
  const FragmentFileMeta meta = m_pipe->load_fragment(path, expected_columns(), schema_hash());
  reference_cell_bytes = meta.reference_cell_bytes; min_begin = meta.min_begin; max_end = meta.max_end; num_cells = meta.ncells; has_cells = num_cells > 0;
  m_src = Source();
}

// ---- windowed array access ------------------------------------------------------------------------------------------------
CombineEngine::~CombineEngine() {
  try { (void)join_prefetch(false); } catch (...) {}
  if (m_src.fd >= 0) ::close(m_src.fd);
  if (m_src.chunk) (void)hipHostFree(m_src.chunk);
}

uint64_t CombineEngine::staging_budget_bytes() const {
  if (const char* e = getenv("GDBAMD_STAGE_BUDGET_BYTES")) return (uint64_t)std::max<long long>(1, atoll(e));
  if (const char* e = getenv("GDBAMD_STAGE_BUDGET_MB")) return (uint64_t)std::max<long long>(1, atoll(e)) << 20;
  return 8192ull << 20;   // cell bytes per window; the columnar fragment, its row index and the per-piece buffers come on top
}

void CombineEngine::open_memory_cells(const uint8_t* cells, uint64_t nbytes) {
  (void)join_prefetch(false);
  if (m_src.fd >= 0) ::close(m_src.fd);
  const Source keep = m_src;
  m_src = Source();
  m_src.chunk = keep.chunk; m_src.chunk_cap = keep.chunk_cap;
  m_src.kind = SRC_CELLS_MEMORY; m_src.mem = cells; m_src.size = nbytes;
  rewind_source();
}

void CombineEngine::open_cell_callback(CellChunkFn fn, void* user) {
  (void)join_prefetch(false);
  if (m_src.fd >= 0) ::close(m_src.fd);
  const Source keep = m_src;
  m_src = Source();
  m_src.chunk = keep.chunk; m_src.chunk_cap = keep.chunk_cap;
  m_src.kind = SRC_CALLBACK; m_src.fn = fn; m_src.fn_user = user;
  rewind_source();
}

void CombineEngine::open_array(const std::string& dir) {
  (void)join_prefetch(false);
  if (m_src.fd >= 0) ::close(m_src.fd);
  const Source keep = m_src;
  m_src = Source();
  m_src.chunk = keep.chunk; m_src.chunk_cap = keep.chunk_cap;
  const std::string frag = dir + "/fragment.gdbamd", cells = dir + "/cells.bin";
  struct stat cs;
  const bool have_cells = ::stat(cells.c_str(), &cs) == 0;
  // columnar fragment first: file -> HBM copies, no parsing.  It has to serve this query (all rows, callset order), to have been
  // written under the same vid / callset mapping and, when it names a cells.bin, from the one that is there now.
  bool identity = true;
  for (uint64_t q = 0; q < m_qc.get_num_rows_to_query(); ++q) if (m_qc.get_array_row_idx_for_query_row_idx(q) != (int64_t)q) identity = false;
  struct stat fs;
  if (identity && ::stat(frag.c_str(), &fs) == 0) {
    try {
      const FragmentFileMeta meta = m_pipe->open_fragment_file(frag, expected_columns(), schema_hash());
      const bool stale = have_cells && meta.source_bytes != 0 && (meta.source_bytes != (uint64_t)cs.st_size || meta.source_mtime != (int64_t)cs.st_mtime);
      if (stale) m_pipe->close_fragment_file();
      else {
        m_src.kind = SRC_FRAGMENT_FILE; m_src.ncells_file = meta.ncells; m_src.size = (uint64_t)fs.st_size;
        min_begin = meta.min_begin; max_end = meta.max_end;
      }
    } catch (const std::exception&) {
      if (!have_cells) throw;    // nothing to fall back to
    }
  }
  if (m_src.kind == SRC_NONE) {
    m_src.fd = ::open(cells.c_str(), O_RDONLY);
    if (m_src.fd < 0) throw std::runtime_error("cannot open " + cells);
    m_src.kind = SRC_CELLS_FILE; m_src.size = (uint64_t)cs.st_size;
  }
  rewind_source();
}

void CombineEngine::rewind_source() {
  m_src.cursor = 0; m_src.cell_cursor = 0; m_src.window_valid = false; m_src.eof = false; m_src.cov = Coverage{INT64_MIN, INT64_MIN};
  m_src.pending_valid = false; m_window_eof = false;
  reference_cell_bytes = 0; num_cells = 0; has_cells = false;
  if (m_src.kind != SRC_FRAGMENT_FILE) { min_begin = INT64_MAX; max_end = 0; }
}

// Stages the next column window of the source into `dst`; the cells still live at `carry_from` come from the fragment `carry_src`
// holds (dst itself, or - overlapped staging - the pipeline that is computing on the previous window).  Touches the source's
// cursors and `r` only: the engine's counters and the coverage are updated by whoever takes the window into use.
void CombineEngine::stage_window(DevicePipeline& dst, DevicePipeline& carry_src, int64_t carry_from, StagedWindow& r) {
  Source& S = m_src;
  DevicePipeline* const m_pipe = &dst;                      // (everything below stages into dst)
  uint64_t& reference_cell_bytes = r.reference_cell_bytes;
  int64_t& min_begin = r.min_begin; int64_t& max_end = r.max_end;
  dst.begin_staging_from(carry_src, carry_from);
  // The first window of a source has nothing to hide behind (no window is being computed yet): a quarter of the budget gets the
  // device started sooner, the windows behind it are staged under its kernels.  (Sources parsed on the way in only; budgets below
  // 256 MiB - the tests' - are taken as they are.)
  const uint64_t full_budget = staging_budget_bytes();
  const bool ramp = !S.window_valid && S.cursor == 0 && S.cell_cursor == 0 && overlap_enabled() && full_budget >= ((uint64_t)256 << 20);
  const uint64_t budget = ramp ? full_budget / 4 : full_budget;
  int64_t next_begin = INT64_MAX, new_cells = 0;
  if (S.kind == SRC_FRAGMENT_FILE) {
    const DevicePipeline::FragmentWindow w = m_pipe->append_fragment_cells(S.cell_cursor, budget);
    S.cell_cursor = w.c1; next_begin = w.next_begin; new_cells = w.ncells;
    reference_cell_bytes += w.reference_cell_bytes;
    S.eof = S.cell_cursor >= S.ncells_file;
  } else if (S.kind == SRC_CALLBACK) {
    const CellStreamLayout& L = layout();
    uint64_t taken = 0;
    for (;;) {
      if (!S.pending_valid) {
        S.pending = nullptr; S.pending_bytes = 0;
        if (!S.fn(S.fn_user, &S.pending, &S.pending_bytes)) { S.eof = true; break; }
        S.pending_valid = true;
        if (S.pending_bytes == 0) { S.pending_valid = false; continue; }
      }
      if (taken > 0 && taken + S.pending_bytes > budget) break;     // the chunk in hand opens the next window
      const DevicePipeline::CellStreamInfo info = m_pipe->append_cells(S.pending, S.pending_bytes, L.schema, L.attr_to_field, L.row_map);
      reference_cell_bytes += info.reference_cell_bytes;
      new_cells += info.ncells;
      if (info.ncells > 0) { min_begin = std::min(min_begin, info.min_begin); max_end = std::max(max_end, info.max_end); }
      taken += S.pending_bytes;
      S.pending_valid = false;
    }
    if (S.pending_valid) { int64_t col; memcpy(&col, S.pending + 8, 8); next_begin = col; }
  } else {
    // the binary cell stream in sub-chunks (whole begin columns each): read -> walk the sizes -> HBM, until the window is full
    const CellStreamLayout& L = layout();
    static const uint64_t sub_mb = []() { const char* e = getenv("GDBAMD_STAGE_SUB_MB"); return e && *e ? (uint64_t)std::max<long long>(1, std::min<long long>(2048, atoll(e))) : (uint64_t)1024; }();
    const uint64_t sub = std::min<uint64_t>(budget, sub_mb << 20);
    uint64_t taken = 0;
    std::vector<uint64_t> offs;
    uint64_t want = sub;
    while (taken < budget && S.cursor < S.size) {
      const uint64_t n = std::min<uint64_t>(want, S.size - S.cursor);
      const bool to_end = S.cursor + n >= S.size;
      const uint8_t* buf;
      if (S.kind == SRC_CELLS_MEMORY) buf = S.mem + S.cursor;
      else {
        if (S.chunk_cap < n) {
          if (S.chunk) (void)hipHostFree(S.chunk);
          S.chunk = nullptr; S.chunk_cap = 0;
          if (hipHostMalloc((void**)&S.chunk, (size_t)n, hipHostMallocDefault) != hipSuccess) throw GenomicsDBDeviceException("cannot pin the cell read buffer");
          S.chunk_cap = (size_t)n;
        }
        for (uint64_t got = 0; got < n;) {
          const ssize_t k = ::pread(S.fd, S.chunk + got, (size_t)(n - got), (off_t)(S.cursor + got));
          if (k <= 0) throw std::runtime_error("short read from cells.bin");
          got += (uint64_t)k;
        }
        buf = S.chunk;
      }
      // the sizes are walked on the device (GDBAMD_HOST_WALK=1: by one host thread, for comparison)
      static const bool host_walk = getenv("GDBAMD_HOST_WALK") != nullptr;
      DevicePipeline::CellWalk wk;
      DevicePipeline::CellStreamInfo info;
      if (host_walk) {
        wk = DevicePipeline::walk_cells(buf, n, L.row_map, !to_end, offs);
        if (wk.single_column) { want *= 2; continue; }     // one begin column wider than the sub-chunk: offer more bytes
        info = m_pipe->append_cells(buf, wk.bytes_taken, L.schema, L.attr_to_field, L.row_map, &offs, &wk);
      } else {
        info = m_pipe->append_cells(buf, n, L.schema, L.attr_to_field, L.row_map, nullptr, nullptr, !to_end, &wk);
        if (wk.single_column) { want *= 2; continue; }
      }
      reference_cell_bytes += info.reference_cell_bytes;
      new_cells += info.ncells;
      if (info.ncells > 0) { min_begin = std::min(min_begin, info.min_begin); max_end = std::max(max_end, info.max_end); }
      S.cursor += wk.bytes_taken; taken += wk.bytes_taken;
      next_begin = to_end ? INT64_MAX : wk.next_begin;
      want = sub;
    }
    S.eof = S.cursor >= S.size;
    if (S.eof) next_begin = INT64_MAX;
  }
  m_pipe->finish_staging();
  r.new_cells = new_cells;
  r.has_cells = new_cells + m_pipe->carried_cells() > 0;
  r.carry_from = carry_from;
  r.hi = next_begin == INT64_MAX ? INT64_MAX - 1 : next_begin - 1;
  r.eof = S.eof;
}

void CombineEngine::take_window(const StagedWindow& r) {
  Source& S = m_src;
  reference_cell_bytes += r.reference_cell_bytes;
  num_cells += r.new_cells;
  has_cells = r.has_cells;
  if (r.min_begin != INT64_MAX) min_begin = std::min(min_begin, r.min_begin);
  max_end = std::max(max_end, r.max_end);
  if (r.carry_from != INT64_MIN && min_begin != INT64_MAX) min_begin = std::min(min_begin, r.carry_from);
  S.cov.lo = S.window_valid ? S.cov.hi + 1 : INT64_MIN;
  S.cov.hi = r.hi;
  S.window_valid = true;
  m_window_eof = r.eof;
  ++windows_staged;
}

// Overlapped staging: while the caller computes on the window in m_pipe, a thread stages the next one into m_pipe2 (its own HIP
// stream, device blocks from its own pool, carried cells read from m_pipe's fragment); cover() then only swaps the two.  For the
// sources that are parsed on the way in (cells.bin, cells in memory, cell callback); GDBAMD_OVERLAP_STAGING=0 switches it off.
bool CombineEngine::overlap_enabled() const {
  if (const char* e = getenv("GDBAMD_OVERLAP_STAGING")) if (*e == '0') return false;
  return m_src.kind == SRC_CELLS_FILE || m_src.kind == SRC_CELLS_MEMORY || m_src.kind == SRC_CALLBACK;
}
void CombineEngine::start_prefetch() {
  if (m_prefetch.joinable() || m_src.eof || !overlap_enabled()) return;
  if (!m_pipe2) m_pipe2.reset(new DevicePipeline(m_hp, m_device));
  (void)layout();                                            // (built here, not on the thread)
  const int64_t carry_from = m_src.cov.hi + 1;
  m_next = StagedWindow();
  m_prefetch_error = nullptr;
  m_prefetch = std::thread([this, carry_from]() {
    try { stage_window(*m_pipe2, *m_pipe, carry_from, m_next); }
    catch (...) { m_prefetch_error = std::current_exception(); }
  });
}
bool CombineEngine::join_prefetch(bool take) {
  if (!m_prefetch.joinable()) return false;
  m_prefetch.join();
  if (m_prefetch_error) { std::exception_ptr e = m_prefetch_error; m_prefetch_error = nullptr; std::rethrow_exception(e); }
  if (!take) return false;
  std::swap(m_pipe, m_pipe2);
  ++pipeline_generation;
  if (m_user_ref && !m_pipe_has_user_ref[pipeline_generation & 1]) { m_pipe->set_reference_window(m_user_ref_begin, m_user_ref_bases); m_pipe_has_user_ref[pipeline_generation & 1] = true; }
  take_window(m_next);
  return true;
}

void CombineEngine::advance_window() {
  if (join_prefetch(true)) { start_prefetch(); return; }
  StagedWindow r;
  stage_window(*m_pipe, *m_pipe, m_src.window_valid ? m_src.cov.hi + 1 : INT64_MIN, r);
  take_window(r);
  start_prefetch();
}


CombineEngine::Coverage CombineEngine::cover(int64_t column) {
  if (m_src.kind == SRC_NONE) return Coverage{INT64_MIN, INT64_MAX - 1};   // staged by hand (stage_cells*, load_fragment, adopt): all of it
  if (m_src.window_valid && column < m_src.cov.lo) {                       // an earlier interval than the current window: start over
    if (m_src.kind == SRC_CALLBACK) throw GenomicsDBConfigException("a cell callback source is read once, front to back");
    (void)join_prefetch(false);                                              // (a window staged ahead is dropped with the cursors)
    rewind_source();
  }
  // (m_window_eof: the window in use is the source's last one; m_src.eof may already be true for the one being staged ahead)
  while (!m_src.window_valid || (column > m_src.cov.hi && !m_window_eof)) advance_window();
  return m_src.cov;
}

void CombineEngine::column_histogram(uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, std::vector<uint64_t>& counts) {
  if (bin_size == 0 || hist_end < hist_begin) throw GenomicsDBConfigException("column histogram: empty range or bin size 0");
  counts.assign((size_t)((hist_end - hist_begin) / bin_size + 1), 0);
  if (m_qc.get_num_column_intervals() > 0) {
    // the reference hands the operator the cells of the QUERY's column intervals (iterate_over_cells(ad, query_config, op),
    // tools/src/gt_mpi_gather.cc:404-411): per interval the cells that begin in it and the intervals that reach its begin, the
    // selection of --print-calls (for_each_interval_text)
    for (unsigned i = 0; i < m_qc.get_num_column_intervals(); ++i) {
      const int64_t qb = m_qc.get_column_begin(i), qe = m_qc.get_column_end(i);
      bool first_piece = true;
      for (int64_t pos = qb; pos <= qe;) {
        const Coverage cov = cover(pos);
        const int64_t piece[2] = {pos, std::min(cov.hi, qe)};
        m_pipe->column_histogram(hist_begin, hist_end, bin_size, counts.data(), counts.size(), true, piece, first_piece);
        first_piece = false;
        if (piece[1] >= qe || m_src.kind == SRC_NONE || m_window_eof || cov.hi >= INT64_MAX - 1) break;
        pos = piece[1] + 1;
      }
    }
    return;
  }
  int64_t col = INT64_MIN;
  for (;;) {                                                      // window by window; a window's carried-over cells are not counted again
    const Coverage cov = cover(col);
    m_pipe->column_histogram(hist_begin, hist_end, bin_size, counts.data(), counts.size(), true);
    if (m_src.kind == SRC_NONE || m_window_eof || cov.hi >= INT64_MAX - 1) break;
    col = cov.hi + 1;
  }
}

// the cells of every query interval through DevicePipeline::cells_text, piece by piece when the array passes through HBM in column windows
// (a piece behind the first prints only the cells that begin in it)
void CombineEngine::for_each_interval_text(int mode, int arg, const std::function<void(int64_t, int64_t, const std::string&)>& fn) {
  std::vector<std::pair<int64_t, int64_t>> ivs;
  for (unsigned i = 0; i < m_qc.get_num_column_intervals(); ++i) ivs.emplace_back(m_qc.get_column_begin(i), m_qc.get_column_end(i));
  const bool whole_array = ivs.empty();       // a scan of the whole array has no interval begin to intersect (genomicsdb_iterators.cc:188-190)
  if (whole_array) ivs.emplace_back(0, INT64_MAX - 1);
  std::vector<int64_t> q2a;                   // (the reference prints the row of the ARRAY; a staged cell knows its query row)
  { const CellStreamLayout& L = layout(); for (size_t r = 0; r < L.row_map.size(); ++r) { const int32_t q = L.row_map[r]; if (q >= 0) { if ((size_t)q >= q2a.size()) q2a.resize((size_t)q + 1, 0); q2a[(size_t)q] = (int64_t)r; } } }
  for (const auto& iv : ivs) {
    std::string text;
    bool first_piece = true;
    for (int64_t pos = iv.first; pos <= iv.second;) {
      const Coverage cov = cover(pos);
      m_pipe->set_array_rows(q2a);            // (cover() may have swapped the engine's two pipelines)
      const int64_t hi = std::min(cov.hi, iv.second);
      text += m_pipe->cells_text(pos, hi, mode, arg, first_piece && !whole_array);
      first_piece = false;
      if (hi >= iv.second || m_src.kind == SRC_NONE || m_window_eof || cov.hi >= INT64_MAX - 1) break;
      pos = hi + 1;
    }
    fn(iv.first, iv.second, text);
  }
}

std::string CombineEngine::print_calls() {
  const std::string ip = "    ";
  std::string o = "{\n" + ip + "\"variant_calls\": [\n";
  const std::string p0 = ip + ip, p1 = p0 + ip;
  unsigned printed = 0;
  for_each_interval_text(0, 16, [&](int64_t lo, int64_t hi, const std::string& cells) {
    if (cells.empty()) return;                                    // (an interval without a cell prints nothing: the header is printed WITH its first cell)
    if (printed) o += "\n" + p1 + "]\n" + p0 + "},\n";
    o += p0 + "{\n" + p1 + "\"query_interval\": [ " + std::to_string(lo) + ", " + std::to_string(hi) + " ],\n" + p1 + "\"variant_calls\": [\n";
    o.append(cells, 2, std::string::npos);                        // (every cell comes with ",\n" in front of it)
    ++printed;
  });
  if (printed) o += "\n" + p1 + "]\n" + p0 + "}";
  o += "\n" + ip + "]\n}\n";
  return o;
}

std::string CombineEngine::print_csv() {
  std::string o;
  for_each_interval_text(1, 0, [&](int64_t, int64_t, const std::string& t) { o += t; });
  return o;
}

std::string CombineEngine::print_allele_counts() {
  if (m_hp.plan.f_GT < 0) throw GenomicsDBConfigException("GT field must be queried for AlleleCountOperator");   // (variant_operations.cc:909-911)
  const int gt_step = m_hp.plan.field[m_hp.plan.f_GT].length == GDB_VL_PP ? 2 : 1;
  std::string o;
  for_each_interval_text(2, gt_step, [&](int64_t, int64_t, const std::string& t) {
    std::map<int64_t, std::map<std::pair<std::string, std::string>, uint64_t>> counts;      // (m_column_to_REF_ALT_to_count_vec's element of this interval)
    for (size_t b = 0; b < t.size();) {
      const size_t e = t.find('\n', b), t1 = t.find('\t', b), t2 = t.find('\t', t1 + 1);
      ++counts[strtoll(t.c_str() + b, nullptr, 10)][std::make_pair(t.substr(t1 + 1, t2 - t1 - 1), t.substr(t2 + 1, e - t2 - 1))];
      b = e + 1;
    }
    for (const auto& col : counts)
      for (const auto& ra : col.second) o += std::to_string(col.first) + " " + ra.first.first + " " + ra.first.second + " " + std::to_string(ra.second) + "\n";
  });
  return o;
}

void CombineEngine::set_reference_window(int64_t begin, const std::string& bases) {
  m_user_ref = true; m_user_ref_begin = begin; m_user_ref_bases = bases;
  m_pipe->set_reference_window(begin, bases);
  m_pipe_has_user_ref[pipeline_generation & 1] = true; m_pipe_has_user_ref[(pipeline_generation + 1) & 1] = false;
  for (Lane& L : m_lanes) L.has_user_ref = false;
}

void CombineEngine::stage_reference_on(DevicePipeline& pipe, int64_t qb, int64_t qe) {
  if (!m_ref.is_initialized() || !has_cells) return;
  int64_t b = std::max(qb, min_begin), e = std::min(qe, max_end);
  if (e < b) return;
  pipe.set_reference_window(b, m_ref.window(m_qc.get_vid_mapper(), b, e - b + 1));
}
void CombineEngine::stage_reference_for(int64_t qb, int64_t qe) { stage_reference_on(*m_pipe, qb, qe); }

std::vector<IntervalStats> CombineEngine::run_intervals(const std::vector<std::pair<int64_t, int64_t>>& intervals, uint64_t arena_bytes, int lanes,
                                                        const std::function<void(size_t, const char*, uint64_t)>& on_page) {
  struct PageCtx { const std::function<void(size_t, const char*, uint64_t)>* fn; size_t index; };
  const PageCallback relay = [](void* user, const char* dev, uint64_t nbytes) { PageCtx* c = (PageCtx*)user; (*c->fn)(c->index, dev, nbytes); };
  const size_t n = intervals.size();
  std::vector<IntervalStats> out(n);
  if (n == 0) return out;
  lanes = std::max(1, std::min<int>({lanes, 4, (int)n}));
  // A lane pipeline that does not exist yet will allocate its own page arenas, entry table, matrix and sweep buffers (grow-only): no more
  // NEW lanes than the free HBM holds (lane_footprint_bytes: the estimate bench.py used to make on the caller's side), else the run would
  // end in a hipMalloc failure half-way through.  Lanes that exist keep what they hold.
  {
    int64_t widest = 1;
    for (const auto& iv : intervals) widest = std::max<int64_t>(widest, iv.second - iv.first + 1);
    const uint64_t per_lane = lane_footprint_bytes(widest, arena_bytes);
    int existing = 0;
    for (const Lane& L : m_lanes) if (L.pipe) ++existing;
    size_t free_b = 0, total_b = 0;
    if (lanes - 1 > existing && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const int asked = lanes;
      while (lanes - 1 > existing && (uint64_t)(lanes - 1 - existing) * per_lane + (4ull << 30) > (uint64_t)free_b) --lanes;
      if (lanes < asked)
        fprintf(stderr, "genomicsdb_amd: run_intervals: %d lanes asked, %d used (%.1f GB free, ~%.1f GB per new lane)\n", asked, lanes, free_b / 1e9, per_lane / 1e9);
    }
  }
  std::vector<DevicePipeline*> pipes((size_t)lanes, nullptr);
  pipes[0] = m_pipe.get();
  if ((int)m_lanes.size() < lanes - 1) m_lanes.resize((size_t)lanes - 1);
  for (int l = 1; l < lanes; ++l) {
    Lane& L = m_lanes[(size_t)l - 1];
    if (!L.pipe) L.pipe.reset(new DevicePipeline(m_hp, m_device));
    if (L.src != m_pipe.get() || L.gen != m_pipe->fragment_generation()) {     // another fragment since the lane last ran: adopt it (the cells stay where they are)
      L.pipe->adopt_fragment(m_pipe->fragment_view());
      L.src = m_pipe.get(); L.gen = m_pipe->fragment_generation();
    }
    if (m_user_ref && !L.has_user_ref) { L.pipe->set_reference_window(m_user_ref_begin, m_user_ref_bases); L.has_user_ref = true; }
    pipes[(size_t)l] = L.pipe.get();
  }
  for (DevicePipeline* p : pipes) p->set_page_priority(lanes > 1);
  // (the priority stream is for pipelines that share the device: a later run_interval on the engine's own pipeline is alone again)
  struct PriorityReset { std::vector<DevicePipeline*>& v; ~PriorityReset() { for (DevicePipeline* p : v) p->set_page_priority(false); } } priority_reset{pipes};
  std::vector<std::exception_ptr> errors((size_t)lanes);
  auto work = [&](int l) {
    try {
      for (size_t i = (size_t)l; i < n; i += (size_t)lanes) {
        if (!m_user_ref) stage_reference_on(*pipes[(size_t)l], intervals[i].first, intervals[i].second);
        PageCtx pc{&on_page, i};
        out[i] = pipes[(size_t)l]->run_interval(intervals[i].first, intervals[i].second, arena_bytes, on_page ? relay : (PageCallback) nullptr, &pc);
      }
    } catch (...) { errors[(size_t)l] = std::current_exception(); }
  };
  std::vector<std::thread> th;
  for (int l = 1; l < lanes; ++l) th.emplace_back(work, l);
  work(0);
  for (auto& t : th) t.join();
  for (auto& e : errors) if (e) std::rethrow_exception(e);
  return out;
}

uint64_t CombineEngine::lane_footprint_bytes(int64_t interval_columns, uint64_t arena_bytes) const {
  // ~45 bytes of output and up to ~18 bytes of tables (entry slots, resolved matrix, chunk sizes, sweep buffers) per sample and position of
  // the interval, measured at the widths of BASELINE configs[1] and [2]; the page arena is capped by arena_bytes (two of them alternate, but
  // the second is only allocated when a page overflows into it: counted once here like the first)
  const uint64_t cells = std::max<uint64_t>(1, m_qc.get_num_rows_to_query()) * (uint64_t)std::max<int64_t>(1, interval_columns);
  return std::min<uint64_t>(arena_bytes, 45ull * cells + (1ull << 30)) + 18ull * cells + (2ull << 30);
}

void CombineEngine::release_lanes() { m_lanes.clear(); }

GenomicsDBBCFGenerator::GenomicsDBBCFGenerator(const std::string& loader_config_file, const std::string& query_config_file, const char* chr,
                                               const int start, const int end, int my_rank, size_t buffer_capacity, size_t, const char* output_format,
                                               const bool produce_header_only, const bool use_missing_values_only_not_vector_end, const bool keep_idx_fields_in_bcf_header,
                                               const bool bgzf_stream)
    : m_buffer_capacity(buffer_capacity) {
  std::string fmt = output_format ? output_format : "";
  if (!bgzf_stream) { if (fmt == "z") fmt = ""; else if (fmt == "b") fmt = "bu"; }   // (VCFSerializedBufferAdapter never compresses, vcf_adapter.cc:475-505)
  output_format = fmt.c_str();
  GenomicsDBImportConfig loader;
  if (!loader_config_file.empty()) loader.read_from_file(loader_config_file, my_rank);
  // one process per GPU: the device is the launcher's LOCAL_RANK (torchrun / mpirun wrappers export it), else GDBAMD_DEVICE, else 0
  int device = 0;
  if (const char* e = getenv("GDBAMD_DEVICE")) device = atoi(e);
  else if (const char* e2 = getenv("LOCAL_RANK")) device = atoi(e2);
  m_engine.reset(new CombineEngine(mini_json::parse_file(query_config_file), device, loader_config_file.empty() ? nullptr : &loader, my_rank,
                                   output_format ? output_format : "", use_missing_values_only_not_vector_end));
  VariantQueryConfig& qc = m_engine->query_config();
  if (chr && strlen(chr) > 0u) {
    ContigInfo ci;
    if (!qc.get_vid_mapper().get_contig_info(chr, ci)) throw GenomicsDBJNIException(std::string("Could not find TileDB column interval for contig: ") + chr);
    qc.set_column_interval_to_query(ci.m_tiledb_column_offset + (int64_t)start - 1, ci.m_tiledb_column_offset + (int64_t)end - 1);
  }
  // array storage of this build: <workspace>/<array>/fragment.gdbamd (columnar, validated) or cells.bin = begin-cells in the
  // reference binary-cell layout (the Intel TileDB fork's on-disk format is not available: SURVEY 8(f) rank 1); either is read
  // in column windows of the staging budget
  m_engine->open_array(qc.get_workspace(my_rank) + "/" + qc.get_array_name(my_rank));
  m_engine->cover(INT64_MIN);   // first window staged at construction, like the reference opens its array here
  common_init(produce_header_only, keep_idx_fields_in_bcf_header);
}

GenomicsDBBCFGenerator::GenomicsDBBCFGenerator(const std::string& query_json_text, const uint8_t* cells, uint64_t nbytes, size_t buffer_capacity, bool produce_header_only,
                                               const char* output_format, bool use_missing_values_only_not_vector_end, bool keep_idx_fields_in_bcf_header)
    : m_buffer_capacity(buffer_capacity) {
  m_engine.reset(new CombineEngine(mini_json::parse(query_json_text), 0, nullptr, 0, output_format ? output_format : "", use_missing_values_only_not_vector_end));
  if (nbytes > m_engine->staging_budget_bytes()) {   // several windows: the stream comes back for the bytes, so it keeps them
    m_owned_cells.assign(cells, cells + nbytes);
    cells = m_owned_cells.data();
  }
  m_engine->open_memory_cells(cells, nbytes);
  m_engine->cover(INT64_MIN);                         // the first window (for most arrays: all of it) is staged here; malformed cells fail here
  common_init(produce_header_only, keep_idx_fields_in_bcf_header);
}

#define GEN_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw GenomicsDBDeviceException(std::string(#expr) + " failed: " + hipGetErrorString(_e)); } while (0)

void GenomicsDBBCFGenerator::common_init(bool produce_header_only, bool keep_idx_fields_in_bcf_header) {
  m_produce_header_only = produce_header_only;
  // first bytes = header (vcf_adapter.cc:475-488): the VCF text, or "BCF\2\2" + length + text (IDX keys kept or dropped) + NUL
  std::string h = m_engine->plan().plan.bcf_mode ? m_engine->plan().bcf_header_bytes(keep_idx_fields_in_bcf_header) : m_engine->plan().header_text;
  if (m_engine->plan().bgzf) {
    // "z" / "b": the header is a BGZF block of its own (zlib on the host, a few KB); the body follows as blocks compressed on the
    // device; the empty EOF block ends the stream (SAM specification 4.1.2; htslib's bgzf_close writes it)
    h = bgzf_compress_host(h);
    if (produce_header_only) h.append((const char*)kBgzfEofBlock, sizeof(kBgzfEofBlock));
    else m_trailer.assign((const char*)kBgzfEofBlock, sizeof(kBgzfEofBlock));
  }
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
  // (m_engine->pipeline() is looked up at every use: cover() may swap the engine's two pipelines - overlapped staging)
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
      // (an array larger than the staging budget passes through HBM in column windows: a piece ends where the staged window does)
      const bool trace = getenv("GDBAMD_STREAM_TRACE") != nullptr;
      auto now = []() { return std::chrono::steady_clock::now(); };
      auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
      const auto t0 = now();
      const CombineEngine::Coverage cov = m_engine->cover(m_piece_begin);
      const auto t1 = now();
      const int64_t pe = m_engine->pipeline().split_point(m_piece_begin, std::min(qe, cov.hi), max_window_columns());
      const auto t2 = now();
      m_engine->stage_reference_for(m_piece_begin, pe);
      const auto t3 = now();
      m_engine->pipeline().prepare_interval(m_piece_begin, pe);
      if (trace) fprintf(stderr, "[gdbamd stream] piece [%lld, %lld]: cover %.3f s, split_point %.3f s, reference %.3f s, prepare_interval %.3f s\n", (long long)m_piece_begin, (long long)pe,
                         secs(t0, t1), secs(t1, t2), secs(t2, t3), secs(t3, now()));
      m_piece_end = pe;
      m_interval_end = qe;
      m_interval_active = true;
    }
    if (m_engine->pipeline().begin_page(page_cap, m_arena_toggle, &m_page)) { m_arena_toggle ^= 1; have = true; }
    else {
      m_interval_active = false;
      if (m_piece_end >= m_interval_end) { ++m_query_column_interval_idx; m_piece_begin = INT64_MIN; }
      else m_piece_begin = m_piece_end + 1;
    }
  }
  m_engine->pipeline().finish_page(m_page);
  m_page_owner = &m_engine->pipeline();
  // the page behind it goes into the other arena while this one drains (pages of the next piece follow once this piece is done:
  // prepare_interval needs the host)
  if (m_interval_active) {
    if (m_engine->pipeline().begin_page(page_cap, m_arena_toggle, &m_next_page)) { m_arena_toggle ^= 1; m_next_valid = true; }
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
      if (!more) {
        if (!m_trailer.empty()) {     // the bytes that close the stream (BGZF: the EOF block) travel as a last ring slot
          RingSlot& slot = m_ring[(m_ring_head + m_ring_count) % m_ring.size()];
          if (slot.cap < m_trailer.size()) {
            if (slot.host) GEN_HIP(hipHostFree(slot.host));
            slot.host = nullptr; slot.cap = 0;
            GEN_HIP(hipHostMalloc((void**)&slot.host, 4096, hipHostMallocDefault));
            slot.cap = 4096;
          }
          memcpy(slot.host, m_trailer.data(), m_trailer.size());
          slot.len = m_trailer.size(); slot.waited = true;       // (host bytes: nothing to wait for)
          ++m_ring_count;
          m_trailer.clear();
        }
        return;
      }
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
      m_page_owner->set_arena_release_event(m_page.arena, m_arena_read[m_page.arena]);   // (the pipeline that assembled the page: the engine may have swapped since)
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
