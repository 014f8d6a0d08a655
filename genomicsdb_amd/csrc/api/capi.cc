// capi.cc - extern "C" surface declared in include/genomicsdb_amd.h
#include "../../../include/genomicsdb_amd.h"

#include <functional>
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

#include "genomicsdb_bcf_generator.h"
#include "../kernels/gdb_bgzf.h"
#include "../host/vcf_importer.h"
#include "../host/vcf_index.h"

using namespace genomicsdb_amd;

namespace {
thread_local std::string g_last_error;
template <class F, class R> R guarded(F f, R on_error) {
  try { g_last_error.clear(); return f(); }
  catch (const std::exception& e) { g_last_error = e.what(); return on_error; }
  catch (...) { g_last_error = "unknown exception"; return on_error; }
}
struct EngineHandle { std::unique_ptr<CombineEngine> eng; std::string calls_text; int calls_text_mode = -1; };   // calls_text: the document of gdbamd_engine_print_cells(mode) between its sizing and its copying call
}  // namespace

extern "C" {

const char* gdb_mi355_last_error(void) { return g_last_error.c_str(); }
int gdb_mi355_device_count(void) { return DevicePipeline::device_count(); }

void* gdb_mi355_init(const char* loader_json_file, const char* query_json_file, const char* chr, int start, int end, int rank, uint64_t buffer_capacity,
                     uint64_t segment_size, int is_bcf, int produce_header_only, int use_missing, int keep_idx) {
  return guarded([&]() -> void* {
    return new GenomicsDBBCFGenerator(loader_json_file ? loader_json_file : "", query_json_file ? query_json_file : "", chr, start, end, rank, buffer_capacity,
                                      segment_size, is_bcf ? "bu" : "", produce_header_only != 0, is_bcf && use_missing, is_bcf && keep_idx);
  }, (void*)nullptr);
}
void* gdb_mi355_init_from_memory(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, uint64_t buffer_capacity, int produce_header_only) {
  return guarded([&]() -> void* { return new GenomicsDBBCFGenerator(std::string(query_json_text), cells, nbytes, buffer_capacity, produce_header_only != 0); }, (void*)nullptr);
}
void* gdb_mi355_init_from_memory_format(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, uint64_t buffer_capacity, int produce_header_only, int is_bcf,
                                        int use_missing, int keep_idx) {
  return guarded([&]() -> void* {
    return new GenomicsDBBCFGenerator(std::string(query_json_text), cells, nbytes, buffer_capacity, produce_header_only != 0, is_bcf ? "bu" : "", is_bcf && use_missing, keep_idx != 0);
  }, (void*)nullptr);
}
void* gdb_mi355_init_output_format(const char* loader_json_file, const char* query_json_file, const char* chr, int start, int end, int rank, uint64_t buffer_capacity,
                                   uint64_t segment_size, const char* output_format, int produce_header_only, int use_missing, int keep_idx) {
  return guarded([&]() -> void* {
    const std::string fmt = output_format ? output_format : "";
    const bool bcf = fmt == "bu" || fmt == "b";
    return new GenomicsDBBCFGenerator(loader_json_file ? loader_json_file : "", query_json_file ? query_json_file : "", chr, start, end, rank, buffer_capacity,
                                      segment_size, fmt.c_str(), produce_header_only != 0, bcf && use_missing, bcf && keep_idx, true);
  }, (void*)nullptr);
}
void* gdb_mi355_init_from_memory_output_format(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, uint64_t buffer_capacity, int produce_header_only,
                                               const char* output_format, int use_missing, int keep_idx) {
  return guarded([&]() -> void* {
    const std::string fmt = output_format ? output_format : "";
    const bool bcf = fmt == "bu" || fmt == "b";
    return new GenomicsDBBCFGenerator(std::string(query_json_text), cells, nbytes, buffer_capacity, produce_header_only != 0, fmt.c_str(), bcf && use_missing, keep_idx != 0);
  }, (void*)nullptr);
}
uint64_t gdb_mi355_close(void* h) { delete (GenomicsDBBCFGenerator*)h; return 0; }
uint64_t gdb_mi355_get_num_bytes_available(void* h) { return h ? ((GenomicsDBBCFGenerator*)h)->get_buffer_capacity() : 0; }
int gdb_mi355_read_next_byte(void* h) {
  if (!h) return -1;
  return guarded([&]() -> int { auto* g = (GenomicsDBBCFGenerator*)h; if (g->end()) return -1; return (int)g->read_next_byte(); }, -1);
}
int64_t gdb_mi355_read(void* h, uint8_t* dst, uint64_t offset, uint64_t n) {
  if (!h) return 0;
  return guarded([&]() -> int64_t { return (int64_t)((GenomicsDBBCFGenerator*)h)->read_and_advance(dst, offset, n); }, (int64_t)-1);
}
int64_t gdb_mi355_skip(void* h, uint64_t n) {
  if (!h) return 0;
  return guarded([&]() -> int64_t { return (int64_t)((GenomicsDBBCFGenerator*)h)->read_and_advance(nullptr, 0, n); }, (int64_t)-1);
}

int gdb_mi355_peek(void* h, const uint8_t** ptr, uint64_t* n) {
  if (!h || !ptr || !n) return -1;
  return guarded([&]() -> int {
    auto* g = (GenomicsDBBCFGenerator*)h;
    for (;;) {
      const GenomicsDBBCFGenerator::RWBuffer b = g->get_read_batch();
      if (b.m_num_valid_bytes > b.m_next_read_idx) { *ptr = b.m_buffer + b.m_next_read_idx; *n = b.m_num_valid_bytes - b.m_next_read_idx; return 1; }
      if (g->end() || b.m_buffer == nullptr) { *ptr = nullptr; *n = 0; return 0; }
      g->read_and_advance(nullptr, 0, SIZE_MAX);     // an exhausted batch: produce the next one
    }
  }, -1);
}
int gdb_mi355_get_stream_stats(void* h, gdb_mi355_stream_stats* out) {
  if (!h || !out) return -1;
  const GenomicsDBBCFGenerator::DrainStats& d = ((GenomicsDBBCFGenerator*)h)->drain_stats();
  out->pages = d.pages; out->chunks = d.chunks; out->bytes = d.bytes;
  out->seconds_waiting_for_copies = d.seconds_waiting_for_copies; out->seconds_producing = d.seconds_producing;
  return 0;
}

void* gdbamd_engine_create(const char* query_json_text, int device) {
  return guarded([&]() -> void* { auto* e = new EngineHandle; e->eng.reset(new CombineEngine(mini_json::parse(query_json_text), device)); return e; }, (void*)nullptr);
}
void* gdbamd_engine_create_format(const char* query_json_text, int device, int is_bcf, int use_missing) {
  return guarded([&]() -> void* {
    auto* e = new EngineHandle;
    e->eng.reset(new CombineEngine(mini_json::parse(query_json_text), device, nullptr, 0, is_bcf ? "bu" : "", is_bcf && use_missing));
    return e;
  }, (void*)nullptr);
}
void* gdbamd_engine_create_output_format(const char* query_json_text, int device, const char* output_format, int use_missing) {
  return guarded([&]() -> void* {
    const std::string fmt = output_format ? output_format : "";
    auto* e = new EngineHandle;
    e->eng.reset(new CombineEngine(mini_json::parse(query_json_text), device, nullptr, 0, fmt, (fmt == "bu" || fmt == "b") && use_missing));
    return e;
  }, (void*)nullptr);
}
// BGZF blocks of n host bytes, compressed by the device kernels (a utility and the test hook of kernels/gdb_bgzf.hip)
int gdbamd_bgzf_compress(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_len, float* ms_kernels) {
  return gdbamd_bgzf_compress_mode(src, n, dst, dst_cap, dst_len, ms_kernels, 0);
}
int gdbamd_bgzf_compress_mode(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_len, float* ms_kernels, int vcf_text) {
  return guarded([&]() -> int {
    if (DevicePipeline::device_count() <= 0) throw GenomicsDBDeviceException("no HIP device visible");
    const uint64_t bound = bgzf_bound(n);
    char* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)bound + 64) != hipSuccess) throw GenomicsDBDeviceException("hipMalloc failed");
    uint64_t got = 0;
    try {
      if (n && hipMemcpy(d, src, (size_t)n, hipMemcpyHostToDevice) != hipSuccess) throw GenomicsDBDeviceException("copy to the device failed");
      BgzfDeviceCompressor c;
      c.set_text(vcf_text != 0);
      got = c.compress(d, n, d, nullptr, ms_kernels);
      if (got > dst_cap) throw GenomicsDBDeviceException("destination too small: " + std::to_string(got) + " bytes needed");
      if (got && hipMemcpy(dst, d, (size_t)got, hipMemcpyDeviceToHost) != hipSuccess) throw GenomicsDBDeviceException("copy from the device failed");
    } catch (...) { (void)hipFree(d); throw; }
    (void)hipFree(d);
    *dst_len = got;
    return 0;
  }, -1);
}
uint64_t gdbamd_bgzf_bound(uint64_t n) { return bgzf_bound(n); }
void gdbamd_engine_destroy(void* e) { delete (EngineHandle*)e; }
int gdbamd_engine_num_fields(void* e) { return e ? ((EngineHandle*)e)->eng->plan().plan.nfields : -1; }
const char* gdbamd_engine_field_name(void* e, int f) {
  if (!e) return nullptr;
  const HostPlan& hp = ((EngineHandle*)e)->eng->plan();
  return (f >= 0 && f < hp.plan.nfields) ? hp.field_names[(size_t)f].c_str() : nullptr;
}
int gdbamd_engine_field_info(void* e, int f, int* elem_type, int* is_var, int* fixed_num) {
  if (!e) return -1;
  const CombinePlan& pl = ((EngineHandle*)e)->eng->plan().plan;
  if (f < 0 || f >= pl.nfields) return -1;
  *elem_type = pl.field[f].elem;
  *is_var = pl.field[f].length != GDB_VL_FIXED;
  *fixed_num = pl.field[f].fixed_num;
  return 0;
}
uint64_t gdbamd_engine_header(void* e, char* dst, uint64_t cap) {
  if (!e) return 0;
  const std::string& h = ((EngineHandle*)e)->eng->plan().header_text;
  if (dst && cap) memcpy(dst, h.data(), std::min<uint64_t>(cap, h.size()));
  return h.size();
}
int gdbamd_engine_stage_cells(void* e, const uint8_t* cells, uint64_t nbytes) {
  return guarded([&]() -> int { ((EngineHandle*)e)->eng->stage_cells(cells, nbytes); return 0; }, 1);
}
int gdbamd_engine_stage_cells_begin(void* e) { return guarded([&]() -> int { ((EngineHandle*)e)->eng->stage_cells_begin(); return 0; }, 1); }
int gdbamd_engine_stage_cells_append(void* e, const uint8_t* cells, uint64_t nbytes) {
  return guarded([&]() -> int { ((EngineHandle*)e)->eng->stage_cells_append(cells, nbytes); return 0; }, 1);
}
int gdbamd_engine_stage_cells_end(void* e) { return guarded([&]() -> int { ((EngineHandle*)e)->eng->stage_cells_end(); return 0; }, 1); }
int gdbamd_engine_adopt_device_fragment(void* e, int64_t ncells, const int32_t* row, const int64_t* begin, const int64_t* end, const gdbamd_device_column* cols,
                                        int ncols, uint64_t reference_cell_bytes) {
  return guarded([&]() -> int {
    CombineEngine& eng = *((EngineHandle*)e)->eng;
    if (ncols != eng.plan().plan.nfields) throw GenomicsDBDeviceException("adopt_device_fragment: ncols != number of plan fields");
    FragmentView v;
    memset(&v, 0, sizeof(v));
    v.ncells = ncells; v.row = row; v.begin = begin; v.end = end;
    for (int f = 0; f < ncols; ++f) { v.col[f].data = cols[f].data; v.col[f].off = cols[f].off; }
    eng.pipeline().adopt_fragment(v);
    eng.reference_cell_bytes = reference_cell_bytes;
    eng.has_cells = ncells > 0;
    eng.num_cells = ncells;
    return 0;
  }, 1);
}
int gdbamd_engine_open_array(void* e, const char* dir) { return guarded([&]() -> int { ((EngineHandle*)e)->eng->open_array(dir ? dir : ""); return 0; }, 1); }
int gdbamd_engine_open_memory_cells(void* e, const uint8_t* cells, uint64_t nbytes) {
  return guarded([&]() -> int { ((EngineHandle*)e)->eng->open_memory_cells(cells, nbytes); return 0; }, 1);
}
int gdbamd_pin_host_memory(const void* p, uint64_t nbytes) {
  return guarded([&]() -> int {
    if (!p || !nbytes) return 0;
    if (hipHostRegister(const_cast<void*>(p), (size_t)nbytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); throw GenomicsDBDeviceException("hipHostRegister failed (locked-memory limit?)"); }
    return 0;
  }, 1);
}
int gdbamd_unpin_host_memory(const void* p) {
  return guarded([&]() -> int {
    if (p && hipHostUnregister(const_cast<void*>(p)) != hipSuccess) { (void)hipGetLastError(); throw GenomicsDBDeviceException("hipHostUnregister failed"); }
    return 0;
  }, 1);
}
int gdbamd_engine_open_cell_callback(void* e, gdbamd_cell_chunk_fn fn, void* user) {
  return guarded([&]() -> int { ((EngineHandle*)e)->eng->open_cell_callback(fn, user); return 0; }, 1);
}
int gdbamd_engine_cover(void* e, int64_t column, int64_t* lo, int64_t* hi) {
  return guarded([&]() -> int {
    const CombineEngine::Coverage c = ((EngineHandle*)e)->eng->cover(column);
    if (lo) *lo = c.lo;
    if (hi) *hi = c.hi;
    return 0;
  }, 1);
}
int gdbamd_engine_staged_info(void* e, int64_t* ncells, uint64_t* reference_cell_bytes) {
  if (!e) return 1;
  CombineEngine& eng = *((EngineHandle*)e)->eng;
  if (ncells) *ncells = eng.num_cells;
  if (reference_cell_bytes) *reference_cell_bytes = eng.reference_cell_bytes;
  return 0;
}
int gdbamd_engine_set_reference(void* e, int64_t begin, const char* bases, uint64_t len) {
  return guarded([&]() -> int { ((EngineHandle*)e)->eng->set_reference_window(begin, std::string(bases, len)); return 0; }, 1);
}

namespace {
struct HostCopy { char* dst; uint64_t cap, len; };
void copy_page(void* user, const char* dev, uint64_t n) {
  HostCopy* hc = (HostCopy*)user;
  if (hc->dst && hc->len < hc->cap) {
    uint64_t k = std::min(n, hc->cap - hc->len);
    if (hipMemcpy(hc->dst + hc->len, dev, k, hipMemcpyDeviceToHost) != hipSuccess) throw GenomicsDBDeviceException("page copy to host failed");
  }
  hc->len += n;
}
}  // namespace

static void fill_interval_stats(gdbamd_interval_stats* out, const IntervalStats& s, CombineEngine& eng) {
  out->num_cells = s.num_cells; out->num_cells_in_window = s.num_cells_in_window; out->num_records = s.num_records;
  out->num_heavy_incidences = s.num_heavy_incidences; out->bytes_out = s.bytes_out; out->bytes_in_reference_cells = eng.reference_cell_bytes;
  out->pages = s.pages; out->write_launches = s.write_launches; out->err_bits = s.err_bits;
  out->ms_sweep = s.ms_sweep; out->ms_site = s.ms_site; out->ms_size = s.ms_size; out->ms_write = s.ms_write; out->ms_total = s.ms_total;
  out->ms_write_kernel_avg = s.ms_write_kernel_avg;
  out->num_record_types = s.num_record_types; out->resolved_entry_bytes = s.resolved_entry_bytes;
  out->num_text_slots = s.num_text_slots; out->text_pool_bytes = s.text_pool_bytes;
  out->num_remap_elements = s.num_remap_elements;
  out->bytes_compressed = s.bytes_compressed; out->ms_compress = s.ms_compress; out->reserved1 = 0;
  for (int i = 0; i < GDBAMD_GT_NUM_STATS; ++i) out->gt_profile_stats[i] = s.gt_profile[i];
}
int gdbamd_engine_run_interval(void* e, int64_t qb, int64_t qe, uint64_t arena_bytes, char* host_out, uint64_t host_cap, uint64_t* host_len,
                               gdbamd_interval_stats* out) {
  return guarded([&]() -> int {
    CombineEngine& eng = *((EngineHandle*)e)->eng;
    HostCopy hc{host_out, host_cap, 0};
    eng.stage_reference_for(qb, qe);  // no-op unless the query names a reference_genome and cells were staged from host
    IntervalStats s = eng.pipeline().run_interval(qb, qe, arena_bytes, host_out ? copy_page : nullptr, &hc);
    if (host_len) *host_len = host_out ? hc.len : s.bytes_out;
    if (out) fill_interval_stats(out, s, eng);
    return 0;
  }, 1);
}
int gdbamd_engine_run_intervals(void* e, int n, const int64_t* begins, const int64_t* ends, uint64_t arena_bytes, int lanes, gdbamd_interval_stats* stats,
                                char* const* host_out, const uint64_t* host_cap, uint64_t* host_len) {
  return guarded([&]() -> int {
    CombineEngine& eng = *((EngineHandle*)e)->eng;
    std::vector<std::pair<int64_t, int64_t>> ivs;
    for (int i = 0; i < n; ++i) ivs.emplace_back(begins[i], ends[i]);
    std::vector<HostCopy> hc;
    std::function<void(size_t, const char*, uint64_t)> on_page;
    if (host_out) {      // (interval i's pages go to host_out[i], each interval from one thread only)
      for (int i = 0; i < n; ++i) hc.push_back(HostCopy{host_out[i], host_cap ? host_cap[i] : 0, 0});
      on_page = [&hc](size_t i, const char* dev, uint64_t nbytes) { copy_page(&hc[i], dev, nbytes); };
    }
    const std::vector<IntervalStats> res = eng.run_intervals(ivs, arena_bytes, lanes, on_page);
    if (host_len) for (int i = 0; i < n; ++i) host_len[i] = host_out ? hc[(size_t)i].len : res[(size_t)i].bytes_out;
    if (stats) for (int i = 0; i < n; ++i) fill_interval_stats(&stats[i], res[(size_t)i], eng);
    return 0;
  }, 1);
}

int gdbamd_engine_lane_footprint(void* e, int64_t interval_columns, uint64_t arena_bytes, uint64_t* bytes) {
  return guarded([&]() -> int { *bytes = ((EngineHandle*)e)->eng->lane_footprint_bytes(interval_columns, arena_bytes); return 0; }, -1);
}
int gdbamd_engine_release_lanes(void* e) {
  return guarded([&]() -> int { ((EngineHandle*)e)->eng->release_lanes(); return 0; }, -1);
}

int gdbamd_engine_prepare_interval(void* e, int64_t qb, int64_t qe) {
  return guarded([&]() -> int {
    CombineEngine& eng = *((EngineHandle*)e)->eng;
    eng.stage_reference_for(qb, qe);
    eng.pipeline().prepare_interval(qb, qe);
    return 0;
  }, -1);
}
int gdbamd_engine_next_page(void* e, uint64_t arena_bytes, const void** dev_ptr, uint64_t* nbytes) {
  return guarded([&]() -> int {
    const char* p = nullptr;
    uint64_t n = 0;
    const bool more = ((EngineHandle*)e)->eng->pipeline().next_page(arena_bytes, &p, &n);
    if (dev_ptr) *dev_ptr = more ? p : nullptr;
    if (nbytes) *nbytes = more ? n : 0;
    return more ? 1 : 0;
  }, -1);
}

int gdbamd_engine_split_point(void* engine, int64_t qb, int64_t qe, int64_t max_columns, int64_t* piece_end) {
  try { *piece_end = ((EngineHandle*)engine)->eng->pipeline().split_point(qb, qe, max_columns); return 0; } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
int gdbamd_engine_column_histogram(void* engine, uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, uint64_t* counts, uint64_t nbins, int accumulate) {
  try { ((EngineHandle*)engine)->eng->pipeline().column_histogram(hist_begin, hist_end, bin_size, counts, nbins, accumulate != 0); return 0; } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
// gt_mpi_gather --print-calls: the document is produced by the call with dst == NULL (which returns its length) and kept in the handle
// for the call that copies it
int64_t gdbamd_engine_print_calls(void* engine, char* dst, uint64_t cap) { return gdbamd_engine_print_cells(engine, 0, dst, cap); }
// mode 0: --print-calls, 1: --print-csv, 2: --print-AC
int64_t gdbamd_engine_print_cells(void* engine, int mode, char* dst, uint64_t cap) {
  try {
    EngineHandle* h = (EngineHandle*)engine;
    auto make = [&]() { h->calls_text_mode = mode; return mode == 0 ? h->eng->print_calls() : mode == 1 ? h->eng->print_csv() : h->eng->print_allele_counts(); };
    if (!dst) { h->calls_text = make(); return (int64_t)h->calls_text.size(); }
    if (h->calls_text_mode != mode) h->calls_text = make();     // (nothing kept, or the document of another mode)
    const size_t n = h->calls_text.size();
    if (cap < n) { g_last_error = "print_calls: destination too small"; return -1; }
    memcpy(dst, h->calls_text.data(), n);
    std::string().swap(h->calls_text);                          // handed over: a multi-GB document does not stay in the handle
    h->calls_text_mode = -1;
    return (int64_t)n;
  } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
// ColumnHistogramOperator::equi_partition_and_print_bins (variant_operations.cc:769-796), the text it prints; returns its length (dst may be NULL), -1 when
// num_parts >= nbins (the reference prints a complaint and returns false)
int64_t gdbamd_equi_partition_text(const uint64_t* counts, uint64_t nbins, uint64_t hist_begin, uint64_t bin_size, uint64_t num_parts, char* dst, uint64_t cap) {
  if (num_parts >= nbins || num_parts == 0) return -1;
  unsigned long long total = 0;
  for (uint64_t i = 0; i < nbins; ++i) total += counts[i];
  const double per = (double)total / (double)num_parts;
  std::string out;
  char line[160];
  snprintf(line, sizeof(line), "Total %llu #bins %llu count/bins %.1f\n", total, (unsigned long long)num_parts, per);
  out += line;
  // (no cell at all - an empty array, or query rows without cells: the reference's loop would not advance (its assert(j > i)); the header
  // line alone is the answer here)
  for (uint64_t i = 0; i < nbins && total > 0;) {
    uint64_t j = i;
    unsigned long long cur = 0;
    for (; (double)cur < per && j < nbins; cur += counts[j], ++j) {}
    snprintf(line, sizeof(line), "%llu,%llu,%llu\n", (unsigned long long)(hist_begin + i * bin_size), (unsigned long long)(hist_begin + j * bin_size - 1), cur);
    out += line;
    i = j;
  }
  out += "\n";
  if (dst && cap) { const size_t n = std::min<size_t>(out.size(), (size_t)cap); memcpy(dst, out.data(), n); }
  return (int64_t)out.size();
}
int gdbamd_build_output_index(const char* path, int is_bcf) {
  try { if (is_bcf) build_csi_index(path); else build_tbi_index(path); return 0; } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
int gdbamd_engine_save_fragment(void* engine, const char* path) {
  try { ((EngineHandle*)engine)->eng->save_fragment(path); return 0; } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
int gdbamd_engine_save_fragment_compressed(void* engine, const char* path) {
  try { ((EngineHandle*)engine)->eng->save_fragment(path, true); return 0; } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
int gdbamd_engine_load_fragment(void* engine, const char* path) {
  try { ((EngineHandle*)engine)->eng->load_fragment(path); return 0; } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}

int gdbamd_column_partition(const char* loader_json_text, int rank, int64_t* begin, int64_t* end) {
  try {
    GenomicsDBImportConfig cfg;
    cfg.read_from_json(mini_json::parse(std::string(loader_json_text ? loader_json_text : "")), rank);
    const ColumnRange r = cfg.get_column_partition(rank);
    if (begin) *begin = r.first;
    if (end) *end = r.second;
    return 0;
  } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}

int gdbamd_import_cells(const char* vid_mapping_file, const char* callset_mapping_file, const char* file_root, int treat_deletions_as_intervals,
                        int64_t column_begin, int64_t column_end, uint8_t** cells, uint64_t* nbytes, int64_t* ncells) {
  try {
    if (!vid_mapping_file || !callset_mapping_file || !cells || !nbytes) throw GenomicsDBConfigException("gdbamd_import_cells: null argument");
    VidMapper vid;
    vid.parse_vid_json(mini_json::parse_file(vid_mapping_file));
    vid.parse_callsets_json(mini_json::parse_file(callset_mapping_file));
    ImportOptions opt;
    opt.treat_deletions_as_intervals = treat_deletions_as_intervals != 0;
    opt.column_begin = column_begin;
    opt.column_end = column_end;
    if (file_root) opt.file_root = file_root;
    ImportStats st;
    const std::vector<uint8_t> out = import_callsets_to_cells(vid, opt, &st);
    *cells = (uint8_t*)malloc(out.size() ? out.size() : 1);
    if (!*cells) throw GenomicsDBConfigException("out of memory");
    if (!out.empty()) memcpy(*cells, out.data(), out.size());
    *nbytes = out.size();
    if (ncells) *ncells = st.num_cells;
    g_last_error.clear();
    return 0;
  } catch (const std::exception& e) { g_last_error = e.what(); return -1; }
}
void gdbamd_free(void* p) { free(p); }

}  // extern "C"
