// genomicsdb_bcf_generator.h - pull stream over the device pipeline, source-compatible with the reference's
// GenomicsDBBCFGenerator (reference src/main/cpp/include/vcf/genomicsdb_bcf_generator.h:33-93): header first, then the
// combined-gVCF body of every query column interval, produced in batches of at most buffer_capacity bytes.
#pragma once
#include <functional>
#include <memory>
#include <exception>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../kernels/gdb_pipeline.h"
#include "../host/reference_genome.h"

namespace genomicsdb_amd {

class GenomicsDBJNIException : public std::runtime_error {
 public:
  explicit GenomicsDBJNIException(const std::string& m) : std::runtime_error("GenomicsDBJNIException : " + m) {}
};

// what the engine needs to know about one query: configuration, plan, pipeline
class CombineEngine {
 public:
  // output_format "": VCF text, "bu": uncompressed BCF2 records (what GATK4's BCF2Codec reads); the two flags are the JNI's
  explicit CombineEngine(const mini_json::Value& query_json, int device, const GenomicsDBImportConfig* loader = nullptr, int rank = 0,
                         const std::string& output_format = "", bool use_missing_values_only_not_vector_end = false);
  // from a query configuration the caller has already read (the reference's C++ entry: VariantQueryProcessor + operator);
  // max_diploid_alt_alleles != 0 overrides the configuration's value (the operator constructor's argument)
  CombineEngine(const VariantQueryConfig& query_config, int device, const std::string& output_format, bool use_missing_values_only_not_vector_end,
                unsigned max_diploid_alt_alleles = 0);
  VariantQueryConfig& query_config() { return m_qc; }
  const HostPlan& plan() const { return m_hp; }
  DevicePipeline& pipeline() { return *m_pipe; }
  void stage_cells(const uint8_t* cells, uint64_t nbytes);
  // staging in parts: cells of successive calls must continue the column-major order
  void stage_cells_begin();
  void stage_cells_append(const uint8_t* cells, uint64_t nbytes);
  void stage_cells_end();
  // the staged fragment as a columnar file (<workspace>/<array>/fragment.gdbamd is what the query stream opens first)
  void save_fragment(const std::string& path, bool compress = false);
  void load_fragment(const std::string& path);
  // ---- arrays larger than the staging budget: column windows streamed through HBM -----------------------------------------
  // An array source is read window by window (whole begin columns, about staging_budget_bytes() each); the intervals still live
  // at a window's end are carried into the next one on the device (DevicePipeline::begin_staging(carry_from)), the analogue of
  // the reference's segment-at-a-time iterator + scan state (variant_storage_manager.cc:61-153, query_variants.h:126-191).
  // An array that fits the budget is simply one window.
  void open_array(const std::string& dir);                      // <workspace>/<array>: fragment.gdbamd when valid for this query, else cells.bin
  void open_memory_cells(const uint8_t* cells, uint64_t nbytes);  // the caller keeps the bytes alive
  // cells produced on demand: every call hands out the next chunk (whole begin columns, column-major across chunks; valid until
  // the next call), 0 = no more.  One pass only: a query interval in front of the current window cannot be served.
  typedef int (*CellChunkFn)(void* user, const uint8_t** cells, uint64_t* nbytes);
  void open_cell_callback(CellChunkFn fn, void* user);
  struct Coverage { int64_t lo, hi; };                          // every query position in [lo, hi] sees all its live cells in the staged fragment
  Coverage cover(int64_t column);                               // stages windows until `column` is covered
  // ColumnHistogramOperator over the whole array source (all its column windows, front to back): counts[(hist_end - hist_begin) / bin_size + 1]
  void column_histogram(uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, std::vector<uint64_t>& counts);
  // gt_mpi_gather --print-calls (tools/src/gt_mpi_gather.cc:369-383): the JSON document of the query's cells, interval by interval
  // (VariantCallPrintOperator, variant_operations.cc:803-843); the cells are selected and formatted on the device (DevicePipeline::calls_json)
  std::string print_calls();
  // --print-csv (VariantCallPrintCSVOperator, variant_operations.cc:845-903): one line per cell; --print-AC (AlleleCountOperator, :905-1089): per query
  // interval "column REF ALT count" of the normalised ALT alleles the cells' genotypes name, ordered by column, REF, ALT
  std::string print_csv();
  std::string print_allele_counts();
  uint64_t staging_budget_bytes() const;
  int64_t windows_staged = 0;
  uint64_t pipeline_generation = 0;   // counts the swaps of the two pipelines (overlapped staging): pipeline() is another object afterwards
  int64_t num_cells = 0;
  void stage_reference_for(int64_t qb, int64_t qe);
  // Several query intervals of the staged fragment, up to `lanes` of them in flight at a time: lane l takes intervals l, l + lanes, ...
  // on a device pipeline of its own (own HIP stream, own entry table / matrix / page arenas) that ADOPTS the fragment staged in the
  // engine's pipeline - the cells are in HBM once.  The sweep / site / sizing kernels of one interval (latency-bound) then overlap
  // with the page kernel of another (store-bound).  Pages stay in HBM (no callback); per-interval statistics in interval order.
  // The reference's analogue: nothing - its scan is one thread per partition (tools/src/gt_mpi_gather.cc:322-366).
  // on_page (optional): called from the lane's thread for every page of interval `index` (device pointer, valid during the call)
  std::vector<IntervalStats> run_intervals(const std::vector<std::pair<int64_t, int64_t>>& intervals, uint64_t arena_bytes, int lanes,
                                           const std::function<void(size_t index, const char* dev_ptr, uint64_t nbytes)>& on_page = nullptr);
  // HBM a NEW lane pipeline takes for intervals of `interval_columns` positions at this query's sample count (an estimate, see the
  // definition): run_intervals uses no more new lanes than hipMemGetInfo's free bytes hold (4 GiB kept spare) and says so on stderr.
  uint64_t lane_footprint_bytes(int64_t interval_columns, uint64_t arena_bytes) const;
  // frees the lane pipelines (and their HBM); the next run_intervals creates them again
  void release_lanes();
  // reference bases given by the caller for columns [begin, begin + bases.size()): kept, so that the pipeline a later window is
  // staged into (overlapped staging) sees them too
  void set_reference_window(int64_t begin, const std::string& bases);
  uint64_t reference_cell_bytes = 0;
  int64_t min_begin = 0, max_end = 0;
  bool has_cells = false;
  ~CombineEngine();
 private:
  void for_each_interval_text(int mode, int arg, const std::function<void(int64_t, int64_t, const std::string&)>& fn);
  VariantQueryConfig m_qc;
  HostPlan m_hp;
  std::unique_ptr<DevicePipeline> m_pipe;       // the pipeline whose fragment is in use
  std::unique_ptr<DevicePipeline> m_pipe2;      // overlapped staging: the next column window is staged here meanwhile
  struct Lane { std::unique_ptr<DevicePipeline> pipe; const DevicePipeline* src = nullptr; uint64_t gen = 0; bool has_user_ref = false; };
  std::vector<Lane> m_lanes;                    // run_intervals: pipelines 1 .. lanes - 1 (lane 0 is m_pipe)
  void stage_reference_on(DevicePipeline& pipe, int64_t qb, int64_t qe);
  int m_device = 0;
  std::unique_ptr<CellStreamLayout> m_layout;   // attribute order, plan-field map, row map of the binary cell stream
  ReferenceGenomeInfo m_ref;
  const CellStreamLayout& layout();
  std::vector<ColumnLayout> expected_columns();
  uint64_t schema_hash();
  // array source
  enum SourceKind { SRC_NONE, SRC_CELLS_FILE, SRC_CELLS_MEMORY, SRC_FRAGMENT_FILE, SRC_CALLBACK };
  struct Source {
    SourceKind kind = SRC_NONE;
    int fd = -1; uint64_t size = 0, cursor = 0;      // cells file: byte cursor
    const uint8_t* mem = nullptr;                    // cells in memory
    int64_t cell_cursor = 0, ncells_file = 0;        // fragment file: cell cursor
    uint8_t* chunk = nullptr; size_t chunk_cap = 0;  // pinned read buffer (cells file)
    CellChunkFn fn = nullptr; void* fn_user = nullptr;   // callback: one chunk of lookahead
    const uint8_t* pending = nullptr; uint64_t pending_bytes = 0; bool pending_valid = false;
    bool window_valid = false, eof = false;
    Coverage cov{INT64_MIN, INT64_MIN};
  } m_src;
  void rewind_source();
  void advance_window();
  // one staged column window, before it is taken into use
  struct StagedWindow {
    int64_t hi = INT64_MIN, carry_from = INT64_MIN, new_cells = 0, min_begin = INT64_MAX, max_end = 0;
    uint64_t reference_cell_bytes = 0;
    bool has_cells = false, eof = false;
  };
  void stage_window(DevicePipeline& dst, DevicePipeline& carry_src, int64_t carry_from, StagedWindow& r);
  void take_window(const StagedWindow& r);
  bool overlap_enabled() const;
  void start_prefetch();
  bool join_prefetch(bool take);
  std::thread m_prefetch;
  std::exception_ptr m_prefetch_error;
  StagedWindow m_next;
  bool m_window_eof = false;
  bool m_user_ref = false, m_pipe_has_user_ref[2] = {false, false}; int64_t m_user_ref_begin = 0; std::string m_user_ref_bases;
};

class GenomicsDBBCFGenerator {
 public:
  GenomicsDBBCFGenerator(const std::string& loader_config_file, const std::string& query_config_file, const char* chr, const int start,
                         const int end, int my_rank = 0, size_t buffer_capacity = 1048576u, size_t tiledb_segment_size = 1048576u,
                         const char* output_format = "bu", const bool produce_header_only = false,
                         const bool use_missing_values_only_not_vector_end = false, const bool keep_idx_fields_in_bcf_header = true,
                         const bool bgzf_stream = false);
  // bgzf_stream: the reference's stream object serialises through VCFSerializedBufferAdapter, which never compresses - bcf_hdr_serialize /
  // bcf_serialize look at m_is_bcf only (vcf_adapter.cc:475-505): "z" gives plain VCF text there and "b" plain BCF2.  So this constructor (the
  // reference's signature) reads "z" as "" and "b" as "bu" unless bgzf_stream says the caller wants the build's extension, BGZF blocks deflated
  // on the device (gdb_mi355_init_output_format, gt_mpi_gather -O z: what the file-writing VCFAdapter of the reference produces)
  // in-memory flavour: query JSON text + begin-cells (reference binary-cell layout)
  GenomicsDBBCFGenerator(const std::string& query_json_text, const uint8_t* cells, uint64_t nbytes, size_t buffer_capacity, bool produce_header_only,
                         const char* output_format = "", bool use_missing_values_only_not_vector_end = false, bool keep_idx_fields_in_bcf_header = true);
  GenomicsDBBCFGenerator(const GenomicsDBBCFGenerator&) = delete;
  GenomicsDBBCFGenerator& operator=(const GenomicsDBBCFGenerator&) = delete;
  ~GenomicsDBBCFGenerator();
  // n == SIZE_MAX: only produce the next batch
  size_t read_and_advance(uint8_t* dst, size_t offset, size_t n);
  uint8_t read_next_byte();
  bool end() const { return m_done && m_ring_count == 0 && m_next_read_idx >= m_header.size(); }
  size_t get_buffer_capacity() const { return m_buffer_capacity; }
  // the bytes of the current batch that have not been read yet (reference: get_read_batch() hands out the RWBuffer being read;
  // include/vcf/genomicsdb_bcf_generator.h:33-93).  Valid until the next read / skip call.
  struct RWBuffer { const uint8_t* m_buffer; size_t m_num_valid_bytes; size_t m_next_read_idx; };   // (member names of the reference's RWBuffer)
  RWBuffer get_read_batch();
  // drain statistics of the stream (bench: t_drain, end-to-end rate)
  struct DrainStats { uint64_t pages = 0, chunks = 0, bytes = 0; double seconds_waiting_for_copies = 0, seconds_producing = 0; };
  const DrainStats& drain_stats() const { return m_drain; }
  CombineEngine& engine() { return *m_engine; }   // (gt_mpi_gather --produce-histogram counts the array's cells through it)
 private:
  void common_init(bool produce_header_only, bool keep_idx_fields_in_bcf_header = true);
  // Stream machinery: the device assembles pages of up to device_page_bytes() (independent of the caller's buffer_capacity)
  // alternately into two HBM arenas; a page leaves in chunks over a copy stream into a ring of pinned host buffers while the
  // next page is being assembled; read() is served from the ring.
  struct RingSlot { uint8_t* host = nullptr; size_t cap = 0, len = 0; void* done = nullptr; bool waited = false; };
  bool advance_page();          // next page (of this or the next piece / interval) ready for draining; false: stream exhausted
  void fill_ring();             // issue as many chunk copies as there are free slots
  void pop_slot();
  uint64_t device_page_bytes() const;
  std::vector<uint8_t> m_owned_cells;   // in-memory flavour: the cells (declared before the engine: it outlives it)
  std::unique_ptr<CombineEngine> m_engine;
  size_t m_buffer_capacity;
  std::vector<uint8_t> m_header;  // first bytes of the stream
  std::string m_trailer;          // last bytes of the stream (BGZF output formats: the EOF block), handed out once the pages are through
  size_t m_next_read_idx = 0;     // into the header while it lasts, then into the ring's head slot
  std::vector<RingSlot> m_ring;
  size_t m_ring_head = 0, m_ring_count = 0, m_slot_bytes = 0;
  void* m_copy_stream = nullptr;                  // hipStream_t
  void* m_arena_read[2] = {nullptr, nullptr};     // hipEvent_t: last copy out of arena i
  DevicePipeline::PageTicket m_page, m_next_page;
  DevicePipeline* m_page_owner = nullptr;
  bool m_page_valid = false, m_next_valid = false;
  uint64_t m_page_off = 0;
  int m_arena_toggle = 0;
  DrainStats m_drain;
  bool m_done = false, m_interval_active = false, m_produce_header_only = false;
  unsigned m_query_column_interval_idx = 0;
  int64_t m_piece_begin = INT64_MIN, m_piece_end = 0, m_interval_end = 0;   // current piece of the current query interval
  int64_t max_window_columns() const;
};

}  // namespace genomicsdb_amd
