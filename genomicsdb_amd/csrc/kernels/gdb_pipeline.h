// gdb_pipeline.h - the MI355X scan-and-combine pipeline (host-visible interface, no HIP types).
//
// One DevicePipeline = one column partition on one GPU: the device-side replacement of
// VariantQueryProcessor::scan_and_operate + BroadCombinedGVCFOperator (reference
// src/main/cpp/src/genomicsdb/query_variants.cc:334-476, src/query_operations/broad_combined_gvcf.cc:765-901).
// run_interval() consumes a staged columnar fragment and produces the bit-exact VCF body for one query column
// interval in pages of at most `arena_bytes` (the analogue of combined_vcf_records_buffer_size_limit / RWBuffer).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../core/gdb_types.h"
#include "../host/combine_plan.h"
#include "../host/fragment.h"

namespace genomicsdb_amd {

class GenomicsDBDeviceException : public std::runtime_error {
 public:
  explicit GenomicsDBDeviceException(const std::string& m) : std::runtime_error("GenomicsDBDeviceException : " + m) {}
};

struct IntervalStats {
  int64_t num_cells = 0;          // begin-cells staged
  int64_t num_cells_in_window = 0;
  int64_t num_records = 0;        // output positions
  int64_t num_heavy_incidences = 0;
  uint64_t bytes_out = 0;         // VCF body bytes
  int pages = 0;
  uint32_t err_bits = 0;
  // device time (hipEvent) per phase, ms
  float ms_sweep = 0, ms_site = 0, ms_size = 0, ms_write = 0, ms_total = 0;
  float ms_write_kernel_avg = 0;  // average duration of one entry-write launch
  int write_launches = 0;
  // entry text table of the interval
  int num_record_types = 0;       // distinct (FORMAT mask, #merged alleles, remap flags)
  int resolved_entry_bytes = 0;   // bytes per (record, sample) of the resolved matrix: 8, 5 (compact layout) or 0 (matrix-free / event paths)
  int64_t num_text_slots = 0;     // (cell, type) + (record, variant call) + no-call texts
  int64_t text_pool_bytes = 0;
  uint64_t num_remap_elements = 0; // sum over re-indexed records of (calls with PL) x (merged genotypes)
  uint64_t bytes_compressed = 0;   // BGZF output formats: bytes of the pages after compression (bytes_out stays the uncompressed size)
  float ms_compress = 0;           // device time of the compression kernels
  // the reference's GTProfileStats counters (query_variants.h:67-124), per interval; see include/genomicsdb_amd.h
  uint64_t gt_profile[6] = {0, 0, 0, 0, 0, 0};
};
enum GTStatIdx { GT_NUM_CELLS = 0, GT_NUM_CELLS_IN_LEFT_SWEEP, GT_NUM_VALID_CELLS_IN_QUERY, GT_NUM_ATTR_CELLS_ACCESSED,
                 GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS, GT_NUM_OPERATOR_INVOCATIONS, GT_NUM_STATS };

// what the engine needs to know about a staged fragment besides the columns
struct FragmentFileMeta {
  uint64_t reference_cell_bytes = 0; int64_t min_begin = INT64_MAX, max_end = 0, ncells = 0;
  uint64_t schema_hash = 0;                    // vid + callset mapping the file was written under (0: unknown)
  uint64_t source_bytes = 0; int64_t source_mtime = 0;   // size / mtime of the cells.bin it was made from (0: none)
};
// layout of one attribute column as the array schema defines it (what a fragment file is checked against)
struct ColumnLayout { bool var = false; int elem_size = 4; int fixed_num = 1; };

// page consumer: `dev_ptr` points to `nbytes` of VCF text in HBM, valid until the callback returns
typedef void (*PageCallback)(void* user, const char* dev_ptr, uint64_t nbytes);

class DevicePipeline {
 public:
  explicit DevicePipeline(const HostPlan& hp, int device = 0);
  ~DevicePipeline();
  DevicePipeline(const DevicePipeline&) = delete;
  DevicePipeline& operator=(const DevicePipeline&) = delete;
  // copy a host fragment to HBM (replaces the previously staged one)
  void stage_fragment(const HostFragment& hf);
  // staging in parts (column-major order across parts): begin, append host fragments, finish = one contiguous fragment in HBM
  // carry_from != INT64_MIN: the cells of the fragment staged so far whose intervals reach column carry_from (at most one per
  // sample) open the new fragment - windowed streaming of an array larger than the staging budget
  void begin_staging(int64_t carry_from = INT64_MIN);
  // the same, with the carried cells taken from the fragment `source` (another pipeline of the same plan and device) has staged;
  // `source` may go on computing on that fragment meanwhile (overlapped staging of the next column window)
  void begin_staging_from(DevicePipeline& source, int64_t carry_from);
  int64_t carried_cells() const;
  void append_fragment(const HostFragment& hf);
  // the same from the reference's binary cell stream: the bytes are copied to HBM as they are and taken apart there
  // (one thread per cell); the host only walks the cell sizes.  row_map: array row -> query row (-1: not queried)
  struct CellStreamInfo { int64_t ncells = 0; uint64_t reference_cell_bytes = 0; int64_t min_begin = INT64_MAX, max_end = 0; };
  // result of walking the cell sizes of a buffer on the host (the format's one sequential dependency)
  struct CellWalk {
    int64_t nkept = 0, nmark = 0;                // cells of queried rows / of other array rows (boundary markers)
    uint64_t reference_cell_bytes = 0, bytes_taken = 0;
    int64_t first_begin = INT64_MAX, last_begin = INT64_MIN;
    int64_t next_begin = INT64_MAX;              // whole_columns_only: begin column of the first cell left in the buffer
    bool single_column = false;                  // whole_columns_only: the buffer holds (part of) one column only, nothing was taken
  };
  static CellWalk walk_cells(const uint8_t* cells, uint64_t nbytes, const std::vector<int32_t>& row_map, bool whole_columns_only, std::vector<uint64_t>& offs);
  // The cells of a buffer into HBM columns.  The sizes are walked ON THE DEVICE (candidate headers + pointer doubling, see
  // gdb_pipeline.hip "the walk of the cell sizes"); whole_columns_only: more bytes follow the buffer, so the last begin column
  // seen (and a cell the buffer cuts through) is left for the next call - walk_out says how many bytes were taken and where the
  // next window begins.  walked / walk_info: the walk has been done by the caller on the host (walk_cells; GDBAMD_HOST_WALK=1),
  // offs = cell offsets + end offset.
  CellStreamInfo append_cells(const uint8_t* cells, uint64_t nbytes, const VariantArraySchemaLite& schema, const std::vector<int>& attr_to_field,
                              const std::vector<int32_t>& row_map, const std::vector<uint64_t>* walked = nullptr, const CellWalk* walk_info = nullptr,
                              bool whole_columns_only = false, CellWalk* walk_out = nullptr);
  void finish_staging();
  // the staged fragment as a columnar file / a columnar file straight into HBM (format: gdb_pipeline.hip, 'columnar fragment file')
  // compress: version 3 - every data section as DEFLATE tiles (stored / fixed-Huffman blocks) that are inflated on the device when read
  void save_fragment(const std::string& path, const FragmentFileMeta& meta, bool compress = false);
  FragmentFileMeta load_fragment(const std::string& path, const std::vector<ColumnLayout>& expected, uint64_t expected_schema_hash);
  // the same window by window: open (validates the whole header against the file size and the expected layouts), then between
  // begin_staging() and finish_staging() append the cells [c0, c1) that make up whole begin columns and about budget_bytes
  struct FragmentFile;
  struct FragmentWindow { int64_t c0 = 0, c1 = 0, ncells = 0, first_begin = INT64_MAX, last_begin = INT64_MIN, next_begin = INT64_MAX; uint64_t reference_cell_bytes = 0; };
  FragmentFileMeta open_fragment_file(const std::string& path, const std::vector<ColumnLayout>& expected, uint64_t expected_schema_hash);
  FragmentWindow append_fragment_cells(int64_t c0, uint64_t budget_bytes);
  int64_t fragment_file_lower_bound(int64_t column);
  void close_fragment_file();
  // adopt a fragment that already lives in HBM (e.g. torch tensors); the caller keeps ownership
  void adopt_fragment(const FragmentView& device_view);
  // reference bases for TileDB columns [begin, begin + bases.size())
  void set_reference_window(int64_t begin, const std::string& bases);
  // pull interface: prepare_interval() runs the sweep, the site pass and the sizing pass; next_page() then writes the
  // next <= arena_bytes of VCF text (always whole records) into HBM and returns false when the interval is exhausted
  void prepare_interval(int64_t qb, int64_t qe);
  // last column of the first piece of [qb, qe] that can be processed on its own with byte-identical output (cut before a cell begin)
  int64_t split_point(int64_t qb, int64_t qe, int64_t max_columns);
  // ColumnHistogramOperator (variant_operations.cc:732-767) over the staged fragment's begin-cells: counts[(end - begin) / bin_size + 1];
  // accumulate: add to what counts holds (an array streamed in windows is counted window by window)
  // interval != nullptr: only the cells of the query interval [interval[0], interval[1]] (with_intersecting: and the intervals that began in
  // front of it and reach its begin), carried-over cells included
  void column_histogram(uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, uint64_t* counts, uint64_t nbins, bool accumulate = false,
                        const int64_t* interval = nullptr, bool with_intersecting = false);
  // gt_mpi_gather --print-calls (VariantCallPrintOperator, variant_operations.cc:803-843): the cells of [qb, qe] in the reference's iterator
  // order - first the intervals that began before qb and intersect it, then the cells that begin inside - as JSON objects separated by
  // ",\n", every line indented by `indent` spaces; ncells: how many.  with_intersecting = false: only the cells that begin inside (the
  // continuation of an interval whose first piece has been printed from an earlier column window)
  std::string calls_json(int64_t qb, int64_t qe, int indent, bool with_intersecting = true, int64_t* ncells = nullptr);
  // the same selection printed otherwise: mode 1 = the lines of --print-csv (VariantCallPrintCSVOperator, variant_operations.cc:845-903), mode 2 = one line
  // "column<TAB>REF<TAB>ALT" per GT element naming an ALT allele, REF / ALT normalised (AlleleCountOperator, :951-1056; indent = the GT step);
  // mode 0 = the JSON objects with ",\n" in front of each
  std::string cells_text(int64_t qb, int64_t qe, int mode, int indent, bool with_intersecting);
  // the staged cells carry query rows; the printers above print array rows: the map (empty: identical)
  void set_array_rows(const std::vector<int64_t>& query_row_to_array_row);
  bool next_page(uint64_t arena_bytes, const char** dev_ptr, uint64_t* nbytes);
  // the same in two steps (asynchronous page production, two arenas): see gdb_pipeline.hip
  struct PageTicket { int arena = 0; const char* dev = nullptr; uint64_t nbytes = 0; void* done_event = nullptr; };   // done_event: hipEvent_t recorded behind the page's kernels
  bool begin_page(uint64_t arena_bytes, int arena_idx, PageTicket* ticket);
  // waits for the page's kernels; for the BGZF output formats ("z" / "b") the page is then compressed in place on the device
  // (kernels/gdb_bgzf.hip) and ticket.nbytes becomes the size of the compressed page
  void finish_page(PageTicket& ticket);
  // hip_event (hipEvent_t) marks the consumer's last read of the arena; the next begin_page() into it waits for the event on the compute stream
  void set_arena_release_event(int arena_idx, void* hip_event);
  const IntervalStats& interval_stats() const;
  // the staged fragment as device pointers (owned by this pipeline: valid until it stages, loads or adopts another one) and a counter
  // of the fragments held so far - what a second pipeline needs to work on the same fragment (adopt_fragment; CombineEngine::run_intervals)
  FragmentView fragment_view() const;
  uint64_t fragment_generation() const;
  // several pipelines at work on one device (CombineEngine::run_intervals): launch the page kernel on a high-priority stream, so that it is
  // not slowed down by the other lanes' sweeps and sizing passes (the step time is the same; the kernel runs at its stand-alone duration)
  void set_page_priority(bool on);
  // push interface built on the two calls above
  IntervalStats run_interval(int64_t qb, int64_t qe, uint64_t arena_bytes, PageCallback cb, void* user);
  static int device_count();
  struct Impl;
 private:
  void classify_fragment();
  Impl* m_;
};

}  // namespace genomicsdb_amd
