/* genomicsdb_amd.h - C ABI of the MI355X variant-combine engine (libgenomicsdb_amd.so).
 *
 * Drop-in boundary for the scan/combine hot path of Intel-HLS/GenomicsDB v0.10.2.  Plain pointers and sizes only.
 * Every function returns an error through gdb_mi355_last_error() (thread-local) instead of throwing across the ABI.
 * There is NO CPU fallback: without a HIP device every engine entry point fails.
 *
 * (1) Query stream - the six entry points the reference exports over JNI
 *     (reference src/main/jni/src/genomicsdb_GenomicsDBQueryStream.cc:29-111, header
 *      src/main/jni/include/genomicsdb_GenomicsDBQueryStream.h:17-58), each a thin shim over GenomicsDBBCFGenerator
 *     (reference src/main/cpp/include/vcf/genomicsdb_bcf_generator.h:33-93).  A JNI stub forwards 1:1.
 * (2) Engine - explicit staging / per-interval execution with the output left in HBM, for callers that own device
 *     memory (bench, multi-GPU drivers).  Replaces the VariantQueryProcessor::scan_and_operate +
 *     BroadCombinedGVCFOperator pair (reference src/main/cpp/include/genomicsdb/query_variants.h:241-243,
 *     include/query_operations/broad_combined_gvcf.h:59-61).
 */
#ifndef GENOMICSDB_AMD_H
#define GENOMICSDB_AMD_H
#include <stddef.h>
#include <stdint.h>

/* ---- limits of this build that the reference does not have ------------------------------------------------------------
 * The reference's per-record tables resize (include/utils/lut.h:65-343); the device's are fixed (core/gdb_types.h, the
 * values below mirror it and tests/test_capi_cpu.py checks that they agree).  A query that needs more is refused when its
 * plan is built (GenomicsDBException naming the limit) or, for data-dependent limits, ends with an error that spells the
 * limit out ("device error bits ...") - never with truncated output. */
#define GDBAMD_MAX_QUERIED_FIELDS 48        /* attributes of one query (plan time) */
#define GDBAMD_MAX_INFO_FIELDS 24           /* ... of which INFO */
#define GDBAMD_MAX_FORMAT_FIELDS 24         /* ... of which FORMAT */
#define GDBAMD_MAX_MERGED_ALLELES 128       /* alleles of one output record, REF and <NON_REF> included (data) */
#define GDBAMD_MAX_INPUT_ALLELES 64         /* alleles of one input cell, REF included (data) */
#define GDBAMD_MAX_PLOIDY 8                 /* general-ploidy genotype enumeration for G-length fields (data) */
#define GDBAMD_MAX_INFO_VECTOR 64           /* elements of an element_wise_sum INFO vector (data) */
#define GDBAMD_MAX_ID_TOKENS 16             /* distinct ';'-separated ID tokens united in one record (data) */
#define GDBAMD_MAX_FILTER_IDS 16            /* distinct FILTER ids united in one record (data) */
#define GDBAMD_MAX_HISTOGRAM_FIELDS 8       /* (bins, counts) INFO fields reduced with histogram_sum (plan time) */
#define GDBAMD_MAX_PIPELINES_PER_PROCESS 256 /* device pipelines alive in one process: one per query stream / engine, two for an engine that streams an array through HBM
                                              * in windows with overlapped staging (each owns a 3.5 KB element of a __constant__ array - 0.9 MB of device memory read
                                              * with scalar loads; handles do not serialise on it) */

#ifdef __cplusplus
extern "C" {
#endif

const char* gdb_mi355_last_error(void);
int gdb_mi355_device_count(void);

/* ---- (1) query stream ------------------------------------------------------------------------------------ */
/* jniGenomicsDBInit: returns a handle or NULL.  chr == "" keeps the intervals of the query JSON; otherwise the
 * interval is contig offset + start-1 .. end-1 (1-based, inclusive).  is_bcf != 0: uncompressed BCF2 ("bu": "BCF\2\2" header,
 * typed records - what GATK4's BCF2Codec reads; vcf_adapter.cc:475-509), with the two htsjdk switches of the JNI
 * (use_missing_values_only_not_vector_end, keep_idx_fields_in_bcf_header); text VCF otherwise.  buffer_capacity is what
 * get_num_bytes_available reports (the reference's RWBuffer size); it does not size the device pages. */
void* gdb_mi355_init(const char* loader_json_file, const char* query_json_file, const char* chr, int start, int end, int rank,
                     uint64_t buffer_capacity, uint64_t segment_size, int is_bcf, int produce_header_only,
                     int use_missing_values_only_not_vector_end, int keep_idx_fields_in_bcf_header);
/* same stream, configuration and cells handed over in memory (tests, embedding) */
void* gdb_mi355_init_from_memory(const char* query_json_text, const uint8_t* cells, uint64_t cells_nbytes, uint64_t buffer_capacity,
                                 int produce_header_only);
void* gdb_mi355_init_from_memory_format(const char* query_json_text, const uint8_t* cells, uint64_t cells_nbytes, uint64_t buffer_capacity,
                                        int produce_header_only, int is_bcf, int use_missing_values_only_not_vector_end, int keep_idx_fields_in_bcf_header);
/* The same two with the reference's vcf_output_format string instead of the JNI's is_bcf flag: "" VCF text, "bu" BCF2, and the
 * BGZF-compressed flavours "z" (VCF text) / "b" (BCF2) that its VCFAdapter writes through htslib (vcf_adapter.cc:340-372,
 * genomicsdb_config_base.cc:34,156-165).  "z" / "b": header = one BGZF block (host), body = BGZF blocks of 8 192 input bytes (GDBAMD_BGZF_BLOCK = 4096 / 6144 / 8192 / 16384)
 * deflated ON THE DEVICE (only compressed bytes cross PCIe), then the 28-byte EOF block.  The compressed bytes are this build's
 * own; the inflated stream equals the "" / "bu" stream. */
void* gdb_mi355_init_output_format(const char* loader_json_file, const char* query_json_file, const char* chr, int start, int end, int rank,
                                   uint64_t buffer_capacity, uint64_t segment_size, const char* output_format, int produce_header_only,
                                   int use_missing_values_only_not_vector_end, int keep_idx_fields_in_bcf_header);
void* gdb_mi355_init_from_memory_output_format(const char* query_json_text, const uint8_t* cells, uint64_t cells_nbytes, uint64_t buffer_capacity,
                                               int produce_header_only, const char* output_format, int use_missing_values_only_not_vector_end,
                                               int keep_idx_fields_in_bcf_header);
uint64_t gdb_mi355_close(void* handle);                         /* jniGenomicsDBClose */
uint64_t gdb_mi355_get_num_bytes_available(void* handle);       /* jniGenomicsDBGetNumBytesAvailable: buffer capacity */
int gdb_mi355_read_next_byte(void* handle);                     /* jniGenomicsDBReadNextByte: byte or -1 */
int64_t gdb_mi355_read(void* handle, uint8_t* dst, uint64_t offset, uint64_t n); /* jniGenomicsDBRead: bytes copied, -1 on error */
int64_t gdb_mi355_skip(void* handle, uint64_t n);               /* jniGenomicsDBSkip */
/* The unread bytes of the current batch where they lie (a pinned ring buffer), without copying: *ptr / *n are valid until the
 * next call on this handle; consume them with gdb_mi355_skip.  1 = bytes available, 0 = end of the stream, -1 = error.
 * (reference: GenomicsDBBCFGenerator::get_read_batch, include/vcf/genomicsdb_bcf_generator.h:60-63, which the JNI read loop uses) */
int gdb_mi355_peek(void* handle, const uint8_t** ptr, uint64_t* n);
/* How the stream was drained so far (not part of the reference's surface; bench / diagnostics).  The device assembles pages
 * of GDBAMD_DEVICE_PAGE_MB (default 2048) MiB alternately into two HBM arenas - independently of buffer_capacity - and drains them
 * through a ring of pinned host buffers (GDBAMD_RING_SLOTS x GDBAMD_RING_SLOT_MB, default 4 x 64 MiB) on a copy stream while the
 * next page is assembled; gdb_mi355_read is served from the ring. */
typedef struct gdb_mi355_stream_stats {
  uint64_t pages, chunks, bytes;                 /* device pages produced, ring chunks copied, body bytes copied to the host */
  double seconds_waiting_for_copies;             /* time read() spent blocked on a copy that had not landed yet */
  double seconds_producing;                      /* host time inside the sweep / sizing / page launches (reader blocked) */
} gdb_mi355_stream_stats;
int gdb_mi355_get_stream_stats(void* handle, gdb_mi355_stream_stats* out);

/* ---- (2) engine ------------------------------------------------------------------------------------------ */
typedef struct gdbamd_interval_stats {
  int64_t num_cells, num_cells_in_window, num_records, num_heavy_incidences;
  uint64_t bytes_out, bytes_in_reference_cells;
  int32_t pages, write_launches;
  uint32_t err_bits;
  float ms_sweep, ms_site, ms_size, ms_write, ms_total, ms_write_kernel_avg;
  int32_t num_record_types, resolved_entry_bytes;   /* entry text table: distinct record types, slots, pool bytes; bytes per (record, sample) of the
                                                      * resolved matrix: 8 (offset + length), 5 (compact: no entry longer than 255 bytes), 0 (no matrix) */
  int64_t num_text_slots, text_pool_bytes;
  uint64_t num_remap_elements;           /* sum over re-indexed records of (calls with PL) x (merged genotypes): the PL remap work */
  uint64_t bytes_compressed;             /* output formats "z" / "b": bytes of the pages after BGZF compression (bytes_out: before) */
  float ms_compress; int32_t reserved1;  /* device time of the compression kernels */
  /* the six counters of the reference's GTProfileStats (src/main/cpp/include/genomicsdb/query_variants.h:67-124, -DDO_PROFILING),
   * indexed by the enum below; counted per interval on the device.  The reference counts cell VISITS of its iterators; here every
   * cell is touched once per interval, so: NUM_CELLS = cells of the staged window considered for the interval, IN_LEFT_SWEEP =
   * cells that begin before the interval and are still live at its first column (what gt_get_column has to find),
   * VALID_CELLS_IN_QUERY = cells that contribute to at least one record, ATTR_CELLS_ACCESSED = those x queried attributes,
   * PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS = cells cut short by the next cell of their own sample, OPERATOR_INVOCATIONS = records */
  uint64_t gt_profile_stats[6];
} gdbamd_interval_stats;
enum { GDBAMD_GT_NUM_CELLS = 0, GDBAMD_GT_NUM_CELLS_IN_LEFT_SWEEP, GDBAMD_GT_NUM_VALID_CELLS_IN_QUERY, GDBAMD_GT_NUM_ATTR_CELLS_ACCESSED,
       GDBAMD_GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS, GDBAMD_GT_NUM_OPERATOR_INVOCATIONS, GDBAMD_GT_NUM_STATS };

/* one attribute column in device memory; off == NULL for fixed-length attributes */
typedef struct gdbamd_device_column { const void* data; const uint32_t* off; } gdbamd_device_column;

void* gdbamd_engine_create(const char* query_json_text, int device);          /* NULL on error */
void* gdbamd_engine_create_format(const char* query_json_text, int device, int is_bcf, int use_missing_values_only_not_vector_end);  /* pages hold BCF2 records */
void* gdbamd_engine_create_output_format(const char* query_json_text, int device, const char* output_format, int use_missing_values_only_not_vector_end);  /* "", "bu", "z", "b": with "z" / "b" the pages hold BGZF blocks (no header, no EOF block) */
/* n host bytes as BGZF blocks (no EOF block), compressed by the device kernels; gdbamd_bgzf_bound(n) bytes of dst always suffice */
int gdbamd_bgzf_compress(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_len, float* ms_kernels);
/* the same with the kernel chosen: vcf_text != 0 = the anchored kernel the "z" stream uses for its pages of VCF text (matches begin at tabs /
 * newlines; any bytes give a valid stream), 0 = the byte-level kernel ("b", and what gdbamd_bgzf_compress runs) */
int gdbamd_bgzf_compress_mode(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t dst_cap, uint64_t* dst_len, float* ms_kernels, int vcf_text);
uint64_t gdbamd_bgzf_bound(uint64_t n);
void gdbamd_engine_destroy(void* engine);
int gdbamd_engine_num_fields(void* engine);                                     /* plan fields = staged attribute columns */
const char* gdbamd_engine_field_name(void* engine, int f);                      /* array attribute name of plan field f */
int gdbamd_engine_field_info(void* engine, int f, int* elem_type, int* is_var, int* fixed_num);
uint64_t gdbamd_engine_header(void* engine, char* dst, uint64_t cap);           /* VCF header text; returns its length */
/* stage begin-cells given in the reference binary-cell layout (host memory, column-major order) */
int gdbamd_engine_stage_cells(void* engine, const uint8_t* cells, uint64_t nbytes);
/* the same in parts (successive calls continue the column-major order); _end() makes one contiguous fragment in HBM */
int gdbamd_engine_stage_cells_begin(void* engine);
int gdbamd_engine_stage_cells_append(void* engine, const uint8_t* cells, uint64_t nbytes);
int gdbamd_engine_stage_cells_end(void* engine);
/* adopt a columnar fragment that already lives in HBM: row = QUERY row idx, begin/end = columns, cols[num_fields] */
int gdbamd_engine_adopt_device_fragment(void* engine, int64_t ncells, const int32_t* row, const int64_t* begin, const int64_t* end,
                                        const gdbamd_device_column* cols, int ncols, uint64_t reference_cell_bytes);
/* Last column of the first piece of [column_begin, column_end] that can be run on its own with byte-identical output: the cut
 * sits right before a cell begin >= column_begin + max_columns (the reference's sweep closes its interval at every cell
 * begin, query_variants.cc:478-505).  *piece_end = column_end when the interval is narrow enough or no cell begins in between. */
int gdbamd_engine_split_point(void* engine, int64_t column_begin, int64_t column_end, int64_t max_columns, int64_t* piece_end);
/* ColumnHistogramOperator (src/main/cpp/src/query_operations/variant_operations.cc:732-767; gt_mpi_gather --produce-histogram, tools/src/gt_mpi_gather.cc:404-411):
 * the staged begin-cells counted on the device by the bin of their begin column.  nbins must be (hist_end - hist_begin) / bin_size + 1; a cell beginning at or
 * before hist_begin counts for bin 0, one at or behind hist_end for the last bin.  accumulate != 0 adds to what counts holds (arrays streamed in windows). */
int gdbamd_engine_column_histogram(void* engine, uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, uint64_t* counts, uint64_t nbins, int accumulate);
/* VariantCallPrintOperator (src/main/cpp/src/query_operations/variant_operations.cc:803-843; gt_mpi_gather --print-calls, tools/src/gt_mpi_gather.cc:369-383):
 * the cells of the query's column intervals as the reference's JSON document - per interval first the intervals that began in front of it and
 * intersect its begin, then the cells that begin inside (SingleCellTileDBIterator, src/main/cpp/src/genomicsdb/genomicsdb_iterators.cc:181-510); selected and
 * formatted on the device, one thread per cell.  Call with dst == NULL for the length, then with a buffer of at least that size.  -1: error. */
int64_t gdbamd_engine_print_calls(void* engine, char* dst, uint64_t cap);
/* The other two SingleCellOperatorBase printers over the same cells, same calling convention.  mode 0: --print-calls; 1: --print-csv
 * (VariantCallPrintCSVOperator, variant_operations.cc:845-903: "row,begin,end,<fields>" per cell); 2: --print-AC (AlleleCountOperator, :905-1089:
 * "column REF ALT count" per normalised ALT allele named by a genotype, per query interval in the order of column, REF, ALT).  The reference's tests
 * hold no golden for modes 1 and 2: parity is against the oracle's restatement only. */
int64_t gdbamd_engine_print_cells(void* engine, int mode, char* dst, uint64_t cap);
/* "index_output_VCF" (src/main/cpp/src/config/json_config.cc:648, src/main/cpp/src/vcf/vcf_adapter.cc:275-295): the index htslib builds from a finished BGZF
 * file - <path>.tbi for a bgzip'ed VCF (tbx_index_build with the VCF preset), <path>.csi with min_shift 14 for a BGZF BCF2 file (bcf_index_build(.., 14)).
 * The file-writing VCFAdapter and gt_mpi_gather call it when the query JSON says "index_output_VCF": true and the format is "z" / "b". */
int gdbamd_build_output_index(const char* path, int is_bcf);
/* ColumnHistogramOperator::equi_partition_and_print_bins (variant_operations.cc:769-796): the text the reference prints - "Total T #bins P count/bins X.X", one
 * "first_column,last_column,count" line per partition of about equal cell count, an empty line.  Returns the text's length (dst may be NULL), -1 when
 * num_parts >= nbins.  The lines are what column_partitions of a loader JSON should be for P ranks of equal load. */
int64_t gdbamd_equi_partition_text(const uint64_t* counts, uint64_t nbins, uint64_t hist_begin, uint64_t bin_size, uint64_t num_parts, char* dst, uint64_t cap);
/* the staged fragment as a columnar file, and back: file -> HBM copies without parsing.  gdb_mi355_init opens
 * <workspace>/<array>/fragment.gdbamd when present (else cells.bin).  This is the build's own format (the
 * Intel TileDB fork's on-disk format of the reference, variant_storage_manager.cc:61-153, is not available). */
int gdbamd_engine_save_fragment(void* engine, const char* path);
/* the same file with every data section cut into 8 KiB tiles, each a raw DEFLATE stream (stored / fixed-Huffman blocks): compressed
 * bytes cross PCIe and are inflated on the device (fixed and dynamic Huffman codes), one thread per tile (the place of the gzip'd attribute tiles of the reference's
 * TileDB arrays, genomicsdb_iterators.cc:334-423) */
int gdbamd_engine_save_fragment_compressed(void* engine, const char* path);
int gdbamd_engine_load_fragment(void* engine, const char* path);
/* Arrays larger than the staging budget (GDBAMD_STAGE_BUDGET_MB, default 8192 MiB of cells per window): instead of staging by
 * hand, name a source and let the engine pass it through HBM in column windows, carrying the intervals that are still live at a
 * window's end into the next one on the device (the analogue of the reference's segment-at-a-time array iterator and
 * VariantQueryProcessorScanState: variant_storage_manager.cc:61-153, query_variants.h:126-191).
 *   open_array:         <dir>/fragment.gdbamd when it is valid for this query (checked against the array schema, the vid /
 *                       callset mapping and the cells.bin it was made from), else <dir>/cells.bin
 *   open_memory_cells:  begin-cells in host memory (the caller keeps them alive until the engine is destroyed)
 *   open_cell_callback: cells produced on demand; fn hands out the next chunk of whole begin columns (valid until the next call)
 *                       and returns 1, or returns 0 when there are no more; one pass, front to back
 * cover(column) stages windows until `column` is covered and returns the range [*lo, *hi] of query positions the staged
 * fragment serves: run intervals inside it, then ask again with *hi + 1. */
typedef int (*gdbamd_cell_chunk_fn)(void* user, const uint8_t** cells, uint64_t* nbytes);
int gdbamd_engine_open_array(void* engine, const char* dir);
int gdbamd_engine_open_memory_cells(void* engine, const uint8_t* cells, uint64_t nbytes);
int gdbamd_engine_open_cell_callback(void* engine, gdbamd_cell_chunk_fn fn, void* user);
/* Page-lock a range of the caller's own host memory (the cells handed to open_memory_cells / stage_cells_append, a chunk buffer of
 * a cell callback) so that the engine's host -> HBM copies are DMA transfers that overlap the kernels of the window being
 * computed; copies from pageable memory go through the runtime's bounce buffers on the calling thread and hold up the other
 * pipeline's launches.  The range stays valid and pinned until gdbamd_unpin_host_memory; 0 on success (on failure the memory is
 * left as it was: the copies still work, only slower). */
int gdbamd_pin_host_memory(const void* p, uint64_t nbytes);
int gdbamd_unpin_host_memory(const void* p);
int gdbamd_engine_cover(void* engine, int64_t column, int64_t* lo, int64_t* hi);
/* what is staged: #begin-cells and the sum of their reference binary-cell sizes ("bytes_in" of the byte accounting) */
int gdbamd_engine_staged_info(void* engine, int64_t* ncells, uint64_t* reference_cell_bytes);
/* reference bases for TileDB columns [begin, begin+len) (host pointer) */
int gdbamd_engine_set_reference(void* engine, int64_t begin, const char* bases, uint64_t len);
/* scan + combine one column interval.  The VCF body is produced page by page in HBM (arena_bytes per page); when
 * host_out != NULL the pages are copied there (at most host_cap bytes, *host_len = total body bytes). */
int gdbamd_engine_run_interval(void* engine, int64_t column_begin, int64_t column_end, uint64_t arena_bytes, char* host_out,
                               uint64_t host_cap, uint64_t* host_len, gdbamd_interval_stats* stats);

/* n query intervals of the staged fragment, up to `lanes` (1 .. 4) of them in flight at a time: lane l takes intervals l, l + lanes, ... on a
 * device pipeline of its own (own stream, entry table, matrix and page arenas) that works on the SAME staged fragment - the latency-bound
 * sweep / site / sizing kernels of one interval overlap with the store-bound page kernel of another.  The pages stay in HBM (each lane's
 * last page is valid until the lane's next interval) unless host_out != NULL: then interval i's body is copied to host_out[i] (at most
 * host_cap[i] bytes; host_len[i] = its length either way).  stats[i] belongs to interval i.  lanes = 1 is a loop of gdbamd_engine_run_interval.
 * (The reference scans a partition with one thread: tools/src/gt_mpi_gather.cc:322-366; this is the device's way to keep two batches of
 * the same partition in flight.) */
int gdbamd_engine_run_intervals(void* engine, int n, const int64_t* column_begins, const int64_t* column_ends, uint64_t arena_bytes, int lanes,
                                gdbamd_interval_stats* stats, char* const* host_out, const uint64_t* host_cap, uint64_t* host_len);

/* HBM a NEW lane pipeline takes for intervals of `interval_columns` positions at the query's sample count (~45 bytes of page and ~18 bytes
 * of tables per sample and position, the page capped by arena_bytes, + 2 GiB).  gdbamd_engine_run_intervals creates no more new lanes than
 * the device's free memory holds (4 GiB kept spare) and reports the clamp on stderr; lane pipelines stay allocated (grow-only) until
 * gdbamd_engine_release_lanes or the engine's destruction. */
int gdbamd_engine_lane_footprint(void* engine, int64_t interval_columns, uint64_t arena_bytes, uint64_t* bytes);
int gdbamd_engine_release_lanes(void* engine);

/* the same in two steps, for consumers that take the pages where they are (HBM): prepare = sweep, site and sizing passes of
 * the interval; next_page = the next <= arena_bytes of whole records.  *dev_ptr is device memory, valid until the next call
 * on this engine.  next_page returns 1 (a page), 0 (interval exhausted) or -1 (error).
 * (reference: the RWBuffer hand-over of GenomicsDBBCFGenerator::produce_next_batch, src/main/cpp/src/vcf/genomicsdb_bcf_generator.cc:96-125) */
int gdbamd_engine_prepare_interval(void* engine, int64_t column_begin, int64_t column_end);
int gdbamd_engine_next_page(void* engine, uint64_t arena_bytes, const void** dev_ptr, uint64_t* nbytes);

/* ---- (3) host-only helpers (no device needed) ----------------------------------------------------------- */
/* column partition of `rank` as the loader JSON defines it: begin from "column_partitions"[rank], end = next sorted begin - 1
 * (reference: GenomicsDBImportConfig::get_column_partition, src/main/cpp/src/config/json_config.cc:340-417) */
int gdbamd_column_partition(const char* loader_json_text, int rank, int64_t* begin, int64_t* end);

/* (g)VCF import: the files of the callset mapping -> begin-cells (reference binary cell layout, column-major) whose begin
 * column lies in [column_begin, column_end]; what vcf2tiledb's conversion step produces for one column partition
 * (reference: VCF2Binary::convert_VCF_to_binary_for_callset, src/main/cpp/src/vcf/vcf2binary.cc:991-1196, fields :715-989;
 * hand-over order of VCF2TileDBLoader, src/main/cpp/src/loader/tiledb_loader.cc:845-965).  file_root prefixes relative
 * "filename" entries (NULL / "": as they are).  *cells is malloc'ed: release with gdbamd_free.  0 on success. */
int gdbamd_import_cells(const char* vid_mapping_file, const char* callset_mapping_file, const char* file_root, int treat_deletions_as_intervals,
                        int64_t column_begin, int64_t column_end, uint8_t** cells, uint64_t* nbytes, int64_t* ncells);
void gdbamd_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
