// gdb_types.h - plain-old-data descriptors shared by the host layer and the HIP kernels.
//
// CombinePlan is the device-side digest of what the reference keeps in VariantQueryConfig +
// BroadCombinedGVCFOperator's INFO/FORMAT tuple vectors (reference
// src/main/cpp/src/query_operations/broad_combined_gvcf.cc:140-264): which queried attribute plays
// which role, in which order INFO / FORMAT fields are emitted, and the query flags.
//
// FragmentView is a column-interval of the sparse array staged in HBM as structure-of-arrays:
// begin-cells only, in TileDB column-major (col,row) order, one dense array per fixed-length
// attribute and offsets+payload per variable-length attribute (the layout TileDB fragments have on
// disk, and the replacement for the reference's AoS BufferVariantCell / VariantCall model,
// include/genomicsdb/variant_cell.h:36-137, variant.h:75-281).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GDB_HD __host__ __device__ __forceinline__
#define GDB_HD_NOINLINE __host__ __device__
#else
#define GDB_HD inline
#define GDB_HD_NOINLINE
#endif

#define GDB_MAX_FIELDS 48          // queried attributes per plan
#define GDB_MAX_INFO_FIELDS 24
#define GDB_MAX_FORMAT_FIELDS 24
#define GDB_MAX_MERGED_ALLELES 128  // per record (REF included)
#define GDB_MAX_INPUT_ALLELES 64    // per cell (REF included)
#define GDB_MAX_PLOIDY 8            // general-ploidy genotype enumeration (G-length fields, min-PL genotype)
#define GDB_MAX_INFO_VECTOR 64       // elements of an element_wise_sum INFO vector
#define GDB_MAX_ID_TOKENS 16        // distinct ';'-separated ID tokens per output record
#define GDB_MAX_FILTER_IDS 16       // distinct FILTER ids united in one output record
#define GDB_MAX_PIPELINES_PER_PROCESS 256  // elements of the __constant__ context array (kernels/gdb_pipeline.hip: c_ex), one per live pipeline (on gfx9 the constant
                                           // address space is ordinary device memory read with scalar loads: nothing limits it to 64 KB)
#define GDB_MAX_HISTOGRAM_FIELDS 8  // composite (bins, counts) INFO fields reduced with histogram_sum

// htslib / TileDB sentinels (reference include/vcf/vcf.h:59-218)
#define GDB_BCF_INT32_MISSING ((int32_t)0x80000000)
#define GDB_BCF_INT32_VECTOR_END ((int32_t)0x80000001)
#define GDB_BCF_FLOAT_MISSING_BITS 0x7F800001u
#define GDB_BCF_FLOAT_VECTOR_END_BITS 0x7F800002u
#define GDB_TILEDB_NULL_INT32 ((int32_t)0x7FFFFFFF)
#define GDB_TILEDB_NULL_FLOAT_BITS 0x7F7FFFFFu
#define GDB_TILEDB_NULL_CHAR ((char)127)

enum GdbLength { GDB_VL_FIXED = 0, GDB_VL_VAR = 1, GDB_VL_A = 2, GDB_VL_G = 3, GDB_VL_R = 4, GDB_VL_P = 5, GDB_VL_PP = 6 };
enum GdbElem { GDB_ET_INT = 0, GDB_ET_FLOAT = 1, GDB_ET_CHAR = 2, GDB_ET_FLAG = 3 };
enum GdbCombineOp {
  GDB_OP_SUM = 0, GDB_OP_MEAN, GDB_OP_MEDIAN, GDB_OP_DP, GDB_OP_MOVE_TO_FORMAT, GDB_OP_ELEMENT_WISE_SUM,
  GDB_OP_CONCATENATE, GDB_OP_HISTOGRAM_SUM, GDB_OP_UNKNOWN
};

// error bits raised by kernels (never silently ignored: the host turns them into exceptions)
enum GdbErr {
  GDB_ERR_OVERLAP_NOT_REFBLOCK_OR_DELETION = 1u << 0,  // query_variants.cc:533-535
  GDB_ERR_TOO_MANY_MERGED_ALLELES = 1u << 1,
  GDB_ERR_TOO_MANY_INPUT_ALLELES = 1u << 2,
  GDB_ERR_UNSUPPORTED_PLOIDY = 1u << 3,               // ploidy above GDB_MAX_PLOIDY
  GDB_ERR_FLOAT_RANGE = 1u << 4,                      // retired (round 4: put_float prints every float, "%g" restated exactly); the bit stays reserved
  GDB_ERR_ARENA_OVERFLOW = 1u << 5,
  GDB_ERR_INTERNAL = 1u << 6,
  GDB_ERR_TOO_MANY_ID_TOKENS = 1u << 7,               // more than GDB_MAX_ID_TOKENS distinct ID tokens in one record
  GDB_ERR_INFO_VECTOR_TOO_LONG = 1u << 8,             // element_wise_sum over more than GDB_MAX_INFO_VECTOR elements
  GDB_ERR_CELL_STREAM = 1u << 9,                      // malformed / unsorted binary cell stream at staging
  GDB_ERR_TOO_MANY_FILTER_IDS = 1u << 10              // more than GDB_MAX_FILTER_IDS distinct FILTER ids in one record
};

struct GdbFieldDesc {
  int32_t elem;        // GdbElem
  int32_t length;      // GdbLength
  int32_t fixed_num;   // #elements when length == FIXED
  int32_t combine_op;  // GdbCombineOp (INFO fields)
  int32_t is_info, is_format;
  int32_t known_enum;  // reference KnownVariantFieldsEnum or -1
  // 2-D fields (allele-specific annotations): elem is GDB_ET_CHAR (the attribute is a byte blob, gdb_asa.hpp), `length` the
  // descriptor of dimension 0, elem2d the type of the values inside, delim0 / delim1 the VCF delimiters of the two dimensions
  int32_t ndim;
  int32_t elem2d;
  int32_t delim0, delim1;
};

// One attribute column of a staged fragment.  var-length: off[c]..off[c+1] elements of `data`.
struct GdbColumn {
  const void* data;
  const uint32_t* off;  // null for fixed-length columns
};

struct FragmentView {
  int64_t ncells;
  const int32_t* row;    // array row idx per cell
  const int64_t* begin;  // column (genomic position) of the cell
  const int64_t* end;    // END attribute
  GdbColumn col[GDB_MAX_FIELDS];  // indexed by plan field idx
  // begin columns (ascending) of the cells of array rows the query does NOT ask for: the reference's scan still sees those
  // cells and closes the current interval at each of their begins (query_variants.cc:478-505 runs before the row filter of
  // :507), so they survive staging as position-only boundary markers
  int64_t nmarkers;
  const int64_t* marker_begin;
};

struct CombinePlan {
  int32_t nfields;
  GdbFieldDesc field[GDB_MAX_FIELDS];
  // roles (plan field idx or -1)
  int32_t f_REF, f_ALT, f_GT, f_PL, f_DP, f_MIN_DP, f_DP_FORMAT, f_FILTER, f_QUAL, f_ID;
  // emission order
  int32_t n_info;
  int32_t info_field[GDB_MAX_INFO_FIELDS];    // INFO fields with a combine op, query order (DP excluded)
  int32_t n_histogram;                        // histogram_sum pairs, emitted after the other INFO fields (broad_combined_gvcf.cc:562-600)
  int32_t histogram_bin_field[GDB_MAX_HISTOGRAM_FIELDS], histogram_count_field[GDB_MAX_HISTOGRAM_FIELDS];
  int32_t n_format;
  int32_t format_field[GDB_MAX_FORMAT_FIELDS];  // FORMAT fields, query order, INFO-DP pseudo entry last (if queried)
  // flags
  int32_t produce_GT_field, produce_FILTER_field, sites_only_query, min_PL_GT_for_spanning_deletions;
  int32_t max_diploid_alt_alleles;
  int32_t id_order_unordered_set;   // ID union in the order of a Release build of the reference (std::unordered_set<std::string>) instead of sorted
  int32_t qual_combine_op;  // GDB_OP_UNKNOWN unless the vid configures one
  int32_t num_query_rows;   // N: sample columns of the output
  // BCF2 ("bu") output: typed binary records instead of text (vcf_adapter.cc:475-509); ids of the header dictionary
  int32_t bcf_mode;                           // 0: VCF text
  int32_t use_missing_values_not_vector_end;  // the JNI flag for htsjdk (variant_field_handler.cc:846-866)
  int32_t bcf_n_sample;                       // n_sample of every record (0 for sites-only queries)
  int32_t bcf_end_id, bcf_dp_id;              // dictionary index of END / DP
  int32_t bcf_id[GDB_MAX_FIELDS];             // dictionary index of the VCF name of plan field f (-1: none)
};

// Query-row mapping + contig table live in device memory next to the fragment.
struct GdbContig { int64_t offset, length; int32_t name_off, name_len; int32_t rid, pad; };   // rid: index in the header's contig dictionary

struct QueryWindow {
  int64_t qb, qe;               // inclusive column interval being scanned
  const int32_t* row_to_qrow;   // array row -> query row idx or -1
  int32_t num_array_rows;
  const GdbContig* contigs;     // sorted by offset
  int32_t ncontigs;
  const char* contig_names;
  const char* ref_bases;        // reference bases for [ref_begin, ref_begin + ref_len)
  int64_t ref_begin, ref_len;
};
