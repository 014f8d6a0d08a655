// vcf_importer.h - product host layer: (g)VCF text -> begin-cells in the reference's binary cell layout, the step that
// CREATES what the scan-and-combine path consumes.
//
// What it restates: VCF2Binary::convert_VCF_to_binary_for_callset (reference src/main/cpp/src/vcf/vcf2binary.cc:991-1196; field
// conversion :715-989; deletions as intervals :1043-1059; INFO values of multi-sample files divided among the samples :34-53)
// for the array schema of VidMapper::build_tiledb_array_schema (src/main/cpp/src/utils/vid_mapper.cc:354-442), and the
// column-major hand-over of VCF2TileDBLoader (src/main/cpp/src/loader/tiledb_loader.cc:845-965).
//
//   cell = [row i64][col i64][cell_size u64][END i64][REF: i32 n + chars][ALT: i32 n + 'A|C|&' ('&' = <NON_REF>)]
//          [ID: i32 n + chars (only when the vid declares ID)][QUAL f32][FILTER: i32 n + n x i32 field idx]
//          [INFO attributes in vid order][FORMAT attributes in vid order]
//          fixed-length attribute = num x element (missing: TileDB null), var-length = i32 num + num x element (missing: num 0)
//
// Not done (documented in DESIGN.md): htslib's record-level checks, CSV / buffer-stream inputs, intervals that cross a column
// partition boundary (a cell belongs to the partition of its begin column), multi-dimensional (allele-specific) fields.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "vid_mapper.h"

namespace genomicsdb_amd {

class VCF2BinaryException : public std::runtime_error {
 public:
  explicit VCF2BinaryException(const std::string& m) : std::runtime_error("VCF2BinaryException : " + m) {}
};

struct ImportOptions {
  bool treat_deletions_as_intervals = false;       // loader JSON key of the same name
  int64_t column_begin = 0, column_end = INT64_MAX - 1;   // column partition: cells that begin inside are kept
  std::string file_root;                           // prefix of relative "filename" entries of the callset mapping
};
struct ImportStats { int64_t num_files = 0, num_records = 0, num_cells = 0, num_spanning_cells = 0; uint64_t num_bytes = 0; };   // num_spanning_cells: intervals replayed at the partition begin

// every callset of vid's callset mapping (file, idx_in_file, row_idx); cells in column-major (column, row) order
std::vector<uint8_t> import_callsets_to_cells(const VidMapper& vid, const ImportOptions& opt, ImportStats* stats = nullptr);

}  // namespace genomicsdb_amd
