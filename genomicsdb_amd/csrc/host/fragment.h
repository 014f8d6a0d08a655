// fragment.h - product host layer: a column interval of the sparse array as structure-of-arrays
// ("columnar fragment"), the form that is DMA'd to HBM.  Built from the reference's binary-cell stream
// (cell layout: reference src/main/cpp/src/vcf/vcf2binary.cc:991-1196, parser variant_cell.cc:79-117).
// END-copy duplicates (load_operators.cc:161-187) never enter a fragment: only begin cells are staged.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "combine_plan.h"

namespace genomicsdb_amd {

struct HostColumn {
  std::vector<uint8_t> data;
  std::vector<uint32_t> off;  // [ncells+1] element offsets, empty for fixed-length columns
  bool var = false;
  int elem_size = 4;
  int fixed_num = 1;
};

struct HostFragment {
  std::vector<int32_t> row;  // QUERY row idx (array rows outside the query are dropped at staging)
  std::vector<int64_t> begin, end;
  std::vector<HostColumn> cols;  // per plan field
  std::vector<int64_t> marker_begin;  // begins of the cells of array rows outside the query (boundary markers, see FragmentView)
  uint64_t reference_cell_bytes = 0;  // sum of the reference binary-cell sizes of the staged cells ("bytes_in")
  int64_t ncells() const { return (int64_t)row.size(); }
  void append_cells_from(const HostFragment& o, int64_t first, int64_t last);
};

class VariantArraySchemaLite {  // attribute order / typing of the array (reference vid_mapper.cc:354-442)
 public:
  struct Attr { std::string name; GdbElem elem; bool var; int num; int elem_size; };
  explicit VariantArraySchemaLite(const VidMapper& vid);
  std::vector<Attr> attrs;
  int find(const std::string& n) const;
};

// what a parser of the binary cell stream needs to know: attribute order / typing, which plan field each attribute feeds,
// and the array row -> query row map (-1: row not queried)
struct CellStreamLayout {
  VariantArraySchemaLite schema;
  std::vector<int> attr_to_field;
  std::vector<int32_t> row_map;
  CellStreamLayout(const VariantQueryConfig& qc, const HostPlan& hp);
};

// cells: begin-cells in column-major order.  Only rows the query asks for are kept.  (Host-side parser: used by the CPU test
// harness; the product stages through DevicePipeline::append_cells, which takes the stream apart on the device.)
HostFragment fragment_from_cells(const uint8_t* cells, size_t nbytes, const VariantQueryConfig& qc, const HostPlan& hp);

// the inverse (used to hand the same synthetic data to the CPU oracle in bench/tests): plan fields only are not enough to
// rebuild a full cell, so this lives with the synthetic generator instead.

std::vector<uint8_t> read_binary_file(const std::string& path);

}  // namespace genomicsdb_amd
