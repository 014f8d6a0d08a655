#include "fragment.h"

#include <cstring>
#include <fstream>

namespace genomicsdb_amd {

VariantArraySchemaLite::VariantArraySchemaLite(const VidMapper& vid) {
  for (const auto& n : vid.schema_attribute_names()) {
    Attr a;
    a.name = n;
    if (n == "END") { a.elem = GDB_ET_INT; a.var = false; a.num = 1; a.elem_size = 8; attrs.push_back(a); continue; }
    const FieldInfo* fi = vid.get_field_info(n);
    if (!fi) throw VidMapperException("schema attribute " + n + " has no field info");
    a.elem = fi->m_num_dimensions > 1 ? GDB_ET_CHAR : fi->m_element_type;   // multi-D fields: a variable number of bytes
    a.var = fi->m_num_dimensions > 1 || !fi->is_fixed_length_field();
    a.num = a.var ? 0 : (int)fi->m_num_elements;
    a.elem_size = (a.elem == GDB_ET_INT || a.elem == GDB_ET_FLOAT) ? 4 : 1;
    attrs.push_back(a);
  }
}
int VariantArraySchemaLite::find(const std::string& n) const {
  for (size_t i = 0; i < attrs.size(); ++i) if (attrs[i].name == n) return (int)i;
  return -1;
}

CellStreamLayout::CellStreamLayout(const VariantQueryConfig& qc, const HostPlan& hp) : schema(qc.get_vid_mapper()) {
  attr_to_field.assign(schema.attrs.size(), -1);
  for (int f = 0; f < hp.plan.nfields; ++f) {
    int ai = schema.find(hp.field_names[(size_t)f]);
    if (ai < 0) throw UnknownQueryAttributeException("Invalid query attribute : " + hp.field_names[(size_t)f]);
    attr_to_field[(size_t)ai] = f;
  }
  row_map.assign((size_t)std::max<int64_t>(qc.get_num_rows_in_array(), 1), -1);
  for (uint64_t q = 0; q < qc.get_num_rows_to_query(); ++q) {
    int64_t r = qc.get_array_row_idx_for_query_row_idx(q);
    if (r >= 0 && (size_t)r < row_map.size()) row_map[(size_t)r] = (int32_t)q;
  }
}

HostFragment fragment_from_cells(const uint8_t* cells, size_t nbytes, const VariantQueryConfig& qc, const HostPlan& hp) {
  const VidMapper& vid = qc.get_vid_mapper();
  VariantArraySchemaLite schema(vid);
  const CombinePlan& pl = hp.plan;
  HostFragment fr;
  fr.cols.resize((size_t)pl.nfields);
  std::vector<int> attr_to_field(schema.attrs.size(), -1);
  for (int f = 0; f < pl.nfields; ++f) {
    int ai = schema.find(hp.field_names[(size_t)f]);
    if (ai < 0) throw UnknownQueryAttributeException("Invalid query attribute : " + hp.field_names[(size_t)f]);
    attr_to_field[(size_t)ai] = f;
    HostColumn& c = fr.cols[(size_t)f];
    c.var = schema.attrs[(size_t)ai].var;
    c.elem_size = schema.attrs[(size_t)ai].elem_size;
    c.fixed_num = schema.attrs[(size_t)ai].num;
    if (c.var) c.off.push_back(0);
  }
  // array row -> query row
  std::vector<int32_t> row_map((size_t)std::max<int64_t>(qc.get_num_rows_in_array(), 1), -1);
  for (uint64_t q = 0; q < qc.get_num_rows_to_query(); ++q) {
    int64_t r = qc.get_array_row_idx_for_query_row_idx(q);
    if (r >= 0 && (size_t)r < row_map.size()) row_map[(size_t)r] = (int32_t)q;
  }
  size_t off = 0;
  int64_t prev_col = INT64_MIN;
  int32_t prev_row = -1;
  while (off < nbytes) {
    if (off + 32 > nbytes) throw std::runtime_error("truncated cell stream");
    int64_t row, col;
    uint64_t cell_size;
    memcpy(&row, cells + off, 8);
    memcpy(&col, cells + off + 8, 8);
    memcpy(&cell_size, cells + off + 16, 8);
    if (cell_size < 32 || off + cell_size > nbytes) throw std::runtime_error("truncated cell stream");
    const uint8_t* p = cells + off + 24;
    const uint8_t* const cell_end = cells + off + cell_size;      // every read below is checked against the cell it belongs to
    int32_t qrow = (row >= 0 && (size_t)row < row_map.size()) ? row_map[(size_t)row] : -1;
    if (qrow >= 0) {
      if (col < prev_col || (col == prev_col && qrow <= prev_row)) throw std::runtime_error("cells are not in column-major (col,row) order");
      prev_col = col;
      prev_row = qrow;
    }
    for (size_t ai = 0; ai < schema.attrs.size(); ++ai) {
      const auto& a = schema.attrs[ai];
      uint32_t n = (uint32_t)a.num;
      if (a.var) {
        if (p + 4 > cell_end) throw std::runtime_error("cell size mismatch while parsing the cell stream");
        int32_t len; memcpy(&len, p, 4); p += 4;
        if (len < 0) throw std::runtime_error("cell size mismatch while parsing the cell stream");
        n = (uint32_t)len;
      }
      size_t bytes = (size_t)n * (size_t)a.elem_size;
      if (bytes > (size_t)(cell_end - p) || (ai == 0 && bytes < 8)) throw std::runtime_error("cell size mismatch while parsing the cell stream");
      if (qrow >= 0) {
        if (ai == 0) { int64_t e; memcpy(&e, p, 8); fr.end.push_back(e); }
        int f = attr_to_field[ai];
        if (f >= 0) {
          HostColumn& c = fr.cols[(size_t)f];
          c.data.insert(c.data.end(), p, p + bytes);
          if (c.var) c.off.push_back(c.off.back() + n);
        }
      }
      p += bytes;
    }
    if ((uint64_t)(p - (cells + off)) != cell_size) throw std::runtime_error("cell size mismatch while parsing the cell stream");
    if (qrow < 0 && row >= 0 && (size_t)row < row_map.size()) fr.marker_begin.push_back(col);
    if (qrow >= 0) {
      fr.row.push_back(qrow);
      fr.begin.push_back(col);
      fr.reference_cell_bytes += cell_size;
    }
    off += cell_size;
  }
  return fr;
}

std::vector<uint8_t> read_binary_file(const std::string& path) {
  std::ifstream ifs(path.c_str(), std::ios::binary);
  if (!ifs.is_open()) throw std::runtime_error("cannot open " + path);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
}

}  // namespace genomicsdb_amd
