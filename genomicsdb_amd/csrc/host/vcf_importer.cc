// vcf_importer.cc - see vcf_importer.h
#include "vcf_importer.h"

#include <zlib.h>

#include <algorithm>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <strings.h>
#include <unordered_map>

namespace genomicsdb_amd {
namespace {

constexpr int32_t kTileDBNullInt = INT32_MAX;
constexpr uint32_t kTileDBNullFloatBits = 0x7F7FFFFFu;  // FLT_MAX
constexpr char kTileDBNullChar = 127;
constexpr int32_t kBcfIntMissing = INT32_MIN;
constexpr uint32_t kBcfFloatMissingBits = 0x7F800001u;

struct Tok { const char* p; size_t n; };
bool tok_is(const Tok& t, const char* s) { return t.n == strlen(s) && memcmp(t.p, s, t.n) == 0; }
std::string tok_str(const Tok& t) { return std::string(t.p, t.n); }
void split(const char* p, size_t n, char sep, std::vector<Tok>& out) {
  out.clear();
  size_t b = 0;
  for (size_t i = 0; i <= n; ++i)
    if (i == n || p[i] == sep) { out.push_back(Tok{p + b, i - b}); b = i + 1; }
}

template <class T> void put(std::vector<uint8_t>& o, T v) { const uint8_t* b = (const uint8_t*)&v; o.insert(o.end(), b, b + sizeof(T)); }
void put_chars(std::vector<uint8_t>& o, const char* p, size_t n) { put<int32_t>(o, (int32_t)n); o.insert(o.end(), p, p + n); }

std::string read_text_maybe_gz(const std::string& path) {
  gzFile f = gzopen(path.c_str(), "rb");      // transparent for plain text; reads every member of a bgzip file
  if (!f) throw VCF2BinaryException("cannot open " + path);
  std::string s;
  std::vector<char> buf(1 << 20);
  for (;;) {
    int n = gzread(f, buf.data(), (unsigned)buf.size());
    if (n < 0) { gzclose(f); throw VCF2BinaryException("read error in " + path); }
    if (n == 0) break;
    s.append(buf.data(), (size_t)n);
  }
  gzclose(f);
  return s;
}

struct Attr {            // one INFO / FORMAT attribute of the schema, in cell order
  const FieldInfo* fi;
  bool info;
  bool sum_like;         // FieldInfo::is_VCF_field_combine_operation_sum (vid_mapper.cc:1187-1193)
  const FieldInfo* parent = nullptr;   // the composite field of a flattened tuple element
};

// bcf_get_variant_type(line, j) == VCF_INDEL && strlen(REF) > strlen(ALT) (vcf2binary.cc:1046-1057; htslib bcf_set_variant_type),
// for plain base alleles: symbolic alleles and '*' are never INDELs
bool deletion_indel(const Tok& ref, const Tok& alt) {
  if (alt.n == 0 || alt.p[0] == '<' || tok_is(alt, "*") || tok_is(alt, ".")) return false;
  if (ref.n == 1 && alt.n == 1) return false;
  auto up = [](char c) { return (char)toupper((unsigned char)c); };
  size_t r = 0, a = 0;
  while (r < ref.n && a < alt.n && up(ref.p[r]) == up(alt.p[a])) { ++r; ++a; }
  if (a < alt.n && r == ref.n) return false;      // insertion
  if (r < ref.n && a == alt.n) return true;       // pure deletion
  if (r == ref.n && a == alt.n) return false;
  size_t re = ref.n - 1, ae = alt.n - 1;
  while (re > r && ae > a && up(ref.p[re]) == up(alt.p[ae])) { --re; --ae; }
  if (ae == a) { if (re == r) return false; return up(ref.p[re]) == up(alt.p[ae]) && ref.n > alt.n; }
  if (re == r) return up(ref.p[re]) == up(alt.p[ae]) && ref.n > alt.n;
  return false;
}

int64_t parse_int(const Tok& t, const std::string& what) {
  std::string s = tok_str(t);
  char* e = nullptr;
  long long v = strtoll(s.c_str(), &e, 10);
  if (s.empty() || *e) throw VCF2BinaryException("not an integer: '" + s + "' in " + what);
  return v;
}
double parse_double(const Tok& t, const std::string& what) {
  std::string s = tok_str(t);
  char* e = nullptr;
  double v = strtod(s.c_str(), &e);
  if (s.empty() || *e) throw VCF2BinaryException("not a number: '" + s + "' in " + what);
  return v;
}

// A 2-dimensional value ("1.0,2.0|3.0|4.0", the elements of a type tuple alternating inside an inner vector) -> the byte blob of
// THIS tuple element: <u64 size of data><inner vectors><u64 #entries><u64 offsets x (#entries + 1)>
// (GenomicsDBMultiDVectorField::parse_and_store_numeric, genomicsdb_multid_vector_field.cc:238-464; vcf2binary.cc:857-913).
// "" and "NaN" are bcf missing (str_to_element, :31-87), shorter tuple elements are padded with missing, and a sum-like INFO
// field of a multi-sample VCF is divided up among the samples (histogram_sum: the counts only, vcf2binary.cc:862-884).
void encode_2d(std::vector<uint8_t>& o, const Attr& a, const FieldInfo& parent, bool present, const Tok& text, int n_samples, int sample_idx) {
  const FieldInfo& f = *a.fi;
  if (!present || tok_is(text, ".")) { put<int32_t>(o, 0); return; }
  const unsigned ntuple = parent.get_num_elements_in_tuple(), me = f.is_flattened_field() ? f.m_element_index_in_tuple : 0u;
  const bool is_int = f.m_element_type == GDB_ET_INT;
  bool divide = a.sum_like && a.info && n_samples > 1;
  if (divide && f.m_VCF_field_combine_operation == GDB_OP_HISTOGRAM_SUM) divide = me == 1u;
  std::vector<uint8_t> data;
  std::vector<uint64_t> offsets(1, 0);
  std::vector<Tok> inners, toks;
  split(text.p, text.n, f.m_vcf_delimiter[0], inners);
  if (inners.empty()) inners.push_back(Tok{text.p, 0});
  for (const Tok& inner : inners) {
    toks.clear();
    split(inner.p, inner.n, f.m_vcf_delimiter[1], toks);
    if (toks.empty()) toks.push_back(Tok{inner.p, 0});
    const size_t n = (toks.size() + ntuple - 1) / ntuple;     // elements of the longest tuple element
    for (size_t i = 0; i < n; ++i) {
      const size_t at = i * ntuple + me;
      const Tok t = at < toks.size() ? toks[at] : Tok{inner.p, 0};
      const bool missing = t.n == 0 || (t.n == 3 && strncasecmp(t.p, "NaN", 3) == 0);
      if (is_int) {
        int64_t v = kBcfIntMissing;
        if (!missing) {
          v = strtoll(tok_str(t).c_str(), nullptr, 0);
          if (divide) { int64_t q = v / n_samples, r = v % n_samples; v = q + (sample_idx < r ? 1 : 0); }
        }
        put<int32_t>(data, (int32_t)v);
      } else if (missing) put<uint32_t>(data, kBcfFloatMissingBits);
      else { float v = strtof(tok_str(t).c_str(), nullptr); if (divide) v = v / (float)n_samples; put<float>(data, v); }
    }
    offsets.push_back((uint64_t)data.size());
  }
  const size_t blob = 8 + data.size() + 8 + 8 * offsets.size();
  put<int32_t>(o, (int32_t)blob);
  put<uint64_t>(o, (uint64_t)data.size());
  o.insert(o.end(), data.begin(), data.end());
  put<uint64_t>(o, (uint64_t)(offsets.size() - 1));
  for (uint64_t x : offsets) put<uint64_t>(o, x);
}

// one INFO / FORMAT value string -> attribute bytes (vcf2binary.cc:771-969).  present = the key exists in the record
void encode_values(std::vector<uint8_t>& o, const Attr& a, bool present, const Tok& text, int n_samples, int sample_idx, std::vector<Tok>& scratch) {
  const FieldInfo& f = *a.fi;
  if (f.m_num_dimensions == 2) { encode_2d(o, a, a.parent ? *a.parent : f, present, text, n_samples, sample_idx); return; }
  const bool missing = !present || tok_is(text, ".");
  if (f.m_element_type == GDB_ET_FLAG) { o.push_back((uint8_t)(present ? 1 : kTileDBNullChar)); return; }
  if (f.m_element_type == GDB_ET_CHAR) {
    if (missing) put<int32_t>(o, 0); else put_chars(o, text.p, text.n);
    return;
  }
  const bool is_int = f.m_element_type == GDB_ET_INT;
  const bool fixed = f.is_fixed_length_field();
  scratch.clear();
  if (!missing) split(text.p, text.n, ',', scratch);
  if (scratch.empty()) {
    if (fixed) for (unsigned i = 0; i < f.m_num_elements; ++i) { if (is_int) put<int32_t>(o, kTileDBNullInt); else put<uint32_t>(o, kTileDBNullFloatBits); }
    else put<int32_t>(o, 0);
    return;
  }
  if (fixed && scratch.size() != f.m_num_elements)
    throw VCF2BinaryException("field " + f.m_name + ": " + std::to_string(scratch.size()) + " values, expected " + std::to_string(f.m_num_elements));
  if (!fixed) put<int32_t>(o, (int32_t)scratch.size());
  const bool divide = a.sum_like && a.info && n_samples > 1;     // divide_up_among_samples (vcf2binary.cc:34-53)
  for (const Tok& t : scratch) {
    const bool dot = tok_is(t, ".");
    if (is_int) {
      int64_t v = dot ? (int64_t)kBcfIntMissing : parse_int(t, f.m_name);
      if (divide && !dot) {
        int64_t q = v / n_samples, r = v % n_samples;
        if (r < 0) { r += n_samples; --q; }                     // floor division
        v = q + (sample_idx < r ? 1 : 0);
      }
      put<int32_t>(o, (int32_t)v);
    } else if (dot) {
      put<uint32_t>(o, kBcfFloatMissingBits);
    } else {
      float v = (float)parse_double(t, f.m_name);
      if (divide) v = v / (float)n_samples;
      put<float>(o, v);
    }
  }
}

// GT -> allele indices, phase flags interleaved for length "PP" (vcf2binary.cc:923-959)
void encode_gt(std::vector<uint8_t>& o, const FieldInfo& f, bool present, const Tok& text) {
  std::vector<int32_t> alleles, phases;
  if (!present || tok_is(text, ".")) {
    alleles.push_back(-1);       // htslib parses GT '.' as one missing allele
  } else {
    size_t b = 0;
    for (size_t i = 0; i <= text.n; ++i) {
      if (i == text.n || text.p[i] == '/' || text.p[i] == '|') {
        Tok a{text.p + b, i - b};
        alleles.push_back(tok_is(a, ".") ? -1 : (int32_t)parse_int(a, "GT"));
        if (i < text.n) phases.push_back(text.p[i] == '|' ? 1 : 0);
        b = i + 1;
      }
    }
  }
  std::vector<int32_t> vals;
  if (f.m_length_descriptor == GDB_VL_PP) {
    vals.push_back(alleles[0]);
    for (size_t i = 1; i < alleles.size(); ++i) { vals.push_back(phases[i - 1]); vals.push_back(alleles[i]); }
  } else vals = alleles;
  put<int32_t>(o, (int32_t)vals.size());
  for (int32_t v : vals) put<int32_t>(o, v);
}

struct Cell { int64_t row, col; size_t off, len; };

}  // namespace

std::vector<uint8_t> import_callsets_to_cells(const VidMapper& vid, const ImportOptions& opt, ImportStats* stats) {
  if (!vid.is_initialized() || !vid.is_callset_mapping_initialized()) throw VCF2BinaryException("vid and callset mappings are needed");
  // attribute order of the schema (same walk as VidMapper::schema_attribute_names)
  const bool has_id = vid.get_field_info("ID") != nullptr;
  std::vector<Attr> info_attrs, fmt_attrs;
  for (unsigned i = 0; i < vid.get_num_fields(); ++i) {
    const FieldInfo& f = vid.get_field_info(i);
    if (f.m_name == "END") continue;
    const GdbCombineOp op = f.m_VCF_field_combine_operation;
    const bool sum_like = op == GDB_OP_SUM || op == GDB_OP_DP || op == GDB_OP_ELEMENT_WISE_SUM || op == GDB_OP_HISTOGRAM_SUM;
    if (f.m_unsupported_on_device && (f.m_is_vcf_INFO_field || f.m_is_vcf_FORMAT_field))
      throw VCF2BinaryException("field " + f.m_name + ": fields of more than 2 dimensions are not imported by this build");
    if (f.get_num_elements_in_tuple() > 1u) {   // a composite is not an attribute, its flattened tuple elements are
      if (f.m_num_dimensions != 2) throw VCF2BinaryException("field " + f.m_name + ": tuple elements are only imported for 2-dimensional fields");
      continue;
    }
    const FieldInfo* parent = f.is_flattened_field() ? &vid.get_field_info((unsigned)f.m_parent_composite_field_idx) : nullptr;
    if (f.m_is_vcf_INFO_field) { Attr a{&f, true, sum_like}; a.parent = parent; info_attrs.push_back(a); }
  }
  for (unsigned i = 0; i < vid.get_num_fields(); ++i) {
    const FieldInfo& f = vid.get_field_info(i);
    if (f.get_num_elements_in_tuple() > 1u) continue;
    const FieldInfo* parent = f.is_flattened_field() ? &vid.get_field_info((unsigned)f.m_parent_composite_field_idx) : nullptr;
    if (f.m_name != "END" && f.m_is_vcf_FORMAT_field) { Attr a{&f, false, false}; a.parent = parent; fmt_attrs.push_back(a); }
  }
  // callsets grouped by file, in mapping order
  std::vector<std::string> files;
  std::unordered_map<std::string, std::vector<const CallSetInfo*>> by_file;
  for (const CallSetInfo& cs : vid.get_callsets()) {
    if (cs.m_filename.empty()) throw VCF2BinaryException("callset " + cs.m_name + " has no \"filename\"");
    if (!by_file.count(cs.m_filename)) files.push_back(cs.m_filename);
    by_file[cs.m_filename].push_back(&cs);
  }
  std::vector<uint8_t> bytes;
  std::vector<Cell> cells;
  ImportStats st;
  // Intervals that begin in front of the column partition and reach into it are part of the partition: per row the LATEST cell
  // beginning at or before the partition begin counts - if it still covers the partition begin it is handed over first, at its own
  // coordinates; a later cell of the row that ends before the partition drops it
  // (LoaderOperatorBase::handle_intervals_spanning_partition_begin, load_operators.cc:33-79).
  struct Spanning { int64_t col = INT64_MIN, end = INT64_MIN; std::vector<uint8_t> cell; };
  std::unordered_map<int64_t, Spanning> spanning;
  std::vector<Tok> cols, alts, info_kv, fmt_keys, svals, scratch;
  std::vector<uint8_t> body;
  for (const std::string& fn : files) {
    const std::string path = (!fn.empty() && fn[0] != '/' && !opt.file_root.empty()) ? opt.file_root + "/" + fn : fn;
    const std::string text = read_text_maybe_gz(path);
    ++st.num_files;
    std::vector<int64_t> sample_row;   // file sample idx -> array row (-1: not imported)
    int n_samples = 0;
    size_t pos = 0;
    while (pos < text.size()) {
      size_t eol = text.find('\n', pos);
      if (eol == std::string::npos) eol = text.size();
      const char* lp = text.data() + pos;
      size_t ln = eol - pos;
      pos = eol + 1;
      if (ln && lp[ln - 1] == '\r') --ln;
      if (ln == 0) continue;
      if (lp[0] == '#') {
        if (ln > 6 && memcmp(lp, "#CHROM", 6) == 0) {
          split(lp, ln, '\t', cols);
          n_samples = cols.size() > 9 ? (int)cols.size() - 9 : 0;
          sample_row.assign((size_t)n_samples, -1);
          // callset -> row through idx_in_file (the name in the mapping may differ from the file's)
          for (const CallSetInfo* cs : by_file[fn]) {
            if (cs->m_idx_in_file < 0 || cs->m_idx_in_file >= n_samples) throw VCF2BinaryException("idx_in_file out of range for callset " + cs->m_name);
            const std::string want = tok_str(cols[9 + (size_t)cs->m_idx_in_file]);
            for (int s = 0; s < n_samples; ++s) if (tok_str(cols[9 + (size_t)s]) == want) sample_row[(size_t)s] = cs->m_row_idx;
          }
        }
        continue;
      }
      split(lp, ln, '\t', cols);
      if (cols.size() < 8) throw VCF2BinaryException("short record line in " + path);
      ++st.num_records;
      ContigInfo ci;
      if (!vid.get_contig_info(tok_str(cols[0]), ci)) throw VCF2BinaryException("contig " + tok_str(cols[0]) + " is not in the vid mapping");
      const int64_t col = ci.m_tiledb_column_offset + parse_int(cols[1], "POS") - 1;
      const Tok& ref = cols[3];
      alts.clear();
      if (!tok_is(cols[4], ".")) split(cols[4].p, cols[4].n, ',', alts);
      info_kv.clear();
      if (!tok_is(cols[7], ".")) split(cols[7].p, cols[7].n, ';', info_kv);
      auto info_find = [&](const std::string& key, Tok& val) -> bool {      // (a repeated key: the last one counts)
        bool found = false;
        for (const Tok& kv : info_kv) {
          const char* eq = (const char*)memchr(kv.p, '=', kv.n);
          const size_t kn = eq ? (size_t)(eq - kv.p) : kv.n;
          if (kn == key.size() && memcmp(kv.p, key.data(), kn) == 0) { val = eq ? Tok{eq + 1, kv.n - kn - 1} : Tok{kv.p + kv.n, 0}; found = true; }
        }
        return found;
      };
      int64_t end = col;
      Tok endv;
      if (info_find("END", endv)) end = ci.m_tiledb_column_offset + parse_int(endv, "END") - 1;
      else if (opt.treat_deletions_as_intervals)
        for (const Tok& a : alts) if (deletion_indel(ref, a)) { end = col + (int64_t)ref.n - 1; break; }
      if (col > opt.column_end) continue;
      const bool before_partition = col < opt.column_begin;
      if (before_partition) {   // only the latest cell per row at or before the partition begin matters: note it, build the bytes only if it reaches in
        bool any_candidate = false;
        for (int s = 0; s < n_samples; ++s) {
          if (sample_row[(size_t)s] < 0) continue;
          Spanning& sp = spanning[sample_row[(size_t)s]];
          if (col >= sp.col) { sp.col = col; sp.end = end; sp.cell.clear(); if (end >= opt.column_begin) any_candidate = true; }
        }
        if (!any_candidate) continue;
      } else if (col == opt.column_begin) {
        for (int s = 0; s < n_samples; ++s) if (sample_row[(size_t)s] >= 0) { Spanning& sp = spanning[sample_row[(size_t)s]]; sp.col = col; sp.end = end; sp.cell.clear(); }   // a cell AT the begin replaces what spanned it
      }
      std::string alt_ser;
      for (size_t i = 0; i < alts.size(); ++i) { if (i) alt_ser += '|'; if (tok_is(alts[i], "<NON_REF>")) alt_ser += '&'; else alt_ser.append(alts[i].p, alts[i].n); }
      fmt_keys.clear();
      if (cols.size() > 8) split(cols[8].p, cols[8].n, ':', fmt_keys);
      for (int s = 0; s < n_samples; ++s) {
        if (sample_row[(size_t)s] < 0) continue;
        body.clear();
        put<int64_t>(body, end);
        put_chars(body, ref.p, ref.n);
        put_chars(body, alt_ser.data(), alt_ser.size());
        if (has_id) { if (cols[2].n && !tok_is(cols[2], ".")) put_chars(body, cols[2].p, cols[2].n); else put<int32_t>(body, 0); }
        if (tok_is(cols[5], ".")) put<uint32_t>(body, kTileDBNullFloatBits); else put<float>(body, (float)parse_double(cols[5], "QUAL"));
        if (tok_is(cols[6], ".")) put<int32_t>(body, 0);
        else {
          split(cols[6].p, cols[6].n, ';', scratch);
          put<int32_t>(body, (int32_t)scratch.size());
          for (const Tok& t : scratch) {
            const FieldInfo* ff = vid.get_field_info(tok_str(t));
            if (!ff) throw VCF2BinaryException("FILTER " + tok_str(t) + " is not in the vid mapping");
            put<int32_t>(body, ff->m_field_idx);
          }
        }
        for (const Attr& a : info_attrs) {
          Tok v{nullptr, 0};
          const bool present = info_find(a.fi->m_vcf_name, v);
          encode_values(body, a, present, v, n_samples, s, scratch);
        }
        svals.clear();
        if (cols.size() > 9 + (size_t)s) split(cols[9 + (size_t)s].p, cols[9 + (size_t)s].n, ':', svals);
        for (const Attr& a : fmt_attrs) {
          Tok v{nullptr, 0};
          bool present = false;
          for (size_t i = 0; i < fmt_keys.size() && i < svals.size(); ++i)
            if (fmt_keys[i].n == a.fi->m_vcf_name.size() && memcmp(fmt_keys[i].p, a.fi->m_vcf_name.data(), fmt_keys[i].n) == 0) { v = svals[i]; present = true; }
          if (a.fi->m_vcf_name == "GT") encode_gt(body, *a.fi, present, v);
          else encode_values(body, a, present, v, 1, 0, scratch);
        }
        const uint64_t cell_size = 16 + 8 + body.size();
        if (before_partition) {
          Spanning& sp = spanning[sample_row[(size_t)s]];
          if (sp.col == col && sp.end >= opt.column_begin) {     // still this row's latest: keep the bytes aside
            sp.cell.clear();
            put<int64_t>(sp.cell, sample_row[(size_t)s]); put<int64_t>(sp.cell, col); put<uint64_t>(sp.cell, cell_size);
            sp.cell.insert(sp.cell.end(), body.begin(), body.end());
          }
          continue;
        }
        Cell c{sample_row[(size_t)s], col, bytes.size(), (size_t)cell_size};
        put<int64_t>(bytes, c.row); put<int64_t>(bytes, c.col); put<uint64_t>(bytes, cell_size);
        bytes.insert(bytes.end(), body.begin(), body.end());
        cells.push_back(c);
      }
    }
  }
  for (auto& kv : spanning) {
    const Spanning& sp = kv.second;
    if (sp.col >= opt.column_begin || sp.end < opt.column_begin || sp.cell.empty()) continue;
    Cell c{kv.first, sp.col, bytes.size(), sp.cell.size()};
    bytes.insert(bytes.end(), sp.cell.begin(), sp.cell.end());
    cells.push_back(c);
    ++st.num_spanning_cells;
  }
  std::stable_sort(cells.begin(), cells.end(), [](const Cell& a, const Cell& b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
  std::vector<uint8_t> out;
  out.reserve(bytes.size());
  for (const Cell& c : cells) out.insert(out.end(), bytes.begin() + (ptrdiff_t)c.off, bytes.begin() + (ptrdiff_t)(c.off + c.len));
  st.num_cells = (int64_t)cells.size();
  st.num_bytes = out.size();
  if (stats) *stats = st;
  return out;
}

}  // namespace genomicsdb_amd
