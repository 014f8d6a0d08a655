// gdb_oracle_capi.cc - TEST ORACLE driver + C entry points (ctypes).  NOT PRODUCT CODE.
//
// Restates the control flow of tools/src/gt_mpi_gather.cc:322-366 (scan_and_produce_Broad_GVCF):
// build the operator, then for every query interval call scan_and_operate until the scan state
// reports done, draining the output buffer after each call (the "-p <page>" batched mode when
// buffer_limit != 0).
#include "gdb_oracle_combine.hpp"
#include "gdb_oracle_print.hpp"

#include <chrono>
#include <cstdio>

using namespace gdb_oracle;

namespace {

struct RunResult { std::string text; uint64_t num_records = 0; uint64_t num_cells = 0; double scan_seconds = 0; };

RunResult run_query(const std::string& query_json_text, const uint8_t* cells, uint64_t nbytes, int64_t partition_begin, int64_t partition_end,
                    uint64_t buffer_limit, bool with_header, const ReferenceGenome* external_ref) {
  oracle_json::Value q = oracle_json::parse(query_json_text);
  VidMapper vid;
  if (q.HasMember("vid_mapping_file")) vid.load_vid(oracle_json::parse_file(q["vid_mapping_file"].GetString()));
  else if (q.HasMember("vid_mapping")) vid.load_vid(q["vid_mapping"]);
  else throw OracleException("query JSON needs vid_mapping_file or vid_mapping");
  if (q.HasMember("callset_mapping_file")) vid.load_callsets(oracle_json::parse_file(q["callset_mapping_file"].GetString()));
  else if (q.HasMember("callset_mapping") || q.HasMember("callsets")) vid.load_callsets(q);
  else throw OracleException("query JSON needs callset_mapping_file or callset_mapping");
  VariantArray array;
  array.schema = build_array_schema(vid);
  array.num_rows = (int64_t)vid.row_to_callset.size();
  array.load(cells, nbytes, partition_begin, partition_end);
  QueryConfig qc;
  qc.read_query_json(q, vid, 0);
  qc.do_query_bookkeeping(array.schema, vid, array.num_rows, 0);
  std::string tmpl;
  if (!qc.vcf_header_filename.empty()) tmpl = oracle_json::read_text_file(qc.vcf_header_filename);
  ReferenceGenome ref_local;
  const ReferenceGenome* ref = external_ref;
  if (!ref && !qc.reference_genome.empty()) { ref_local.load_fasta(qc.reference_genome); ref = &ref_local; }
  BroadCombinedGVCFOperator op(vid, qc, tmpl, ref, qc.max_diploid_alt_alleles_that_can_be_genotyped);
  op.buffer_limit = buffer_limit;
  RunResult rr;
  if (with_header) { rr.text = op.header_text; op.bytes_in_buffer = op.header_text.size(); }
  QueryProcessor qp(&array);
  auto t0 = std::chrono::steady_clock::now();
  ScanState ss;
  unsigned n_int = std::max<unsigned>(1u, (unsigned)qc.column_intervals.size());
  for (unsigned i = 0; i < n_int; ++i) {
    while (!ss.end()) {
      qp.scan_and_operate(qc, op, i, true, &ss);
      rr.text += op.out;  // do_output(); rw_buffer.m_num_valid_bytes = 0
      op.out.clear();
      op.bytes_in_buffer = 0;
    }
    ss.reset();
  }
  rr.scan_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  rr.num_records = op.num_records;
  rr.num_cells = qp.stats.num_cells;
  return rr;
}

}  // namespace

extern "C" {

// Runs one produce-Broad-GVCF query.  Returns 0 on success; *out is malloc'ed (release with oracle_free).
int oracle_run_query(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, int64_t partition_begin, int64_t partition_end,
                     uint64_t buffer_limit, int with_header, char** out, uint64_t* out_len, uint64_t* num_records, double* scan_seconds,
                     char* err, uint64_t errlen) {
  try {
    RunResult rr = run_query(query_json_text, cells, nbytes, partition_begin, partition_end, buffer_limit, with_header != 0, nullptr);
    *out = (char*)malloc(rr.text.size() + 1);
    memcpy(*out, rr.text.data(), rr.text.size());
    (*out)[rr.text.size()] = 0;
    *out_len = rr.text.size();
    if (num_records) *num_records = rr.num_records;
    if (scan_seconds) *scan_seconds = rr.scan_seconds;
    return 0;
  } catch (const std::exception& e) {
    if (err && errlen) snprintf(err, errlen, "%s", e.what());
    return 1;
  }
}

// Same query, but REF bases of "nobody starts here" records come from the synthetic reference of the bench generator
// (genomicsdb_amd/synth/gvcf_synth.cc: base(column) = "ACGT"[hash2(seed ^ 0x5bd1e9955bd1e995, column) & 3], column = contig offset + position).
static inline uint64_t synth_splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
int oracle_run_query_synthetic_reference(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, uint64_t seed, uint64_t buffer_limit, int with_header,
                                         char** out, uint64_t* out_len, uint64_t* num_records, double* scan_seconds, char* err, uint64_t errlen) {
  try {
    // the generator's bases are a function of the COLUMN (flattened genome); the operator asks by (contig, position in contig)
    std::map<std::string, int64_t> contig_offset;
    {
      oracle_json::Value q = oracle_json::parse(query_json_text);
      VidMapper vid;
      if (q.HasMember("vid_mapping_file")) vid.load_vid(oracle_json::parse_file(q["vid_mapping_file"].GetString()));
      else if (q.HasMember("vid_mapping")) vid.load_vid(q["vid_mapping"]);
      for (auto& c : vid.contigs) contig_offset[c.name] = c.offset;
    }
    ReferenceGenome ref;
    ref.custom = [seed, contig_offset](const std::string& contig, int64_t pos) -> char {
      auto it = contig_offset.find(contig);
      const int64_t col = pos + (it == contig_offset.end() ? 0 : it->second);
      uint64_t s = (seed ^ 0x5bd1e9955bd1e995ull) ^ ((uint64_t)col * 0xD6E8FEB86659FD93ull);
      return "ACGT"[synth_splitmix64(s) & 3];
    };
    RunResult rr = run_query(query_json_text, cells, nbytes, 0, INT64_MAX - 1, buffer_limit, with_header != 0, &ref);
    *out = (char*)malloc(rr.text.size() + 1);
    memcpy(*out, rr.text.data(), rr.text.size());
    (*out)[rr.text.size()] = 0;
    *out_len = rr.text.size();
    if (num_records) *num_records = rr.num_records;
    if (scan_seconds) *scan_seconds = rr.scan_seconds;
    return 0;
  } catch (const std::exception& e) {
    if (err && errlen) snprintf(err, errlen, "%s", e.what());
    return 1;
  }
}

// gt_mpi_gather --print-calls (gdb_oracle_print.hpp): the JSON document of the query's cells; *out malloc'ed
int oracle_print_cells(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, int mode, char** out, uint64_t* out_len, char* err, uint64_t errlen);
int oracle_print_calls(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, char** out, uint64_t* out_len, char* err, uint64_t errlen) {
  return oracle_print_cells(query_json_text, cells, nbytes, 0, out, out_len, err, errlen);
}
// mode 0: --print-calls, 1: --print-csv, 2: --print-AC
int oracle_print_cells(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, int mode, char** out, uint64_t* out_len, char* err, uint64_t errlen) {
  try {
    oracle_json::Value q = oracle_json::parse(query_json_text);
    VidMapper vid;
    if (q.HasMember("vid_mapping_file")) vid.load_vid(oracle_json::parse_file(q["vid_mapping_file"].GetString()));
    else if (q.HasMember("vid_mapping")) vid.load_vid(q["vid_mapping"]);
    else throw OracleException("query JSON needs vid_mapping_file or vid_mapping");
    if (q.HasMember("callset_mapping_file")) vid.load_callsets(oracle_json::parse_file(q["callset_mapping_file"].GetString()));
    else if (q.HasMember("callset_mapping") || q.HasMember("callsets")) vid.load_callsets(q);
    else throw OracleException("query JSON needs callset_mapping_file or callset_mapping");
    VariantArray array;
    array.schema = build_array_schema(vid);
    array.num_rows = (int64_t)vid.row_to_callset.size();
    array.load(cells, nbytes, 0, INT64_MAX - 1);
    QueryConfig qc;
    qc.read_query_json(q, vid, 0);
    qc.do_query_bookkeeping(array.schema, vid, array.num_rows, 0);
    std::string o = mode == 0 ? print_calls(array, qc, vid) : mode == 1 ? print_csv(array, qc) : print_allele_counts(array, qc);
    *out = (char*)malloc(o.size() + 1); memcpy(*out, o.data(), o.size()); (*out)[o.size()] = 0; *out_len = o.size();
    return 0;
  } catch (const std::exception& e) { if (err && errlen) snprintf(err, errlen, "%s", e.what()); return 1; }
}

void oracle_free(char* p) { free(p); }

// ---- the oracle's own JSON / gzip readers (oracle_json.hpp; nothing is shared with the product since round 4) laid open to the tests, which
// compare them with Python's json / gzip modules on every fixture file (tests/test_common_utils.py; the product's readers: tests/hostsim) ----
static void json_dump(const oracle_json::Value& v, std::string& o) {
  using V = oracle_json::Value;
  char buf[64];
  switch (v.type) {
    case V::Null: o += "n"; break;
    case V::Bool: o += v.b ? "t" : "f"; break;
    case V::Int: snprintf(buf, sizeof buf, "i%lld", (long long)v.i); o += buf; break;
    case V::Double: snprintf(buf, sizeof buf, "d%.17g", v.d); o += buf; break;
    case V::String: o += "s"; for (unsigned char c : v.s) { snprintf(buf, sizeof buf, "%02x", c); o += buf; } break;
    case V::Array: o += "["; for (auto& e : v.arr) { json_dump(e, o); o += ","; } o += "]"; break;
    case V::Object: o += "{"; for (auto& kv : v.obj) { for (unsigned char c : kv.first) { snprintf(buf, sizeof buf, "%02x", c); o += buf; } o += ":"; json_dump(kv.second, o); o += ","; } o += "}"; break;
  }
}
// canonical dump of the parsed document (members in document order); *out malloc'ed
int oracle_json_dump(const char* text, char** out, uint64_t* out_len, char* err, uint64_t errlen) {
  try {
    std::string o;
    json_dump(oracle_json::parse(text), o);
    *out = (char*)malloc(o.size() + 1); memcpy(*out, o.data(), o.size()); (*out)[o.size()] = 0; *out_len = o.size();
    return 0;
  } catch (const std::exception& e) { if (err && errlen) snprintf(err, errlen, "%s", e.what()); return 1; }
}
// the bytes oracle_json::gz_read_all gives for a (possibly gzip / BGZF compressed) file
int oracle_gz_read_all(const char* path, char** out, uint64_t* out_len, char* err, uint64_t errlen) {
  try {
    std::string o = oracle_json::gz_read_all(path);
    *out = (char*)malloc(o.size() + 1); memcpy(*out, o.data(), o.size()); (*out)[o.size()] = 0; *out_len = o.size();
    return 0;
  } catch (const std::exception& e) { if (err && errlen) snprintf(err, errlen, "%s", e.what()); return 1; }
}

// format_float exposed for the float-format unit tests
int oracle_format_float(float v, char* buf, uint64_t buflen) {
  std::string s;
  format_float(s, v);
  snprintf(buf, buflen, "%s", s.c_str());
  return (int)s.size();
}

// genotype-order known answers (src/test/cpp/src/test_non_diploid_mapper.cc): enumerates the merged
// genotypes for (num_alleles, ploidy) and returns, per merged genotype, the input genotype index or -1.
// lut_m2i[k] = input allele index for merged allele k (-1 = missing).
int oracle_genotype_map(const int64_t* lut_m2i, unsigned num_merged, int non_ref_exists, unsigned ploidy, uint64_t input_size, int64_t* out, uint64_t out_cap) {
  CombineAllelesLUT lut;
  lut.resize_luts_if_needed(1u, num_merged);
  for (unsigned k = 0; k < num_merged; ++k) if (lut_m2i[k] >= 0) lut.add_input_merged_idx_pair(0u, lut_m2i[k], k);
  uint64_t n = 0;
  remap_based_on_genotype(input_size, 0, lut, num_merged, non_ref_exists != 0, ploidy, [&](uint64_t o, bool has, uint64_t i) {
    if (o < out_cap) out[o] = has ? (int64_t)i : -1;
    n = std::max(n, o + 1);
  });
  return (int)n;
}

}  // extern "C"

#ifdef ORACLE_MAIN
int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s <query.json> <cells.bin> [buffer_limit]\n", argv[0]); return 2; }
  std::string q = oracle_json::read_text_file(argv[1]);
  std::string cells = oracle_json::read_text_file(argv[2]);
  uint64_t lim = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;
  try {
    RunResult rr = run_query(q, (const uint8_t*)cells.data(), cells.size(), 0, INT64_MAX - 1, lim, true, nullptr);
    fwrite(rr.text.data(), 1, rr.text.size(), stdout);
    fprintf(stderr, "records %llu scan %.6f s\n", (unsigned long long)rr.num_records, rr.scan_seconds);
  } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
#endif
