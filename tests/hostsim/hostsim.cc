// hostsim.cc - TEST HARNESS, NOT PRODUCT CODE.
//
// Runs the GDB_HD stage/record/entry functions of genomicsdb_amd/csrc/core (the bodies of the HIP kernels) in
// plain serial loops on the CPU, with std::sort / serial scans standing in for the device sorts and scans.
// Purpose: debug the index arithmetic of the device pipeline in a container without a GPU and keep a CPU
// regression of it next to the oracle.  The shipped library (libgenomicsdb_amd.so) contains no such path.
#include <set>
#include <unordered_set>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <numeric>
#include <string>
#include <vector>

#include "../../genomicsdb_amd/csrc/core/gdb_stages.hpp"
#include "../../genomicsdb_amd/csrc/core/gdb_bcf.hpp"
#include "../../genomicsdb_amd/csrc/host/combine_plan.h"
#include "../../genomicsdb_amd/csrc/host/fragment.h"
#include <map>
#include "../../genomicsdb_amd/csrc/core/gdb_calls.hpp"
#include "../../genomicsdb_amd/csrc/host/reference_genome.h"
#include "../../genomicsdb_amd/csrc/common/gz_text.hpp"

using namespace genomicsdb_amd;

namespace {

FragmentView make_view(const HostFragment& fr) {
  FragmentView v;
  memset(&v, 0, sizeof(v));
  v.ncells = fr.ncells();
  v.row = fr.row.data(); v.begin = fr.begin.data(); v.end = fr.end.data();
  v.nmarkers = (int64_t)fr.marker_begin.size(); v.marker_begin = fr.marker_begin.data();
  for (size_t f = 0; f < fr.cols.size(); ++f) {
    v.col[f].data = fr.cols[f].data.data();
    v.col[f].off = fr.cols[f].var ? fr.cols[f].off.data() : nullptr;
  }
  return v;
}

std::string run_interval(const HostPlan& hp, const HostFragment& hf, const VidMapper& vid, const ReferenceGenomeInfo& ref, int64_t qb, int64_t qe,
                         int rows_per_chunk, int records_per_run, uint32_t& err_out) {
  const CombinePlan& pl = hp.plan;
  const FragmentView fr = make_view(hf);
  const int64_t C = fr.ncells;
  const int64_t N = pl.num_query_rows;
  uint32_t err = 0;
  std::string out;
  if (C == 0) return out;
  // S0 classify
  std::vector<uint64_t> vmask(C); std::vector<uint32_t> cflags(C); std::vector<int32_t> dpval(C), k_lo(C), k_hi(C); std::vector<int64_t> eff_end(C);
  CellMeta cm{vmask.data(), cflags.data(), dpval.data(), eff_end.data(), k_lo.data(), k_hi.data()};
  for (int64_t c = 0; c < C; ++c) classify_cell(fr, pl, cm, c, &err);
  // S1 row index (stable sort by row)
  std::vector<int64_t> perm(C), rm_begin(C), row_ptr(N + 1, 0);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return fr.row[a] < fr.row[b]; });
  for (int64_t c = 0; c < C; ++c) row_ptr[fr.row[c] + 1]++;
  for (int64_t r = 0; r < N; ++r) row_ptr[r + 1] += row_ptr[r];
  // S2 effective END
  std::vector<int64_t> span(C);
  for (int64_t j = 0; j < C; ++j) stage_eff_end(fr, cm, perm.data(), j, rm_begin.data(), span.data(), &err);
  // clip the window to what the cells can reach (keeps relative positions small)
  // S3 events
  std::vector<uint64_t> keys(2 * C + 2 * fr.nmarkers);
  for (int64_t c = 0; c < C; ++c) stage_event_keys(fr, cm, c, 0, qb, qe, keys.data());
  for (int64_t m = 0; m < fr.nmarkers; ++m) stage_marker_keys(fr, m, qb, qe, keys.data() + 2 * C + 2 * m);
  std::sort(keys.begin(), keys.end());
  const int64_t NE = 2 * C + 2 * fr.nmarkers;
  std::vector<int64_t> incl(NE);
  std::vector<int32_t> run_end(NE), run_excl(NE);
  { int64_t acc = 0; for (int64_t i = 0; i < NE; ++i) { acc = packed_add(acc, stage_event_delta(keys[i])); incl[i] = acc; } }
  int64_t U = 0;
  for (int64_t i = 0; i < NE; ++i) { run_end[i] = stage_is_run_end(keys.data(), NE, i); run_excl[i] = (int32_t)U; U += run_end[i]; }
  std::vector<int64_t> bpos(U + 1), nrec(U + 1), rbase(U + 1);
  std::vector<int32_t> bcov(U + 1), bdel(U + 1);
  Boundaries bd{bpos.data(), bcov.data(), bdel.data(), nrec.data()};
  for (int64_t i = 0; i < NE; ++i) stage_boundary_write(keys.data(), incl.data(), run_excl.data(), i, run_end[i], bd, qb);
  int64_t P = 0;
  for (int64_t u = 0; u < U; ++u) { nrec[u] = stage_boundary_nrec(bd, U, u); rbase[u] = P; P += nrec[u]; }
  if (P == 0) { err_out |= err; return out; }
  std::vector<int64_t> rstart(P), rend(P);
  for (int64_t k = 0; k < P; ++k) stage_record_expand(bd, rbase.data(), U, k, rstart.data(), rend.data());
  RecordTable rec{P, rstart.data(), rend.data()};
  // S4 ranges + difference arrays
  const int nf = pl.n_format;
  std::vector<int32_t> dfmt((size_t)std::max(nf, 1) * (P + 1), 0), ddp(P + 1, 0), dnr(P + 1, 0);
  DiffArrays da{dfmt.data(), ddp.data(), dnr.data(), P + 1};
  std::vector<int64_t> heavy_count(C), hoff(C + 1, 0);
  for (int64_t c = 0; c < C; ++c) stage_cell_ranges(fr, pl, cm, rec, c, 0, qb, qe, da, heavy_count.data());
  // S5 scans
  for (int i = 0; i < nf; ++i) { int32_t a = 0; for (int64_t k = 0; k <= P; ++k) { a += dfmt[(size_t)i * (P + 1) + k]; dfmt[(size_t)i * (P + 1) + k] = a; } }
  { int32_t a = 0, b = 0; for (int64_t k = 0; k <= P; ++k) { a += ddp[k]; ddp[k] = a; b += dnr[k]; dnr[k] = b; } }
  for (int64_t c = 0; c < C; ++c) hoff[c + 1] = hoff[c] + heavy_count[c];
  const int64_t T = hoff[C];
  // S6 incidences
  std::vector<uint64_t> ikeys(T); std::vector<int64_t> ivals(T);
  for (int64_t c = 0; c < C; ++c) stage_incidence_fill(fr, cm, hoff.data(), c, 0, N, ikeys.data(), ivals.data(), nullptr);
  {
    std::vector<int64_t> order(T);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return ikeys[a] < ikeys[b]; });
    std::vector<uint64_t> k2(T); std::vector<int64_t> v2(T);
    for (int64_t t = 0; t < T; ++t) { k2[t] = ikeys[order[t]]; v2[t] = ivals[order[t]]; }
    ikeys.swap(k2); ivals.swap(v2);
  }
  std::vector<int64_t> hbase(P + 1);
  for (int64_t k = 0; k <= P; ++k) hbase[k] = stage_heavy_base(ikeys.data(), T, N, k);
  std::vector<uint32_t> i2m_off(T + 1, 0);
  for (int64_t t = 0; t < T; ++t) i2m_off[t + 1] = i2m_off[t] + GDB_CF_NALT(cflags[ivals[t]]) + 1;
  std::vector<int8_t> i2m(i2m_off[T] + 1), gto((size_t)GDB_MAX_PLOIDY * T + 2);
  std::vector<uint8_t> iflags(T + 1);
  HeavyLists hl{hbase.data(), ivals.data(), i2m_off.data(), i2m.data(), iflags.data(), gto.data()};
  // S7 site pass 0
  std::vector<uint8_t> num_alleles(P), rflags(P);
  std::vector<uint32_t> fmt_mask(P), prefix_len(P);
  SiteOut so{num_alleles.data(), rflags.data(), fmt_mask.data(), prefix_len.data()};
  // name tables / window
  std::string refwin;
  int64_t ref_begin = rstart[0];
  int64_t ref_len = rstart[P - 1] - rstart[0] + 1;
  if (ref.is_initialized()) refwin = ref.window(vid, ref_begin, ref_len);
  QueryWindow qw;
  memset(&qw, 0, sizeof(qw));
  qw.qb = qb; qw.qe = qe; qw.contigs = hp.contigs.data(); qw.ncontigs = (int)hp.contigs.size(); qw.contig_names = hp.contig_names.data();
  qw.ref_bases = refwin.empty() ? nullptr : refwin.data(); qw.ref_begin = ref_begin; qw.ref_len = refwin.empty() ? 0 : ref_len;
  NameTables nt{hp.names_text.data(), hp.field_name_off.data(), hp.field_name_len.data(), hp.filter_name_off.data(), hp.filter_name_len.data(), (int)hp.filter_name_off.size()};
  PresenceCounts pc{dfmt.data(), ddp.data(), dnr.data(), P + 1};
  SiteCtx sx{fr, pl, cm, rec, hl, pc, nt, qw, so};
  std::vector<float> tie_buf((size_t)16 * (T + 1));
  unsigned long long tie_used = 0;
  sx.tie.buf = tie_buf.data(); sx.tie.used = &tie_used; sx.tie.capacity = tie_buf.size();
  for (int64_t k = 0; k < P; ++k) { CountSink cs; site_emit(sx, k, cs, true, &err); prefix_len[k] = (uint32_t)cs.n; }
  // S8 entry sizing per (record, row chunk); rows walk runs of records
  RowIndex ri{row_ptr.data(), perm.data(), rm_begin.data()};
  EntryCtx ex{fr, pl, cm, hl};
  const int64_t nchunks = (N + rows_per_chunk - 1) / rows_per_chunk;
  std::vector<uint64_t> chunk_size((size_t)P * nchunks, 0), chunk_off((size_t)P * nchunks + 1, 0);
  for (int64_t k0 = 0; k0 < P; k0 += records_per_run) {
    const int64_t k1 = std::min<int64_t>(P, k0 + records_per_run);
    for (int64_t r = 0; r < N; ++r) {
      RowWalker w; w.init(ri, (int32_t)r, rstart[k0]);
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t c = w.live(ri, cm, rstart[k]);
        if (!fmt_mask[k]) continue;
        RecordInfo rinfo = load_record_info(so, hl, k);
        CountSink cs = entry_emit(ex, rinfo, c, CountSink(), &err);
        chunk_size[(size_t)k * nchunks + r / rows_per_chunk] += 1 + cs.n;
      }
    }
    for (int64_t k = k0; k < k1; ++k) { chunk_size[(size_t)k * nchunks] += prefix_len[k]; chunk_size[(size_t)k * nchunks + nchunks - 1] += 1; }
  }
  for (size_t i = 0; i < chunk_size.size(); ++i) chunk_off[i + 1] = chunk_off[i] + chunk_size[i];
  out.resize(chunk_off.back());
  // S9 write: prefix (site pass 1) + entries
  for (int64_t k = 0; k < P; ++k) { ByteSink bs(&out[chunk_off[(size_t)k * nchunks]]); site_emit(sx, k, bs, false, &err); out[chunk_off[(size_t)(k + 1) * nchunks] - 1] = '\n'; }
  for (int64_t k0 = 0; k0 < P; k0 += records_per_run) {
    const int64_t k1 = std::min<int64_t>(P, k0 + records_per_run);
    std::vector<uint64_t> cursor((size_t)(k1 - k0) * nchunks);
    for (int64_t k = k0; k < k1; ++k) for (int64_t ch = 0; ch < nchunks; ++ch) cursor[(size_t)(k - k0) * nchunks + ch] = chunk_off[(size_t)k * nchunks + ch] + (ch == 0 ? prefix_len[k] : 0);
    for (int64_t r = 0; r < N; ++r) {
      RowWalker w; w.init(ri, (int32_t)r, rstart[k0]);
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t c = w.live(ri, cm, rstart[k]);
        if (!fmt_mask[k]) continue;
        RecordInfo rinfo = load_record_info(so, hl, k);
        uint64_t& cur = cursor[(size_t)(k - k0) * nchunks + r / rows_per_chunk];
        out[cur++] = '\t';
        ByteSink bs = entry_emit(ex, rinfo, c, ByteSink(&out[cur]), &err);
        cur = (uint64_t)(bs.p - out.data());
      }
    }
  }
  err_out |= err;
  return out;
}

}  // namespace

extern "C" {

int hostsim_run_query(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, int with_header, int rows_per_chunk, int records_per_run,
                      char** out, uint64_t* out_len, uint32_t* err_bits, char* errmsg, uint64_t errlen) {
  try {
    VariantQueryConfig qc;
    qc.read_from_json(mini_json::parse(query_json_text), 0, "");
    qc.do_query_bookkeeping(qc.get_vid_mapper().get_num_callsets(), 0);
    std::string tmpl;
    if (!qc.get_vcf_header_filename().empty()) tmpl = mini_json::read_text_file(qc.get_vcf_header_filename());
    HostPlan hp = build_combine_plan(qc, tmpl);
    HostFragment hf = fragment_from_cells(cells, nbytes, qc, hp);
    ReferenceGenomeInfo ref;
    if (!qc.get_reference_genome().empty()) ref.initialize(qc.get_reference_genome());
    std::string text = with_header ? hp.header_text : std::string();
    uint32_t err = 0;
    unsigned nint = qc.get_num_column_intervals();
    if (nint == 0) text += run_interval(hp, hf, qc.get_vid_mapper(), ref, 0, INT64_MAX - 1, rows_per_chunk, records_per_run, err);
    for (unsigned i = 0; i < nint; ++i)
      text += run_interval(hp, hf, qc.get_vid_mapper(), ref, qc.get_column_begin(i), qc.get_column_end(i), rows_per_chunk, records_per_run, err);
    *out = (char*)malloc(text.size() + 1);
    memcpy(*out, text.data(), text.size());
    *out_len = text.size();
    *err_bits = err;
    return 0;
  } catch (const std::exception& e) {
    snprintf(errmsg, errlen, "%s", e.what());
    return 1;
  }
}
// gt_mpi_gather --print-calls with the kernel bodies on the host (core/gdb_calls.hpp: calls_select + calls_emit_cell; the document frame as
// CombineEngine::print_calls writes it)
int hostsim_print_cells(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, int mode, char** out, uint64_t* out_len, char* errmsg, uint64_t errlen);
int hostsim_print_calls(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, char** out, uint64_t* out_len, char* errmsg, uint64_t errlen) {
  return hostsim_print_cells(query_json_text, cells, nbytes, 0, out, out_len, errmsg, errlen);
}
// mode 1: --print-csv lines, mode 2: --print-AC (the lines of calls_emit_allele_lines counted like CombineEngine::print_allele_counts)
int hostsim_print_cells(const char* query_json_text, const uint8_t* cells, uint64_t nbytes, int mode, char** out, uint64_t* out_len, char* errmsg, uint64_t errlen) {
  try {
    VariantQueryConfig qc;
    qc.read_from_json(mini_json::parse(query_json_text), 0, "");
    qc.do_query_bookkeeping(qc.get_vid_mapper().get_num_callsets(), 0);
    HostPlan hp = build_combine_plan(qc, "");
    HostFragment hf = fragment_from_cells(cells, nbytes, qc, hp);
    const CombinePlan& pl = hp.plan;
    const FragmentView fr = make_view(hf);
    const int64_t C = fr.ncells, N = pl.num_query_rows;
    uint32_t err = 0;
    std::vector<uint64_t> vmask(C); std::vector<uint32_t> cflags(C); std::vector<int32_t> dpval(C), k_lo(C), k_hi(C); std::vector<int64_t> eff_end(C);
    CellMeta cm{vmask.data(), cflags.data(), dpval.data(), eff_end.data(), k_lo.data(), k_hi.data()};
    for (int64_t c = 0; c < C; ++c) classify_cell(fr, pl, cm, c, &err);
    std::vector<int64_t> perm(C), rm_begin(C), span(C);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return fr.row[a] < fr.row[b]; });
    (void)N;
    for (int64_t j = 0; j < C; ++j) stage_eff_end(fr, cm, perm.data(), j, rm_begin.data(), span.data(), &err);
    std::string names_text; std::vector<int32_t> names_off;
    for (const auto& nm : hp.field_names) { names_off.push_back((int32_t)names_text.size()); names_text += nm; }
    names_off.push_back((int32_t)names_text.size());
    std::vector<int64_t> q2a;
    { CellStreamLayout L(qc, hp); for (size_t r = 0; r < L.row_map.size(); ++r) if (L.row_map[r] >= 0) { if ((size_t)L.row_map[r] >= q2a.size()) q2a.resize((size_t)L.row_map[r] + 1, 0); q2a[(size_t)L.row_map[r]] = (int64_t)r; } }
    CallsNames names{names_text.data(), names_off.data(), q2a.empty() ? nullptr : q2a.data()};
    QueryWindow qw;
    memset(&qw, 0, sizeof(qw));
    qw.contigs = hp.contigs.data(); qw.ncontigs = (int32_t)hp.contigs.size(); qw.contig_names = hp.contig_names.data();
    const std::string ip = "    ", p0 = ip + ip, p1 = p0 + ip;
    std::string o = mode == 0 ? "{\n" + ip + "\"variant_calls\": [\n" : std::string();
    const int gt_step = pl.f_GT >= 0 && pl.field[pl.f_GT].length == GDB_VL_PP ? 2 : 1;
    std::vector<std::pair<int64_t, int64_t>> ivs;
    for (unsigned i = 0; i < qc.get_num_column_intervals(); ++i) ivs.emplace_back(qc.get_column_begin(i), qc.get_column_end(i));
    const bool whole = ivs.empty();
    if (whole) ivs.emplace_back(0, INT64_MAX - 1);
    unsigned printed = 0;
    for (const auto& iv : ivs) {
      std::string body;
      for (int64_t c = 0; c < C; ++c) {
        int64_t end;
        if (!calls_select(fr, eff_end.data(), c, iv.first, iv.second, !whole, end)) continue;
        if (mode != 0) {
          CountSink cs;
          if (mode == 1) calls_emit_csv(cs, fr, pl, names, c, end); else calls_emit_allele_lines(cs, fr, pl, c, gt_step, &err);
          std::string t((size_t)cs.n, '\0');
          ByteSink bs(&t[0]);
          if (mode == 1) calls_emit_csv(bs, fr, pl, names, c, end); else calls_emit_allele_lines(bs, fr, pl, c, gt_step, &err);
          body += t;
          continue;
        }
        CountSink cs; calls_emit_cell(cs, fr, pl, qw, names, c, end, 16);
        std::string cell((size_t)cs.n, '\0');
        ByteSink bs(&cell[0]); calls_emit_cell(bs, fr, pl, qw, names, c, end, 16);
        if (!body.empty()) body += ",\n";
        body += cell;
      }
      if (mode == 1) { o += body; continue; }
      if (mode == 2) {
        std::map<int64_t, std::map<std::pair<std::string, std::string>, uint64_t>> counts;
        for (size_t b = 0; b < body.size();) {
          const size_t e = body.find('\n', b), t1 = body.find('\t', b), t2 = body.find('\t', t1 + 1);
          ++counts[strtoll(body.c_str() + b, nullptr, 10)][std::make_pair(body.substr(t1 + 1, t2 - t1 - 1), body.substr(t2 + 1, e - t2 - 1))];
          b = e + 1;
        }
        for (const auto& col : counts)
          for (const auto& ra : col.second) o += std::to_string(col.first) + " " + ra.first.first + " " + ra.first.second + " " + std::to_string(ra.second) + "\n";
        continue;
      }
      if (body.empty()) continue;
      if (printed) o += "\n" + p1 + "]\n" + p0 + "},\n";
      o += p0 + "{\n" + p1 + "\"query_interval\": [ " + std::to_string(iv.first) + ", " + std::to_string(iv.second) + " ],\n" + p1 + "\"variant_calls\": [\n" + body;
      ++printed;
    }
    if (mode == 0) { if (printed) o += "\n" + p1 + "]\n" + p0 + "}"; o += "\n" + ip + "]\n}\n"; }
    if (err) throw std::runtime_error("device error bits " + std::to_string(err));
    *out = (char*)malloc(o.size() + 1); memcpy(*out, o.data(), o.size()); *out_len = o.size();
    return 0;
  } catch (const std::exception& e) { snprintf(errmsg, errlen, "%s", e.what()); return 1; }
}
void hostsim_free(char* p) { free(p); }

// the restated libstdc++ selection against the library itself: 1 when the whole array ends up bit-identical
// depth_limit < 0: the public entry point; >= 0: the library's internal loop with that depth budget (reaches its heap-select branch)
int hostsim_nth_element_same(const float* values, int64_t n, int64_t nth, int depth_limit) {
  std::vector<float> a(values, values + n), b(values, values + n);
  if (depth_limit < 0) {
    gdb_nth_element_libstdcxx(a.data(), n, nth);
    std::nth_element(b.begin(), b.begin() + nth, b.end());
  } else {
    gdb_introselect_libstdcxx(a.data(), n, nth, depth_limit);
    std::__introselect(b.begin(), b.begin() + nth, b.end(), (long)depth_limit, __gnu_cxx::__ops::__iter_less_iter());
  }
  return memcmp(a.data(), b.data(), (size_t)n * sizeof(float)) == 0 ? 1 : 0;
}

// genotype-order known answers (tests/golden/genotype_tables.json): the device's own enumeration (gdb_next_genotype /
// gdb_genotype_index / the closed forms of ploidy 1 and 2, as bin_remap_genotypes and emit_remap_genotypes use them) run over an
// input vector whose element g holds g: out[merged genotype] = input genotype index or -1.  m2i[k] = input allele of merged allele k
int hostsim_genotype_map(const int32_t* m2i, int num_merged, int nr_in, int ploidy, int64_t* out, int cap) {
  EntryMaps em;
  memset(&em, 0, sizeof(em));
  int8_t m2i_store[GDB_MAX_MERGED_ALLELES];
  em.m2i = m2i_store;
  for (int k = 0; k < GDB_MAX_MERGED_ALLELES; ++k) em.m2i[k] = (int8_t)(k < num_merged ? m2i[k] : -1);
  em.nr_in = nr_in; em.light = false; em.remap = true;
  std::vector<int32_t> data(100000);
  for (size_t i = 0; i < data.size(); ++i) data[i] = (int32_t)i;
  std::vector<char> buf(4 * 100000);
  ByteSink bs(buf.data());
  BinTrack tr;
  tr.reset();
  uint32_t err = 0;
  bin_remap_genotypes(bs, tr, data.data(), (int)data.size(), em, num_merged, ploidy, &err);
  if (err) return -1;
  const int n = (int)tr.n;
  for (int i = 0; i < n && i < cap; ++i) { int32_t v; memcpy(&v, buf.data() + 4 * i, 4); out[i] = v == GDB_BCF_INT32_MISSING ? -1 : v; }
  return n;
}

// gdb_core.hpp's restatement of the iteration order of libstdc++'s std::unordered_set<int> against the library itself:
// `nranges` range inserts (lens[r] ids each, flattened in ids).  out_mine / out_lib receive the two orders; returns the set's size,
// -1 when the restatement reports an overflow (more than GDB_MAX_FILTER_IDS distinct ids / a bucket count off its table)
int hostsim_uset_order(const int32_t* ids, const int32_t* lens, int nranges, int32_t* out_mine, int32_t* out_lib, int cap) {
  GdbUSetOrder us;
  gdb_uset_init(us);
  std::unordered_set<int> lib;
  const int32_t* p = ids;
  for (int r = 0; r < nranges; ++r) {
    gdb_uset_insert_range(us, p, lens[r]);
    lib.insert(p, p + lens[r]);
    p += lens[r];
  }
  int n = 0;
  for (int v : lib) { if (n < cap) out_lib[n] = v; ++n; }
  if (us.overflow) return -1;
  for (int i = 0; i < us.n && i < cap; ++i) out_mine[i] = us.key[i];
  return us.n == n ? n : -2;
}

// gdb_core.hpp's restatement of a Release build's ID union (std::hash<std::string> + the iteration order of libstdc++'s
// std::unordered_set<std::string>) against the library itself: `ntok` tokens inserted one by one (text = the tokens back to back,
// lens[i] bytes each).  The two orders come back as token numbers (first occurrence) in out_mine / out_lib; *hash_mismatch = number of
// tokens whose restated hash differs from std::hash<std::string>.  Returns the set's size, -1 beyond GDB_MAX_ID_TOKENS.
int hostsim_id_union_order(const char* text, const int32_t* lens, int ntok, int release_order, int32_t* out_mine, int32_t* out_lib, int cap, int32_t* hash_mismatch) {
  const char* tp[GDB_MAX_ID_TOKENS]; int tn[GDB_MAX_ID_TOKENS]; int nt = 0;
  GdbUSetHashOrder us;
  gdb_useth_init(us);
  std::unordered_set<std::string> lib;
  std::set<std::string> lib_sorted;
  std::vector<std::string> first_seen;
  uint32_t err = 0;
  *hash_mismatch = 0;
  const char* p = text;
  for (int i = 0; i < ntok; ++i) {
    const std::string tok(p, (size_t)lens[i]);
    if (gdb_libstdcxx_hash_bytes(p, lens[i]) != (uint64_t)std::hash<std::string>()(tok)) ++*hash_mismatch;
    gdb_id_union_add(p, lens[i], tp, tn, nt, us, release_order, &err);
    lib.insert(tok);
    lib_sorted.insert(tok);
    if (std::find(first_seen.begin(), first_seen.end(), tok) == first_seen.end()) first_seen.push_back(tok);
    p += lens[i];
  }
  if (err) return -1;
  auto number_of = [&](const std::string& t) { return (int32_t)(std::find(first_seen.begin(), first_seen.end(), t) - first_seen.begin()); };
  int n = 0;
  if (release_order) { for (auto& t : lib) { if (n < cap) out_lib[n] = number_of(t); ++n; } }
  else { for (auto& t : lib_sorted) { if (n < cap) out_lib[n] = number_of(t); ++n; } }
  for (int i = 0; i < nt && i < cap; ++i) { const int w = gdb_id_union_at(us, release_order, i); out_mine[i] = number_of(std::string(tp[w], (size_t)tn[w])); }
  return nt == n ? n : -2;
}

// gdb_asa.hpp's "%.3f" of a float (exact integer arithmetic on the mantissa) for the formatting test
int hostsim_fixed3(float v, char* buf, uint64_t cap) {
  struct S { char* p; uint64_t cap, n; void put(char c) { if (n + 1 < cap) p[n] = c; ++n; } void write(const char* q, int k) { for (int i = 0; i < k; ++i) put(q[i]); } } s{buf, cap, 0};
  uint32_t err = 0;
  put_fixed3(s, v, &err);
  if (s.n < cap) buf[s.n] = 0;
  return err ? -1 : (int)s.n;
}

// gdb_core.hpp's put_float (kputd's rule inside [0.0001, 999999], an exact "%g" outside) on a whole array of bit patterns: the
// texts come back NUL-separated; *mismatch = index of the first one that differs from what the C library prints through kputd's
// own logic (sign, then "%g" of the magnitude; -1: none).  One call checks millions of patterns.
int64_t hostsim_put_float_check(const uint32_t* bits, int64_t n, char* first_bad_mine, char* first_bad_libc, uint64_t cap) {
  struct S { char* p; uint64_t cap, n; void put(char c) { if (n + 1 < cap) p[n] = c; ++n; } void write(const char* q, int k) { for (int i = 0; i < k; ++i) put(q[i]); } };
  for (int64_t i = 0; i < n; ++i) {
    float f; memcpy(&f, &bits[i], 4);
    char mine[64], libc[64];
    S s{mine, sizeof(mine), 0};
    put_float(s, f);
    mine[s.n < sizeof(mine) ? s.n : sizeof(mine) - 1] = 0;
    // kputd (htslib kstring.c), restated with the C library doing "%g": what the reference prints
    double d = f;
    int at = 0;
    if (d == 0) snprintf(libc, sizeof(libc), "%s", std::signbit(d) ? "-0" : "0");
    else {
      if (d < 0) { libc[at++] = '-'; d = -d; }
      if (!(d >= 0.0001 && d <= 999999)) snprintf(libc + at, sizeof(libc) - at, "%g", d);
      else { libc[at] = 0; strcpy(libc, mine); }   // (the six-digit rule is pinned by the goldens and the oracle tests)
    }
    if (strcmp(mine, libc) != 0) {
      snprintf(first_bad_mine, cap, "%s", mine); snprintf(first_bad_libc, cap, "%s", libc);
      return i;
    }
  }
  return -1;
}
int hostsim_put_float(float v, char* buf, uint64_t cap) {
  struct S { char* p; uint64_t cap, n; void put(char c) { if (n + 1 < cap) p[n] = c; ++n; } void write(const char* q, int k) { for (int i = 0; i < k; ++i) put(q[i]); } } s{buf, cap, 0};
  put_float(s, v);
  if (s.n < cap) buf[s.n] = 0;
  return (int)s.n;
}

// gdb_nth_element_by_lists (the data-parallel statement of libstdc++'s introselect that k_site_huge runs) against std::nth_element itself:
// 1 = same selected value (bit for bit) AND same permutation left behind
int hostsim_nth_by_lists_same(const float* values, int64_t n, int64_t nth) {
  std::vector<float> a(values, values + n), b(values, values + n);
  std::vector<uint32_t> pl((size_t)n + 1), pr((size_t)n + 1);
  const float mine = gdb_nth_element_by_lists(a.data(), n, nth, pl.data(), pr.data());
  std::nth_element(b.begin(), b.begin() + nth, b.end());
  if (memcmp(&mine, &b[(size_t)nth], 4) != 0) return 0;
  return memcmp(a.data(), b.data(), (size_t)n * sizeof(float)) == 0 ? 1 : 0;
}

// the typed-value encoder of the BCF2 path (gdb_core.hpp: bcf_enc_vint / bcf_enc_size / bcf_enc_int1) for the byte-level known answers of
// tests/test_bcf_typed_values.py (derived from the BCFv2.2 specification, not from the tests' own decoder)
int hostsim_bcf_enc_vint(const int32_t* a, int n, uint8_t* out, int cap) {
  struct S { uint8_t* p; int cap, n; void put(char c) { if (n < cap) p[n] = (uint8_t)c; ++n; } } s{out, cap, 0};
  bcf_enc_vint(s, a, n);
  return s.n;
}
int hostsim_bcf_enc_size(int size, int type, uint8_t* out, int cap) {
  struct S { uint8_t* p; int cap, n; void put(char c) { if (n < cap) p[n] = (uint8_t)c; ++n; } } s{out, cap, 0};
  bcf_enc_size(s, size, type);
  return s.n == bcf_enc_size_bytes(size) ? s.n : -1;
}

// the PRODUCT's JSON / gzip readers (csrc/common/mini_json.hpp, gz_text.hpp) laid open to the tests the same way the oracle lays open its own
// (oracle/oracle_json.hpp): both are compared with Python's json / gzip modules (tests/test_common_utils.py)
static void hostsim_json_dump_value(const mini_json::Value& v, std::string& o) {
  using V = mini_json::Value;
  char buf[64];
  switch (v.type) {
    case V::Null: o += "n"; break;
    case V::Bool: o += v.b ? "t" : "f"; break;
    case V::Int: snprintf(buf, sizeof buf, "i%lld", (long long)v.i); o += buf; break;
    case V::Double: snprintf(buf, sizeof buf, "d%.17g", v.d); o += buf; break;
    case V::String: o += "s"; for (unsigned char c : v.s) { snprintf(buf, sizeof buf, "%02x", c); o += buf; } break;
    case V::Array: o += "["; for (auto& e : v.arr) { hostsim_json_dump_value(e, o); o += ","; } o += "]"; break;
    case V::Object: o += "{"; for (auto& kv : v.obj) { for (unsigned char c : kv.first) { snprintf(buf, sizeof buf, "%02x", c); o += buf; } o += ":"; hostsim_json_dump_value(kv.second, o); o += ","; } o += "}"; break;
  }
}
int hostsim_json_dump(const char* text, char** out, uint64_t* out_len, char* err, uint64_t errlen) {
  try {
    std::string o;
    hostsim_json_dump_value(mini_json::parse(text), o);
    *out = (char*)malloc(o.size() + 1); memcpy(*out, o.data(), o.size()); (*out)[o.size()] = 0; *out_len = o.size();
    return 0;
  } catch (const std::exception& e) { if (err && errlen) snprintf(err, errlen, "%s", e.what()); return 1; }
}
int hostsim_gz_read_all(const char* path, char** out, uint64_t* out_len, char* err, uint64_t errlen) {
  try {
    std::string o = gz_text::read_all(path);
    *out = (char*)malloc(o.size() + 1); memcpy(*out, o.data(), o.size()); (*out)[o.size()] = 0; *out_len = o.size();
    return 0;
  } catch (const std::exception& e) { if (err && errlen) snprintf(err, errlen, "%s", e.what()); return 1; }
}

}  // extern "C"
