// gdb_stages.hpp - per-element bodies of the sweep stages (GDB_HD; see gdb_core.hpp for the rules).
//
// The reference sweeps one cell at a time through a priority queue ordered by END
// (VariantQueryProcessor::scan_and_operate / handle_gvcf_ranges, reference
// src/main/cpp/src/genomicsdb/query_variants.cc:296-476).  Here the same intervals come out of sorts and
// scans over all cells of the staged column interval:
//   S2  overlap override + END-copy lookup  -> eff_end per cell        (query_variants.cc:512-543, 845-941)
//   S3  begin / END+1 events -> boundaries -> output records           (handle_gvcf_ranges :305-331, deletions :310)
//   S4  record range of every cell, presence / DP difference arrays    (what each operate() call would see)
//   S6  (record, heavy cell) incidences in (record,row) order          (row-ascending allele merge order)
//   S8  per-(record,row) live-cell walk, sizing and writing the sample columns
#pragma once
#include "gdb_core.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
#define GDB_ATOMIC_ADD_I32(p, v) atomicAdd((int*)(p), (int)(v))
#define GDB_ATOMIC_OR_U32(p, v) atomicOr((unsigned*)(p), (unsigned)(v))
#define GDB_ATOMIC_ADD_U64(p, v) atomicAdd((unsigned long long*)(p), (unsigned long long)(v))
#else
#define GDB_ATOMIC_ADD_I32(p, v) (*(p) += (v))
#define GDB_ATOMIC_OR_U32(p, v) (*(p) |= (v))
#define GDB_ATOMIC_ADD_U64(p, v) (*(p) += (uint64_t)(v))
#endif

#define GDB_EVENT_SENTINEL 0xFFFFFFFFFFFFFFFFull

// ---- S2: effective END (row-major walk).  perm = cells stably sorted by row ---------------------------------
GDB_HD void stage_eff_end(const FragmentView& fr, const CellMeta& cm, const int64_t* perm, int64_t j, int64_t* rm_begin, int64_t* span, uint32_t* err) {
  const int64_t C = fr.ncells;
  const int64_t c = perm[j];
  int64_t eff = fr.end[c];
  if (j + 1 < C) {
    const int64_t nx = perm[j + 1];
    if (fr.row[nx] == fr.row[c] && fr.begin[nx] <= eff) {  // the next cell of this sample overrides the rest of this interval
      eff = fr.begin[nx] - 1;
      if (!(cm.cflags[c] & (GDB_CF_DELETION | GDB_CF_REFBLOCK))) *err |= GDB_ERR_OVERLAP_NOT_REFBLOCK_OR_DELETION;
    }
  }
  cm.eff_end[c] = eff;
  span[c] = eff - fr.begin[c];  // its maximum bounds how far back a query window has to look for live cells
  rm_begin[j] = fr.begin[c];
}
GDB_HD bool cell_in_window(const FragmentView& fr, const CellMeta& cm, int64_t c, int64_t qb, int64_t qe) {
  return fr.begin[c] <= qe && cm.eff_end[c] >= qb && cm.eff_end[c] >= fr.begin[c];
}

// ---- S3a: two event keys per cell: (position << 2) | (is_end << 1) | is_deletion ---------------------------
// (keys are indexed relative to c_base: only the cells that can reach the window take part in an interval)
GDB_HD void stage_event_keys(const FragmentView& fr, const CellMeta& cm, int64_t c, int64_t c_base, int64_t qb, int64_t qe, uint64_t* keys) {
  const uint32_t f = cm.cflags[c];
  keys += 2 * (c - c_base) - 2 * c;
  if (!cell_in_window(fr, cm, c, qb, qe)) { keys[2 * c] = GDB_EVENT_SENTINEL; keys[2 * c + 1] = GDB_EVENT_SENTINEL; return; }
  const int64_t b = fr.begin[c] > qb ? fr.begin[c] : qb;
  const int64_t e = cm.eff_end[c] < qe ? cm.eff_end[c] : qe;
  const uint64_t del = (f & GDB_CF_DELETION) ? 1u : 0u;
  keys[2 * c] = ((uint64_t)(b - qb) << 2) | del;
  keys[2 * c + 1] = ((uint64_t)(e + 1 - qb) << 2) | 2u | del;
}
// a boundary marker = a begin and an end event at the same position: a boundary with no change of coverage
GDB_HD void stage_marker_keys(const FragmentView& fr, int64_t m, int64_t qb, int64_t qe, uint64_t* two_keys) {
  const int64_t p = fr.marker_begin[m];
  if (p < qb || p > qe) { two_keys[0] = GDB_EVENT_SENTINEL; two_keys[1] = GDB_EVENT_SENTINEL; return; }
  two_keys[0] = (uint64_t)(p - qb) << 2;
  two_keys[1] = ((uint64_t)(p - qb) << 2) | 2u;
}
// S3b: deltas of a sorted key: packed (coverage delta << 32) + deletion-coverage delta, both as wrapped int32 lanes
GDB_HD int64_t stage_event_delta(uint64_t key) {
  if (key == GDB_EVENT_SENTINEL) return 0;
  const int32_t dc = (key & 2u) ? -1 : 1;
  const int32_t dd = (key & 1u) ? dc : 0;
  return (int64_t)(((uint64_t)(uint32_t)dc << 32) | (uint64_t)(uint32_t)dd);
}
GDB_HD int64_t packed_add(int64_t a, int64_t b) {  // lane-wise int32 add
  const uint32_t hi = (uint32_t)((uint64_t)a >> 32) + (uint32_t)((uint64_t)b >> 32);
  const uint32_t lo = (uint32_t)a + (uint32_t)b;
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
// S3c: element i closes a boundary if the next key has another position
GDB_HD int32_t stage_is_run_end(const uint64_t* keys, int64_t n_events, int64_t i) {
  if (keys[i] == GDB_EVENT_SENTINEL) return 0;
  if (i + 1 >= n_events || keys[i + 1] == GDB_EVENT_SENTINEL) return 1;
  return (keys[i] >> 2) != (keys[i + 1] >> 2);
}
struct Boundaries {  // U boundaries: position, live-call coverage and deletion coverage right after it
  int64_t* pos; int32_t* cov; int32_t* del; int64_t* nrec;
};
GDB_HD void stage_boundary_write(const uint64_t* keys, const int64_t* incl_scan, const int32_t* run_end_excl, int64_t i, int32_t is_end, const Boundaries& b, int64_t qb) {
  if (!is_end) return;
  const int64_t u = run_end_excl[i];
  b.pos[u] = (int64_t)(keys[i] >> 2) + qb;
  b.cov[u] = (int32_t)(uint32_t)((uint64_t)incl_scan[i] >> 32);
  b.del[u] = (int32_t)(uint32_t)incl_scan[i];
}
// S3d: #records that start in [pos_u, pos_{u+1}): none when nothing is live, one per bp while a deletion is live
GDB_HD int64_t stage_boundary_nrec(const Boundaries& b, int64_t U, int64_t u) {
  if (u + 1 >= U || b.cov[u] <= 0) return 0;
  return b.del[u] > 0 ? (b.pos[u + 1] - b.pos[u]) : 1;
}
GDB_HD void stage_record_expand(const Boundaries& b, const int64_t* rbase, int64_t U, int64_t k, int64_t* rstart, int64_t* rend) {
  int64_t lo = 0, hi = U;  // last u with rbase[u] <= k
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (rbase[mid] <= k) lo = mid + 1; else hi = mid; }
  const int64_t u = lo - 1;
  const int64_t t = k - rbase[u];
  if (b.del[u] > 0) { rstart[k] = b.pos[u] + t; rend[k] = b.pos[u] + t; }
  else { rstart[k] = b.pos[u]; rend[k] = b.pos[u + 1] - 1; }
}

// ---- S4: record range of a cell + difference arrays ------------------------------------------------------
GDB_HD int presence_field(const CombinePlan& pl, int i) {  // field whose validity decides presence of format entry i
  const int f = pl.format_field[i];
  if (f == pl.f_DP && pl.f_DP >= 0) return pl.f_DP_FORMAT;  // FORMAT DP exists iff some DP_FORMAT is valid (may be -1)
  return f;
}
struct DiffArrays { int32_t* fmt; int32_t* dp; int32_t* nr; int64_t stride; };  // stride = P + 1
// The same difference arrays with several counters per 64-bit word (the +1/-1 pairs dominate the stage: one atomic per word
// instead of one per counter).  Field j = FORMAT presence j (j < n_format) or the <NON_REF> count (j = n_format) sits in word
// j / per_word at bit (j % per_word) * width; width bits hold any count up to the number of samples, and since every prefix
// sum of a field is >= 0 the modulo-2^64 sums never borrow across fields.  The DP sum has a word of its own (last).
struct DiffPacked { uint64_t* w; int64_t stride; int32_t width, per_word, nwords; };

// record range [klo, khi] a cell is live in (false: none); also k_lo / k_hi / heavy_count of the cell.
// first_record_at (optional): [qe - qb + 2] table, entry p = first record whose start is >= qb + p (P if none); it replaces
// the two binary searches over the record starts by two loads
GDB_HD bool stage_cell_range(const FragmentView& fr, const CellMeta& cm, const RecordTable& rec, int64_t c, int64_t c_base, int64_t qb, int64_t qe,
                             int64_t* heavy_count, const int32_t* first_record_at, int64_t& klo, int64_t& khi) {
  cm.k_lo[c] = -1; cm.k_hi[c] = -1; heavy_count[c - c_base] = 0;
  if (!cell_in_window(fr, cm, c, qb, qe) || rec.npos == 0) return false;
  const int64_t b = fr.begin[c] > qb ? fr.begin[c] : qb;
  const int64_t e = cm.eff_end[c] < qe ? cm.eff_end[c] : qe;
  if (first_record_at) {
    klo = first_record_at[b - qb];
    khi = (int64_t)first_record_at[e + 1 - qb] - 1;
  } else {
    int64_t lo = 0, hi = rec.npos;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (rec.start[mid] < b) lo = mid + 1; else hi = mid; }
    klo = lo;
    hi = rec.npos;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (rec.start[mid] <= e) lo = mid + 1; else hi = mid; }
    khi = lo - 1;
  }
  if (khi < klo) return false;
  cm.k_lo[c] = (int32_t)klo; cm.k_hi[c] = (int32_t)khi;
  if (cm.cflags[c] & GDB_CF_HEAVY) heavy_count[c - c_base] = khi - klo + 1;
  return true;
}
// contribution of a cell to packed word `word` (presence of FORMAT fields / <NON_REF>)
GDB_HD uint64_t stage_packed_word(const CombinePlan& pl, const DiffPacked& pk, uint64_t vmask, uint32_t cflags, int word) {
  uint64_t acc = 0;
  const int j0 = word * pk.per_word;
  for (int j = j0; j < j0 + pk.per_word && j <= pl.n_format; ++j) {
    bool on;
    if (j < pl.n_format) { const int pf = presence_field(pl, j); on = pf >= 0 && ((vmask >> pf) & 1); }
    else on = (cflags & GDB_CF_HAS_NR) != 0;
    if (on) acc |= 1ull << ((j - j0) * pk.width);
  }
  return acc;
}
// serial flavour (CPU harness): +v at klo, -v at khi + 1 in the int32 difference arrays
GDB_HD void stage_cell_ranges(const FragmentView& fr, const CombinePlan& pl, const CellMeta& cm, const RecordTable& rec, int64_t c, int64_t c_base,
                              int64_t qb, int64_t qe, const DiffArrays& d, int64_t* heavy_count) {
  int64_t klo, khi;
  if (!stage_cell_range(fr, cm, rec, c, c_base, qb, qe, heavy_count, nullptr, klo, khi)) return;
  const uint32_t f = cm.cflags[c];
  const uint64_t vm = cm.vmask[c];
  for (int i = 0; i < pl.n_format; ++i) {
    const int pf = presence_field(pl, i);
    if (pf < 0 || !((vm >> pf) & 1)) continue;
    GDB_ATOMIC_ADD_I32(d.fmt + (int64_t)i * d.stride + klo, 1);
    GDB_ATOMIC_ADD_I32(d.fmt + (int64_t)i * d.stride + khi + 1, -1);
  }
  const int32_t dp = cm.dpval[c];
  if (dp) { GDB_ATOMIC_ADD_I32(d.dp + klo, dp); GDB_ATOMIC_ADD_I32(d.dp + khi + 1, -dp); }
  if (f & GDB_CF_HAS_NR) { GDB_ATOMIC_ADD_I32(d.nr + klo, 1); GDB_ATOMIC_ADD_I32(d.nr + khi + 1, -1); }
}

// ---- S6: incidence keys: record * N + row, value = cell ----------------------------------------------------
GDB_HD void stage_incidence_fill(const FragmentView& fr, const CellMeta& cm, const int64_t* hoff, int64_t c, int64_t c_base, int64_t nrows,
                                 uint64_t* keys, int64_t* vals, uint32_t* lut_len) {
  if (!(cm.cflags[c] & GDB_CF_HEAVY) || cm.k_lo[c] < 0) return;
  const int64_t base = hoff[c - c_base];
  const int64_t klo = cm.k_lo[c], khi = cm.k_hi[c];
  for (int64_t k = klo; k <= khi; ++k) {
    keys[base + (k - klo)] = (uint64_t)k * (uint64_t)nrows + (uint64_t)fr.row[c];
    vals[base + (k - klo)] = c;
  }
  (void)lut_len;
}
GDB_HD int64_t stage_heavy_base(const uint64_t* sorted_keys, int64_t T, int64_t nrows, int64_t k) {  // first incidence of record k
  const uint64_t want = (uint64_t)k * (uint64_t)nrows;
  int64_t lo = 0, hi = T;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (sorted_keys[mid] < want) lo = mid + 1; else hi = mid; }
  return lo;
}

// ---- S8: the per-row walker ----------------------------------------------------------------------------------
struct RowIndex {           // cells grouped by row, column order inside a row
  const int64_t* row_ptr;   // [N+1]
  const int64_t* rm_cell;   // [C]
  const int64_t* rm_begin;  // [C]
};
struct RowWalker {
  int64_t j, j_begin, j_end;
  GDB_HD void init(const RowIndex& ri, int32_t row, int64_t s0) {
    j_begin = ri.row_ptr[row]; j_end = ri.row_ptr[row + 1];
    int64_t lo = j_begin, hi = j_end;  // last j with rm_begin[j] <= s0
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (ri.rm_begin[mid] <= s0) lo = mid + 1; else hi = mid; }
    j = lo - 1;
  }
  // live cell of this row at position s (s must not decrease between calls), or -1
  GDB_HD int64_t live(const RowIndex& ri, const CellMeta& cm, int64_t s) {
    while (j + 1 < j_end && ri.rm_begin[j + 1] <= s) ++j;
    if (j < j_begin) return -1;
    const int64_t c = ri.rm_cell[j];
    return (s <= cm.eff_end[c]) ? c : -1;
  }
};

GDB_HD RecordInfo load_record_info(const SiteOut& so, const HeavyLists& hl, int64_t k) {
  RecordInfo ri;
  ri.num_merged = so.num_alleles[k];
  ri.rflags = so.rflags[k];
  ri.fmt_mask = so.fmt_mask[k];
  ri.hbase = hl.base[k];
  ri.hend = hl.base[k + 1];
  return ri;
}
