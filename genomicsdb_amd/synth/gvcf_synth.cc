// gvcf_synth.cc - deterministic synthetic gVCF generator: N samples x L bp, one contig, schema of the
// reference's tests/inputs/vid.json.  BENCH / TEST INPUT TOOLING: produces the begin-cells (reference binary-cell layout,
// src/main/cpp/src/vcf/vcf2binary.cc:991-1196) in TileDB column-major order, chunk by chunk, plus the synthetic reference.
// Both the GPU path and the CPU oracle read exactly these bytes.
//
//   per-sample stream seed = splitmix64(seed ^ (row+1) * 0x9E3779B97F4A7C15)
//   records tile [B, B+L) left to right, never overlapping:
//     7/8 reference block: length 1+Geometric(mean 120) capped at 2000, ALT=<NON_REF>, GT 0/0, GQ in {0,20,50,99}
//                          (p = .05,.15,.3,.5), DP in U[10,60], MIN_DP in U[10,DP], PL = [0, 3GQ, 45GQ]
//     1/8 variant: 85% SNV / 8% deletion (REF len 2-10) / 7% insertion (ALT len 2-6); allele from the site pool
//                  pool(pos) (K = 1 for 90% of sites, else 2-3) so samples share alleles; followed by <NON_REF>;
//                  GT 0/1 (2/3) or 1/1 (1/3); PL 0 at the called genotype, others U[10,10000]; AD consistent with DP;
//                  SB 4 x U[0,40]; rank-sum INFOs round(N(0,1),3); MQ round(U[40,60],2); RAW_MQ = MQ^2*DP; MQ0 = 0;
//                  INFO DP = DP; QUAL round(U[30,3000],2)
//   reference base(pos) = "ACGT"[hash(seed, pos) & 3]
//   genome mode (gdbsynth_set_contigs, BASELINE configs[3]): columns are the flattened genome of a contig table (tiledb_column_offset,
//   length); no record crosses the end of its contig (reference blocks are cut there, a deletion that would cross becomes an SNV)
//   and every sample begins a new record at the first column of the next contig - as separate per-contig gVCF records would
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline uint64_t hash2(uint64_t a, uint64_t b) { uint64_t s = a ^ (b * 0xD6E8FEB86659FD93ull); return splitmix64(s); }

struct Rng {
  uint64_t s;
  uint64_t next() { return splitmix64(s); }
  uint32_t below(uint32_t n) { return (uint32_t)(next() % n); }
  int range(int a, int b) { return a + (int)below((uint32_t)(b - a + 1)); }
  double unit() { return ((double)(next() >> 11) + 1.0) / 9007199254740993.0; }  // (0,1]
  double normal() { double u1 = unit(), u2 = unit(); return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};

struct Rec {
  int64_t begin, end;
  int32_t row;
  uint8_t kind, reflen, altlen, hom;
  char ref[12];
  char alt[8];
  float qual, rs[4], mq, raw_mq;
  int32_t dp, gq, min_dp, sb[4], ad[3], pl[6];
  uint8_t nfilter, idlen;       // optional modes (gdbsynth_set_modes): FILTER ids, ID text
  int32_t filter[2];
  char id[24];
};

const int32_t NULL_I32 = 0x7FFFFFFF;
const uint32_t NULL_F32 = 0x7F7FFFFFu;

struct Synth {
  uint64_t seed;
  int32_t n_samples;
  int64_t B, L;
  std::vector<Rng> rng;
  std::vector<int64_t> pos;  // next begin per sample
  int64_t chunk_begin;
  std::vector<uint8_t> cells;  // last chunk
  int64_t last_ncells = 0;

  // c5 of BASELINE.json (high-ALT stress): inside [dense_begin, dense_begin + dense_len) every sample starts a variant at
  // every position that is a multiple of hot_stride, with its ALT drawn from a site pool of dense_K insertion alleles
  int64_t dense_begin = 0, dense_len = 0, hot_stride = 50;
  int dense_K = 0;
  int float_stress_permille = 0;   // share of the variant calls whose floats leave the everyday range: below 1e-4, above 1e6 / 2^63, subnormal, +inf, NaN
  double rs_scale = 1000.0;   // rank sums are rounded to 1 / rs_scale (coarse scales make tied medians, zeros of both signs included)
  // optional modes, all off by default (the plain stream stays byte-identical): decisions come from a hash of (seed, row, begin),
  // not from the per-sample stream.  overlap: the next record of a sample begins INSIDE a reference block / deletion (the scan's
  // overlap override, query_variants.cc:512-543); filter: variant cells carry FILTER = [filter_id]; filter2: some carry
  // [filter_id2] instead (records that unite two different ids); id: variant cells carry one or two ';'-separated ID tokens
  // (the array schema then has an ID attribute between ALT and QUAL)
  int overlap_permille = 0, filter_permille = 0, filter2_permille = 0, id_permille = 0;
  int32_t filter_id = 1, filter_id2 = 0;
  bool with_id = false;
  uint64_t mode_hash(int32_t row, int64_t begin, uint64_t salt) const { return hash2(seed ^ salt, (uint64_t)begin * 0x9E3779B97F4A7C15ull + (uint64_t)row); }
  // genome mode: contig table in column space, sorted by offset; empty = one unbounded contig
  std::vector<std::pair<int64_t, int64_t>> contigs;   // (offset, length)
  // end (exclusive) of the contig holding column p; a column in a gap between contigs is moved to the next contig's offset
  int64_t contig_end(int64_t& p) const {
    if (contigs.empty()) return INT64_MAX;
    size_t lo = 0, hi = contigs.size();
    while (lo < hi) { size_t mid = (lo + hi) >> 1; if (contigs[mid].first <= p) lo = mid + 1; else hi = mid; }
    if (lo > 0 && p < contigs[lo - 1].first + contigs[lo - 1].second) return contigs[lo - 1].first + contigs[lo - 1].second;
    if (lo < contigs.size()) { p = contigs[lo].first; return contigs[lo].first + contigs[lo].second; }
    return INT64_MAX;   // behind the last contig: unbounded (the caller's [B, B+L) ends the stream)
  }
  bool in_dense(int64_t p) const { return dense_len > 0 && p >= dense_begin && p < dense_begin + dense_len; }
  bool is_hot(int64_t p) const { return in_dense(p) && (p % hot_stride) == 0; }

  char base(int64_t p) const { return "ACGT"[hash2(seed ^ 0x5bd1e9955bd1e995ull, (uint64_t)p) & 3]; }

  void next_record(int32_t row, Rec& r) {
    Rng& g = rng[row];
    memset(&r, 0, sizeof(r));
    r.row = row;
    const int64_t cend = contig_end(pos[row]);   // (may move pos[row] out of a gap)
    r.begin = pos[row];
    const bool hot = is_hot(r.begin);
    if (!hot && g.below(8) != 0) {  // reference block
      double u = g.unit();
      int64_t len = 1 + (int64_t)std::floor(std::log(u) / std::log(1.0 - 1.0 / 120.0));
      if (len > 2000) len = 2000;
      if (len < 1) len = 1;
      if (dense_len > 0) {  // a block must not run over the next hot position
        int64_t nh = ((r.begin / hot_stride) + 1) * hot_stride;
        if (in_dense(nh) && r.begin + len > nh) len = nh - r.begin;
      }
      if (r.begin + len > cend) len = cend - r.begin;
      r.kind = 0;
      r.end = r.begin + len - 1;
      r.reflen = 1; r.ref[0] = base(r.begin);
      r.altlen = 1; r.alt[0] = '&';
      uint32_t q = g.below(100);
      r.gq = q < 5 ? 0 : q < 20 ? 20 : q < 50 ? 50 : 99;
      r.dp = g.range(10, 60);
      r.min_dp = g.range(10, r.dp);
      r.pl[0] = 0; r.pl[1] = 3 * r.gq; r.pl[2] = 45 * r.gq;
    } else {
      uint32_t t = g.below(100);
      uint64_t site = hash2(seed, (uint64_t)r.begin);
      int K = (site % 100) < 90 ? 1 : 2 + (int)((site >> 8) & 1);
      if (hot) { K = dense_K; t = 99; }   // insertion from the dense pool
      int pick = (int)g.below((uint32_t)K);
      uint64_t ah = hash2(site, (uint64_t)pick + 17);
      if (t >= 85 && t < 93 && r.begin + 2 + (int64_t)(ah % 9) > cend) t = 0;   // a deletion would run over the contig's end
      char rb = base(r.begin);
      if (t < 85) {  // SNV
        r.kind = 1; r.end = r.begin; r.reflen = 1; r.ref[0] = rb;
        const char* bases = "ACGT";
        int bi = (int)(strchr(bases, rb) - bases);
        r.altlen = 1; r.alt[0] = bases[(bi + 1 + (int)((ah % 3 + pick) % 3)) & 3];
      } else if (t < 93) {  // deletion
        int len = 2 + (int)(ah % 9);
        r.kind = 2; r.end = r.begin + len - 1; r.reflen = (uint8_t)len;
        for (int i = 0; i < len; ++i) r.ref[i] = base(r.begin + i);
        r.altlen = 1; r.alt[0] = rb;
      } else {  // insertion
        int ins = 1 + (int)(ah % 5);
        if (hot) { ins = 5; ah = (uint64_t)pick * 0x9E3779B97F4A7C15ull + 12345; }   // K distinct 5-mers
        r.kind = 3; r.end = r.begin; r.reflen = 1; r.ref[0] = rb;
        r.altlen = (uint8_t)(1 + ins); r.alt[0] = rb;
        for (int i = 0; i < ins; ++i) r.alt[1 + i] = "ACGT"[hot ? ((pick >> (2 * i)) & 3) : ((ah >> (8 + 2 * i)) & 3)];
      }
      r.hom = g.below(3) == 0;
      r.dp = g.range(10, 60);
      r.min_dp = NULL_I32;
      for (int i = 0; i < 4; ++i) r.rs[i] = (float)(std::round(g.normal() * rs_scale) / rs_scale);
      r.mq = (float)(std::round((40.0 + 20.0 * g.unit()) * 100.0) / 100.0);
      r.raw_mq = r.mq * r.mq * (float)r.dp;
      r.qual = (float)(std::round((30.0 + 2970.0 * g.unit()) * 100.0) / 100.0);
      if (float_stress_permille > 0) {
        const uint64_t h = mode_hash(row, r.begin, 0xf10a7f10a7ull);
        if ((int)(h % 1000) < float_stress_permille) {
          switch ((h >> 12) % 6) {
            case 0: for (int i = 0; i < 4; ++i) r.rs[i] *= 1e-6f; r.qual = 5e-5f; break;
            case 1: for (int i = 0; i < 4; ++i) r.rs[i] *= 1e24f; r.qual = 1e25f; r.raw_mq = 3e38f; break;
            case 2: r.qual = std::numeric_limits<float>::infinity(); r.mq = 1.5e-7f; break;
            case 3: { uint32_t nanbits = 0x7FC00000u; memcpy(&r.qual, &nanbits, 4); r.mq = 1e-10f; break; }
            case 4: for (int i = 0; i < 4; ++i) r.rs[i] = (i & 1) ? 1e-41f : -3e-39f; r.qual = 9.9999994e-5f; break;
            default: r.qual = 999999.06f; r.mq = 1234567.0f; r.raw_mq = 1.8446744e19f; break;
          }
        }
      }
      for (int i = 0; i < 4; ++i) r.sb[i] = g.range(0, 40);
      int called = r.hom ? 2 : 1;  // genotype index of 1/1 = 2, 0/1 = 1
      for (int i = 0; i < 6; ++i) r.pl[i] = (i == called) ? 0 : g.range(10, 10000);
      int second = 0x7FFFFFFF;
      for (int i = 0; i < 6; ++i) if (i != called) second = std::min(second, r.pl[i]);
      r.gq = std::min(99, second);
      if (r.hom) { r.ad[0] = g.range(0, 2); r.ad[1] = r.dp - r.ad[0]; }
      else { r.ad[1] = r.dp / 2 + g.range(-3, 3); if (r.ad[1] < 1) r.ad[1] = 1; if (r.ad[1] > r.dp) r.ad[1] = r.dp; r.ad[0] = r.dp - r.ad[1]; }
      r.ad[2] = 0;
    }
    pos[row] = r.end + 1;
    if (overlap_permille > 0 && (r.kind == 0 || r.kind == 2) && r.end - r.begin >= 2 && !in_dense(r.begin)) {
      const uint64_t h = mode_hash(row, r.begin, 0x0f0f1234abcdull);
      if ((int)(h % 1000) < overlap_permille) pos[row] = r.begin + 1 + (int64_t)((h >> 20) % (uint64_t)(r.end - r.begin));   // in (begin, end]
    }
    if (r.kind != 0) {
      if (filter_permille > 0 || filter2_permille > 0) {
        const int v = (int)(mode_hash(row, r.begin, 0x77aa55ull) % 1000);
        if (v < filter2_permille) { r.nfilter = 1; r.filter[0] = filter_id2; }
        else if (v < filter2_permille + filter_permille) { r.nfilter = 1; r.filter[0] = filter_id; }
      }
      if (with_id && id_permille > 0) {
        const uint64_t h = mode_hash(row, r.begin, 0x1d1d1dull);
        if ((int)(h % 1000) < id_permille) {
          // tokens shared per site (so that unions meet equal and different tokens): rs<site hash % 4>, sometimes a second one
          const uint64_t sh = hash2(seed ^ 0x1d5eedull, (uint64_t)r.begin);
          int n = snprintf(r.id, sizeof(r.id), "rs%u", (unsigned)(100 + (sh + (h >> 12)) % 4));
          if ((h >> 30) & 1) n += snprintf(r.id + n, sizeof(r.id) - (size_t)n, ";x%u", (unsigned)((h >> 33) % 3));
          r.idlen = (uint8_t)n;
        }
      }
    }
  }

  // one cell in the reference binary layout, attribute order of tests/inputs/vid.json:
  // END REF ALT QUAL FILTER | BaseQRankSum ClippingRankSum MQRankSum ReadPosRankSum MQ RAW_MQ MQ0 DP | DP_FORMAT GQ SB AD PL PGT PID MIN_DP GT
  struct Out {
    uint8_t* p;
    void put(const void* src, size_t n) { memcpy(p, src, n); p += n; }
    template <class T> void v(T x) { memcpy(p, &x, sizeof(T)); p += sizeof(T); }
    void c(char ch) { *p++ = (uint8_t)ch; }
  };
  static uint32_t cell_size(const Rec& r) {
    // coords 24 + END 8 + REF 4+len + ALT 4+len + QUAL 4 + FILTER 4 + 8 INFO words 32 + DP_FORMAT 4 + GQ 4 + SB 16 + AD + PL + PGT 4 + PID 4 + MIN_DP 4 + GT 12
    const uint32_t alt = r.kind == 0 ? 1u : (uint32_t)r.altlen + 2u;
    const uint32_t ad_pl = r.kind == 0 ? (4u + 4u + 12u) : (4u + 12u + 4u + 24u);
    return 24u + 8u + 4u + r.reflen + 4u + alt + 4u + 4u + 32u + 4u + 4u + 16u + ad_pl + 4u + 4u + 4u + 12u + 4u * r.nfilter + (r.idlen & 0x80 ? 4u + (r.idlen & 0x7Fu) : 0u);
  }
  // (idlen bit 7 = the schema has an ID attribute; set for every record of such a stream by next_record's caller)
  static void write_cell(uint8_t* dst, const Rec& r) {
    Out o{dst};
    o.v<int64_t>(r.row); o.v<int64_t>(r.begin); o.v<uint64_t>(0);
    o.v<int64_t>(r.end);
    o.v<int32_t>(r.reflen); o.put(r.ref, r.reflen);
    if (r.kind == 0) { o.v<int32_t>(1); o.c('&'); }
    else { o.v<int32_t>(r.altlen + 2); o.put(r.alt, r.altlen); o.c('|'); o.c('&'); }
    if (r.idlen & 0x80) { o.v<int32_t>(r.idlen & 0x7F); o.put(r.id, r.idlen & 0x7Fu); }   // ID (only when the schema has it)
    if (r.kind == 0) o.v<uint32_t>(NULL_F32); else o.v<float>(r.qual);
    o.v<int32_t>(r.nfilter); for (int i = 0; i < r.nfilter; ++i) o.v<int32_t>(r.filter[i]);  // FILTER
    if (r.kind == 0) { for (int i = 0; i < 6; ++i) o.v<uint32_t>(NULL_F32); o.v<int32_t>(NULL_I32); o.v<int32_t>(NULL_I32); }
    else {
      for (int i = 0; i < 4; ++i) o.v<float>(r.rs[i]);
      o.v<float>(r.mq); o.v<float>(r.raw_mq); o.v<int32_t>(0); o.v<int32_t>(r.dp);
    }
    o.v<int32_t>(r.dp);  // DP_FORMAT
    o.v<int32_t>(r.gq);
    if (r.kind == 0) { for (int i = 0; i < 4; ++i) o.v<int32_t>(NULL_I32); o.v<int32_t>(0); o.v<int32_t>(3); for (int i = 0; i < 3; ++i) o.v<int32_t>(r.pl[i]); }
    else {
      for (int i = 0; i < 4; ++i) o.v<int32_t>(r.sb[i]);
      o.v<int32_t>(3); for (int i = 0; i < 3; ++i) o.v<int32_t>(r.ad[i]);
      o.v<int32_t>(6); for (int i = 0; i < 6; ++i) o.v<int32_t>(r.pl[i]);
    }
    o.v<int32_t>(0); o.v<int32_t>(0);  // PGT, PID
    o.v<int32_t>(r.min_dp);
    o.v<int32_t>(2);
    if (r.kind == 0) { o.v<int32_t>(0); o.v<int32_t>(0); }
    else { o.v<int32_t>(r.hom ? 1 : 0); o.v<int32_t>(1); }
    const uint64_t sz = (uint64_t)(o.p - dst);
    memcpy(dst + 16, &sz, 8);
  }

  // all cells with chunk_begin <= begin < col_end, column-major order.  Every thread generates a block of consecutive samples;
  // a counting pass over (column, thread) gives each thread the place of its cells inside every column (within a thread the
  // cells of one column come out in ascending sample order), so generation, ordering and serialisation all run in parallel.
  void next_chunk(int64_t col_end, int nthreads) {
    if (col_end > B + L) col_end = B + L;
    const int64_t ncols = std::max<int64_t>(0, col_end - chunk_begin);
    const int T = std::max(1, std::min(nthreads, (int)n_samples));
    std::vector<std::vector<Rec>> recs((size_t)T);
    std::vector<std::vector<uint32_t>> cnt((size_t)T), bytes((size_t)T);
    auto gen = [&](int t) {
      const int32_t r0 = (int32_t)((int64_t)n_samples * t / T), r1 = (int32_t)((int64_t)n_samples * (t + 1) / T);
      std::vector<Rec>& out = recs[(size_t)t];
      cnt[(size_t)t].assign((size_t)ncols, 0); bytes[(size_t)t].assign((size_t)ncols, 0);
      Rec r;
      for (int32_t row = r0; row < r1; ++row)
        while (true) {
          contig_end(pos[row]);            // (moves a position out of a gap between contigs before the chunk bound is looked at)
          if (pos[row] >= col_end) break;
          next_record(row, r);
          if (with_id) r.idlen |= 0x80;
          out.push_back(r);
          const size_t c = (size_t)(r.begin - chunk_begin);
          cnt[(size_t)t][c]++; bytes[(size_t)t][c] += cell_size(r);
        }
    };
    {
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back(gen, t);
      for (auto& x : th) x.join();
    }
    // place of (column, thread) in the chunk: bytes[t][c] becomes the byte offset of thread t's first cell of column c
    std::vector<uint64_t> col_base((size_t)ncols + 1, 0);
    uint64_t at = 0; int64_t total = 0;
    for (int64_t c = 0; c < ncols; ++c) {
      col_base[(size_t)c] = at;
      uint32_t within = 0;
      for (int t = 0; t < T; ++t) { const uint32_t b = bytes[(size_t)t][(size_t)c]; bytes[(size_t)t][(size_t)c] = within; within += b; total += cnt[(size_t)t][(size_t)c]; }
      at += within;
    }
    cells.resize((size_t)at);
    uint8_t* base = cells.data();
    auto ser = [&](int t) {
      std::vector<uint32_t>& off = bytes[(size_t)t];
      for (const Rec& r : recs[(size_t)t]) {
        const size_t c = (size_t)(r.begin - chunk_begin);
        write_cell(base + col_base[c] + off[c], r);
        off[c] += cell_size(r);
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back(ser, t);
      for (auto& x : th) x.join();
    }
    last_ncells = total;
    chunk_begin = col_end;
  }
};

}  // namespace

extern "C" {

void* gdbsynth_create(uint64_t seed, int32_t n_samples, int64_t B, int64_t L);
void* gdbsynth_create_dense(uint64_t seed, int32_t n_samples, int64_t B, int64_t L, int64_t dense_begin, int64_t dense_len, int64_t hot_stride, int32_t K) {
  Synth* s = (Synth*)gdbsynth_create(seed, n_samples, B, L);
  s->dense_begin = dense_begin; s->dense_len = dense_len; s->hot_stride = hot_stride > 0 ? hot_stride : 50; s->dense_K = K > 0 ? (K > 1024 ? 1024 : K) : 1;
  return s;
}
void* gdbsynth_create(uint64_t seed, int32_t n_samples, int64_t B, int64_t L) {
  Synth* s = new Synth;
  s->seed = seed; s->n_samples = n_samples; s->B = B; s->L = L; s->chunk_begin = B;
  s->rng.resize((size_t)n_samples); s->pos.assign((size_t)n_samples, B);
  for (int32_t r = 0; r < n_samples; ++r) { uint64_t st = seed ^ ((uint64_t)(r + 1) * 0x9E3779B97F4A7C15ull); s->rng[(size_t)r].s = splitmix64(st); }
  return s;
}
void gdbsynth_destroy(void* h) { delete (Synth*)h; }
// genome mode: n contigs as (tiledb_column_offset, length) pairs; call before the first chunk
void gdbsynth_set_contigs(void* h, const int64_t* offsets, const int64_t* lengths, int32_t n) {
  Synth* s = (Synth*)h;
  s->contigs.clear();
  for (int32_t i = 0; i < n; ++i) s->contigs.emplace_back(offsets[i], lengths[i]);
  std::sort(s->contigs.begin(), s->contigs.end());
}
void gdbsynth_set_float_stress(void* h, int permille) { ((Synth*)h)->float_stress_permille = permille; }
void gdbsynth_set_rank_sum_scale(void* h, double scale) { ((Synth*)h)->rs_scale = scale > 0 ? scale : 1000.0; }
void gdbsynth_set_modes(void* h, int overlap_permille, int filter_permille, int filter2_permille, int id_permille, int filter_id, int filter_id2, int with_id) {
  Synth* s = (Synth*)h;
  s->overlap_permille = overlap_permille; s->filter_permille = filter_permille; s->filter2_permille = filter2_permille; s->id_permille = id_permille;
  s->filter_id = filter_id; s->filter_id2 = filter_id2; s->with_id = with_id != 0;
}
// generates the next chunk (cells beginning before col_end); returns #cells, *cells / *nbytes valid until the next call
int64_t gdbsynth_next_chunk(void* h, int64_t col_end, int nthreads, const uint8_t** cells, uint64_t* nbytes) {
  Synth* s = (Synth*)h;
  s->next_chunk(col_end, std::max(1, nthreads));
  *cells = s->cells.data();
  *nbytes = s->cells.size();
  return s->last_ncells;
}
void gdbsynth_reference(uint64_t seed, int64_t begin, int64_t len, char* out) {
  Synth s; s.seed = seed;
  for (int64_t i = 0; i < len; ++i) out[i] = s.base(begin + i);
}

}  // extern "C"
