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
#include <algorithm>
#include <cmath>
#include <cstdint>
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
  double rs_scale = 1000.0;   // rank sums are rounded to 1 / rs_scale (coarse scales make tied medians, zeros of both signs included)
  bool in_dense(int64_t p) const { return dense_len > 0 && p >= dense_begin && p < dense_begin + dense_len; }
  bool is_hot(int64_t p) const { return in_dense(p) && (p % hot_stride) == 0; }

  char base(int64_t p) const { return "ACGT"[hash2(seed ^ 0x5bd1e9955bd1e995ull, (uint64_t)p) & 3]; }

  void next_record(int32_t row, Rec& r) {
    Rng& g = rng[row];
    memset(&r, 0, sizeof(r));
    r.row = row;
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
  }

  static void put(std::vector<uint8_t>& o, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; o.insert(o.end(), b, b + n); }
  template <class T> static void putv(std::vector<uint8_t>& o, T v) { put(o, &v, sizeof(T)); }

  // one cell in the reference binary layout, attribute order of tests/inputs/vid.json:
  // END REF ALT QUAL FILTER | BaseQRankSum ClippingRankSum MQRankSum ReadPosRankSum MQ RAW_MQ MQ0 DP | DP_FORMAT GQ SB AD PL PGT PID MIN_DP GT
  static void write_cell(std::vector<uint8_t>& o, const Rec& r) {
    size_t start = o.size();
    putv<int64_t>(o, r.row); putv<int64_t>(o, r.begin); putv<uint64_t>(o, 0);
    putv<int64_t>(o, r.end);
    putv<int32_t>(o, r.reflen); put(o, r.ref, r.reflen);
    if (r.kind == 0) { putv<int32_t>(o, 1); o.push_back('&'); }
    else { putv<int32_t>(o, r.altlen + 2); put(o, r.alt, r.altlen); o.push_back('|'); o.push_back('&'); }
    if (r.kind == 0) putv<uint32_t>(o, NULL_F32); else putv<float>(o, r.qual);
    putv<int32_t>(o, 0);  // FILTER: none
    if (r.kind == 0) { for (int i = 0; i < 6; ++i) putv<uint32_t>(o, NULL_F32); putv<int32_t>(o, NULL_I32); putv<int32_t>(o, NULL_I32); }
    else {
      for (int i = 0; i < 4; ++i) putv<float>(o, r.rs[i]);
      putv<float>(o, r.mq); putv<float>(o, r.raw_mq); putv<int32_t>(o, 0); putv<int32_t>(o, r.dp);
    }
    putv<int32_t>(o, r.dp);  // DP_FORMAT
    putv<int32_t>(o, r.gq);
    if (r.kind == 0) { for (int i = 0; i < 4; ++i) putv<int32_t>(o, NULL_I32); putv<int32_t>(o, 0); putv<int32_t>(o, 3); for (int i = 0; i < 3; ++i) putv<int32_t>(o, r.pl[i]); }
    else {
      for (int i = 0; i < 4; ++i) putv<int32_t>(o, r.sb[i]);
      putv<int32_t>(o, 3); for (int i = 0; i < 3; ++i) putv<int32_t>(o, r.ad[i]);
      putv<int32_t>(o, 6); for (int i = 0; i < 6; ++i) putv<int32_t>(o, r.pl[i]);
    }
    putv<int32_t>(o, 0); putv<int32_t>(o, 0);  // PGT, PID
    putv<int32_t>(o, r.min_dp);
    putv<int32_t>(o, 2);
    if (r.kind == 0) { putv<int32_t>(o, 0); putv<int32_t>(o, 0); }
    else { putv<int32_t>(o, r.hom ? 1 : 0); putv<int32_t>(o, 1); }
    uint64_t sz = o.size() - start;
    memcpy(&o[start + 16], &sz, 8);
  }

  // all cells with chunk_begin <= begin < col_end, column-major order
  void next_chunk(int64_t col_end, int nthreads) {
    if (col_end > B + L) col_end = B + L;
    std::vector<std::vector<Rec>> per_thread((size_t)nthreads);
    auto work = [&](int t) {
      std::vector<Rec>& out = per_thread[(size_t)t];
      Rec r;
      for (int32_t row = t; row < n_samples; row += nthreads)
        while (pos[row] < col_end) { next_record(row, r); out.push_back(r); }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(work, t);
    for (auto& x : th) x.join();
    std::vector<const Rec*> all;
    size_t total = 0;
    for (auto& v : per_thread) total += v.size();
    all.reserve(total);
    for (auto& v : per_thread) for (auto& r : v) all.push_back(&r);
    std::sort(all.begin(), all.end(), [](const Rec* a, const Rec* b) { return a->begin < b->begin || (a->begin == b->begin && a->row < b->row); });
    cells.clear();
    cells.reserve(total * 160);
    for (auto* r : all) write_cell(cells, *r);
    last_ncells = (int64_t)total;
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
void gdbsynth_set_rank_sum_scale(void* h, double scale) { ((Synth*)h)->rs_scale = scale > 0 ? scale : 1000.0; }
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
