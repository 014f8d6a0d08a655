// gdb_pipeline.hip - HIP kernels + stage orchestration of the MI355X variant-combine path (gfx950).
//
// Kernel bodies are the GDB_HD functions of ../core; this file adds the grid mapping, the wavefront (64-lane)
// reductions / scans, and the device-wide sorts and scans (rocPRIM).  All of it is integer / byte work bound by
// HBM traffic: no MFMA, no GEMM reshaping (see DESIGN.md for the per-kernel byte accounting).
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/reverse_iterator.hpp>

#include "../core/gdb_stages.hpp"
#include "../core/gdb_bcf.hpp"
#include "../core/gdb_calls.hpp"
#include "gdb_pipeline.h"
#include "gdb_bgzf.h"

namespace genomicsdb_amd {

// error bits (GdbErr, core/gdb_types.h) spelled out for the exception text
static std::string err_bits_text(uint32_t bits) {
  static const char* names[] = {"an interval is overlapped by one of the same sample that is neither a reference block nor a deletion",
                                "more than GDB_MAX_MERGED_ALLELES alleles in one record", "more than GDB_MAX_INPUT_ALLELES alleles in one cell",
                                "ploidy above GDB_MAX_PLOIDY", "(retired: every float has a text since round 4)", "page arena overflow", "internal inconsistency",
                                "more than GDB_MAX_ID_TOKENS distinct ID tokens in one record", "an element_wise_sum INFO vector longer than GDB_MAX_INFO_VECTOR",
                                "malformed or unsorted cell stream", "more than GDB_MAX_FILTER_IDS distinct FILTER ids in one record"};
  std::string out = "device error bits " + std::to_string(bits) + " (";
  bool first = true;
  for (unsigned b = 0; b < 32; ++b)
    if (bits & (1u << b)) { if (!first) out += "; "; first = false; out += b < sizeof(names) / sizeof(names[0]) ? names[b] : "unknown"; }
  return out + "; limits: include/genomicsdb_amd.h, GdbErr: core/gdb_types.h)";
}


#define HIP_CHECK(expr)                                                                                            \
  do {                                                                                                             \
    hipError_t _e = (expr);                                                                                        \
    if (_e != hipSuccess)                                                                                          \
      throw GenomicsDBDeviceException(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " at " + __FILE__ + ":" + std::to_string(__LINE__)); \
  } while (0)

namespace {

constexpr int kBlock = 256;          // threads per workgroup = 4 wavefronts of 64
constexpr int kCountSpread = 64;     // addresses a device-wide counter is spread over

// Device blocks of the staging path (the parts of a fragment being staged, the merged fragment) are kept and handed out again:
// hipFree synchronises the whole device, and with a second pipeline staging the next column window while this one computes
// (CombineEngine's overlapped staging) a free on the staging thread would wait for - and serialise with - the page kernels.
// (put() keeps the memory: after finish_staging the blocks of the parts stay with the pool next to the merged fragment, and an engine over a
//  windowed source owns two pipelines with a pool each - resident HBM for staging is up to ~4 x one window of GDBAMD_STAGE_BUDGET_MB; a block
//  more than twice a request + 4 MB large is not handed out for it.  Size the budget with that factor in mind.)
struct BlockPool {
  struct Block { void* p; size_t cap; bool used; };
  std::vector<Block> blocks;
  void* get(size_t bytes) {
    bytes = std::max<size_t>(bytes, 256);
    int best = -1;
    for (size_t i = 0; i < blocks.size(); ++i)
      if (!blocks[i].used && blocks[i].cap >= bytes && (best < 0 || blocks[i].cap < blocks[(size_t)best].cap)) best = (int)i;
    if (best >= 0 && blocks[(size_t)best].cap <= 2 * bytes + ((size_t)4 << 20)) { blocks[(size_t)best].used = true; return blocks[(size_t)best].p; }
    void* d = nullptr;
    const size_t cap = (bytes + bytes / 8 + 255) & ~(size_t)255;
    HIP_CHECK(hipMalloc(&d, cap));
    blocks.push_back(Block{d, cap, true});
    return d;
  }
  void put(void* p) { for (auto& b : blocks) if (b.p == p) { b.used = false; return; } if (p) (void)hipFree(p); }
  void release_all() { for (auto& b : blocks) (void)hipFree(b.p); blocks.clear(); }
};

template <class T> struct DevBuf {   // grow-only device allocation
  T* p = nullptr;
  size_t cap = 0;
  void ensure(size_t n) {
    if (n <= cap) return;
    if (p) (void)hipFree(p);
    p = nullptr;
    size_t want = n + n / 8 + 16;
    HIP_CHECK(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  ~DevBuf() { release(); }
};

inline bool debug_sync() { static int v = -1; if (v < 0) { const char* e = getenv("GDBAMD_DEBUG_SYNC"); v = (e && *e && *e != '0') ? 1 : 0; } return v == 1; }
#define STAGE(name) do { if (debug_sync()) { HIP_CHECK(hipStreamSynchronize(st)); fprintf(stderr, "[gdbamd] stage %s\n", name); fflush(stderr); } } while (0)
// Records one wavefront takes in a row.  Measured on c2 (200 kb windows): the sizing pass, which starts every run with a
// binary search per sample, is fastest at 64; the page pass (no per-run set-up beyond one 64-lane scalar fetch) at 32 -
// more, shorter runs balance better across the 256 CUs than longer ones amortise.  GDBAMD_RUN / GDBAMD_RUN_W override.
inline int size3_rounds();
inline int size_run_length() { const char* e = getenv("GDBAMD_RUN"); return e && *e ? std::max(1, atoi(e)) : size3_rounds() ? 128 : 64; }   // (the piece-wise kernel: 128 measured 0.3 ms faster than 64, 256 slower)
inline int write_run_length() { const char* e = getenv("GDBAMD_RUN_W"); return e && *e ? std::max(1, atoi(e)) : 32; }
// record types that get text-table slots (GDBAMD_MAX_TYPES < 64 forces the direct path in tests)
inline int max_tabled_types() {
  const char* e = getenv("GDBAMD_MAX_TYPES");   // read per interval: tests flip it in-process
  return e && *e ? std::max(0, std::min(64, atoi(e))) : 64;
}
// HBM the resolved (record, sample) matrix of a whole interval may take (8 bytes x records x samples rounded up to 64);
// wider intervals resolve page by page.  GDBAMD_RESOLVED_MB overrides (tests use 0 to force the per-page path).
inline uint64_t resolved_budget_bytes() {
  const char* e = getenv("GDBAMD_RESOLVED_MB");
  return e && *e ? (uint64_t)atoll(e) << 20 : (uint64_t)32 << 30;
}
// records per (run, chunk) of the change-list page assembly (<= 64); GDBAMD_EVENTS=1 selects it instead of the dense matrix
inline int event_run_length() { const char* e = getenv("GDBAMD_EV_RUN"); return e && *e ? std::max(1, std::min(64, atoi(e))) : 32; }
// (off by default: measured slower than the dense matrix, with scalar loads of the changes and with vector loads + v_readlane)
inline bool events_enabled() { const char* e = getenv("GDBAMD_EVENTS"); return e && *e && *e != '0'; }
// How an interval's sample columns are sized and assembled (GDBAMD_ASM_PATH, read per interval; measured in DESIGN.md section 5):
//   0 (default)  the sizing pass of rounds 1-3 (k_assemble_size: every (record, sample) pair walked, sizes and matrix in one pass), pages by k_assemble_write
//   1            text only: sizes from pieces (k_size2), no matrix at all, the page pass walks the pieces itself (k_write2)
//   2            sizes from pieces (k_size2), the matrix from a piece walker per lane (k_fill2), pages by k_assemble_write / the BCF kernels
//   3            text only: sizes from pieces (k_size2), piece lists (k_plist), pages by k_write3 (no matrix, no walking in the page pass)
inline int asm_path_wanted() { const char* e = getenv("GDBAMD_ASM_PATH"); return e && *e ? std::max(0, std::min(3, atoi(e))) : 0; }
// sizing pass: 3 = k_assemble_size3 in rounds of 8 records (default), 16 = rounds of 16, 0 = k_assemble_size
inline int size3_rounds() { const char* e = getenv("GDBAMD_SIZE3"); const int v = e && *e ? atoi(e) : 8; return v == 0 ? 0 : v >= 16 ? 16 : 8; }
inline bool res_chunk_major() { static const bool v = []() { const char* e = getenv("GDBAMD_RES_LAYOUT"); return e && *e == '1'; }(); return v; }   // (measured: no difference; record-major stays)
// compact resolved matrix (ResMatrix): 1 (default) whenever the interval allows it, 0 never (A/B runs); the check mode of the sizing
// kernels compares the wide layout word for word, so it keeps that one
inline bool size3_check();
inline bool res_compact_wanted();
inline bool size3_check() { const char* e = getenv("GDBAMD_SIZE3_CHECK"); return e && *e && *e != '0'; }
inline bool res_compact_wanted() { if (size3_check()) return false; const char* e = getenv("GDBAMD_RES_COMPACT"); return !(e && *e == '0'); }
inline int fill_run_length() { const char* e = getenv("GDBAMD_RUN_F"); return e && *e ? std::max(1, atoi(e)) : 128; }
inline int write2_run_length() { const char* e = getenv("GDBAMD_RUN_W2"); return e && *e ? std::max(1, atoi(e)) : 64; }
// wavefronts (= neighbouring 64-sample chunks) per workgroup of the page assembly: 4 measured best on the store-only model
inline int write_waves_per_group() { const char* e = getenv("GDBAMD_WRITE_WAVES"); return e && *e ? atoi(e) : 1; }
inline bool slot_regroup() { static const bool v = []() { const char* e = getenv("GDBAMD_SLOT_REGROUP"); return !(e && *e == '0'); }(); return v; }
// the page kernel on a high-priority stream: GDBAMD_PAGE_PRIORITY=0 / 1 overrides what the engine asked for (set_page_priority)
inline int page_priority_env() { const char* e = getenv("GDBAMD_PAGE_PRIORITY"); return e && (*e == '0' || *e == '1') ? *e - '0' : -1; }
inline bool coop_unroll_wanted() { const char* e = getenv("GDBAMD_COOP_UNROLL"); return !(e && *e == '0'); }   // (A/B: 0 keeps the word-by-word copy of long entries)
inline bool xcd_aware_numbering() { const char* e = getenv("GDBAMD_XCD_AWARE"); return !(e && *e == '0'); }
// LDS image of one (record, 64-sample chunk) of the page kernel.  GDBAMD_WRITE_IMAGE_KB = 4 / 6 / 8 forces one; otherwise by the
// interval's average chunk: 4 KiB while a chunk (c2: 2.8 KB) and its alignment slack fit - more resident wavefronts per CU, 0.4-0.6 ms
// of a 12 ms launch, three A/B rounds inside one call (profiles/r5_ab_image.txt) - and 8 KiB for wider entries (c3's width: ~6 KB
// per chunk would take two passes through a 4 KiB image)
inline int write_image_kb(uint64_t avg_chunk_bytes = 0) {
  const char* e = getenv("GDBAMD_WRITE_IMAGE_KB");
  if (e && *e) return atoi(e);
  return avg_chunk_bytes && avg_chunk_bytes <= 3400 ? 4 : 8;
}
inline int order_block_log2() {
  const char* e = getenv("GDBAMD_ORDER_BLOCK_LOG2");
  return e && *e ? std::max(0, std::min(30, atoi(e))) : 12;
}
inline unsigned blocks_for(int64_t n, int b = kBlock) { return (unsigned)std::max<int64_t>(1, (n + b - 1) / b); }
inline int bits_for(uint64_t max_value) { int b = 1; while (b < 64 && (max_value >> b)) ++b; return std::min(64, b + 1); }

// ---- wavefront scan / reduce on DPP (no LDS traffic) --------------------------------------------------------------------
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v, int n) {   // lane i <- lane i-n of the same 16-lane row, else 0
  switch (n) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
  }
}
__device__ __forceinline__ uint32_t wave_inclusive_scan_dpp(uint32_t v) {
  v += dpp_row_shr(v, 1);
  v += dpp_row_shr(v, 2);
  v += dpp_row_shr(v, 4);
  v += dpp_row_shr(v, 8);
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_total(uint32_t inclusive) { return (uint32_t)__builtin_amdgcn_readlane((int)inclusive, 63); }

// unaligned 4 / 8 byte accesses (gfx950 serves them in hardware)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed)) PackedU32 { uint32_t v; };
struct __attribute__((packed)) PackedU64 { uint64_t v; };

// ---- elementwise stage kernels -----------------------------------------------------------------------------
__global__ void k_classify(FragmentView fr, CombinePlan pl, CellMeta cm, uint32_t* err) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= fr.ncells) return;
  uint32_t e = 0;
  classify_cell(fr, pl, cm, c, &e);
  if (e) atomicOr(err, e);
}
__global__ void k_iota_rows(FragmentView fr, int32_t* keys, int64_t* vals) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= fr.ncells) return;
  keys[c] = fr.row[c];
  vals[c] = c;
}
__global__ void k_row_ptr(const int32_t* sorted_rows, int64_t C, int32_t N, int64_t* row_ptr) {
  int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > N) return;
  int64_t lo = 0, hi = C;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (sorted_rows[mid] < (int32_t)r) lo = mid + 1; else hi = mid; }
  row_ptr[r] = lo;
}
__global__ void k_eff_end(FragmentView fr, CellMeta cm, const int64_t* perm, int64_t* rm_begin, int64_t* span, uint32_t* err) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= fr.ncells) return;
  uint32_t e = 0;
  stage_eff_end(fr, cm, perm, j, rm_begin, span, &e);
  if (e) atomicOr(err, e);
}
// cells that can intersect [qb,qe]: begin in [qb - max_span, qe]  (cells are sorted by begin)
__global__ void k_cell_window(const int64_t* begin, int64_t C, int64_t lo_pos, int64_t hi_pos, int64_t* out) {
  if (blockIdx.x || threadIdx.x) return;
  int64_t lo = 0, hi = C;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (begin[mid] < lo_pos) lo = mid + 1; else hi = mid; }
  out[0] = lo;
  hi = C;
  while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (begin[mid] <= hi_pos) lo = mid + 1; else hi = mid; }
  out[1] = lo;
}
// smallest begin >= pos of a sorted begin array (INT64_MAX: none): candidate split point of a wide query interval
__global__ void k_first_begin_at_or_after(const int64_t* begin, int64_t C, int64_t pos, int64_t* out) {
  if (blockIdx.x || threadIdx.x) return;
  int64_t lo = 0, hi = C;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (begin[mid] < pos) lo = mid + 1; else hi = mid; }
  *out = lo < C ? begin[lo] : INT64_MAX;
}
__global__ void k_event_keys(FragmentView fr, CellMeta cm, int64_t c_base, int64_t n, int64_t qb, int64_t qe, uint64_t* keys) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  stage_event_keys(fr, cm, c_base + i, c_base, qb, qe, keys);
}
__global__ void k_marker_keys(FragmentView fr, int64_t m_base, int64_t n, int64_t qb, int64_t qe, uint64_t* keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) stage_marker_keys(fr, m_base + i, qb, qe, keys + 2 * i);
}
__global__ void k_event_delta(const uint64_t* keys, int64_t n, int64_t* delta, int32_t* run_end) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  delta[i] = stage_event_delta(keys[i]);
  run_end[i] = stage_is_run_end(keys, n, i);
}
struct PackedAdd { __host__ __device__ int64_t operator()(const int64_t& a, const int64_t& b) const { return packed_add(a, b); } };
__global__ void k_boundary_write(const uint64_t* keys, const int64_t* incl, const int32_t* run_end, const int32_t* run_excl, int64_t n, Boundaries b, int64_t qb) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  stage_boundary_write(keys, incl, run_excl, i, run_end[i], b, qb);
}
__global__ void k_boundary_nrec(Boundaries b, int64_t U) {
  int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  b.nrec[u] = stage_boundary_nrec(b, U, u);
}
__global__ void k_record_expand(Boundaries b, const int64_t* rbase, int64_t U, int64_t P, int64_t* rstart, int64_t* rend) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  stage_record_expand(b, rbase, U, k, rstart, rend);
}
// first_record_at[p] := k for the record starting at qb + p (the starts are distinct and ascending); the suffix-minimum scan
// that follows turns the marks into "first record starting at or behind qb + p"
__global__ void k_mark_record_starts(const int64_t* rstart, int64_t P, int64_t qb, int32_t* first_record_at) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < P) first_record_at[rstart[k] - qb] = (int32_t)k;
}
__global__ void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
// packed prefix sums -> the int32 arrays the site kernels read (PresenceCounts)
__global__ void k_unpack_counts(DiffPacked pk, int n_format, int64_t n, DiffArrays d) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint64_t mask = pk.width >= 64 ? ~0ull : ((1ull << pk.width) - 1ull);
  for (int j = 0; j <= n_format; ++j) {
    const uint64_t w = pk.w[(int64_t)(j / pk.per_word) * pk.stride + k];
    const int32_t v = (int32_t)((w >> ((j % pk.per_word) * pk.width)) & mask);
    if (j < n_format) d.fmt[(int64_t)j * d.stride + k] = v; else d.nr[k] = v;
  }
  d.dp[k] = (int32_t)(int64_t)pk.w[(int64_t)(pk.nwords - 1) * pk.stride + k];
}
// Cells arrive sorted by begin, so neighbouring lanes mostly share klo (about ten cells start per record): the +words of a
// wavefront are first summed per run of equal klo (segmented suffix sum over the sorted key), and only the first lane of a
// run issues the atomic.  The -words at khi + 1 are scattered and go out one per lane.
__device__ __forceinline__ uint64_t wave_run_sum(uint32_t key, uint64_t v, int lane) {   // key = run id: equal ids are contiguous
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t ok = __shfl_down(key, off, 64);
    const uint64_t ov = __shfl_down(v, off, 64);
    if (lane + off < 64 && ok == key) v += ov;
  }
  return v;
}
__global__ void k_cell_ranges(FragmentView fr, CombinePlan pl, CellMeta cm, RecordTable rec, int64_t c_base, int64_t n, int64_t qb, int64_t qe,
                              int64_t* heavy_count, int32_t* in_window_count, const int32_t* first_record_at, DiffPacked pk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  bool live = false;
  int64_t klo = -1 - (int64_t)lane, khi = -1;      // dead lanes: distinct keys, never merged with anything
  uint64_t vm = 0; uint32_t cf = 0; int32_t dp = 0;
  if (i < n) {
    const int64_t c = c_base + i;
    live = stage_cell_range(fr, cm, rec, c, c_base, qb, qe, heavy_count, first_record_at, klo, khi);
    if (live) { vm = cm.vmask[c]; cf = cm.cflags[c]; dp = cm.dpval[c]; } else klo = -1 - (int64_t)lane;
  }
  const int64_t prev_key = __shfl_up(klo, 1, 64);
  const bool head = lane == 0 || prev_key != klo;
  const uint32_t run_id = wave_inclusive_scan_dpp(head ? 1u : 0u);   // runs of equal klo (dead lanes are runs of their own)
  // The -words: normally scattered (cells that begin together end apart), one atomic per lane.  At a hot site of BASELINE configs[4]
  // 50 000 cells begin AND end together: 50 000 x nwords atomics on one address per site (k_cell_ranges 6.0 ms per 2 000-column piece
  // of c5, two thirds of its sweep).  When most of the wavefront's neighbours share khi as well, the -words are summed per run like
  // the +words (runs of equal (klo, khi): a subdivision of the klo runs).
  const int64_t prev_khi = __shfl_up(khi, 1, 64);
  const bool same_khi = live && lane > 0 && prev_key == klo && prev_khi == khi;
  const bool runs_end_together = __popcll(__ballot(same_khi)) >= 32;                 // uniform
  const bool head2 = head || prev_khi != khi;
  const uint32_t run_id2 = runs_end_together ? wave_inclusive_scan_dpp(head2 ? 1u : 0u) : 0u;
  for (int word = 0; word < pk.nwords; ++word) {           // uniform
    uint64_t acc = 0;
    if (live) acc = word + 1 < pk.nwords ? stage_packed_word(pl, pk, vm, cf, word) : (uint64_t)(int64_t)dp;
    const uint64_t run = wave_run_sum(run_id, acc, lane);
    uint64_t* w = pk.w + (int64_t)word * pk.stride;
    if (live && head && run) atomicAdd((unsigned long long*)(w + klo), (unsigned long long)run);
    if (runs_end_together) {
      const uint64_t run2 = wave_run_sum(run_id2, acc, lane);
      if (live && head2 && run2) atomicAdd((unsigned long long*)(w + khi + 1), (unsigned long long)(0 - run2));
    } else if (live && acc) atomicAdd((unsigned long long*)(w + khi + 1), (unsigned long long)(0 - acc));
  }
  // one atomic per wavefront, spread over kCountSpread addresses (same-address atomics serialise in L2: ~10 ns each)
  const uint64_t m = __ballot(live);
  if (lane == 0 && m) atomicAdd(in_window_count + 16 + ((blockIdx.x * 4u + (threadIdx.x >> 6)) & (kCountSpread - 1)), (int32_t)__popcll(m));
  // GTProfileStats (query_variants.h:67-124): cells the left sweep contributes (begin before the interval, live at its first
  // column) and cells cut short by the next cell of their sample (the reference flushes its PQ there); both are rare
  bool left = false, cut = false;
  if (live) { const int64_t c = c_base + i; left = fr.begin[c] < qb; cut = cm.eff_end[c] < fr.end[c]; }
  const uint64_t ml = __ballot(left), mc = __ballot(cut);
  if (lane == 0 && ml) atomicAdd(in_window_count + 4, (int32_t)__popcll(ml));
  if (lane == 0 && mc) atomicAdd(in_window_count + 5, (int32_t)__popcll(mc));
}
__global__ void k_incidence_fill(FragmentView fr, CellMeta cm, const int64_t* hoff, int64_t c_base, int64_t n, int64_t nrows, uint64_t* keys, int64_t* vals) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  stage_incidence_fill(fr, cm, hoff, c_base + i, c_base, nrows, keys, vals, nullptr);
}
__global__ void k_heavy_base(const uint64_t* sorted_keys, int64_t T, int64_t nrows, int64_t P, int64_t* hbase) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > P) return;
  hbase[k] = stage_heavy_base(sorted_keys, T, nrows, k);
}
__global__ void k_lut_len(const int64_t* inc_cell, const uint32_t* cflags, int64_t T, uint32_t* len) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  len[t] = GDB_CF_NALT(cflags[inc_cell[t]]) + 1;
}

// ---- sorted medians (records with many variant calls) ------------------------------------------------------------------
// reduce_scalar's median scan is quadratic in the number of variant calls of a record: fine for ~10, ruinous for the
// thousands that tens of thousands of samples bring.  When the interval averages more than kSortedMedianThreshold calls per
// record every median field is ordered once by a device-wide radix sort (key layout at MedianOrder).
constexpr int kSortedMedianThreshold = 16;
__global__ void k_median_keys(FragmentView fr, CombinePlan pl, CellMeta cm, RecordTable rec, const uint64_t* inc_keys_sorted, const int64_t* inc_cell, int64_t T,
                              int64_t nrows, int f, int keep_spanning, uint64_t* keys, uint32_t* idx) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const int64_t k = (int64_t)(inc_keys_sorted[t] / (uint64_t)nrows);
  const int64_t c = inc_cell[t];
  bool valid = field_valid(cm, c, f);
  if (valid && !keep_spanning && (cm.cflags[c] & GDB_CF_DELETION) && rec.start[k] > fr.begin[c]) valid = false;   // inc_is_spanning
  uint32_t bits = 0;
  if (valid) {
    int n;
    if (pl.field[f].elem == GDB_ET_FLOAT) { const float v = cell_field<float>(fr, pl, f, c, n)[0]; valid = gdb_float_valid(v); bits = gdb_orderable_bits(v); }
    else { const int32_t v = cell_field<int32_t>(fr, pl, f, c, n)[0]; valid = gdb_int_valid(v); bits = gdb_orderable_bits(v); }
  }
  keys[t] = ((uint64_t)k << 33) | (valid ? 0ull : 1ull << 32) | (valid ? bits : 0u);
  idx[t] = (uint32_t)t;
}

// ---- scalar reducer inputs, one value per (record, variant call) incidence (ScalarPre): the site thread then sums / counts
// over a contiguous array instead of chasing cell -> validity mask -> field offset -> value once per call and field
__global__ void k_scalar_values(FragmentView fr, CombinePlan pl, CellMeta cm, RecordTable rec, const uint64_t* inc_keys_sorted, const int64_t* inc_cell, int64_t T,
                                int64_t nrows, int f, int keep_spanning, uint32_t* out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const bool is_float = pl.field[f].elem == GDB_ET_FLOAT;
  const int64_t c = inc_cell[t];
  uint32_t bits = is_float ? GDB_BCF_FLOAT_MISSING_BITS : (uint32_t)GDB_BCF_INT32_MISSING;
  bool valid = field_valid(cm, c, f);
  if (valid && !keep_spanning && (cm.cflags[c] & GDB_CF_DELETION)) {
    const int64_t k = (int64_t)(inc_keys_sorted[t] / (uint64_t)nrows);
    if (rec.start[k] > fr.begin[c]) valid = false;                                  // inc_is_spanning
  }
  if (valid) {
    int n;
    if (is_float) { const float v = cell_field<float>(fr, pl, f, c, n)[0]; if (gdb_float_valid(v)) bits = gdb_f2u(v); }
    else { const int32_t v = cell_field<int32_t>(fr, pl, f, c, n)[0]; if (gdb_int_valid(v)) bits = (uint32_t)v; }
  }
  out[t] = bits;
}

// ---- medians of records with very many variant calls: one workgroup per (big record, median field) ------------------------
constexpr int kBigRecord = 48;          // variant calls from which a record counts as big
constexpr int kBigCapacity = 4096;      // values a workgroup holds in LDS; larger records keep the per-thread scan
constexpr int kMaxBigRecords = 4096;
__global__ void k_big_records(const int64_t* hbase, int64_t P, int32_t* big_index, int32_t* big_list, int32_t* counter) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t n = hbase[k + 1] - hbase[k];
  int32_t idx = -1;
  if (n > kBigRecord && n <= kBigCapacity) { idx = atomicAdd(counter, 1); if (idx < kMaxBigRecords) big_list[idx] = (int32_t)k; else idx = -1; }
  big_index[k] = idx;
}
// The rule of reduce_scalar: the median is the value v with #less <= n_valid/2 < #less-or-equal; among equal values the
// first call in row order supplies the bits.
__global__ void __launch_bounds__(kBlock) k_big_medians(FragmentView fr, CombinePlan pl, CellMeta cm, RecordTable rec, HeavyLists hl, const int32_t* big_list,
                                                        const int32_t* counter, int f, int slot, int keep_spanning, int64_t stride, uint32_t* value, uint8_t* ok) {
  __shared__ uint32_t s_bits[kBigCapacity];     // order-preserving keys of the valid values
  __shared__ uint32_t s_raw[kBigCapacity];      // their raw bit patterns, in row order
  __shared__ int32_t s_n, s_best;
  __shared__ uint32_t s_zero_signs;             // bit 0: a +0 among the values, bit 1: a -0
  const int count = min(*counter, kMaxBigRecords);
  for (int bi = blockIdx.x; bi < count; bi += gridDim.x) {   // uniform
    const int64_t k = big_list[bi];
    const int64_t b = hl.base[k], e = hl.base[k + 1];
    const int64_t s_k = rec.start[k];
    if (threadIdx.x == 0) { s_n = 0; s_best = 0x7FFFFFFF; s_zero_signs = 0; }
    __syncthreads();
    // gather the valid values in row order: one thread walks the (sorted) incidences - a few hundred to a few thousand
    if (threadIdx.x == 0) {
      int n = 0;
      for (int64_t t = b; t < e; ++t) {
        const int64_t c = hl.cell[t];
        if (!keep_spanning && (cm.cflags[c] & GDB_CF_DELETION) && s_k > fr.begin[c]) continue;
        if (!field_valid(cm, c, f)) continue;
        int nn;
        if (pl.field[f].elem == GDB_ET_FLOAT) { const float v = cell_field<float>(fr, pl, f, c, nn)[0]; if (!gdb_float_valid(v)) continue; s_bits[n] = gdb_orderable_bits(v); s_raw[n] = gdb_f2u(v); }
        else { const int32_t v = cell_field<int32_t>(fr, pl, f, c, nn)[0]; if (!gdb_int_valid(v)) continue; s_bits[n] = gdb_orderable_bits(v); s_raw[n] = (uint32_t)v; }
        ++n;
      }
      s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    const int mid = n / 2;
    const bool is_float = pl.field[f].elem == GDB_ET_FLOAT;
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const uint32_t v = s_bits[i];
      int less = 0, leq = 0;
      for (int j = 0; j < n; ++j) { const uint32_t w = s_bits[j]; less += w < v; leq += w <= v; }
      if (less <= mid && mid < leq) atomicMin(&s_best, i);
      if (is_float && (s_raw[i] << 1) == 0u) atomicOr(&s_zero_signs, s_raw[i] ? 2u : 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const int64_t at = (int64_t)slot * stride + bi;
      // a zero median with both signs present: which zero the reference's nth_element leaves in the middle is worked out by
      // reduce_scalar (gdb_nth_element_libstdcxx) - ok = 2 sends the site thread there
      const bool tie = n > 0 && is_float && s_zero_signs == 3u && (s_raw[s_best] << 1) == 0u;
      ok[at] = n > 0 ? (tie ? 2 : 1) : 0;
      value[at] = n > 0 ? s_raw[s_best] : 0u;
    }
    __syncthreads();
  }
}

// ---- records with very many variant calls: one WORKGROUP per record does everything that walks the calls ----------------------
// One thread walking the ~10 000 calls of a hot site is a chain of dependent loads, ~14 us per call (BASELINE configs[4]: 142 of
// 146 ms of an interval).  Here lane = call: the ALT alleles of all calls go through an LDS hash table keyed by the allele hash,
// every entry keeps the smallest (call, token) position it was seen at, and the merged list is the entries in that order - the
// first-appearance order of the reference's merge_alt_alleles (variant_operations.cc:134-228); a second pass looks every
// candidate up, writes the input -> merged LUTs (strings compared against the entry's representative: a true hash collision
// sends the record back to the serial walk) and the flags / min-PL genotypes.  Scalar INFO reducers: counts and integer sums by
// reduction, float sums in call order by one thread out of LDS-staged chunks (float addition is not associative), medians by a
// 4-pass radix select over the order-preserving keys; FILTER union by reduction.  site_emit then only formats (HugeSites).
constexpr int kHugeRecord = 256;          // variant calls from which a record counts as huge
constexpr int kMaxHugeRecords = 8192;
constexpr int kHugeTable = 512;           // LDS hash table slots (at most 127 distinct alleles per record)
constexpr uint32_t kHugeEmpty = 0xFFFFFFFFu;
__global__ void k_huge_records(const int64_t* hbase, int64_t P, int32_t threshold, int32_t* huge_index, int32_t* huge_list, int32_t* counter) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t n = hbase[k + 1] - hbase[k];
  int32_t idx = -1;
  if (n > threshold) { idx = atomicAdd(counter, 1); if (idx < kMaxHugeRecords) huge_list[idx] = (int32_t)k; else idx = -1; }
  huge_index[k] = idx;
}
struct HugeTable { uint32_t* hash; uint32_t* order; int16_t* midx; int32_t* overflow; };
__device__ __forceinline__ uint32_t huge_fix_hash(uint32_t h) { return h == kHugeEmpty ? 0xFFFFFFFEu : h; }
struct HugeInsert {          // pass A: note every candidate allele with the position it appears at
  HugeTable tb; uint32_t order_base; uint32_t i;
  __device__ int operator()(const AlleleRef& cand) {
    const uint32_t h = huge_fix_hash(cand.hash);
    uint32_t s = h & (kHugeTable - 1);
    for (int probe = 0; probe < kHugeTable; ++probe, s = (s + 1) & (kHugeTable - 1)) {
      uint32_t cur = tb.hash[s];
      if (cur == kHugeEmpty) cur = atomicCAS(&tb.hash[s], kHugeEmpty, h);
      if (cur == kHugeEmpty || cur == h) { atomicMin(&tb.order[s], order_base + i); ++i; return 0; }
    }
    *tb.overflow = 1;
    ++i;
    return 0;
  }
};
struct HugePick {            // the i-th candidate of a call (the representative of a table entry)
  uint32_t want, i; AlleleRef out; bool found;
  __device__ int operator()(const AlleleRef& cand) { if (i == want) { out = cand; found = true; } ++i; return 0; }
};
struct HugeLookup {          // pass C: merged index of every candidate
  HugeTable tb; const AlleleRef* merged; const char* mref; int mref_len; int32_t* fallback;
  __device__ int operator()(const AlleleRef& cand) {
    const uint32_t h = huge_fix_hash(cand.hash);
    uint32_t s = h & (kHugeTable - 1);
    for (int probe = 0; probe < kHugeTable; ++probe, s = (s + 1) & (kHugeTable - 1)) {
      if (tb.hash[s] == h) {
        const int m = tb.midx[s];
        if (m < 1 || !allele_equal(merged[m], cand, mref, mref_len)) { *fallback = 1; return 1; }   // two alleles, one hash: serial walk
        return m;
      }
      if (tb.hash[s] == kHugeEmpty) break;
    }
    *fallback = 1;
    return 1;
  }
};
// (1 024 lanes per hot record: at the stated size of BASELINE configs[4] a record has 50 000 calls and a 2 kb piece holds 40 such
// records - fewer than there are CUs - so the lanes of ONE record are what fills the device)
constexpr int kHugeBlock = 1024;
// exclusive scan of one value per thread over the workgroup of k_site_huge (16 wavefronts); s_w: 17 LDS words
__device__ __forceinline__ uint32_t huge_block_scan(uint32_t v, uint32_t* s_w, uint32_t& total) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t incl = wave_inclusive_scan_dpp(v);
  if (lane == 63) s_w[wv] = incl;
  __syncthreads();
  if (tid == 0) { uint32_t acc = 0; for (int i = 0; i < kHugeBlock / 64; ++i) { const uint32_t x = s_w[i]; s_w[i] = acc; acc += x; } s_w[kHugeBlock / 64] = acc; }
  __syncthreads();
  const uint32_t r = s_w[wv] + incl - v;
  total = s_w[kHugeBlock / 64];
  __syncthreads();
  return r;
}
// The zero the reference's std::nth_element leaves at the middle when -0 and +0 tie there, by the whole workgroup: gdb_nth_element_by_lists
// (gdb_core.hpp: libstdc++'s introselect with every Hoare sweep stated as two position lists, pairwise swaps and one comparison for the
// cut) with the lists built by workgroup scans.  One thread walking the 50 000 calls of a hot site of BASELINE configs[4] took 1.3 ms per
// record and field; 2 000 such records were 2.7 s of a 4.0 s run.  val[hb .. he): the calls' values in call order (`missing`: none);
// a / posL / posR: scratch of nvalid entries each.  Every thread returns the selected value's bits.
__device__ uint32_t huge_tie_median(const uint32_t* __restrict__ val, int64_t hb, int64_t he, uint32_t missing, float* a, uint32_t* posL, uint32_t* posR,
                                    uint32_t* s_w, uint32_t* s_m, uint32_t* s_res) {
  const int tid = threadIdx.x;
  uint32_t total;
  {   // the valid values, in call order
    const int64_t ninc = he - hb, per = (ninc + kHugeBlock - 1) / kHugeBlock;
    const int64_t b = hb + min(ninc, (int64_t)tid * per), e = hb + min(ninc, (int64_t)(tid + 1) * per);
    uint32_t cnt = 0;
    for (int64_t t = b; t < e; ++t) cnt += val[t] != missing;
    uint32_t at = huge_block_scan(cnt, s_w, total);
    for (int64_t t = b; t < e; ++t) { const uint32_t u = val[t]; if (u != missing) a[at++] = __uint_as_float(u); }
  }
  __syncthreads();
  const int64_t n = total, nth = n / 2;
  int depth = 0;
  for (int64_t m = n; m > 1; m >>= 1) ++depth;
  depth *= 2;
  int64_t first = 0, last = n;
  while (last - first > 3) {                                   // uniform
    if (depth == 0) break;                                     // (heap select: left to one thread below)
    --depth;
    if (tid == 0) { gdb_median3_to_first(a, first, last); *s_m = 0u; }
    __syncthreads();
    const float pivot = a[first];
    const int64_t len = last - first - 1, per = (len + kHugeBlock - 1) / kHugeBlock;
    const int64_t b = first + 1 + min(len, (int64_t)tid * per), e = first + 1 + min(len, (int64_t)(tid + 1) * per);
    uint32_t cl = 0, cr = 0;
    for (int64_t i = b; i < e; ++i) { const float v = a[i]; cl += !(v < pivot); cr += !(pivot < v); }
    uint32_t nl, nr;
    uint32_t atl = huge_block_scan(cl, s_w, nl);
    const uint32_t fwd = huge_block_scan(cr, s_w, nr);
    uint32_t atr = nr - fwd - cr;                              // right stops in front of mine in the descending list: those of the segments behind
    for (int64_t i = b; i < e; ++i) if (!(a[i] < pivot)) posL[atl++] = (uint32_t)i;
    for (int64_t i = e - 1; i >= b; --i) if (!(pivot < a[i])) posR[atr++] = (uint32_t)i;
    __syncthreads();
    const uint32_t both = min(nl, nr);
    uint32_t mine = 0;
    for (uint32_t k = tid; k < both; k += kHugeBlock) mine += posL[k] < posR[k];
    if (mine) atomicAdd(s_m, mine);
    __syncthreads();
    const uint32_t m = *s_m;                                   // (posL ascends, posR descends: the pairs that swap are the first m)
    for (uint32_t k = tid; k < m; k += kHugeBlock) { const uint32_t i = posL[k], j = posR[k]; const float x = a[i]; a[i] = a[j]; a[j] = x; }
    const int64_t cut = (m < nl && (m == 0 || posL[m] < posR[m - 1])) ? (int64_t)posL[m] : (int64_t)posR[m - 1];
    __syncthreads();
    if (cut <= nth) first = cut; else last = cut;
  }
  if (tid == 0) *s_res = __float_as_uint(gdb_introselect_libstdcxx(a + first, last - first, nth - first, last - first > 3 ? 0 : 1));
  __syncthreads();
  const uint32_t r = *s_res;
  __syncthreads();
  return r;
}
__global__ void __launch_bounds__(kHugeBlock) k_site_huge(const SiteCtx* __restrict__ sxp, const int32_t* __restrict__ huge_list, const int32_t* __restrict__ counter,
                                                      HugeSiteOut* __restrict__ out, uint32_t* err) {
  const SiteCtx& cx = *sxp;
  const CombinePlan& pl = cx.pl;
  __shared__ uint32_t tab_hash[kHugeTable], tab_order[kHugeTable];
  __shared__ int16_t tab_midx[kHugeTable];
  __shared__ unsigned long long s_key, s_cnt[4], s_filter;
  __shared__ int32_t s_flags[4];              // [0] table overflow, [1] fallback, [2] filter multi, [3] any id
  __shared__ int32_t s_nocc, s_isum;
  __shared__ const char* s_mref;
  __shared__ int32_t s_mref_len;
  __shared__ uint32_t s_hist[256], s_prefix, s_mask, s_first_t;
  __shared__ long long s_rank;
  __shared__ uint32_t s_stage[2048];
  __shared__ uint32_t s_scan_w[kHugeBlock / 64 + 1], s_tie_m, s_tie_res;
  __shared__ unsigned long long s_tie_at;
  const int tid = threadIdx.x;
  const int count = min(*counter, kMaxHugeRecords);
  uint32_t e = 0;
  for (int bi = blockIdx.x; bi < count; bi += gridDim.x) {      // uniform
    const int64_t k = huge_list[bi];
    const int64_t hb = cx.hl.base[k], he = cx.hl.base[k + 1];
    const int64_t s_k = cx.rec.start[k];
    HugeSiteOut& o = out[bi];
    // ---- merged REF: longest REF among the calls starting here, the first one at equal length ---------------------------------
    if (tid == 0) {
      int len0 = 0;
      s_mref = site_first_ref(cx, s_k, len0);
      s_mref_len = len0;
      s_key = ((unsigned long long)(uint32_t)len0 << 32) | 0xFFFFFFFFull;
      s_flags[0] = s_flags[1] = s_flags[2] = s_flags[3] = 0;
      s_nocc = 0;
      s_filter = ~0ull;
    }
    for (int s = tid; s < kHugeTable; s += kHugeBlock) { tab_hash[s] = kHugeEmpty; tab_order[s] = 0xFFFFFFFFu; tab_midx[s] = 0; }
    __syncthreads();
    for (int64_t t = hb + tid; t < he; t += kHugeBlock) {
      const int64_t c = cx.hl.cell[t];
      if (cx.fr.begin[c] != s_k || !field_valid(cx.cm, c, pl.f_REF)) continue;
      int n;
      cell_field<char>(cx.fr, pl, pl.f_REF, c, n);
      atomicMax(&s_key, ((unsigned long long)(uint32_t)n << 32) | (unsigned long long)(0xFFFFFFFEu - (uint32_t)(t - hb)));
    }
    __syncthreads();
    if (tid == 0 && (uint32_t)s_key != 0xFFFFFFFFu) {
      const int64_t t = hb + (int64_t)(0xFFFFFFFEu - (uint32_t)s_key);
      int n;
      s_mref = cell_field<char>(cx.fr, pl, pl.f_REF, cx.hl.cell[t], n);
      s_mref_len = n;
    }
    __syncthreads();
    const char* mref = s_mref;
    const int mref_len = s_mref_len;
    HugeTable tb{tab_hash, tab_order, tab_midx, &s_flags[0]};
    // ---- pass A: candidates -> table ---------------------------------------------------------------------------------------------
    for (int64_t t = hb + tid; t < he; t += kHugeBlock) {
      HugeInsert fa{tb, (uint32_t)(t - hb) << 6, 0u};
      site_merge_call(cx, t, s_k, mref, mref_len, false, fa, &e);
      if (fa.i > 63u) s_flags[1] = 1;                          // more tokens than the position key holds: serial walk
    }
    __syncthreads();
    // a record that goes back to the serial walk (table overflow, a call with more tokens than the position key holds) takes no
    // further part: its table is not trustworthy, the passes below would read merged[] entries nobody wrote (uniform per record)
    const bool skip_a = (s_flags[0] | s_flags[1]) != 0;
    // ---- merged order = ascending first position; representatives -----------------------------------------------------------------
    for (int s = tid; s < kHugeTable && !skip_a; s += kHugeBlock) {
      if (tab_hash[s] == kHugeEmpty) continue;
      const uint32_t mine = tab_order[s];
      int rank = 0;
      for (int s2 = 0; s2 < kHugeTable; ++s2) rank += (tab_hash[s2] != kHugeEmpty && tab_order[s2] < mine) ? 1 : 0;
      atomicAdd(&s_nocc, 1);
      if (rank + 1 >= GDB_MAX_MERGED_ALLELES - 1) { e |= GDB_ERR_TOO_MANY_MERGED_ALLELES; tab_midx[s] = 1; continue; }
      tab_midx[s] = (int16_t)(rank + 1);
      HugePick pk{mine & 63u, 0u, AlleleRef{nullptr, 0, -1, 0u}, false};
      uint32_t e2 = 0;
      site_merge_call(cx, hb + (int64_t)(mine >> 6), s_k, mref, mref_len, false, pk, &e2);
      if (pk.found) o.merged[rank + 1] = pk.out; else s_flags[1] = 1;
    }
    __syncthreads();
    const int nmerged = min(1 + s_nocc, GDB_MAX_MERGED_ALLELES - 1);
    const int num_merged = nmerged + (cx.pc.nr_cnt[k] > 0 ? 1 : 0);
    const bool skip_b = (s_flags[0] | s_flags[1]) != 0;        // (read behind the barrier: uniform)
    // ---- pass C: LUTs, flags, min-PL genotypes; <NON_REF> last -----------------------------------------------------------------------
    for (int64_t t = hb + tid; t < he && !skip_b; t += kHugeBlock) {
      HugeLookup fc{tb, o.merged, mref, mref_len, &s_flags[1]};
      site_merge_call(cx, t, s_k, mref, mref_len, true, fc, &e);
      const int64_t c = cx.hl.cell[t];
      const uint32_t cf = cx.cm.cflags[c];
      if (cx.pc.nr_cnt[k] > 0 && (cf & GDB_CF_HAS_NR)) {
        int alt_len;
        const char* alt = cell_field<char>(cx.fr, pl, pl.f_ALT, c, alt_len);
        const char* tok; int tl;
        for (int i = 0; alt_token(alt, alt_len, i, tok, tl); ++i)
          if (tl > 0 && tok[0] == '&' && i + 1 <= (int)GDB_CF_NALT(cf)) cx.hl.i2m[cx.hl.i2m_off[t] + i + 1] = (int8_t)(num_merged - 1);
      }
    }
    // ---- FILTER union, ID presence ----------------------------------------------------------------------------------------------------
    const bool do_filter = pl.produce_FILTER_field && pl.f_FILTER >= 0;
    if (do_filter || pl.f_ID >= 0)
      for (int64_t t = hb + tid; t < he; t += kHugeBlock) {
        const int64_t c = cx.hl.cell[t];
        if (pl.f_ID >= 0 && field_valid(cx.cm, c, pl.f_ID)) s_flags[3] = 1;
        if (do_filter && field_valid(cx.cm, c, pl.f_FILTER)) {
          int n;
          const int32_t* p = cell_field<int32_t>(cx.fr, pl, pl.f_FILTER, c, n);
          if (n > 0) atomicMin(&s_filter, ((unsigned long long)(uint32_t)(t - hb) << 32) | (unsigned long long)(uint32_t)p[0]);
        }
      }
    __syncthreads();
    if (do_filter && s_filter != ~0ull) {
      const int32_t first_id = (int32_t)(uint32_t)s_filter;
      for (int64_t t = hb + tid; t < he; t += kHugeBlock) {
        const int64_t c = cx.hl.cell[t];
        if (!field_valid(cx.cm, c, pl.f_FILTER)) continue;
        int n;
        const int32_t* p = cell_field<int32_t>(cx.fr, pl, pl.f_FILTER, c, n);
        for (int i = 0; i < n; ++i) if (p[i] != first_id) s_flags[2] = 1;
      }
    }
    __syncthreads();
    if (tid == 0) {
      o.mref = mref; o.mref_len = mref_len; o.nmerged = nmerged;
      o.filter_first_id = (do_filter && s_filter != ~0ull) ? (int32_t)(uint32_t)s_filter : -1;
      o.filter_multi = s_flags[2]; o.any_id = s_flags[3];
      o.fallback = (s_flags[0] || s_flags[1] || s_flags[2]) ? 1 : 0;   // several different FILTER ids: their order needs the calls in sequence (filter_union_order)
    }
    // ---- scalar reducers over the gathered per-call values (ScalarPre) ---------------------------------------------------------------
    if (cx.pre.enabled)
      for (int f = 0; f < pl.nfields; ++f) {                    // uniform
        const int slot = cx.pre.slot[f];
        if (slot < 0) continue;
        const bool is_float = pl.field[f].elem == GDB_ET_FLOAT;
        const int op = (f == pl.f_QUAL) ? pl.qual_combine_op : pl.field[f].combine_op;
        const uint32_t missing = is_float ? GDB_BCF_FLOAT_MISSING_BITS : (uint32_t)GDB_BCF_INT32_MISSING;
        const uint32_t* val = cx.pre.val + (int64_t)slot * cx.pre.stride;
        if (tid < 4) s_cnt[tid] = 0;
        if (tid == 0) { s_isum = 0; s_first_t = 0xFFFFFFFFu; }
        __syncthreads();
        unsigned long long nvalid = 0, nbelow = 0, nneg0 = 0, npos0 = 0;
        int32_t isum = 0;
        for (int64_t t = hb + tid; t < he; t += kHugeBlock) {
          const uint32_t u = val[t];
          if (u == missing) continue;
          ++nvalid;
          if (is_float) { nneg0 += u == 0x80000000u; npos0 += u == 0u; nbelow += __uint_as_float(u) < 0.0f; }
          else isum += (int32_t)u;
        }
        if (nvalid) atomicAdd(&s_cnt[0], nvalid);
        if (nbelow) atomicAdd(&s_cnt[1], nbelow);
        if (nneg0) atomicAdd(&s_cnt[2], nneg0);
        if (npos0) atomicAdd(&s_cnt[3], npos0);
        if (isum) atomicAdd(&s_isum, isum);
        __syncthreads();
        const long long total_valid = (long long)s_cnt[0];
        uint32_t sum_bits = (uint32_t)s_isum;
        if (is_float && (op == GDB_OP_SUM || op == GDB_OP_MEAN)) {
          // float sum in call order: chunks staged in LDS by everybody, added up by one thread
          float fsum = 0.0f;
          for (int64_t t0 = hb; t0 < he; t0 += 2048) {             // uniform
            const int64_t m = min((int64_t)2048, he - t0);
            for (int64_t j = tid; j < m; j += kHugeBlock) s_stage[j] = val[t0 + j];
            __syncthreads();
            if (tid == 0) for (int64_t j = 0; j < m; ++j) { const uint32_t u = s_stage[j]; if (u != missing) fsum += __uint_as_float(u); }
            __syncthreads();
          }
          sum_bits = __float_as_uint(fsum);
        }
        uint32_t median_bits = 0;
        int32_t median_ok = 0;
        if (op == GDB_OP_MEDIAN && total_valid > 0) {
          // radix select of the element of rank total_valid / 2 (std::nth_element at the mid point), 8 bits per pass
          if (tid == 0) { s_prefix = 0; s_mask = 0; s_rank = total_valid / 2; }
          __syncthreads();
          for (int pass = 0; pass < 4; ++pass) {                  // uniform
            const int shift = 24 - 8 * pass;
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix, mask = s_mask;
            for (int64_t t = hb + tid; t < he; t += kHugeBlock) {
              const uint32_t u = val[t];
              if (u == missing) continue;
              const uint32_t key = is_float ? gdb_orderable_bits(__uint_as_float(u)) : gdb_orderable_bits((int32_t)u);
              if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
              long long r = s_rank;
              uint32_t b = 0;
              for (; b < 255u && r >= (long long)s_hist[b]; ++b) r -= s_hist[b];
              s_rank = r; s_prefix = prefix | (b << shift); s_mask = mask | (255u << shift);
            }
            __syncthreads();
          }
          const uint32_t want = s_prefix;
          for (int64_t t = hb + tid; t < he; t += kHugeBlock) {       // among equal values the first call supplies the bits
            const uint32_t u = val[t];
            if (u == missing) continue;
            const uint32_t key = is_float ? gdb_orderable_bits(__uint_as_float(u)) : gdb_orderable_bits((int32_t)u);
            if (key == want) atomicMin(&s_first_t, (uint32_t)(t - hb));
          }
          __syncthreads();
          if (s_first_t != 0xFFFFFFFFu) {
            median_bits = val[hb + s_first_t];
            const bool zero = is_float && (median_bits << 1) == 0u;
            median_ok = (zero && s_cnt[2] && s_cnt[3]) ? 2 : 1;   // zeros of both signs: the reference's nth_element decides
          }
          if (median_ok == 2 && cx.tie.buf) {                   // uniform: ... and this workgroup runs it (else: reduce_scalar, one thread)
            if (tid == 0) s_tie_at = atomicAdd(cx.tie.used, 3ull * (unsigned long long)total_valid);
            __syncthreads();
            const unsigned long long at = s_tie_at;
            if (at + 3ull * (unsigned long long)total_valid <= cx.tie.capacity) {
              float* a = cx.tie.buf + at;
              median_bits = huge_tie_median(val, hb, he, missing, a, (uint32_t*)(a + total_valid), (uint32_t*)(a + 2 * total_valid), s_scan_w, &s_tie_m, &s_tie_res);
              median_ok = 1;
            }
            __syncthreads();
          }
        }
        if (tid == 0) {
          HugeScalar& hv = o.scalar[slot];
          hv.nvalid = total_valid; hv.nbelow = (long long)s_cnt[1]; hv.nneg0 = (long long)s_cnt[2]; hv.npos0 = (long long)s_cnt[3];
          hv.sum_bits = sum_bits; hv.median_bits = median_bits; hv.median_ok = median_ok; hv.pad = 0;
        }
        __syncthreads();
      }
    __syncthreads();
  }
  if (e) atomicOr(err, e);
}

// ---- site kernels: one thread per record -----------------------------------------------------------------------
// Pass 0 runs the record logic ONCE: allele merge, LUTs, per-record flags - and the text of the fixed columns, formatted
// through a capped LDS sink into a lane-private strip and parked in a fixed-stride staging slot.  The page pass then only
// copies the parked text (records whose fixed columns exceed the slot - long allele lists - are formatted again).
constexpr int kSiteStride = 256;                   // distance of two records' staging slots
// bytes of the fixed columns that are parked in the lane's LDS strip (the rest goes to the record's tail slot in global memory, word by
// word: LdsSpillSink).  The site pass is a latency-bound kernel of one wavefront per workgroup, and the strip is what bounds its
// residency: 256 bytes per lane = 9 wavefronts per CU, 192 = 12 (the register limit at 149), 128 and the registers capped at 128
// (amdgpu_waves_per_eu(4)) = 16.  c2, site phase, A/B inside one call: 2.79 (256) / 2.48 (192) / 2.53 (128) / 2.37 ms (128, capped).  Until
// round 4 anything below ~190 bytes was a cliff (160 bytes: 9.1 ms) - c2's fixed columns are ~170 bytes at variant sites, and the tails
// went byte by byte through a pool of 32 768 chunks handed out by one atomic counter (exhausted after a tenth of the spilling records,
// the rest formatted a second time by the page pass).  With a tail slot per record there is no counter and no second formatting.
#ifndef GDBAMD_SITE_CAP
#define GDBAMD_SITE_CAP 128
#endif
constexpr int kSiteCap = GDBAMD_SITE_CAP;
constexpr int kSiteStripWords = kSiteCap / 4 + 1;
// The lanes of a wavefront wait for the record with the most variant calls among their 64: records are dealt out in the order of
// their call counts (`order`, a radix sort over 8-bit keys), so that a wavefront's records cost about the same.
__global__ void k_site_order_keys(const int64_t* __restrict__ hbase, int64_t P, uint32_t* __restrict__ key, int32_t* __restrict__ val) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const int64_t n = hbase[k + 1] - hbase[k];
  key[k] = (uint32_t)(n > 255 ? 255 : n);
  val[k] = (int32_t)k;
}
#ifndef GDBAMD_SITE_ATTR
#define GDBAMD_SITE_ATTR __attribute__((amdgpu_waves_per_eu(4)))
#endif
__global__ void __launch_bounds__(64) GDBAMD_SITE_ATTR k_site_size(const SiteCtx* __restrict__ sxp, const int32_t* __restrict__ order, char* __restrict__ staging, SpillPool spill, int32_t* __restrict__ spill_chunk, uint32_t* err) {
  const SiteCtx& sx = *sxp;
  __shared__ uint32_t strip[64 * kSiteStripWords];
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= sx.rec.npos) return;
  if (order) k = order[k];
  uint32_t e = 0;
  uint32_t* mine = strip + threadIdx.x * kSiteStripWords;
  LdsSpillSink cs((gdb_lds_char*)mine, (uint32_t)kSiteCap, spill, k);
  site_emit(sx, k, cs, true, &e);
  cs.flush();
  sx.so.prefix_len[k] = cs.n;
  // longer texts: the tail lies in a chunk of the spill pool (or, when that did not work out, the page pass formats the record again)
  spill_chunk[k] = (cs.n > (uint32_t)kSiteCap && cs.complete()) ? 0 : -1;    // >= 0: the tail lies in the record's tail slot
  const uint32_t nstaged = min(cs.n, (uint32_t)kSiteCap);
  uint4* dst = reinterpret_cast<uint4*>(staging + (size_t)k * kSiteStride);
  for (uint32_t q = 0; (q << 4) < nstaged; ++q) dst[q] = make_uint4(mine[4 * q], mine[4 * q + 1], mine[4 * q + 2], mine[4 * q + 3]);
  if (e) atomicOr(err, e);
}
// The parked fixed columns of a record (<= 256 bytes, 256-byte aligned staging slot) -> their place in the page, one wavefront
// per record: lane j assembles the j-th ALIGNED destination word from its own source word and its left neighbour's
// (v_alignbyte), so the record leaves as one coalesced store; only the two edge words go out bytewise.  Also writes the
// newline of every record.
__global__ void __launch_bounds__(256) k_site_copy(const uint32_t* __restrict__ prefix_len, const char* __restrict__ staging, const char* __restrict__ spill_buf,
                                                   const int32_t* __restrict__ spill_chunk, int64_t k_begin, int64_t k_end,
                                                   const uint64_t* __restrict__ chunk_off, int nchunks, uint64_t page_base, char* __restrict__ arena) {
  const int64_t k = k_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63u;
  if (k >= k_end) return;
  const uint32_t n_all = prefix_len[k];
  char* rec = arena + (chunk_off[k * nchunks] - page_base);
  if (lane == 0) arena[chunk_off[(k + 1) * nchunks] - page_base - 1] = '\n';
  uint32_t n = n_all;
  if (n_all > (uint32_t)kSiteCap) {             // a longer text: its head is parked like any other, its tail lies in a chunk of the spill pool
    const int32_t sc = spill_chunk[k];
    if (sc < 0) return;                         // (no chunk: k_site_write formats the record again)
    const char* tail = spill_buf + (size_t)k * kSpillTail;
    for (uint32_t i = lane; i < n_all - (uint32_t)kSiteCap; i += 64u) rec[(uint32_t)kSiteCap + i] = tail[i];
    n = (uint32_t)kSiteCap;
  }
  const uint32_t a = (uint32_t)((uintptr_t)rec & 3u);
  const uint32_t cur = reinterpret_cast<const uint32_t*>(staging + (size_t)k * kSiteStride)[lane];
  uint32_t prev = (uint32_t)__shfl_up((int)cur, 1);
  if (lane == 0) prev = 0;
  const uint32_t w = a ? __builtin_amdgcn_alignbyte(cur, prev, 4u - a) : cur;      // destination word `lane` = record bytes [4 lane - a, 4 lane - a + 4)
  const int32_t r0 = (int32_t)(4u * lane) - (int32_t)a;
  if (r0 >= 0 && r0 + 3 < (int32_t)n) {
    *reinterpret_cast<uint32_t*>(rec + r0) = w;
  } else {
#pragma unroll
    for (int b = 0; b < 4; ++b) if (r0 + b >= 0 && r0 + b < (int32_t)n) rec[r0 + b] = (char)((w >> (8 * b)) & 0xFFu);
  }
  if (lane == 63 && a && 256 - (int32_t)a < (int32_t)n)                              // the 65th destination word of a full, misaligned record
    for (int32_t r = 256 - (int32_t)a; r < (int32_t)n; ++r) rec[r] = (char)((cur >> (8 * (r & 3))) & 0xFFu);
}
__global__ void k_site_write(const SiteCtx* __restrict__ sxp, const char* __restrict__ staging, const char* __restrict__ spill_buf, const int32_t* __restrict__ spill_chunk, int64_t k_begin, int64_t k_end, const uint64_t* chunk_off, int nchunks,
                             uint64_t page_base, char* arena, uint32_t* err) {
  const SiteCtx& sx = *sxp;
  int64_t k = k_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= k_end) return;
  uint32_t e = 0;
  char* dst = arena + (chunk_off[k * nchunks] - page_base);
  const uint32_t n = sx.so.prefix_len[k];
  if (n <= (uint32_t)kSiteCap) {
    return;                                     // parked whole in its staging slot: k_site_copy has written it (and the '\n')
  } else if (spill_chunk[k] >= 0) {
    return;                                     // head parked, tail in the spill pool: k_site_copy has put both in place
  } else {
    ByteSink bs(dst);
    site_emit(sx, k, bs, false, &e);
  }
  arena[chunk_off[(k + 1) * nchunks] - page_base - 1] = '\n';
  if (e) atomicOr(err, e);
}

// ---- sample columns -------------------------------------------------------------------------------------------------
// Per-interval context in constant memory: plan, column pointers, per-cell metadata pointers.  Uniform accesses to it are
// scalar loads through the scalar cache (a by-reference argument would turn each of them into a vector memory instruction).
// One slot per pipeline (leased for the pipeline's lifetime): the handles of one process no longer take turns on a single symbol.
constexpr int kCtxSlots = GDB_MAX_PIPELINES_PER_PROCESS;
__constant__ EntryCtx c_ex[kCtxSlots];
static std::mutex g_ctx_slot_mutex;
static bool g_ctx_slot_used[kCtxSlots];

// the instantiations are kept out of line: one copy each instead of one per call site
__device__ __noinline__ void entry_store(int cs, RecordInfo ri, int64_t c, char* dst, uint32_t* e) {
  (void)entry_emit(c_ex[cs], ri, c, ByteSink(dst), e);
}
// length of the text; its first `cap` bytes are left at dst (LDS)
__device__ __noinline__ uint32_t entry_store_lds_capped(int cs, RecordInfo ri, int64_t c, gdb_lds_char* dst, uint32_t cap, uint32_t* e) {
  return entry_emit(c_ex[cs], ri, c, LdsCapSink(dst, cap), e).n;
}

// binary (BCF) flavour of the same two wrappers
__device__ __noinline__ void entry_bin_store(int cs, RecordInfo ri, int64_t c, char* dst, uint32_t* e) {
  (void)entry_emit_bin(c_ex[cs], ri, c, ByteSink(dst), e);
}
__device__ __noinline__ uint32_t entry_bin_lds_capped(int cs, RecordInfo ri, int64_t c, gdb_lds_char* dst, uint32_t cap, uint32_t* e) {
  return entry_emit_bin(c_ex[cs], ri, c, LdsCapSink(dst, cap), e).n;
}

// ---- entry text table ---------------------------------------------------------------------------------------------
// The sample columns are >99 % of the output bytes, and almost all of them repeat: a reference-block call is live in ~100
// consecutive records and its text depends only on (cell, record type) with type = (FORMAT mask, remap flags, #merged
// alleles).  So the field emitters run once per distinct (cell, type) pair, once per (record, heavy call) incidence and once
// per type for the no-call column, into a text pool in HBM ("slots", 16-byte aligned); the P x N assembly pass that follows
// is a pure gather-copy: slot id -> (offset, length) -> LDS -> page.
constexpr int kTypeHash = 4096;           // open-addressing table of record type keys
constexpr int kMaxTypes = 64;             // types with a slot bitmask position; records of further types get one slot per sample
constexpr uint32_t kUntabledType = 255u;
constexpr uint64_t kEmptyKey = ~0ull;

__device__ __forceinline__ uint64_t record_type_key(uint32_t fmt_mask, uint32_t num_merged, uint32_t rflags) {
  return (uint64_t)fmt_mask | ((uint64_t)num_merged << 32) | ((uint64_t)rflags << 40);
}
__device__ __forceinline__ uint32_t type_hash(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 29;
  return (uint32_t)k & (kTypeHash - 1);
}
__device__ __forceinline__ void type_insert_global(uint64_t key, int32_t k, unsigned long long* hkeys, int32_t* hrep) {
  uint32_t h = type_hash(key);
  for (int probe = 0; probe < kTypeHash; ++probe, h = (h + 1) & (kTypeHash - 1)) {
    unsigned long long cur = __atomic_load_n(&hkeys[h], __ATOMIC_RELAXED);
    if (cur == kEmptyKey) cur = atomicCAS(&hkeys[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
    if (cur == kEmptyKey || cur == key) {   // representative record: the smallest index
      if (__atomic_load_n(&hrep[h], __ATOMIC_RELAXED) > k) atomicMin(&hrep[h], k);
      return;
    }
  }
}
// A workgroup first folds its 256 records into a small LDS table (a window has ~10 distinct types), then only the
// occupied LDS buckets go to the global table: a few thousand global atomics per interval instead of one per record.
__global__ void __launch_bounds__(kBlock) k_type_insert(SiteOut so, int64_t P, unsigned long long* hkeys, int32_t* hrep) {
  constexpr int kLocal = 128;
  __shared__ unsigned long long lkey[kLocal];
  __shared__ int32_t lrep[kLocal];
  if (threadIdx.x < kLocal) { lkey[threadIdx.x] = kEmptyKey; lrep[threadIdx.x] = INT32_MAX; }
  __syncthreads();
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < P) {
    const uint64_t key = record_type_key(so.fmt_mask[k], so.num_alleles[k], so.rflags[k]);
    uint32_t h = type_hash(key) & (kLocal - 1);
    bool placed = false;
    for (int probe = 0; probe < kLocal && !placed; ++probe, h = (h + 1) & (kLocal - 1)) {
      unsigned long long cur = lkey[h];
      if (cur == kEmptyKey) cur = atomicCAS(&lkey[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
      if (cur == kEmptyKey || cur == key) { atomicMin(&lrep[h], (int32_t)k); placed = true; }
    }
    if (!placed) type_insert_global(key, (int32_t)k, hkeys, hrep);   // more than 128 types inside one workgroup
  }
  __syncthreads();
  if (threadIdx.x < kLocal && lkey[threadIdx.x] != kEmptyKey) type_insert_global(lkey[threadIdx.x], lrep[threadIdx.x], hkeys, hrep);
}
// dense ids for the first max_types occupied buckets (bucket order); later ones are "untabled".  One wavefront: every lane
// counts its kTypeHash/64 consecutive buckets, a wavefront scan gives its first id.
__global__ void k_type_assign(const unsigned long long* hkeys, const int32_t* hrep, int max_types, uint8_t* hid, int32_t* type_rep, int32_t* ntypes) {
  if (blockIdx.x) return;
  constexpr int per = kTypeHash / 64;
  const int lane = threadIdx.x;
  uint32_t cnt = 0;
  for (int j = 0; j < per; ++j) cnt += hkeys[lane * per + j] != kEmptyKey;
  const uint32_t incl = wave_inclusive_scan_dpp(cnt);
  int n = (int)(incl - cnt);
  for (int j = 0; j < per; ++j) {
    const int h = lane * per + j;
    if (hkeys[h] == kEmptyKey) continue;
    if (n < max_types) { hid[h] = (uint8_t)n; type_rep[n] = hrep[h]; } else hid[h] = (uint8_t)kUntabledType;
    ++n;
  }
  if (lane == 63) *ntypes = min((int)incl, max_types);
}
__global__ void k_type_lookup(SiteOut so, int64_t P, const unsigned long long* hkeys, const uint8_t* hid, uint8_t* rtype, uint32_t* untabled) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const uint64_t key = record_type_key(so.fmt_mask[k], so.num_alleles[k], so.rflags[k]);
  uint32_t h = type_hash(key);
  uint8_t t = (uint8_t)kUntabledType;
  for (int probe = 0; probe < kTypeHash; ++probe, h = (h + 1) & (kTypeHash - 1)) {
    const unsigned long long cur = hkeys[h];
    if (cur == key) { t = hid[h]; break; }
    if (cur == kEmptyKey) break;
  }
  rtype[k] = t;
  untabled[k] = t == (uint8_t)kUntabledType ? 1u : 0u;
}
__global__ void k_untabled_records(const uint32_t* untabled, const uint32_t* ubase, int64_t P, int32_t* urec) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < P && untabled[k]) urec[ubase[k]] = (int32_t)k;
}
// which record types does every plain (non-heavy) cell of the window meet?
// Which record types does every plain (non-heavy) cell of the window meet?  A cell is live in ~100 records, so instead of
// reading their types one by one the kernel asks per type "is there a record of this type in [k_lo, k_hi]?" of a prefix
// count table: occ[t][k] = #records of type t before k (one exclusive scan over the 64 x (P+1) one-hot matrix; differences
// inside a row do not care about the row's starting value).
__global__ void k_type_onehot(const uint8_t* rtype, int64_t P, uint32_t* occ) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const uint32_t t = rtype[k];
  if (t != kUntabledType) occ[(int64_t)t * (P + 1) + k] = 1u;
}
__global__ void k_cell_types(const uint32_t* cflags, const int32_t* k_lo, const int32_t* k_hi, const uint32_t* occ, int64_t P, const int32_t* ntypes_p,
                             int64_t c_base, int64_t n, uint64_t* tmask, uint32_t* nslots) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = c_base + i;
  uint64_t m = 0;
  if (!(cflags[c] & GDB_CF_HEAVY) && k_lo[c] >= 0) {
    const int64_t lo = k_lo[c], hi1 = (int64_t)k_hi[c] + 1;
    const int ntypes = *ntypes_p;
    for (int t = 0; t < ntypes; ++t) {
      const uint32_t* row = occ + (int64_t)t * (P + 1);
      if (row[hi1] != row[lo]) m |= 1ull << t;
    }
  }
  tmask[i] = m;
  nslots[i] = (uint32_t)__popcll(m);
}
constexpr int kSlotBlock = 64;     // threads per workgroup of the slot kernels
constexpr int kSlotStride = 128;   // bytes of an inline slot: slot s lives at pool + s * kSlotStride; longer texts go to the overflow pool
constexpr int kStripWords = kSlotStride / 4 + 1;   // LDS words per lane: odd stride, no bank conflicts between lanes
constexpr uint32_t kOverflowBit = 0x80000000u;     // descriptor offsets with this bit are 16-byte units into the overflow pool
// Text mode: the last 8 bytes of an inline slot are its TAG {u32 where, u32 len}: where = 0 (the text is the slot's first len
// bytes, len <= kInlineText) or kOverflowBit | 16-byte unit of the overflow pool.  The matrix-free page assembly (k_write2) needs
// nothing but the slot number to fetch a text: the tag arrives with the slot's own cache line.  BCF entries keep all 128 bytes.
constexpr int kSlotTagAt = kSlotStride - 8;
constexpr int kText2Chunks = 8;                        // 16-byte chunks of a slot k_write2 keeps in registers (the last one holds the tag)
constexpr int kInlineText = kSlotTagAt;                // = 120
struct SlotTable {
  uint32_t* len;          // [S]  entry bytes incl. the leading tab; 0: the record has no FORMAT columns
  const uint32_t* ovf16;  // [S]  overflow-pool offset in 16-byte units (exclusive scan; only meaningful for len > kSlotStride)
  char* pool;             // inline texts, fixed stride
  char* pool_ovf;
  uint32_t light_base;    // = kMaxTypes: slots [0,kMaxTypes) are the no-call texts per type
  uint32_t heavy_base;
  uint32_t row_base;      // slots of (untabled record, sample): row_base + u * N + row
  uint32_t* ovf16_w;      // the same array, written by pass 0 for the texts it places itself (bump allocation, below)
  unsigned int* bump;     // per shard (kBumpShards x 4, workgroup -> shard by blockIdx: one hot counter would queue up ~10^6 atomics):
                          // [0] units taken from the shard's part of the overflow pool, [1] long texts not placed, [2] their units, [3] long texts in all
  uint32_t bump_cap;      // units of ONE shard's part of the overflow pool (0: pass 0 places nothing, every long text waits for pass 1)
  int32_t ctx;            // this pipeline's slot of c_ex
  int32_t bcf;            // entries are binary (gdb_bcf.hpp) instead of text
  uint32_t inline_max;    // longest entry that stays in its inline slot: kInlineText (text, tagged slots) or kSlotStride (BCF)
  // heavy slots are numbered in FILL order (cell by cell, record by record: hoff[cell] + k - k_lo[cell]), so that a walker names the
  // slot of a heavy call from its registers, without the fill-order -> (record, row)-order table of earlier rounds
  const int64_t* hoff; const int32_t* k_lo; int64_t c_base;
  unsigned int* over255;  // [kBumpShards] set by pass 0 when an entry is longer than 255 bytes (the compact resolved matrix keeps lengths in a byte)
};
// PASS 0: ONE run of the field emitters gives the length of the text and, when it fits kSlotStride bytes (nearly always),
// the text itself: formatted into a lane-private LDS strip, it leaves as 16-byte stores into the lane's inline slot.
// A text longer than an inline slot that still fits the strip (STRIPW words per lane: 128 or 256 bytes) is placed by pass 0 too:
// the wavefront adds up the 16-byte units its long texts need, takes them from the overflow pool with ONE atomic and every lane
// stores its text there - at the width of BASELINE configs[2] (10 000 samples, ~6 merged alleles per record) half of all texts
// are between 128 and 256 bytes, and formatting them a second time (pass 1) was 5 ms of a 33 ms window.
// PASS 1: only when pass 0 could not place every long text (longer than the strip, or the pool ran out): all texts longer than
// an inline slot are formatted again, straight into an overflow pool sized by a scan of their lengths.
// One template, four enumerations (types / plain cells x met types / heavy incidences / samples of untabled records).
constexpr int kStripWordsWide = 256 / 4 + 1;
constexpr uint32_t kSlotUnplaced = 0xFFFFFFFFu;   // ovf16 of a long text that has no place yet
constexpr int kBumpShards = 16;
// pass 1 formats: every long text (scan mode, bump_cap == 0: the places come from a scan over all lengths) or only the ones pass 0
// could not place (bump mode)
__device__ __forceinline__ bool slot_needs_pass1(const SlotTable& st, uint32_t s) {
  return st.len[s] > st.inline_max && (st.bump_cap == 0u || st.ovf16[s] == kSlotUnplaced);
}
template <int PASS, int STRIPW> __device__ __forceinline__ uint32_t slot_fill(const SlotTable& st, uint32_t s, const RecordInfo& rinfo, int64_t c, uint32_t* mine, uint32_t* e) {
  uint32_t len = 0;
  if (PASS == 0) {
    constexpr uint32_t cap = (uint32_t)(STRIPW - 1) * 4u;
    if (rinfo.fmt_mask) {
      gdb_lds_char* txt = (gdb_lds_char*)mine;
      if (st.bcf) len = entry_bin_lds_capped(st.ctx, rinfo, c, txt, cap, e);
      else {
      *txt = '\t';
      len = 1u + entry_store_lds_capped(st.ctx, rinfo, c, txt + 1, cap - 1u, e);
      }
      if (len <= st.inline_max) {
        uint4* dst = reinterpret_cast<uint4*>(st.pool + (size_t)s * kSlotStride);
        uint32_t nq = (len + 15u) >> 4;
        if (!st.bcf) {                                             // the tag leaves with the slot's last 16 bytes
          mine[kSlotTagAt / 4] = 0u; mine[kSlotTagAt / 4 + 1] = len;
          dst[kSlotStride / 16 - 1] = make_uint4(mine[kSlotStride / 4 - 4], mine[kSlotStride / 4 - 3], mine[kSlotStride / 4 - 2], mine[kSlotStride / 4 - 1]);
          if (nq > (uint32_t)(kSlotStride / 16 - 1)) nq = kSlotStride / 16 - 1;
        }
        for (uint32_t q = 0; q < nq; ++q) dst[q] = make_uint4(mine[4 * q], mine[4 * q + 1], mine[4 * q + 2], mine[4 * q + 3]);
      }
    }
    st.len[s] = len;
  } else if (slot_needs_pass1(st, s)) {
    uint32_t off = st.ovf16[s];
    if (st.bump_cap) {                                           // bump mode: the few texts pass 0 left take their place now
      const uint32_t sh = blockIdx.x & (kBumpShards - 1);        // (the same shard as in pass 0: the host has made room for them there)
      off = sh * st.bump_cap + atomicAdd(&st.bump[sh * 4], (st.len[s] + 15u) >> 4);
      st.ovf16_w[s] = off;
    }
    char* dst = st.pool_ovf + (size_t)off * 16;
    if (st.bcf) { entry_bin_store(st.ctx, rinfo, c, dst, e); return 0; }
    *dst = '\t';
    entry_store(st.ctx, rinfo, c, dst + 1, e);
  }
  return len;
}
// pass 0, behind slot_fill, by EVERY lane of the wavefront (active: the lane has a slot): place the long texts that fit the strip
template <int STRIPW> __device__ __forceinline__ void slot_place_long(const SlotTable& st, bool active, uint32_t s, uint32_t len, const uint32_t* mine) {
  constexpr uint32_t cap = (uint32_t)(STRIPW - 1) * 4u;
  const bool is_long = active && len > st.inline_max;
  if (!__any((int)is_long)) return;                             // uniform
  if (__any((int)(active && len > 255u)) && (threadIdx.x & 63) == 0) atomicOr(&st.over255[blockIdx.x & (kBumpShards - 1)], 1u);
  const bool fits = is_long && st.bump_cap != 0u && len <= cap;
  const uint32_t units = fits ? (len + 15u) >> 4 : 0u;
  const uint32_t incl = wave_inclusive_scan_dpp(units);
  const uint32_t total = wave_total(incl);
  unsigned int* const cnt = st.bump + (blockIdx.x & (kBumpShards - 1)) * 4;
  const uint32_t shard_base = (blockIdx.x & (kBumpShards - 1)) * st.bump_cap;
  uint32_t base = 0;
  if (total) {                                                   // uniform
    if ((threadIdx.x & 63) == 0) base = atomicAdd(&cnt[0], total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  }
  bool placed = false;
  if (fits) {
    const uint32_t off = base + incl - units;
    if (off + units <= st.bump_cap) {
      uint4* dst = reinterpret_cast<uint4*>(st.pool_ovf + (size_t)(shard_base + off) * 16);
      for (uint32_t q = 0; q < units; ++q) dst[q] = make_uint4(mine[4 * q], mine[4 * q + 1], mine[4 * q + 2], mine[4 * q + 3]);
      st.ovf16_w[s] = shard_base + off;
      placed = true;
    }
  }
  const bool left = is_long && !placed;
  if (left) st.ovf16_w[s] = kSlotUnplaced;
  const uint64_t longs = __ballot(is_long), failed = __ballot(left);
  const uint32_t lu = failed ? wave_total(wave_inclusive_scan_dpp(left ? (len + 15u) >> 4 : 0u)) : 0u;
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&cnt[3], (unsigned int)__popcll(longs));
    if (failed) { atomicAdd(&cnt[1], (unsigned int)__popcll(failed)); atomicAdd(&cnt[2], lu); }
  }
}
template <int PASS, int STRIPW> __global__ void k_slots_nocall(SlotTable st, SiteOut so, const int32_t* type_rep, int ntypes, uint32_t* err) {
  __shared__ uint32_t strip[kSlotBlock * STRIPW];
  uint32_t* mine = strip + threadIdx.x * STRIPW;
  const int t = threadIdx.x;
  if (blockIdx.x) return;
  uint32_t e = 0, len = 0;
  const bool live = t < ntypes && t < kMaxTypes;
  if (live) len = slot_fill<PASS, STRIPW>(st, (uint32_t)t, load_record_info(so, c_ex[st.ctx].hl, type_rep[t]), -1, mine, &e);
  else if (PASS == 0 && t < kMaxTypes) st.len[t] = 0;
  if (PASS == 0) slot_place_long<STRIPW>(st, live, (uint32_t)t, len, mine);
  if (e) atomicOr(err, e);
}
// Plain cells: one thread per (cell, type) slot, so every lane formats a text (a cell meets ~5 of the ~60 types of an
// interval: one thread per cell walking the types in a uniform order kept ~8 % of the lanes busy per round and measured
// 0.7 ms slower per 1 Mb window); the lanes of a wavefront work on different types.  slot_cell[s] = window index of the cell
// that owns light slot s.
__global__ void k_slot_cells(const uint32_t* tbase, const uint32_t* nslots, int64_t n, uint32_t* slot_cell) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t b = tbase[i], m = nslots[i];
  for (uint32_t q = 0; q < m; ++q) slot_cell[b + q] = (uint32_t)i;
}
// (cell, type) slots of plain cells.  Consecutive slots are the types of one cell, so a wavefront of 64 consecutive slots mixes cheap
// types (reference-only records: three genotypes) with dear ones (PL re-indexed to 6 - 10 genotypes, AD, SB) and every lane waits
// for the dearest: pass 0 regroups the 256 slots of a workgroup by type first (counting sort in LDS), which gives each of its four
// wavefronts one to three types.
#ifndef GDBAMD_LIGHT_BLOCK
#define GDBAMD_LIGHT_BLOCK 256
#endif
#ifndef GDBAMD_LIGHT_BLOCK_WIDE
#define GDBAMD_LIGHT_BLOCK_WIDE 192
#endif
constexpr int kLightBlock = GDBAMD_LIGHT_BLOCK;              // threads per workgroup with the narrow strip ...
constexpr int kLightBlockWide = GDBAMD_LIGHT_BLOCK_WIDE;     // ... and with the wide one: 192 threads = 51 KB of strips = three workgroups = 9 wavefronts per CU (256: two workgroups = 8); 10 000 samples,
                                                             // device-only 1.59-1.62 -> 1.64 M positions/s (profiles/r6_ab_micro_variants.txt; the same file: the sizing pass at 64 registers and the
                                                             // narrow-strip kernel at 192 threads x 5 wavefronts per SIMD change nothing / lose)
template <int STRIPW> constexpr int light_block() { return STRIPW <= kSlotStride / 4 + 1 ? kLightBlock : kLightBlockWide; }
// With the narrow strip the kernel holds 16 wavefronts per CU by LDS (35 KB per workgroup of four) but 133 registers allowed only 12:
// capped at 128 (amdgpu_waves_per_eu(4), no spills) the c2 sizing phase is 0.4 ms shorter (10.5 -> 10.1 ms, two A/B pairs on one box).
// The wide strip (67 KB: 8 wavefronts per CU whatever the registers) keeps its registers.
#ifndef GDBAMD_LIGHT_WAVES
#define GDBAMD_LIGHT_WAVES 4
#endif
template <int PASS, int STRIPW> __global__ void __launch_bounds__(light_block<STRIPW>()) __attribute__((amdgpu_waves_per_eu(STRIPW <= kStripWords ? GDBAMD_LIGHT_WAVES : 1))) k_slots_light(SlotTable st, SiteOut so, const int32_t* type_rep, const uint64_t* tmask, const uint32_t* tbase,
                                                      const uint32_t* slot_cell, int64_t c_base, int64_t SL, int regroup, uint32_t* err) {
  constexpr int kThreads = light_block<STRIPW>();
  __shared__ uint32_t strip[kThreads * STRIPW];
  __shared__ uint32_t tcount[kMaxTypes + 1], job[kThreads];
  uint32_t* mine = strip + threadIdx.x * STRIPW;
  int64_t sidx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool live = sidx < SL;
  if (PASS == 1) { if (!__any((int)(live && slot_needs_pass1(st, st.light_base + (uint32_t)(live ? sidx : 0))))) return; }
  uint32_t i = 0;
  int t = kMaxTypes;                                          // (slots behind the table's end sort last)
  if (live) {
    i = slot_cell[sidx];
    uint64_t m = tmask[i];
    for (uint32_t q = (uint32_t)sidx - tbase[i]; q; --q) m &= m - 1;     // the slot's rank among the cell's types -> its type
    t = __builtin_ctzll(m);
  }
  if (PASS == 0 && regroup) {                                 // uniform
    if (threadIdx.x <= (unsigned)kMaxTypes) tcount[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t r = atomicAdd(&tcount[t], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {
      const uint32_t v = tcount[threadIdx.x];
      const uint32_t incl = wave_inclusive_scan_dpp(v);
      tcount[threadIdx.x] = incl - v;
      if (threadIdx.x == 63) tcount[kMaxTypes] = incl;
    }
    __syncthreads();
    job[tcount[t] + r] = threadIdx.x | ((uint32_t)t << 16);
    __syncthreads();
    const uint32_t jb = job[threadIdx.x];
    sidx = (int64_t)blockIdx.x * blockDim.x + (jb & 0xFFFFu);
    t = (int)(jb >> 16);
    live = sidx < SL;
    if (live) i = slot_cell[sidx];
  }
  const uint32_t s = st.light_base + (uint32_t)(live ? sidx : 0);
  uint32_t e = 0, len = 0;
  if (live) len = slot_fill<PASS, STRIPW>(st, s, load_record_info(so, c_ex[st.ctx].hl, type_rep[t]), c_base + (int64_t)i, mine, &e);
  if (PASS == 0) slot_place_long<STRIPW>(st, live, s, len, mine);
  if (e) atomicOr(err, e);
}
template <int PASS, int STRIPW> __global__ void k_slots_heavy(SlotTable st, SiteOut so, const uint64_t* inc_keys_sorted, const int64_t* inc_cell, int64_t T, int64_t nrows,
                                                 uint32_t* err) {
  __shared__ uint32_t strip[kSlotBlock * STRIPW];
  uint32_t* mine = strip + threadIdx.x * STRIPW;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < T;
  uint32_t e = 0, len = 0;
  uint32_t s = st.heavy_base;
  if (live) {
    const int64_t k = (int64_t)(inc_keys_sorted[i] / (uint64_t)nrows);
    const int64_t c = inc_cell[i];
    s += (uint32_t)(st.hoff[c - st.c_base] + (k - (int64_t)st.k_lo[c]));
    len = slot_fill<PASS, STRIPW>(st, s, load_record_info(so, c_ex[st.ctx].hl, k), c, mine, &e);
  }
  if (PASS == 0) slot_place_long<STRIPW>(st, live, s, len, mine);
  if (e) atomicOr(err, e);
}
template <int PASS, int STRIPW> __global__ void k_slots_untabled(SlotTable st, SiteOut so, RowIndex ri, RecordTable rec, const int32_t* urec, int64_t U, int32_t N, uint32_t* err) {
  __shared__ uint32_t strip[kSlotBlock * STRIPW];
  uint32_t* mine = strip + threadIdx.x * STRIPW;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool in_range = i < U * (int64_t)N;
  uint32_t e = 0, len = 0;
  bool live = false;
  const uint32_t s = st.row_base + (uint32_t)(in_range ? i : 0);
  if (in_range) {
    const int64_t u = i / N;
    const int32_t row = (int32_t)(i - u * N);
    const int64_t k = urec[u];
    RowWalker w;
    w.init(ri, row, rec.start[k]);
    const int64_t c = w.live(ri, c_ex[st.ctx].cm, rec.start[k]);
    if (c >= 0 && (c_ex[st.ctx].cm.cflags[c] & GDB_CF_HEAVY)) { if (PASS == 0) st.len[s] = 0; }   // heavy calls have their incidence slot
    else { len = slot_fill<PASS, STRIPW>(st, s, load_record_info(so, c_ex[st.ctx].hl, k), c, mine, &e); live = true; }
  }
  if (PASS == 0) slot_place_long<STRIPW>(st, live, s, len, mine);
  if (e) atomicOr(err, e);
}
// the overflow pool has grown: shard sh's part moved from sh * old_su to sh * new_su (units); placed texts keep their offset inside the part
__global__ void k_slot_rebase(const uint32_t* len, uint32_t* ovf16, int64_t S, uint32_t old_su, uint32_t new_su, uint32_t inline_max) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S || len[s] <= inline_max) return;
  const uint32_t o = ovf16[s];
  if (o == kSlotUnplaced) return;
  const uint32_t sh = o / old_su;
  ovf16[s] = sh * new_su + (o - sh * old_su);
}
__global__ void k_slot_units(const uint32_t* len, int64_t S, uint32_t* units, uint32_t inline_max) {   // overflow-pool units of every slot
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < S) units[s] = len[s] > inline_max ? (len[s] + 15u) >> 4 : 0u;
}
// descriptors (off16, len) of all slots (desc != nullptr: the matrix kernels and the BCF path read them) and, in text mode
// (tag_pool != nullptr), the tags of the slots the formatting pass did not tag itself: empty entries and overflow texts, whose
// final place is only known here (after pass 1 / a grown pool)
__global__ void k_slot_desc(const uint32_t* len, const uint32_t* ovf16, int64_t S, uint2* desc, char* tag_pool, uint32_t inline_max) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const uint32_t l = len[s];
  const bool is_long = l > inline_max;
  if (desc) desc[s] = is_long ? make_uint2(kOverflowBit | ovf16[s], l) : make_uint2((uint32_t)s * (kSlotStride / 16), l);
  if (tag_pool && (is_long || l == 0u)) *reinterpret_cast<uint2*>(tag_pool + (size_t)s * kSlotStride + kSlotTagAt) = make_uint2(is_long ? (kOverflowBit | ovf16[s]) : 0u, l);
}

// ---- assembly ------------------------------------------------------------------------------------------------------
// Row-major walk list: what a lane needs when its sample moves to the next cell, in ONE 32-byte element.  eff_end / heavy are
// properties of the staged fragment; the rest is refreshed per interval for the cells of the window and stamped with the
// interval's epoch (a cell outside the window keeps the stamp of an earlier interval: it is dead for every record of this one).
struct __attribute__((aligned(16))) WalkCell {
  int64_t eff_end;
  uint64_t aux;     // plain cell: bitmask of the record types it has slots for; heavy cell: unused
  int32_t k_lo, k_hi;   // records the cell is live in (k_lo < 0: none)
  uint32_t base;    // plain cell: first slot (relative to light_base); heavy cell: first slot of its calls (relative to heavy_base; record k: base + k - k_lo)
  uint32_t flags;   // bit 0: heavy; bits 1..3: 16-byte chunks that hold the longest inline text of the cell's slots (<= 7; the tag sits in the 8th);
                    // bits 8..31: epoch of the interval that filled k_lo / k_hi / base / aux
};
constexpr uint32_t kWalkHeavy = 1u;
constexpr int kWalkChunksShift = 1;
constexpr uint32_t kWalkChunksMask = 7u;
// (an overflow text: only its tag is read from the slot; bytes 112 .. 119 of an inline text arrive with the tag's chunk)
__device__ __forceinline__ uint32_t inline_chunks(uint32_t len) { return len <= (uint32_t)kInlineText ? min((len + 15u) >> 4, kWalkChunksMask) : 0u; }
// Cell j of the row-major order (row r) sits at walk[j + r]: behind the cells of every row there is one SENTINEL element that never
// carries an interval's stamp - a walker that runs off its row's end reads "nothing further is live" instead of checking a bound.
__global__ void k_walk_static(const int64_t* perm, const int32_t* row, const int64_t* row_ptr, int32_t N, const int64_t* eff_end, const uint32_t* cflags, int64_t C, WalkCell* walk, int64_t* inv) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  WalkCell w;
  w.aux = 0; w.k_lo = -1; w.k_hi = -1; w.base = 0;
  if (j < C) {
    const int64_t c = perm[j];
    const int64_t at = j + row[c];
    w.eff_end = eff_end[c]; w.flags = (cflags[c] & GDB_CF_HEAVY) ? kWalkHeavy : 0u;
    walk[at] = w;
    inv[c] = at;
  } else if (j < C + N) {
    const int64_t r = j - C;
    w.eff_end = INT64_MIN; w.flags = 0;
    walk[row_ptr[r + 1] + r] = w;
  }
}
__global__ void k_walk_window(const int64_t* inv, const uint32_t* cflags, const int64_t* hoff, const int32_t* k_lo, const int32_t* k_hi, const uint32_t* tbase, const uint64_t* tmask,
                              const uint32_t* slot_len, uint32_t light_base, uint32_t heavy_base, int64_t c_base, int64_t n, uint32_t epoch, WalkCell* walk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = c_base + i;
  WalkCell* w = walk + inv[c];
  const bool heavy = (cflags[c] & GDB_CF_HEAVY) != 0;
  uint32_t chunks = 0;
  if (k_lo[c] >= 0) {
    const uint32_t first = heavy ? heavy_base + (uint32_t)hoff[i] : light_base + tbase[i];
    const uint32_t cnt = heavy ? (uint32_t)(k_hi[c] - k_lo[c] + 1) : (uint32_t)__popcll(tmask[i]);
    for (uint32_t q = 0; q < cnt; ++q) chunks = max(chunks, inline_chunks(slot_len[first + q]));
  }
  const uint4 tail = make_uint4((uint32_t)k_lo[c], (uint32_t)k_hi[c], heavy ? (uint32_t)hoff[i] : tbase[i], (epoch << 8) | (chunks << kWalkChunksShift) | (heavy ? kWalkHeavy : 0u));
  w->aux = heavy ? 0ull : tmask[i];
  *reinterpret_cast<uint4*>(&w->k_lo) = tail;
}

struct AsmCtx {
  const int64_t* row_ptr;   // [N+1] row-major ranges
  const int64_t* rm_begin;  // [C]   begin column in row-major order (binary search at the start of a run)
  const WalkCell* walk;     // [C]
  const int64_t* rec_start; // [P]
  const uint8_t* rtype;     // [P]
  const uint32_t* prefix_len; // [P] bytes of the fixed columns of a record (chunk 0 starts behind them)
  const uint2* desc;        // [S] (off16, len)
  const char* pool;
  const uint32_t* ubase;    // [P]  rank of a record among those with an untabled type
  uint32_t light_base, heavy_base, row_base;
  int32_t nrows;
};

// A lane follows one sample through increasing records.  The live cell and what is needed to name its slot stay in
// registers; memory is touched only when the sample moves to its next cell: one WalkCell plus the begin of the one behind.
struct SlotWalker {
  int64_t j, j_end, next_begin, cur_end;
  uint64_t aux;
  uint32_t base;
  int32_t k_lo, row_;
  bool heavy;
  __device__ __forceinline__ void load(const AsmCtx& a) {
    const WalkCell w = a.walk[j + row_];
    next_begin = (j + 1 < j_end) ? a.rm_begin[j + 1] : INT64_MAX;
    cur_end = w.eff_end; aux = w.aux; base = w.base; k_lo = w.k_lo; heavy = (w.flags & kWalkHeavy) != 0;
  }
  __device__ __forceinline__ void init(const AsmCtx& a, int32_t row, int64_t s0) {
    row_ = row;
    const int64_t j_begin = a.row_ptr[row];
    j_end = a.row_ptr[row + 1];
    int64_t lo = j_begin, hi = j_end;  // last j with rm_begin[j] <= s0
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.rm_begin[mid] <= s0) lo = mid + 1; else hi = mid; }
    j = lo - 1;
    if (j >= j_begin) load(a);
    else { cur_end = INT64_MIN; heavy = false; aux = 0; base = 0; k_lo = 0; next_begin = (j + 1 < j_end) ? a.rm_begin[j + 1] : INT64_MAX; }
  }
  __device__ __forceinline__ void advance(const AsmCtx& a, int64_t s) { while (next_begin <= s) { ++j; load(a); } }
  // (a cell that ended before the window keeps stale base/aux, but it is dead for every record of the window)
  __device__ __forceinline__ uint32_t slot(const AsmCtx& a, int64_t k, int64_t s, uint32_t t, int32_t row) const {
    const bool dead = s > cur_end;
    if (!dead && heavy) return a.heavy_base + base + (uint32_t)(k - (int64_t)k_lo);
    if (t == kUntabledType) return a.row_base + a.ubase[k] * (uint32_t)a.nrows + (uint32_t)row;
    if (dead) return t;
    return a.light_base + base + (uint32_t)__popcll(aux & ((1ull << t) - 1ull));
  }
};

// s_waitcnt vmcnt(0): behind loads issued under a per-lane condition.  The compiler cannot tell whether such a load is still in flight on
// the paths that skipped its use, and waits for it (vmcnt(0): for every store issued since, too) wherever one of its registers is written
// next - in k_bcf_write that was in the middle of every group pass and between the image's store sets.  Placed where nothing the
// kernel wants in flight is in flight.
__device__ __forceinline__ void settle_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }
// Copy n bytes of a 16-byte aligned pool slot to an arbitrary LDS byte address, executed by all lanes of a wavefront
// with per-lane n (0: idle lane).  Destination words are funnel-shifted out of consecutive source words (v_alignbyte).
// The last, partial word is stored whole: it spills at most 3 bytes past the entry, and every spilled byte is a HEAD byte
// (the bytes before the first word boundary) of one of the following entries.  The head bytes are therefore written by
// finish(), after the words of all lanes (LDS executes a wavefront's instructions in order), which repairs the spill.
// The stores are PREDICATED BY ADDRESS, not by branches: a word (or head byte) that is out of range goes to a per-lane scrap word
// behind the image.  As `if (m < nw) dw[m] = ...` every one of the ~20 stores of a record step compiled into v_cmp +
// s_and_saveexec + a taken branch + s_or: the wavefront spent more issue slots on exec-mask bookkeeping than on the copy.
struct SlotCopy {
  gdb_lds_char* dst;
  __attribute__((address_space(3))) uint32_t* dw;
  __attribute__((address_space(3))) uint32_t* scrap;
  uint32_t n, h, nw, carry, head_word;
  __device__ __forceinline__ void begin(gdb_lds_char* d, uint32_t len, gdb_lds_char* scrap_word) {
    dst = d; n = len;
    h = (4u - ((uint32_t)(uintptr_t)d & 3u)) & 3u;
    if (h > n) h = n;
    nw = (n - h + 3u) >> 2;          // destination words, the last one possibly partial
    dw = (__attribute__((address_space(3))) uint32_t*)(d + h);
    scrap = (__attribute__((address_space(3))) uint32_t*)scrap_word;
    carry = 0; head_word = 0;
  }
  __device__ __forceinline__ void put(uint32_t m, uint32_t v) const { *((m < nw) ? dw + m : scrap) = v; }   // (m - 1 with m == 0 wraps: out of range)
  // source chunk q (bytes [16q, 16q+16)) -> destination words 4q-1 .. 4q+2
  __device__ __forceinline__ void chunk(uint32_t q, const uint4& x) {
    if (q == 0) head_word = x.x;
    const uint32_t m1 = q << 2;      // word m takes source bytes [h+4m, h+4m+4): low part in source word m, high part in m+1
    put(m1 - 1u, __builtin_amdgcn_alignbyte(x.x, carry, h));
    put(m1, __builtin_amdgcn_alignbyte(x.y, x.x, h));
    put(m1 + 1u, __builtin_amdgcn_alignbyte(x.z, x.y, h));
    put(m1 + 2u, __builtin_amdgcn_alignbyte(x.w, x.z, h));
    carry = x.w;
  }
  __device__ __forceinline__ bool needs(uint32_t q) const { return (q << 2) <= nw && nw > 0; }   // q >= 1
  __device__ __forceinline__ void finish() const {
    gdb_lds_char* sc = (gdb_lds_char*)scrap;
    *((h > 0) ? dst : sc) = (char)(head_word & 0xFFu);
    *((h > 1) ? dst + 1 : sc) = (char)((head_word >> 8) & 0xFFu);
    *((h > 2) ? dst + 2 : sc) = (char)((head_word >> 16) & 0xFFu);
  }
};
constexpr int kScrapBytes = 64 * 4;   // one scrap word per lane behind the LDS image
__device__ __forceinline__ uint4 load_chunk(const char* __restrict__ src, uint32_t q, uint32_t n) {
  uint4 x = make_uint4(0, 0, 0, 0);
  if ((q << 4) < n) x = *reinterpret_cast<const uint4*>(src + (q << 4));
  return x;
}

// ---- assembly kernels: one wavefront = `run` records of one type x 64 samples, no workgroup barriers ---------------------
// Records are visited in (type, position) order (`order`): along such a run a sample keeps the same slot until its live
// cell changes.  k_assemble_size walks the samples (the only pass with dependent loads: walk list -> incidence -> slot
// descriptor), adds up the chunk sizes and leaves the resolved (pool offset, length) of every (record, sample) in a
// matrix; k_assemble_write streams that matrix (coalesced, prefetched one record ahead), keeps the current text of each
// sample in registers, and reads the pool only when a sample's slot changes.  The per-record scalars (index, start, type,
// destination) are fetched 64 records at a time, one per lane, and broadcast with v_readlane.
constexpr int kAsmRows = 64;         // samples per (record, chunk): one wavefront
constexpr int kWaveLds = 8 * 1024;   // LDS image of one (record, chunk)
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;
constexpr int kCooperativeEntry = 512;   // entry texts longer than this are copied pool -> page by the whole wavefront, not through the LDS image

__device__ __forceinline__ int64_t readlane64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(uint64_t)v, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((uint64_t)v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Row of (record k_rel, chunk ch) in the resolved matrix.  res_rows == 0: record-major (the 16 chunk rows of a record are neighbours: what the
// BCF kernels read); res_rows = #records of the matrix: CHUNK-major - the rows a wavefront writes (sizing) and reads (page assembly) along
// its run of records lie 512 bytes apart instead of 8 KB (GDBAMD_RES_LAYOUT=1; measured on c2: no difference, so record-major is what runs).
__device__ __forceinline__ int64_t res_row(int64_t k_rel, int ch, int nchunks, int64_t res_rows) { return res_rows ? (int64_t)ch * res_rows + k_rel : k_rel * nchunks + ch; }
// The resolved matrix in one of two layouts.  wide: (pool offset, length) as a uint2 per (record, sample) - 8 bytes.  compact
// (text output, chosen per interval when no entry text is longer than 255 bytes - every window of the c2 workload): the pool offsets
// as a plane of u32 and the lengths as a plane of u8 - 5 bytes per pair: 3 of 8 bytes less written by the sizing pass and read
// back by the page pass (8.2 -> 5.1 GB per 1 Mb window of 1 000 samples, each way).  BCF2 and the piece / event paths keep the wide one.
struct ResMatrix {
  uint2* wide;        // nullptr: compact, or no matrix at all (sizes only)
  uint32_t* off;      // compact planes (nullptr: wide)
  uint8_t* len8;
  __device__ __forceinline__ bool any() const { return wide != nullptr || off != nullptr; }
  __device__ __forceinline__ void store(int64_t at, uint2 d) const {
    if (off) { off[at] = d.x; len8[at] = (uint8_t)d.y; } else wide[at] = d;
  }
  __device__ __forceinline__ uint2 load(int64_t at) const {
    if (off) return make_uint2(off[at], (uint32_t)len8[at]);
    return wide[at];
  }
};
// chunk_size == nullptr: resolve only (per-page matrix when the whole interval's matrix would not fit the budget)
__global__ void __launch_bounds__(kAsmRows)
k_assemble_size(AsmCtx a, const int32_t* __restrict__ order, int64_t n, int32_t N, int nchunks, int run, uint64_t* __restrict__ chunk_size,
                ResMatrix resolved, int64_t resolved_base, int64_t res_rows) {
  // chunk is the fast grid dimension: the wavefronts in flight work on the same few records (one 44 KB line is written
  // by its 16 chunk wavefronts at about the same time: DRAM pages, TLB entries and shared boundary lines stay hot)
  const int64_t ib = (int64_t)(blockIdx.x / (unsigned)nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(blockIdx.x % (unsigned)nchunks);
  const int lane = threadIdx.x;
  const int32_t r = ch * kAsmRows + lane;
  SlotWalker w;
  int64_t prev_k = INT64_MAX;
  uint32_t cur_slot = kNoSlot;
  uint2 cur = make_uint2(0, 0);
  for (int64_t i0 = ib; i0 < ie; i0 += 64) {                // uniform
    const int cnt = (int)min((int64_t)64, ie - i0);
    int32_t my_k = 0; int64_t my_s = 0; uint32_t my_t = 0; uint64_t my_total = 0;
    if (lane < cnt) {
      my_k = order[i0 + lane];
      my_s = a.rec_start[my_k];
      my_t = a.rtype[my_k];
      if (ch == 0) my_total = a.prefix_len[my_k];
      if (ch == nchunks - 1) my_total += 1;                 // '\n'
    }
    for (int jj = 0; jj < cnt; ++jj) {                      // uniform
      const int64_t k = __builtin_amdgcn_readlane(my_k, jj);
      const int64_t s = readlane64(my_s, jj);
      const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_t, jj);
      uint2 d = make_uint2(0, 0);
      if (r < N) {
        if (k < prev_k) w.init(a, r, s); else w.advance(a, s);  // uniform branch: a new type restarts at its first record
        const uint32_t sl = w.slot(a, k, s, t, r);
        if (sl != cur_slot) { cur_slot = sl; cur = a.desc[sl]; }
        d = cur;
      }
      prev_k = k;
      if (resolved.any()) resolved.store(res_row(k - resolved_base, ch, nchunks, res_rows) * kAsmRows + lane, d);
      const uint32_t total = wave_total(wave_inclusive_scan_dpp(d.y));
      if (lane == jj) my_total += total;
    }
    if (chunk_size && lane < cnt) chunk_size[(int64_t)my_k * nchunks + ch] = my_total;
  }
}

// The same pass, PIECE-WISE (default; GDBAMD_SIZE3=0 selects the kernel above).  k_assemble_size decides record by record whether a
// sample's entry changes: with 64 samples in a wavefront some lane changes at nearly every record (~1.6 changes per step), so the
// dear path - walk list, incidence / slot arithmetic, descriptor load: two dependent loads - is executed by 1-2 of 64 lanes on
// ~80 % of the 64 steps of a run.  Here the records go in rounds of R: in phase 1 EVERY lane names its own pieces of the round (a piece =
// the records of the round that keep one entry: the lane asks the walker at the piece's first record, exactly as the kernel above
// does at every record, and finds the piece's end among the round's record starts, held in LDS, by binary search), all lanes busy, a
// few iterations per round (the most pieces any lane has); phase 2 is 64 cheap steps (is this record the start of my next piece? -
// one register compare - and the row's coalesced store).  The sizes do not need a scan per record either: every piece adds its length
// at its first record and takes it away behind its last in a difference array over the batch's 64 records (LDS atomics), one scan per batch.
// Bit-identical output (GDBAMD_SIZE3_CHECK=1 runs both and compares).
#ifdef GDBAMD_SIZE3_WAVES               // (variant builds: 8 = at most 64 registers)
#define GDBAMD_SIZE3_ATTR __attribute__((amdgpu_waves_per_eu(GDBAMD_SIZE3_WAVES)))
#else
#define GDBAMD_SIZE3_ATTR
#endif
template <int R> __global__ void __launch_bounds__(kAsmRows) GDBAMD_SIZE3_ATTR
k_assemble_size3(AsmCtx a, const int32_t* __restrict__ order, int64_t n, int32_t N, int nchunks, int run, uint64_t* __restrict__ chunk_size,
                 ResMatrix resolved, int64_t resolved_base, int64_t res_rows) {
  __shared__ int64_t ss[kAsmRows];                     // the batch's record starts
  __shared__ int32_t ks[kAsmRows];                     // ... and record indices
  __shared__ uint32_t diff[kAsmRows + 1];              // difference array of the chunk sizes over the batch
  __shared__ uint2 pl_d[R * kAsmRows];                 // the lanes' pieces of one round: entry ...
  __shared__ uint8_t pl_at[R * kAsmRows];              // ... and first record (index into the batch)
  const int64_t ib = (int64_t)(blockIdx.x / (unsigned)nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(blockIdx.x % (unsigned)nchunks);
  const int lane = threadIdx.x;
  const int32_t r = ch * kAsmRows + lane;
  const bool has_row = r < N;
  SlotWalker w;
  int64_t prev_k = INT64_MAX;                          // (uniform) the record in front
  uint32_t cur_slot = kNoSlot;
  uint2 cur = make_uint2(0, 0);
  for (int64_t i0 = ib; i0 < ie; i0 += 64) {          // uniform
    const int cnt = (int)min((int64_t)64, ie - i0);
    int32_t my_k = 0; int64_t my_s = INT64_MAX; uint32_t my_t = 0; uint64_t my_total = 0;
    if (lane < cnt) {
      my_k = order[i0 + lane];
      my_s = a.rec_start[my_k];
      my_t = a.rtype[my_k];
      if (ch == 0) my_total = a.prefix_len[my_k];
      if (ch == nchunks - 1) my_total += 1;           // '\n'
    }
    ss[lane] = my_s; ks[lane] = my_k; diff[lane] = 0u;
    if (lane == 0) diff[kAsmRows] = 0u;
    // segments: records of one type in increasing order (a new type, or the next block of the order, starts a new one)
    const int32_t left_k = __shfl_up(my_k, 1, 64);
    const uint32_t left_t = (uint32_t)__shfl_up((int)my_t, 1, 64);
    const uint64_t heads = __ballot(lane < cnt && (lane == 0 || my_k < left_k || my_t != left_t));
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int seg_a = 0; seg_a < cnt;) {                // uniform
      const uint64_t behind = seg_a + 1 < 64 ? heads >> (seg_a + 1) : 0ull;
      const int seg_b = behind ? seg_a + 1 + (int)__builtin_ctzll(behind) : cnt;
      const int64_t k_a = __builtin_amdgcn_readlane(my_k, seg_a);
      const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_t, seg_a);
      if (k_a < prev_k && has_row) w.init(a, r, readlane64(my_s, seg_a));   // uniform test: a new type restarts at its first record
      for (int ra = seg_a; ra < seg_b; ra += R) {      // uniform
        const int rb = min(seg_b, ra + R);
        // ---- phase 1: the lane's pieces of [ra, rb) ----
        int pos = ra, np = 0;
        while (__any((int)(pos < rb))) {
          if (pos < rb) {
            uint2 d = make_uint2(0, 0);
            int e = rb;
            if (has_row) {
              const int64_t s = ss[pos];
              const int64_t k = ks[pos];
              w.advance(a, s);
              const uint32_t sl = w.slot(a, k, s, t, r);
              if (sl != cur_slot) { cur_slot = sl; cur = a.desc[sl]; }
              d = cur;
              const bool dead = s > w.cur_end;
              if (t == kUntabledType || (!dead && w.heavy)) e = pos + 1;     // a slot per record
              else {
                // the entry holds while the walker stands (no cell of the row begins) and the cell stays dead / stays live
                const int64_t x = dead ? w.next_begin - 1 : min(w.cur_end, w.next_begin - 1);
                int lo = pos + 1, hi = rb;             // first record of the round that starts behind x
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ss[mid] > x) hi = mid; else lo = mid + 1; }
                e = lo;
              }
            }
            pl_d[np * kAsmRows + lane] = d;
            pl_at[np * kAsmRows + lane] = (uint8_t)pos;
            ++np;
            if (chunk_size && d.y) { atomicAdd(&diff[pos], d.y); atomicAdd(&diff[e], 0u - d.y); }
            pos = e;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        // ---- phase 2: the rows ----
        if (resolved.any()) {
          int q = 0, nxt = ra;                         // (every lane's first piece begins at ra)
          uint2 row = make_uint2(0, 0);
          for (int jj = ra; jj < rb; ++jj) {           // uniform
            if (jj == nxt) {
              row = pl_d[q * kAsmRows + lane];
              ++q;
              nxt = q < np ? (int)pl_at[q * kAsmRows + lane] : 255;
            }
            const int64_t k = __builtin_amdgcn_readlane(my_k, jj);
            resolved.store(res_row(k - resolved_base, ch, nchunks, res_rows) * kAsmRows + lane, row);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
      prev_k = __builtin_amdgcn_readlane(my_k, seg_b - 1);
      seg_a = seg_b;
    }
    if (chunk_size) {
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const uint32_t incl = wave_inclusive_scan_dpp(diff[lane]);      // bytes of the sample columns of record `lane` in this chunk
      if (lane < cnt) chunk_size[(int64_t)my_k * nchunks + ch] = my_total + incl;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}
// GDBAMD_SIZE3_CHECK: the two sizing kernels' outputs, word for word
__global__ void k_compare_words(const uint32_t* a, const uint32_t* b, int64_t nwords, unsigned long long* mismatches) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nwords && a[i] != b[i]) atomicAdd(mismatches, 1ull);
}

constexpr int kTextChunks = 8;       // 16-byte chunks of a slot kept in registers
struct SlotText { uint4 x[kTextChunks]; };

// Page assembly: one wavefront = `run` records x 64 samples (one chunk), one wavefront per workgroup (no s_barrier).
// Variants measured on MI355X this round and dropped because they were slower (c2, 200 kb window, ms per launch, this
// kernel = 3.0-3.2): 2 / 4 chunks per wavefront in lock step 3.3 / 4.7; 2 / 4 records built before one flush 3.9 / 5.4;
// builder + flusher wavefront pairs 5.6; skewed software pipeline with two LDS images 4.2; unaligned ds_write_b128 of whole
// chunks instead of the funnel-shifted words (gfx950 accepts any byte alignment, but it costs 3.8 vs 2.9).  All of them trade resident
// wavefronts for fewer exposed waits, and lose: the kernel is bound by (resident wavefronts) / (per-record latency).
// Which (run, chunk) a wavefront takes.  Workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2: with
// "chunk fast" numbering the 16 chunks of a record - 2.8 KB runs that share their first and last cache line with the neighbour
// chunk - were written from 8 different L2s.  A store-only kernel of exactly this shape (tests/tools/microbench/store_bw.hip)
// goes from 3.4 to 4.6 TB/s when the numbering keeps the chunks of a record run on one XCD, and to 5.1 TB/s with 4 neighbouring
// chunks per workgroup (4 wavefronts, one LDS image each, still no barrier); the page assembly sat exactly on the 3.4.
template <int WAVES> __device__ __forceinline__ int64_t xcd_aware_unit(int64_t total_units, int xcd_aware = 1) {
  const unsigned nb = gridDim.x, per = nb / 8u;
  const unsigned lb = (xcd_aware && blockIdx.x < per * 8u) ? (blockIdx.x % 8u) * per + blockIdx.x / 8u : blockIdx.x;
  const int64_t u = (int64_t)lb * WAVES + (int64_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: scalar registers)
  return u < total_units ? u : -1;
}
// COOP_U: 16-byte words per lane whose source loads are in flight together in the cooperative copy of a long entry (the whole
// wavefront moves one entry pool -> page): 2 where long entries are rare (c2, c3); c5's hot records are ALL long entries (6 KB of PL
// text per sample, 550 of 597 GB): 4 there (chosen by the host from the largest record's average entry), at the price of registers.
struct __attribute__((packed)) GlobalPackedU128_ { u32x4 v; };
typedef __attribute__((address_space(1))) GlobalPackedU128_ GlobalPackedU128;   // an unaligned 16-byte word in GLOBAL memory (global_load, not flat_load)
template <int WAVES, int kWaveLds, int COOP_U = 2> __global__ void __launch_bounds__(kAsmRows * WAVES)
k_assemble_write(const char* __restrict__ pool, const char* __restrict__ pool_ovf, const uint32_t* __restrict__ prefix_len, const ResMatrix resolved, int64_t resolved_base,
                 const int32_t* __restrict__ order, int64_t n, int nchunks, int run, const uint64_t* __restrict__ chunk_off, uint64_t page_base,
                 char* __restrict__ arena, int xcd_aware, int64_t res_rows) {
  const int64_t unit = xcd_aware_unit<WAVES>(((n + run - 1) / run) * (int64_t)nchunks, xcd_aware);
  if (unit < 0) return;
  const int64_t ib = (unit / nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x & 63;
  __shared__ __attribute__((aligned(16))) char lds_all[WAVES][kWaveLds + 16 + kScrapBytes];
  char* const lds_buf = &lds_all[threadIdx.x >> 6][0];
  uint2 cur = make_uint2(0xFFFFFFFFu, 0);
  const char* cur_src = pool;
  SlotText txt;
#pragma unroll
  for (int q = 0; q < kTextChunks; ++q) txt.x[q] = make_uint4(0, 0, 0, 0);
  for (int64_t i0 = ib; i0 < ie; i0 += 64) {                // uniform
    const int cnt = (int)min((int64_t)64, ie - i0);
    int32_t my_k = 0; int64_t my_dst = 0;
    if (lane < cnt) {
      my_k = order[i0 + lane];
      my_dst = (int64_t)(chunk_off[(int64_t)my_k * nchunks + ch] - page_base) + (ch == 0 ? prefix_len[my_k] : 0u);
    }
    uint2 d_next = resolved.load(res_row((int64_t)__builtin_amdgcn_readlane(my_k, 0) - resolved_base, ch, nchunks, res_rows) * kAsmRows + lane);
    for (int jj = 0; jj < cnt; ++jj) {                      // uniform
      const uint2 d = d_next;
      if (jj + 1 < cnt) d_next = resolved.load(res_row((int64_t)__builtin_amdgcn_readlane(my_k, jj + 1) - resolved_base, ch, nchunks, res_rows) * kAsmRows + lane);
      const uint32_t len = d.y;
      if (len && (d.x != cur.x || len != cur.y)) {          // the sample moved to another slot: fetch its text
        cur = d;
        cur_src = ((d.x & kOverflowBit) ? pool_ovf : pool) + (size_t)(d.x & ~kOverflowBit) * 16;
#pragma unroll
        for (int q = 0; q < kTextChunks; ++q) txt.x[q] = load_chunk(cur_src, q, len);
      }
      const uint32_t inc = wave_inclusive_scan_dpp(len);
      const uint32_t excl = inc - len;
      const uint32_t total = wave_total(inc);
      if (total == 0) continue;                             // uniform: no FORMAT columns in this record
      char* const grec = arena + readlane64(my_dst, jj);
      // The chunk normally fits one LDS image.  Wider chunks (long PL vectors) go out in passes over consecutive lane
      // ranges that fit; only a single entry larger than the image is copied straight from the pool.
      uint32_t l0 = 0, base_off = 0;
      while (base_off < total) {                            // uniform
        char* gdst = grec + base_off;
        const uint32_t al = (uint32_t)((uintptr_t)gdst & 15u);
        const bool fits = (uint32_t)lane >= l0 && al + (inc - base_off) <= (uint32_t)kWaveLds;
        const uint64_t fit_mask = __ballot(fits) >> l0;     // inc is non-decreasing: the fitting lanes are a run starting at l0
        const uint32_t len_l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(l0 < 64u ? l0 : 63u));
        if (!(fit_mask & 1ull) || len_l0 > (uint32_t)kCooperativeEntry) {   // a long text (or one that exceeds the image): the whole wavefront copies it
          const char* big_src = (const char*)(uintptr_t)readlane64((int64_t)(uintptr_t)cur_src, (int)l0);   // (pool slots are 16-byte aligned)
          const uint32_t big_len = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)l0);
          uint32_t head = (16u - al) & 15u;                 // bytes up to the first 16-byte boundary of the destination
          if (head > big_len) head = big_len;
          if ((uint32_t)lane < head) gdst[lane] = big_src[lane];
          const uint32_t nwords = (big_len - head) >> 4;
          uint4* gw = reinterpret_cast<uint4*>(gdst + head);
          // aligned 16-byte stores fed by UNALIGNED 16-byte global loads (gfx9 handles the misalignment in the memory pipeline).  Until round 5
          // every word was assembled from two aligned loads with v_alignbyte and a run-time word offset - which the compiler turns into a
          // chain of eight v_cndmask per source word: ~150 instructions per 16 bytes, the reason c5's page assembly ran at 0.19.
          const GlobalPackedU128* gsrc = (const GlobalPackedU128*)(uintptr_t)(big_src + head);
          for (uint32_t w0 = lane; w0 < nwords; w0 += kAsmRows * COOP_U) {
            u32x4 v[COOP_U];
#pragma unroll
            for (int u = 0; u < COOP_U; ++u) {                    // all loads of the batch first ...
              const uint32_t wq = w0 + (uint32_t)u * kAsmRows;
              v[u] = gsrc[wq < nwords ? wq : w0].v;
            }
#pragma unroll
            for (int u = 0; u < COOP_U; ++u) {                    // ... then the stores
              const uint32_t wq = w0 + (uint32_t)u * kAsmRows;
              if (wq < nwords) gw[wq] = make_uint4(v[u][0], v[u][1], v[u][2], v[u][3]);
            }
          }
          const uint32_t tail_at = head + (nwords << 4);
          if ((uint32_t)lane < big_len - tail_at) gdst[tail_at + lane] = big_src[tail_at + lane];
          base_off = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)l0);
          ++l0;
          continue;
        }
        const uint32_t nfit = fit_mask == ~0ull ? 64u - l0 : (uint32_t)__builtin_ctzll(~fit_mask);
        const uint32_t l1 = l0 + nfit;
        const uint32_t pass_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(l1 - 1)) - base_off;
        const bool mine = (uint32_t)lane >= l0 && (uint32_t)lane < l1;
        SlotCopy cp;
        cp.begin((gdb_lds_char*)lds_buf + al + (excl - base_off), mine ? len : 0u, (gdb_lds_char*)lds_buf + kWaveLds + 16 + 4 * lane);
        cp.chunk(0, txt.x[0]);
#pragma unroll
        for (int q = 1; q < kTextChunks; ++q) if (__any((int)cp.needs(q))) cp.chunk(q, txt.x[q]);
        for (uint32_t q = kTextChunks; __any((int)cp.needs(q)); ++q) cp.chunk(q, load_chunk(cur_src, q, mine ? len : 0u));
        cp.finish();
        // One wavefront per workgroup: the LDS unit runs its instructions in order, so the image only needs a compiler-level
        // fence (wavefront scope emits no s_waitcnt: outstanding matrix loads and page stores keep flying across records).
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const char* img = lds_buf + al;
        uint32_t head = (16u - al) & 15u;
        if (head > pass_total) head = pass_total;
        const uint32_t nwords = (pass_total - head) >> 4;
        const uint32_t tail_at = head + (nwords << 4);
        const uint4* lsrc = reinterpret_cast<const uint4*>(img + head);
        uint4* gw = reinterpret_cast<uint4*>(gdst + head);
        for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) gw[wq] = lsrc[wq];
        {   // the bytes in front of the first and behind the last 16-byte word in ONE store instruction: lanes 0-15 head, 16-31 tail
          const uint32_t li = (uint32_t)lane & 15u;
          const bool is_tail = (lane & 16) != 0;
          const uint32_t at = is_tail ? tail_at + li : li;
          if (lane < 32 && li < (is_tail ? pass_total - tail_at : head)) gdst[at] = img[at];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        l0 = l1;
        base_off += pass_total;
      }
    }
  }
}

// ---- matrix-free sizing and page assembly (text output) -------------------------------------------------------------------------
// The (record, sample) matrix of the kernels above costs 8 bytes per pair twice (written by the sizing pass, read back by the page
// pass: 2 x 8.2 GB per 1 Mb window of 1 000 samples) and a sizing pass that touches every pair.  Neither is needed:
//  * a sample's entry changes only where its live cell (or the gap between two cells) changes: a PIECE.  Along a run of same-type
//    records a lane keeps `valid_until`, the last record its present entry holds for; the common step is one compare.
//  * record sizes are sums over pieces, not over pairs: k_size2 adds every piece's length as a +/- pair into a difference array
//    over the type-ordered records of a block (LDS) and scans it - O(cells x types met), not O(records x samples).
//  * the page pass (k_write2) walks the pieces itself; a slot's text and its length arrive together (the slot's tag, kSlotTagAt).
struct PieceCtx {
  const int64_t* row_ptr;     // [N+1] row-major ranges
  const int64_t* rm_begin;    // [C]   begin column in row-major order
  const WalkCell* walk;       // [C]
  const int64_t* rec_start;   // [P]
  const uint8_t* rtype;       // [P]
  const uint32_t* ubase;      // [P]   rank of a record among those with an untabled type
  const uint32_t* occ;        // [kMaxTypes][P+1]  records of a type before k (rows share one scan: only differences inside a row mean something)
  const uint32_t* slot_len;   // [S]
  const uint32_t* prefix_len; // [P]
  int32_t* jhint;             // [ceil(P / kHintStep)][N]  first cell of a row (relative to row_ptr) that can be live at or behind record h * kHintStep: left by
                              // k_size2, which walks every row once anyway; the page pass starts a run there instead of searching the row
  uint32_t light_base, heavy_base, row_base, epoch;
  int32_t nrows;
  int64_t P;
};
constexpr int kHintStep = 256;
struct PieceWalker {
  // W = the cell the walker stands at (walk[j]); Wn = walk[j + 1], requested when the walker moved on to W: it has arrived long
  // before it is looked at.  x = k_lo, y = k_hi, w = flags; W.z = the cell's slot for the run's record type (heavy: of record k_lo),
  // Wn.z = base.  `next` = the cell behind Wn.
  const WalkCell* next;
  int32_t valid_until;       // last record the lane's present slot holds for
  u32x4 W, Wn; uint64_t Wnaux;
  __device__ __forceinline__ void request(const PieceCtx& a) {   // Wn <- *next (no wait here; behind the row's cells lies its sentinel)
    Wn = *reinterpret_cast<const u32x4*>(&next->k_lo); Wnaux = next->aux;
    ++next;
  }
  // W <- Wn, then the cell behind it is requested WITHOUT a wait.  Real moves, pinned in front of the loads: left to itself the
  // compiler renames instead - W takes over Wn's registers, the new Wn is loaded somewhere else and copied back at the end of the
  // block, behind an s_waitcnt vmcnt(0) that the copy then needs.
  // A cell stamped by another interval begins behind the window (the hint never points in front of it): nothing further is live.
  __device__ __forceinline__ void advance(const PieceCtx& a, uint64_t below) {
    uint32_t wx, wy, wz, ww, alo, ahi;
    asm volatile("v_mov_b32 %0, %1" : "=v"(wx) : "v"(Wn.x));
    asm volatile("v_mov_b32 %0, %1" : "=v"(wy) : "v"(Wn.y));
    asm volatile("v_mov_b32 %0, %1" : "=v"(wz) : "v"(Wn.z));
    asm volatile("v_mov_b32 %0, %1" : "=v"(ww) : "v"(Wn.w));
    asm volatile("v_mov_b32 %0, %1" : "=v"(alo) : "v"((uint32_t)Wnaux));
    asm volatile("v_mov_b32 %0, %1" : "=v"(ahi) : "v"((uint32_t)(Wnaux >> 32)));
    __builtin_amdgcn_sched_barrier(0);
    request(a);
    const bool fresh = (ww >> 8) == a.epoch;
    W.x = fresh ? wx : (uint32_t)INT32_MAX; W.y = fresh ? wy : (uint32_t)INT32_MAX; W.w = ww;
    W.z = (ww & kWalkHeavy) ? a.heavy_base + wz : a.light_base + wz + (uint32_t)__popc(alo & (uint32_t)below) + (uint32_t)__popc(ahi & (uint32_t)(below >> 32));
  }
  // k = the first record the lane will be asked about: every cell in front of the hinted one is dead from record (k / kHintStep) * kHintStep on
  __device__ __forceinline__ void init(const PieceCtx& a, int32_t row, int32_t k, uint64_t below) {
    next = a.walk + (a.row_ptr[row] + row + (int64_t)a.jhint[(int64_t)(k / kHintStep) * a.nrows + row]);
    valid_until = INT32_MIN;
    request(a);
    advance(a, below);   // (reads what was requested: the wait for it stays here, in the rare block)
    asm volatile("; Wn has arrived" : : "v"(Wn.x), "v"(Wn.w), "v"((uint32_t)Wnaux), "v"((uint32_t)(Wnaux >> 32)));   // ... and so does the wait for the second request
  }
  __device__ __forceinline__ bool behind(int32_t k) const { return k > (int32_t)W.y || (int32_t)W.x < 0; }   // W ended before k, or is live in no record
  // slot of (row, record k of type t); k > valid_until on entry, k does not decrease between calls.
  // The vector-memory counter of gfx9 is one in-order queue of loads AND stores: a wait for anything waits for the page stores of
  // the previous record too.  So the first advance is peeled out of the loop: what it reads (Wn) was requested a whole piece ago
  // and has been waited for together with that piece's text - the compiler puts no s_waitcnt in front of it; only the rare
  // further steps (a cell passed over: nothing of the run's type inside it) wait for the request they have just made.
  // nq: 16-byte chunks of the slot that hold its text if it is an inline one
  __device__ __forceinline__ uint32_t move(const PieceCtx& a, int32_t k, uint32_t t, uint64_t below, int32_t row, uint32_t nocall_chunks, uint32_t& nq) {
    const bool tabled = t != kUntabledType;
    if (behind(k)) {
      advance(a, below);
      while (behind(k)) advance(a, below);
    }
    const bool heavy = (W.w & kWalkHeavy) != 0, inside = k >= (int32_t)W.x;
    nq = (W.w >> kWalkChunksShift) & kWalkChunksMask;
    if (!tabled) valid_until = k;                                // records of untabled types: one slot per (record, sample)
    else if (!inside) { valid_until = (int32_t)W.x - 1; nq = nocall_chunks; return t; }   // between two cells: the type's no-call entry
    else valid_until = heavy ? k : (int32_t)W.y;
    if (inside && heavy) return W.z + (uint32_t)(k - (int32_t)W.x);
    if (!tabled) { nq = kWalkChunksMask; return a.row_base + a.ubase[k] * (uint32_t)a.nrows + (uint32_t)row; }
    return W.z;
  }
};

// Sizing: one wavefront = kSizeBlock consecutive records x 64 samples.  Inside the block the records are ranked by (type, index)
// (position p = adj[t] + occ[t][k]); a plain cell's slot for type t holds for the type-t records of [k_lo, k_hi], a contiguous
// range of positions: +len - nocall at its first, -(len - nocall) behind its last.  Heavy calls and the samples of untabled
// records have a slot per record.  Every record starts from (#samples of the chunk) x (no-call length of its type).
constexpr int kSizeBlock = 1024;
__global__ void __launch_bounds__(kAsmRows)
k_size2(PieceCtx a, int nchunks, uint64_t* __restrict__ chunk_size) {
  __shared__ uint32_t diff[kSizeBlock + 2];
  __shared__ int32_t adj[kMaxTypes + 1];
  __shared__ uint32_t nolen[kMaxTypes + 1];
  const int lane = threadIdx.x;
  // the chunk wavefronts of a record block read the same lines of `occ`: keep them on one XCD (one L2), next to each other in time
  const int64_t unit = xcd_aware_unit<1>((int64_t)gridDim.x);
  const int64_t sub = unit / nchunks;
  const int ch = (int)(unit % nchunks);
  const int64_t k0 = sub * kSizeBlock, k1 = min(a.P, k0 + (int64_t)kSizeBlock);
  const int nrec = (int)(k1 - k0);
  for (int i = lane; i < kSizeBlock + 2; i += kAsmRows) diff[i] = 0;
  {
    const uint32_t* row = a.occ + (int64_t)lane * (a.P + 1);     // (kMaxTypes == 64 lanes: lane t counts type t)
    const uint32_t o0 = row[k0], cnt = row[k1] - o0;
    const uint32_t incl = wave_inclusive_scan_dpp(cnt);
    adj[lane] = (int32_t)(incl - cnt) - (int32_t)o0;
    nolen[lane] = a.slot_len[lane];
    if (lane == 63) { adj[kMaxTypes] = (int32_t)incl - (int32_t)a.ubase[k0]; nolen[kMaxTypes] = 0; }   // untabled records rank behind the tabled ones
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const int32_t r = ch * kAsmRows + lane;
  if (r < a.nrows) {
    const int64_t j_begin = a.row_ptr[r], j_end = a.row_ptr[r + 1];
    const int64_t s0 = a.rec_start[k0];
    int64_t lo = j_begin, hi = j_end;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.rm_begin[mid] <= s0) lo = mid + 1; else hi = mid; }
    int64_t j = lo - 1;
    bool may_be_stale = j >= j_begin;                            // (the cell found may begin before the window - stale stamp -: it cannot reach it; every
    if (!may_be_stale) j = j_begin;                              //  cell behind it begins inside or behind the window)
    int64_t hint_k = k0;                                         // next record a hint is due for
    for (; j < j_end; ++j) {
      const WalkCell* wp = a.walk + j + r;
      const uint4 tail = *reinterpret_cast<const uint4*>(&wp->k_lo);
      const int32_t klo = (int32_t)tail.x, khi = (int32_t)tail.y;
      const bool stale = (tail.w >> 8) != a.epoch;
      if (stale) { if (may_be_stale) { may_be_stale = false; continue; } break; }
      may_be_stale = false;
      if (klo < 0) continue;
      for (; hint_k <= (int64_t)khi && hint_k < k1; hint_k += kHintStep) a.jhint[(hint_k / kHintStep) * a.nrows + r] = (int32_t)(j - j_begin);
      if ((int64_t)klo >= k1) break;
      if ((int64_t)khi < k0) continue;
      const int64_t ka = max((int64_t)klo, k0), kb = min((int64_t)khi, k1 - 1);
      if (tail.w & kWalkHeavy) {
        for (int64_t k = ka; k <= kb; ++k) {
          const uint32_t t = a.rtype[k];
          const uint32_t tt = t == kUntabledType ? (uint32_t)kMaxTypes : t;
          const uint32_t d = a.slot_len[a.heavy_base + tail.z + (uint32_t)(k - klo)] - nolen[tt];
          const uint32_t p = (uint32_t)(adj[tt] + (int32_t)(t == kUntabledType ? a.ubase[k] : a.occ[(int64_t)t * (a.P + 1) + k]));
          atomicAdd(&diff[p], d); atomicAdd(&diff[p + 1], 0u - d);
        }
      } else {
        uint64_t m = wp->aux;
        uint32_t sl = a.light_base + tail.z;
        while (m) {
          const int t = __builtin_ctzll(m);
          m &= m - 1;
          const uint32_t d = a.slot_len[sl++] - nolen[t];
          const uint32_t* row = a.occ + (int64_t)t * (a.P + 1);
          const uint32_t pa = (uint32_t)(adj[t] + (int32_t)row[ka]), pb = (uint32_t)(adj[t] + (int32_t)row[kb + 1]);
          atomicAdd(&diff[pa], d); atomicAdd(&diff[pb], 0u - d);
        }
      }
    }
    for (; hint_k < k1; hint_k += kHintStep) a.jhint[(hint_k / kHintStep) * a.nrows + r] = (int32_t)(j - j_begin);   // (nothing live further on: the cell the walk stopped at)
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  uint32_t carry = 0;
  for (int b = 0; b < nrec; b += kAsmRows) {                     // uniform
    const uint32_t v = diff[b + lane];
    const uint32_t incl = wave_inclusive_scan_dpp(v) + carry;
    diff[b + lane] = incl;
    carry = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const uint32_t rows_here = (uint32_t)min(kAsmRows, a.nrows - ch * kAsmRows);
  for (int i = lane; i < nrec; i += kAsmRows) {
    const int64_t k = k0 + i;
    const uint32_t t = a.rtype[k];
    uint64_t size;
    if (t != kUntabledType) size = (uint64_t)diff[adj[t] + (int32_t)a.occ[(int64_t)t * (a.P + 1) + k]] + (uint64_t)rows_here * nolen[t];
    else {
      size = diff[adj[kMaxTypes] + (int32_t)a.ubase[k]];
      const uint32_t* rl = a.slot_len + a.row_base + (uint64_t)a.ubase[k] * (uint32_t)a.nrows + (uint32_t)(ch * kAsmRows);
      for (uint32_t q = 0; q < rows_here; ++q) size += rl[q];    // (the slot of a sample with a heavy call here is empty)
    }
    if (ch == 0) size += a.prefix_len[k];
    if (ch == nchunks - 1) size += 1;                            // '\n'
    chunk_size[k * nchunks + ch] = size;
  }
}

// ---- piece lists --------------------------------------------------------------------------------------------------------------------
// k_write2 above pays for deciding, record by record and lane by lane, what the next piece is (stale cells, cells without a record of the
// run's type, gaps, heavy calls: ~130 instructions of a wavefront on nearly every record - and these kernels are bound by the instructions
// they issue).  The decisions do not depend on the page being assembled, so they are made ONCE per interval, per piece: for every block of
// kSizeBlock records, every tabled record type t and every sample row r a LIST of entries (first record it holds from, where the entry's
// text lies, its length), in record order:
//   * a plain cell contributes one entry per type it has a slot for, from the first record of the block it is live in;
//   * a heavy call one entry per record, in the list of that record's type;
//   * wherever no cell is live (between two cells, in front of the first, behind the last) every type gets the no-call entry;
//   * a list ends with an entry that never begins (INT32_MAX).
// Every record of type t in the block therefore has an entry of list (block, t, r) beginning at or in front of it, and the entry in force is
// the last such one.  k_plist<false> counts, a scan turns the counts into list offsets, k_plist<true> fills; k_write3 consumes.
struct PieceEntry { int32_t k_from; uint32_t where, len; int32_t k_next; };   // where: 16-byte unit of the inline pool, or kOverflowBit | unit of the overflow pool; k_next: the k_from of the entry behind (a lane decides to move on from the entry it HOLDS - the one behind it is still on its way)
template <bool FILL> __global__ void __launch_bounds__(kAsmRows)
k_plist(PieceCtx a, const uint2* __restrict__ desc, int NT, int nchunks, uint32_t* __restrict__ counts_or_offsets, PieceEntry* __restrict__ plist) {
  __shared__ uint32_t cur[kMaxTypes][kAsmRows];                 // per (type, lane): entries so far / where the next one goes
  const int lane = threadIdx.x;
  const int64_t unit = xcd_aware_unit<1>((int64_t)gridDim.x);
  const int64_t sub = unit / nchunks;
  const int ch = (int)(unit % nchunks);
  const int64_t k0 = sub * kSizeBlock, k1 = min(a.P, k0 + (int64_t)kSizeBlock);
  const int32_t r = ch * kAsmRows + lane;
  if (r >= a.nrows) return;
  uint32_t* const table = counts_or_offsets + ((int64_t)sub * NT) * a.nrows + r;      // [t] at stride nrows
  for (int t = 0; t < NT; ++t) cur[t][lane] = FILL ? table[(int64_t)t * a.nrows] : 0u;
  auto emit = [&](uint32_t t, int64_t k_from, uint32_t slot) {
    if (FILL) {
      const uint2 d = desc[slot];
      PieceEntry e;
      e.k_from = (int32_t)k_from; e.where = d.x; e.len = d.y; e.k_next = INT32_MAX;
      const uint32_t at = cur[t][lane];
      if (at != table[(int64_t)t * a.nrows]) plist[at - 1].k_next = (int32_t)k_from;
      *reinterpret_cast<uint4*>(&plist[at]) = *reinterpret_cast<const uint4*>(&e);
    }
    ++cur[t][lane];
  };
  const int64_t j_begin = a.row_ptr[r], j_end = a.row_ptr[r + 1];
  const int64_t s0 = a.rec_start[k0];
  int64_t lo = j_begin, hi = j_end;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (a.rm_begin[mid] <= s0) lo = mid + 1; else hi = mid; }
  int64_t j = lo - 1;
  bool may_be_stale = j >= j_begin;                              // (as in k_size2: the cell found may begin in front of the window)
  if (!may_be_stale) j = j_begin;
  int64_t covered = k0 - 1;                                      // last record of the block that has its entries
  for (; j < j_end; ++j) {
    const WalkCell* wp = a.walk + j + r;
    const uint4 tail = *reinterpret_cast<const uint4*>(&wp->k_lo);
    const int32_t klo = (int32_t)tail.x, khi = (int32_t)tail.y;
    if ((tail.w >> 8) != a.epoch) { if (may_be_stale) { may_be_stale = false; continue; } break; }
    may_be_stale = false;
    if (klo < 0) continue;
    if ((int64_t)klo >= k1) break;
    if ((int64_t)khi < k0) continue;
    const int64_t ka = max((int64_t)klo, k0), kb = min((int64_t)khi, k1 - 1);
    if (ka > covered + 1) for (int t = 0; t < NT; ++t) emit((uint32_t)t, covered + 1, (uint32_t)t);   // nothing live: the types' no-call entries
    if (tail.w & kWalkHeavy) {
      for (int64_t k = ka; k <= kb; ++k) { const uint32_t t = a.rtype[k]; if (t != kUntabledType) emit(t, k, a.heavy_base + tail.z + (uint32_t)(k - klo)); }
    } else {
      uint64_t m = wp->aux;
      uint32_t sl = a.light_base + tail.z;
      while (m) { const int t = __builtin_ctzll(m); m &= m - 1; if (t < NT) emit((uint32_t)t, ka, sl); ++sl; }
    }
    covered = kb;
  }
  if (covered < k1 - 1) for (int t = 0; t < NT; ++t) emit((uint32_t)t, covered + 1, (uint32_t)t);
  for (int t = 0; t < NT; ++t) emit((uint32_t)t, (int64_t)INT32_MAX, (uint32_t)t);
  if (!FILL) for (int t = 0; t < NT; ++t) table[(int64_t)t * a.nrows] = cur[t][lane];
}

// The (record, sample) matrix for k_assemble_write / the BCF kernels, from a PieceWalker per lane: what k_assemble_size leaves, at half
// its instructions per step (the kernels of this file are bound by the instructions they issue: ~4.4 cycles of a SIMD each) - one
// compare where nothing changes, no position arithmetic, no incidence look-up, no scan (the sizes come from k_size2).  The descriptor
// of a lane that changes at record jj + 1 is requested during step jj and taken at the head of step jj + 1.
__global__ void __launch_bounds__(kAsmRows)
k_fill2(PieceCtx a, const uint2* __restrict__ desc, const int32_t* __restrict__ order, int64_t n, int nchunks, int run, uint2* __restrict__ resolved, int64_t resolved_base) {
  const int64_t unit = xcd_aware_unit<1>(((n + run - 1) / run) * (int64_t)nchunks);
  if (unit < 0) return;
  const int64_t ib = (unit / nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x;
  const int32_t r = ch * kAsmRows + lane;
  const bool has_row = r < a.nrows;
  PieceWalker w;
  w.next = a.walk; w.valid_until = INT32_MAX; w.W = u32x4{0, 0, 0, 0}; w.Wn = u32x4{0, 0, 0, 0}; w.Wnaux = 0;
  uint64_t below = 0;
  uint32_t prev_t = 0xFFFFFFFFu;
  uint2 cur = make_uint2(0, 0), nxt = make_uint2(0, 0);
  bool changed = false;
  auto request = [&](int32_t k, uint32_t t) {
    if (t != prev_t) {                                      // uniform: the run starts, or its records change type
      prev_t = t;
      below = t != kUntabledType ? (1ull << t) - 1ull : 0ull;
      if (has_row) w.init(a, r, k, below);
    }
    changed = has_row && k > w.valid_until;
    if (changed) { uint32_t nq; nxt = desc[w.move(a, k, t, below, r, 0u, nq)]; }
  };
  for (int64_t i0 = ib; i0 < ie; i0 += 64) {                // uniform
    const int cnt = (int)min((int64_t)64, ie - i0);
    int32_t my_k = 0; uint32_t my_t = 0;
    if (lane < cnt) { my_k = order[i0 + lane]; my_t = a.rtype[my_k]; }
    request(__builtin_amdgcn_readlane(my_k, 0), (uint32_t)__builtin_amdgcn_readlane((int)my_t, 0));
    for (int jj = 0; jj < cnt; ++jj) {                      // uniform
      if (changed) cur = nxt;
      const int64_t k = __builtin_amdgcn_readlane(my_k, jj);
      uint2* const row = resolved + ((k - resolved_base) * nchunks + ch) * kAsmRows;
      if (jj + 1 < cnt) request(__builtin_amdgcn_readlane(my_k, jj + 1), (uint32_t)__builtin_amdgcn_readlane((int)my_t, jj + 1));
      else changed = false;
      row[lane] = cur;
    }
  }
}

// 16 bytes into the registers the lane's text already lives in, as an instruction the compiler does not know to be a load: left to
// itself it gives a text requested ahead (below) registers of its own next to the text still in use - 181 instead of 97 for the
// kernel, two wavefronts per SIMD instead of four.  The price: the compiler does not wait for these loads either; text_arrived()
// does, and ties the wait to the registers so that nothing that reads them can be moved in front of it.
template <int OFF> __device__ __forceinline__ void load16_in_place(u32x4& d, const char* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2 ; in-place text" : "+v"(d) : "v"(p), "n"(OFF) : "memory");   // (the marker: tests/tools/check_inplace_loads.py)
}
template <int N> struct TextRegs { u32x4 x[N]; };
__device__ __forceinline__ void text_arrived() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// The same wait where `behind` (uniform) page-store instructions have been issued SINCE the loads: gfx9 counts loads and stores in one
// in-order counter, so "at most `behind` outstanding" = the loads have landed, and the stores - whose acknowledgement nobody needs - keep
// flying.  (vmcnt(0) here would wait for the acknowledgement of stores issued a moment ago: a full trip to memory on nearly every record,
// which is exactly what asking for the texts a record ahead is meant to avoid.)  `behind` may be smaller than the truth, never larger.
__device__ __forceinline__ void texts_arrived_in_front_of(uint32_t behind) {
  switch (behind) {
    case 0: asm volatile("s_waitcnt vmcnt(0) ; texts arrived" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1) ; texts arrived" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2) ; texts arrived" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3) ; texts arrived" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4) ; texts arrived" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5) ; texts arrived" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(6) ; texts arrived" ::: "memory"); break;
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ uint4 as_uint4(const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); }
// Page assembly without the matrix: k_assemble_write's image / flush loop, fed by a PieceWalker per lane.
template <int WAVES, int kWaveLds> __global__ void __launch_bounds__(kAsmRows * WAVES)
k_write2(PieceCtx a, const char* __restrict__ pool, const char* __restrict__ pool_ovf, const int32_t* __restrict__ order, int64_t n, int nchunks, int run,
         const uint64_t* __restrict__ chunk_off, uint64_t page_base, char* __restrict__ arena, int xcd_aware) {
  const int dbg = xcd_aware >> 1;                            // timing experiments only (GDBAMD_W2_DBG): 1 = no walk behind a run's first record, 2 = slots are named but not fetched, 32 = no request ahead
  const int64_t unit = xcd_aware_unit<WAVES>(((n + run - 1) / run) * (int64_t)nchunks, xcd_aware & 1);
  if (unit < 0) return;
  const int64_t ib = (unit / nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x & 63;
  const int32_t r = ch * kAsmRows + lane;
  const bool has_row = r < a.nrows;
  __shared__ __attribute__((aligned(16))) char lds_all[WAVES][kWaveLds + 16 + kScrapBytes];
  char* const lds_buf = &lds_all[threadIdx.x >> 6][0];
  uint32_t cur_where = 0, cur_len = 0;                      // the tag of the lane's present slot (an inline text is wholly in txt)
  TextRegs<kText2Chunks> txt;
#pragma unroll
  for (int q = 0; q < kText2Chunks; ++q) txt.x[q] = u32x4{0, 0, 0, 0};
  PieceWalker w;
  w.next = a.walk; w.valid_until = INT32_MAX; w.W = u32x4{0, 0, 0, 0}; w.Wn = u32x4{0, 0, 0, 0}; w.Wnaux = 0;
  uint64_t below = 0;                                       // record types in front of the run's one (bit mask: a plain cell's slot = its base + how many of them it meets)
  uint32_t prev_t = 0xFFFFFFFFu, nocall_chunks = 0;
  uint64_t pending = 0;                                     // lanes that have asked for a slot's line and not looked at it yet
  uint32_t behind = 0;                                      // (uniform) page-store instructions issued since the last request
  // A step has two halves.  REQUEST (record jj): samples whose entry changes there name their slot and ask for its line - nothing
  // waits.  TAKE + build + flush: the tags are read (here the loads are waited for), the image is built and written.  The request
  // for record jj + 1 is made between the build and the flush of record jj: the texts of the lanes that change are dead once the
  // image stands, and the random-access latency of the new ones (what this kernel would otherwise sit out on nearly every
  // record: ~1.6 us against 2.5 us for everything else in the step) passes while the image is flushed.
  auto request = [&](int32_t k, uint32_t t) {
    if (t != prev_t) {                                      // uniform: the run starts, or its records change type
      prev_t = t;
      nocall_chunks = t != kUntabledType ? inline_chunks(a.slot_len[t]) : 0u;
      below = t != kUntabledType ? (1ull << t) - 1ull : 0ull;
      if (has_row) w.init(a, r, k, below);
    }
    const bool need = has_row && k > w.valid_until && !((dbg & 2) && w.valid_until != INT32_MIN);
    if (need) {                                             // the sample's entry changes here: name the slot, ask for its tag and text
      uint32_t nq;
      const uint32_t sl = w.move(a, k, t, below, r, nocall_chunks, nq);
      if (dbg & 1) w.valid_until = INT32_MAX;
      const char* src = pool + (size_t)((dbg & 64) && w.valid_until != INT32_MIN && cur_len ? t : sl) * kSlotStride;   // (64: every later fetch reads the type's no-call slot: a line that is always in the cache)
      // (the kernel is bound by the instructions it issues: four chunks without asking, the other three behind one test - not a test per chunk)
      load16_in_place<112>(txt.x[7], src);
      load16_in_place<0>(txt.x[0], src);
      load16_in_place<16>(txt.x[1], src);
      load16_in_place<32>(txt.x[2], src);
      load16_in_place<48>(txt.x[3], src);
      if (nq > 4u) {
        load16_in_place<64>(txt.x[4], src);
        load16_in_place<80>(txt.x[5], src);
        load16_in_place<96>(txt.x[6], src);
      }
    }
    pending = __ballot(need);
    behind = 0;
  };
  // the image of one pass (<= kWaveLds bytes of a record's chunk, with the destination's alignment) -> page
  auto flush = [&](char* gdst, uint32_t al, uint32_t pass_total) {
    const char* img = lds_buf + al;
    uint32_t head = (16u - al) & 15u;
    if (head > pass_total) head = pass_total;
    const uint32_t nwords = (pass_total - head) >> 4;
    const uint32_t tail_at = head + (nwords << 4);
    const uint4* lsrc = reinterpret_cast<const uint4*>(img + head);
    uint4* gw = reinterpret_cast<uint4*>(gdst + head);
    for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) gw[wq] = lsrc[wq];
    behind += (nwords + (uint32_t)kAsmRows - 1u) / (uint32_t)kAsmRows;   // (one store instruction per trip of the loop)
    if (head | (pass_total - tail_at)) {   // (uniform, so that the count is exact: one too few and the wait is for the first store of THIS flush)
      // the bytes in front of the first and behind the last 16-byte word in ONE store instruction: lanes 0-15 head, 16-31 tail
      const uint32_t li = (uint32_t)lane & 15u;
      const bool is_tail = (lane & 16) != 0;
      const uint32_t at = is_tail ? tail_at + li : li;
      if (lane < 32 && li < (is_tail ? pass_total - tail_at : head)) gdst[at] = img[at];
      ++behind;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
  for (int64_t i0 = ib; i0 < ie; i0 += 64) {                // uniform
    const int cnt = (int)min((int64_t)64, ie - i0);
    int32_t my_k = 0; int64_t my_dst = 0; uint32_t my_t = 0;
    if (lane < cnt) {
      my_k = order[i0 + lane];
      my_t = a.rtype[my_k];
      my_dst = (int64_t)(chunk_off[(int64_t)my_k * nchunks + ch] - page_base) + (ch == 0 ? a.prefix_len[my_k] : 0u);
    }
    request(__builtin_amdgcn_readlane(my_k, 0), (uint32_t)__builtin_amdgcn_readlane((int)my_t, 0));
    for (int jj = 0; jj < cnt; ++jj) {                      // uniform
      if (pending) {                                        // uniform
        if (dbg & 32) text_arrived(); else texts_arrived_in_front_of(behind);
        if (((pending >> lane) & 1ull) && !((dbg & 64) && cur_len)) {   // take the tag
          cur_where = txt.x[kText2Chunks - 1].z;
          cur_len = txt.x[kText2Chunks - 1].w;
          if (cur_where & kOverflowBit) {                   // longer than an inline slot: the text lies in the overflow pool
            const char* src = pool_ovf + (size_t)(cur_where & ~kOverflowBit) * 16;
#pragma unroll
            for (int q = 0; q < kText2Chunks; ++q) { const uint4 v = load_chunk(src, q, cur_len); txt.x[q] = u32x4{v.x, v.y, v.z, v.w}; }
          }
        }
        pending = 0;
      }
      const uint32_t len = cur_len;
      const uint32_t inc = wave_inclusive_scan_dpp(len);
      const uint32_t excl = inc - len;
      const uint32_t total = wave_total(inc);
      char* const grec = arena + readlane64(my_dst, jj);
      // The chunk normally fits one LDS image.  Wider chunks (long PL vectors) go out in passes over consecutive lane ranges that
      // fit; only a single entry larger than the image is copied straight from the pool.  The flush of a pass is put off until the
      // next one is about to build (the image is about to be overwritten) or, for the last one, until the next record's request is out.
      uint32_t l0 = 0, base_off = 0;
      bool built = false;
      char* f_dst = grec; uint32_t f_al = 0, f_total = 0;
      while (base_off < total) {                            // uniform
        if (built) { flush(f_dst, f_al, f_total); built = false; }
        char* gdst = grec + base_off;
        const uint32_t al = (uint32_t)((uintptr_t)gdst & 15u);
        const bool fits = (uint32_t)lane >= l0 && al + (inc - base_off) <= (uint32_t)kWaveLds;
        const uint64_t fit_mask = __ballot(fits) >> l0;     // inc is non-decreasing: the fitting lanes are a run starting at l0
        const uint32_t len_l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(l0 < 64u ? l0 : 63u));
        if (!(fit_mask & 1ull) || len_l0 > (uint32_t)kCooperativeEntry) {   // a long text (or one that exceeds the image): the whole wavefront copies it
          const char* big_src = pool_ovf + (size_t)((uint32_t)__builtin_amdgcn_readlane((int)cur_where, (int)l0) & ~kOverflowBit) * 16;   // (only overflow texts are this long)
          const uint32_t big_len = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)l0);
          uint32_t head = (16u - al) & 15u;                 // bytes up to the first 16-byte boundary of the destination
          if (head > big_len) head = big_len;
          if ((uint32_t)lane < head) gdst[lane] = big_src[lane];
          const uint32_t nwords = (big_len - head) >> 4;
          uint4* gw = reinterpret_cast<uint4*>(gdst + head);
          for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) {   // aligned 16-byte stores, source words assembled from two aligned loads
            const char* sp = big_src + head + ((size_t)wq << 4);
            const uint4 lo = *reinterpret_cast<const uint4*>((uintptr_t)sp & ~(uintptr_t)15);
            const uint4 hi = *reinterpret_cast<const uint4*>(((uintptr_t)sp & ~(uintptr_t)15) + 16);
            const uint32_t sh = (uint32_t)((uintptr_t)sp & 15u);
            const uint32_t ww[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t x = ww[(sh >> 2) + q], y = ww[(sh >> 2) + q + 1 < 8 ? (sh >> 2) + q + 1 : 7];
              o[q] = __builtin_amdgcn_alignbyte(y, x, sh & 3u);
            }
            gw[wq] = make_uint4(o[0], o[1], o[2], o[3]);
          }
          const uint32_t tail_at = head + (nwords << 4);
          if ((uint32_t)lane < big_len - tail_at) gdst[tail_at + lane] = big_src[tail_at + lane];
          base_off = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)l0);
          ++l0;
          continue;
        }
        const uint32_t nfit = fit_mask == ~0ull ? 64u - l0 : (uint32_t)__builtin_ctzll(~fit_mask);
        const uint32_t l1 = l0 + nfit;
        const uint32_t pass_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(l1 - 1)) - base_off;
        const bool mine = (uint32_t)lane >= l0 && (uint32_t)lane < l1;
        SlotCopy cp;
        cp.begin((gdb_lds_char*)lds_buf + al + (excl - base_off), mine ? len : 0u, (gdb_lds_char*)lds_buf + kWaveLds + 16 + 4 * lane);
        cp.chunk(0, as_uint4(txt.x[0]));
#pragma unroll
        for (int q = 1; q < kText2Chunks; ++q) if (__any((int)cp.needs(q))) cp.chunk(q, as_uint4(txt.x[q]));
        if (__any((int)cp.needs(kText2Chunks))) {           // texts longer than the registers hold: overflow texts
          const char* cur_src = pool_ovf + (size_t)(cur_where & ~kOverflowBit) * 16;
          for (uint32_t q = kText2Chunks; __any((int)cp.needs(q)); ++q) cp.chunk(q, load_chunk(cur_src, q, mine ? len : 0u));
        }
        cp.finish();
        // One wavefront per workgroup: the LDS unit runs its instructions in order, so the image only needs a compiler-level
        // fence (wavefront scope emits no s_waitcnt: outstanding loads and page stores keep flying across records).
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        built = true; f_dst = gdst; f_al = al; f_total = pass_total;
        l0 = l1;
        base_off += pass_total;
      }
      if (jj + 1 < cnt) request(__builtin_amdgcn_readlane(my_k, jj + 1), (uint32_t)__builtin_amdgcn_readlane((int)my_t, jj + 1));
      if (built) flush(f_dst, f_al, f_total);
    }
  }
}

// Page assembly from the piece lists (k_plist): k_write2's image / flush loop and its requests ahead; a lane's transition is "take the next entry,
// ask for the one behind it, ask for the text" - the entry says where the text lies and how long it is, nothing is decided here.
template <int WAVES, int kWaveLds> __global__ void __launch_bounds__(kAsmRows * WAVES)
k_write3(PieceCtx a, const PieceEntry* __restrict__ plist, const uint32_t* __restrict__ pofs, int NT, const char* __restrict__ pool, const char* __restrict__ pool_ovf, const int32_t* __restrict__ order, int64_t n, int nchunks, int run,
         const uint64_t* __restrict__ chunk_off, uint64_t page_base, char* __restrict__ arena, int xcd_aware, unsigned long long* dbg_counters) {
  const int dbg = xcd_aware >> 1;                            // timing experiments only (GDBAMD_W2_DBG): 1 = no walk behind a run's first record, 2 = slots are named but not fetched, 32 = no request ahead
  const int64_t unit = xcd_aware_unit<WAVES>(((n + run - 1) / run) * (int64_t)nchunks, xcd_aware & 1);
  if (unit < 0) return;
  const int64_t ib = (unit / nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x & 63;
  const int32_t r = ch * kAsmRows + lane;
  const bool has_row = r < a.nrows;
  __shared__ __attribute__((aligned(16))) char lds_all[WAVES][kWaveLds + 16 + kScrapBytes];
  char* const lds_buf = &lds_all[threadIdx.x >> 6][0];
  uint32_t cur_where = 0, cur_len = 0;                      // the tag of the lane's present slot (an inline text is wholly in txt)
  TextRegs<kText2Chunks> txt;
#pragma unroll
  for (int q = 0; q < kText2Chunks; ++q) txt.x[q] = u32x4{0, 0, 0, 0};
  // the lane's place in its piece list: nx = the next entry (x = first record it holds from, y = where its text lies, z = its length), requested
  // when the lane took the one in front of it; ent = the entry behind nx
  const PieceEntry* ent = plist;
  u32x4 nx = u32x4{(uint32_t)INT32_MAX, 0u, 0u, (uint32_t)INT32_MAX};
  int32_t cur_until = INT32_MAX;                            // the first record the lane's present entry does not hold any more
  uint32_t prev_t = 0xFFFFFFFFu;
  int32_t prev_sb = -1;
  uint32_t behind = 0;                                      // (uniform) page-store instructions issued since the last request
  uint64_t t_wait = 0, n_wait = 0, t_begin = (dbg & 64) ? __builtin_readcyclecounter() : 0;   // (GDBAMD_W2_DBG=64: cycle accounting into the first words of the arena's tail - timing runs only)
  // A step has two halves.  REQUEST (record jj): samples whose entry changes there name their slot and ask for its line - nothing
  // waits.  TAKE + build + flush: the tags are read (here the loads are waited for), the image is built and written.  The request
  // for record jj + 1 is made between the build and the flush of record jj: the texts of the lanes that change are dead once the
  // image stands, and the random-access latency of the new ones (what this kernel would otherwise sit out on nearly every
  // record: ~1.6 us against 2.5 us for everything else in the step) passes while the image is flushed.
  // (real moves in front of the load, as in PieceWalker::advance: the entry being taken leaves nx before the next one is loaded into it)
  auto take = [&]() {
    uint32_t w_, l_, u_;
    asm volatile("v_mov_b32 %0, %1" : "=v"(w_) : "v"(nx.y));
    asm volatile("v_mov_b32 %0, %1" : "=v"(l_) : "v"(nx.z));
    asm volatile("v_mov_b32 %0, %1" : "=v"(u_) : "v"(nx.w));
    asm volatile("; the entry's first word stays the entry's" : : "v"(nx.x));   // (unused here - but a register the compiler takes for dead while the load into it is under way costs a wait for that load)
    __builtin_amdgcn_sched_barrier(0);
    load16_in_place<0>(nx, reinterpret_cast<const char*>(ent));   // (in place as well: the compiler would wait for it - and with it for every page store in flight - where the lane next looks at nx)
    ++ent;
    cur_where = w_; cur_len = l_; cur_until = (int32_t)u_;
  };
  auto request = [&](int32_t k, uint32_t t) {
    const int32_t sb = k / kSizeBlock;
    if (t != prev_t || sb != prev_sb) {                     // uniform: the run starts, its records change type or enter the next block of records: another list
      prev_t = t; prev_sb = sb;
      if (has_row) {
        ent = plist + pofs[((int64_t)sb * NT + t) * a.nrows + r];
        load16_in_place<0>(nx, reinterpret_cast<const char*>(ent));
        ++ent;
        text_arrived();                                     // (the wait stays in this rare block)
        cur_until = INT32_MIN;                              // (every record of the list has an entry at or in front of it: the first one is taken below)
      }
    }
    const bool need = has_row && k >= cur_until;
    if (need) {                                             // the sample's entry changes here: take the next one, ask for the one behind it and for the text
      take();
      while (k >= cur_until) { text_arrived(); take(); }    // (entries without a record of this run in between: rare, and only these wait for the entry just asked for)
      const char* src = (cur_where & kOverflowBit) ? pool_ovf + (size_t)(cur_where & ~kOverflowBit) * 16 : pool + (size_t)cur_where * 16;
      load16_in_place<0>(txt.x[0], src);
      load16_in_place<16>(txt.x[1], src);
      load16_in_place<32>(txt.x[2], src);
      if (cur_len > 48u) {
        load16_in_place<48>(txt.x[3], src);
        load16_in_place<64>(txt.x[4], src);
        load16_in_place<80>(txt.x[5], src);
        load16_in_place<96>(txt.x[6], src);
        load16_in_place<112>(txt.x[7], src);
      }
    }
    behind = 0;
  };
  // the image of one pass (<= kWaveLds bytes of a record's chunk, with the destination's alignment) -> page
  auto flush = [&](char* gdst, uint32_t al, uint32_t pass_total) {
    const char* img = lds_buf + al;
    uint32_t head = (16u - al) & 15u;
    if (head > pass_total) head = pass_total;
    const uint32_t nwords = (pass_total - head) >> 4;
    const uint32_t tail_at = head + (nwords << 4);
    const uint4* lsrc = reinterpret_cast<const uint4*>(img + head);
    uint4* gw = reinterpret_cast<uint4*>(gdst + head);
    for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) gw[wq] = lsrc[wq];
    behind += (nwords + (uint32_t)kAsmRows - 1u) / (uint32_t)kAsmRows;   // (one store instruction per trip of the loop)
    if (head | (pass_total - tail_at)) {   // (uniform, so that the count is exact: one too few and the wait is for the first store of THIS flush)
      // the bytes in front of the first and behind the last 16-byte word in ONE store instruction: lanes 0-15 head, 16-31 tail
      const uint32_t li = (uint32_t)lane & 15u;
      const bool is_tail = (lane & 16) != 0;
      const uint32_t at = is_tail ? tail_at + li : li;
      if (lane < 32 && li < (is_tail ? pass_total - tail_at : head)) gdst[at] = img[at];
      ++behind;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
  for (int64_t i0 = ib; i0 < ie; i0 += 64) {                // uniform
    const int cnt = (int)min((int64_t)64, ie - i0);
    int32_t my_k = 0; int64_t my_dst = 0; uint32_t my_t = 0;
    if (lane < cnt) {
      my_k = order[i0 + lane];
      my_t = a.rtype[my_k];
      my_dst = (int64_t)(chunk_off[(int64_t)my_k * nchunks + ch] - page_base) + (ch == 0 ? a.prefix_len[my_k] : 0u);
    }
    request(__builtin_amdgcn_readlane(my_k, 0), (uint32_t)__builtin_amdgcn_readlane((int)my_t, 0));
    for (int jj = 0; jj < cnt; ++jj) {                      // uniform
      // the texts (and the entries) asked for during the previous step: every record waits - where no sample changed, for nothing
      if (dbg & 64) { const uint64_t t0 = __builtin_readcyclecounter(); texts_arrived_in_front_of(behind); t_wait += __builtin_readcyclecounter() - t0; ++n_wait; }
      else if (dbg & 32) text_arrived();
      else texts_arrived_in_front_of(behind);
      const uint32_t len = cur_len;
      const uint32_t inc = wave_inclusive_scan_dpp(len);
      const uint32_t excl = inc - len;
      const uint32_t total = wave_total(inc);
      char* const grec = arena + readlane64(my_dst, jj);
      // The chunk normally fits one LDS image.  Wider chunks (long PL vectors) go out in passes over consecutive lane ranges that
      // fit; only a single entry larger than the image is copied straight from the pool.  The flush of a pass is put off until the
      // next one is about to build (the image is about to be overwritten) or, for the last one, until the next record's request is out.
      uint32_t l0 = 0, base_off = 0;
      bool built = false;
      char* f_dst = grec; uint32_t f_al = 0, f_total = 0;
      while (base_off < total) {                            // uniform
        if (built) { flush(f_dst, f_al, f_total); built = false; }
        char* gdst = grec + base_off;
        const uint32_t al = (uint32_t)((uintptr_t)gdst & 15u);
        const bool fits = (uint32_t)lane >= l0 && al + (inc - base_off) <= (uint32_t)kWaveLds;
        const uint64_t fit_mask = __ballot(fits) >> l0;     // inc is non-decreasing: the fitting lanes are a run starting at l0
        const uint32_t len_l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(l0 < 64u ? l0 : 63u));
        if (!(fit_mask & 1ull) || len_l0 > (uint32_t)kCooperativeEntry) {   // a long text (or one that exceeds the image): the whole wavefront copies it
          const char* big_src = pool_ovf + (size_t)((uint32_t)__builtin_amdgcn_readlane((int)cur_where, (int)l0) & ~kOverflowBit) * 16;   // (only overflow texts are this long)
          const uint32_t big_len = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)l0);
          uint32_t head = (16u - al) & 15u;                 // bytes up to the first 16-byte boundary of the destination
          if (head > big_len) head = big_len;
          if ((uint32_t)lane < head) gdst[lane] = big_src[lane];
          const uint32_t nwords = (big_len - head) >> 4;
          uint4* gw = reinterpret_cast<uint4*>(gdst + head);
          for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) {   // aligned 16-byte stores, source words assembled from two aligned loads
            const char* sp = big_src + head + ((size_t)wq << 4);
            const uint4 lo = *reinterpret_cast<const uint4*>((uintptr_t)sp & ~(uintptr_t)15);
            const uint4 hi = *reinterpret_cast<const uint4*>(((uintptr_t)sp & ~(uintptr_t)15) + 16);
            const uint32_t sh = (uint32_t)((uintptr_t)sp & 15u);
            const uint32_t ww[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t x = ww[(sh >> 2) + q], y = ww[(sh >> 2) + q + 1 < 8 ? (sh >> 2) + q + 1 : 7];
              o[q] = __builtin_amdgcn_alignbyte(y, x, sh & 3u);
            }
            gw[wq] = make_uint4(o[0], o[1], o[2], o[3]);
          }
          const uint32_t tail_at = head + (nwords << 4);
          if ((uint32_t)lane < big_len - tail_at) gdst[tail_at + lane] = big_src[tail_at + lane];
          base_off = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)l0);
          ++l0;
          continue;
        }
        const uint32_t nfit = fit_mask == ~0ull ? 64u - l0 : (uint32_t)__builtin_ctzll(~fit_mask);
        const uint32_t l1 = l0 + nfit;
        const uint32_t pass_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(l1 - 1)) - base_off;
        const bool mine = (uint32_t)lane >= l0 && (uint32_t)lane < l1;
        SlotCopy cp;
        cp.begin((gdb_lds_char*)lds_buf + al + (excl - base_off), mine ? len : 0u, (gdb_lds_char*)lds_buf + kWaveLds + 16 + 4 * lane);
        cp.chunk(0, as_uint4(txt.x[0]));
#pragma unroll
        for (int q = 1; q < kText2Chunks; ++q) if (__any((int)cp.needs(q))) cp.chunk(q, as_uint4(txt.x[q]));
        if (__any((int)cp.needs(kText2Chunks))) {           // texts longer than the registers hold: overflow texts
          const char* cur_src = pool_ovf + (size_t)(cur_where & ~kOverflowBit) * 16;
          for (uint32_t q = kText2Chunks; __any((int)cp.needs(q)); ++q) cp.chunk(q, load_chunk(cur_src, q, mine ? len : 0u));
        }
        cp.finish();
        // One wavefront per workgroup: the LDS unit runs its instructions in order, so the image only needs a compiler-level
        // fence (wavefront scope emits no s_waitcnt: outstanding loads and page stores keep flying across records).
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        built = true; f_dst = gdst; f_al = al; f_total = pass_total;
        l0 = l1;
        base_off += pass_total;
      }
      if (jj + 1 < cnt) request(__builtin_amdgcn_readlane(my_k, jj + 1), (uint32_t)__builtin_amdgcn_readlane((int)my_t, jj + 1));
      if (built) flush(f_dst, f_al, f_total);
    }
  }
  if ((dbg & 64) && lane == 0) {
    unsigned long long* c = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(pofs)) ;   // (the list offsets are not needed any more by this wavefront... other wavefronts still read them: use the tail instead)
    (void)c;
    unsigned long long* acc = dbg_counters;
    atomicAdd(&acc[0], (unsigned long long)(__builtin_readcyclecounter() - t_begin));
    atomicAdd(&acc[1], (unsigned long long)t_wait);
    atomicAdd(&acc[2], (unsigned long long)n_wait);
    atomicAdd(&acc[3], (unsigned long long)(ie - ib));
  }
}

// ---- page assembly from a sparse change list instead of the dense (record, sample) matrix ----------------------------------------
// Along a run of same-type records a sample keeps its slot until its live cell changes: ~2 of the 64 lanes of a chunk per record.
// The sizing pass therefore leaves, per (run, chunk), the first record's row (64 x (off16, len)) and then only the CHANGES
// (step, lane, off16, len), 8 bytes each - a twelfth of the matrix traffic on c2.  More important is what the page pass does
// with them: the change list is wave-uniform data, so it and the texts of the changed lanes are fetched with SCALAR loads
// (s_load, counted by lgkmcnt) and moved into the lanes' registers with v_writelane.  The steady-state loop then has no vector
// load left: gfx9 has one in-order vmcnt for loads and stores, and every vector load in the loop made the wavefront wait for
// the acknowledgement of the previous record's page stores (~75 % of its cycles); now the stores stream.
struct EventBuf {
  uint2* init;       // [blocks * 64]        row of the first record of every (run, chunk)
  uint2* ev;         // [blocks * run * 64]  changes: x = off16 (overflow bit included), y = len (20 bits) | lane << 20 | step << 26
  uint32_t* count;   // [blocks]
  int32_t run;
};
constexpr uint32_t kEvLenBits = 20, kEvLenMask = (1u << kEvLenBits) - 1u;

__global__ void __launch_bounds__(kAsmRows)
k_assemble_size_ev(AsmCtx a, const int32_t* __restrict__ order, int64_t n, int32_t N, int nchunks, int run, uint64_t* __restrict__ chunk_size, EventBuf eb, uint32_t* err) {
  const int64_t ib = (int64_t)(blockIdx.x / (unsigned)nchunks) * run;      // (run <= 64: one batch of per-lane record scalars)
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(blockIdx.x % (unsigned)nchunks);
  const int lane = threadIdx.x;
  const int32_t r = ch * kAsmRows + lane;
  const int64_t b = blockIdx.x;
  uint2* __restrict__ my_ev = eb.ev + b * (int64_t)run * kAsmRows;
  uint32_t nev = 0;
  SlotWalker w;
  int64_t prev_k = INT64_MAX;
  uint32_t cur_slot = kNoSlot;
  uint2 cur = make_uint2(0, 0);
  const int cnt = (int)(ie - ib);
  int32_t my_k = 0; int64_t my_s = 0; uint32_t my_t = 0; uint64_t my_total = 0;
  if (lane < cnt) {
    my_k = order[ib + lane];
    my_s = a.rec_start[my_k];
    my_t = a.rtype[my_k];
    if (ch == 0) my_total = a.prefix_len[my_k];
    if (ch == nchunks - 1) my_total += 1;                 // '\n'
  }
  for (int jj = 0; jj < cnt; ++jj) {                      // uniform
    const int64_t k = __builtin_amdgcn_readlane(my_k, jj);
    const int64_t s = readlane64(my_s, jj);
    const uint32_t t = (uint32_t)__builtin_amdgcn_readlane((int)my_t, jj);
    uint2 d = make_uint2(0, 0);
    bool changed = false;
    if (r < N) {
      if (k < prev_k) w.init(a, r, s); else w.advance(a, s);  // uniform branch: a new type restarts at its first record
      const uint32_t sl = w.slot(a, k, s, t, r);
      if (sl != cur_slot) { cur_slot = sl; const uint2 nd = a.desc[sl]; changed = nd.x != cur.x || nd.y != cur.y; cur = nd; }
      d = cur;
    }
    prev_k = k;
    if (jj == 0) eb.init[b * kAsmRows + lane] = d;
    else {
      const uint64_t m = __ballot(changed);
      if (m) {
        if (changed) {
          const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          my_ev[nev + rank] = make_uint2(d.x, (d.y & kEvLenMask) | ((uint32_t)lane << kEvLenBits) | ((uint32_t)jj << 26));
          if (d.y > kEvLenMask) atomicOr(err, (uint32_t)GDB_ERR_INTERNAL);
        }
        nev += (uint32_t)__popcll(m);
      }
    }
    const uint32_t total = wave_total(wave_inclusive_scan_dpp(d.y));
    if (lane == jj) my_total += total;
  }
  if (lane < cnt) chunk_size[(int64_t)my_k * nchunks + ch] = my_total;
  if (lane == 0) eb.count[b] = nev;
}

// scalar memory reads the compiler does not know about: the values are tied to the wait that makes them valid
__device__ __forceinline__ u32x4 sload_x4(const void* p) { u32x4 v; asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory"); return v; }
__device__ __forceinline__ uint64_t sload_x2(const void* p) { uint64_t v; asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(v) : "s"(p) : "memory"); return v; }
__device__ __forceinline__ void swait(uint64_t& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a) : : "memory"); }
__device__ __forceinline__ void swait(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e) : : "memory"); }
// v_writelane with an SGPR value takes its lane select from M0 (one scalar operand per VALU instruction on gfx9)
// (M0 is saved and restored: the compiler keeps it out of its own allocation but may rely on its value)
__device__ __forceinline__ uint32_t wlane(uint32_t sval, uint32_t l, uint32_t old) {
  uint32_t keep;
  asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(old), "=&s"(keep) : "s"(sval), "s"(l));
  return old;
}
__device__ __forceinline__ void wlane4(uint4& dst, const u32x4& sv, uint32_t l) {
  uint32_t keep;
  asm volatile("s_mov_b32 %4, m0\n\ts_mov_b32 m0, %9\n\tv_writelane_b32 %0, %5, m0\n\tv_writelane_b32 %1, %6, m0\n\tv_writelane_b32 %2, %7, m0\n\tv_writelane_b32 %3, %8, m0\n\ts_mov_b32 m0, %4"
               : "+v"(dst.x), "+v"(dst.y), "+v"(dst.z), "+v"(dst.w), "=&s"(keep) : "s"(sv.x), "s"(sv.y), "s"(sv.z), "s"(sv.w), "s"(l));
}

template <int WAVES> __global__ void __launch_bounds__(kAsmRows * WAVES)
k_assemble_write_ev(const char* __restrict__ pool, const char* __restrict__ pool_ovf, const uint32_t* __restrict__ prefix_len, EventBuf eb, int64_t b0,
                    const int32_t* __restrict__ order, int64_t n, int nchunks, int run, const uint64_t* __restrict__ chunk_off, uint64_t page_base, char* __restrict__ arena) {
  const int64_t unit = xcd_aware_unit<WAVES>(((n + run - 1) / run) * (int64_t)nchunks);
  if (unit < 0) return;
  const int64_t ib = (unit / nchunks) * run;
  const int64_t ie = min(n, ib + (int64_t)run);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x & 63;
  const int64_t b = b0 + unit;
  __shared__ __attribute__((aligned(16))) char lds_all[WAVES][kWaveLds + 16 + kScrapBytes];
  char* const lds_buf = &lds_all[threadIdx.x >> 6][0];
  const int cnt = (int)(ie - ib);
  int32_t my_k = 0; int64_t my_dst = 0;
  if (lane < cnt) {
    my_k = order[ib + lane];
    my_dst = (int64_t)(chunk_off[(int64_t)my_k * nchunks + ch] - page_base) + (ch == 0 ? prefix_len[my_k] : 0u);
  }
  // state of the first record: one coalesced row + the texts (vector loads: nothing of this wavefront is in flight yet)
  const uint2 d0 = eb.init[b * kAsmRows + lane];
  uint32_t len = d0.y, cur_off = d0.x;
  SlotText txt;
  {
    const char* src = ((cur_off & kOverflowBit) ? pool_ovf : pool) + (size_t)(cur_off & ~kOverflowBit) * 16;
#pragma unroll
    for (int q = 0; q < kTextChunks; ++q) txt.x[q] = load_chunk(src, q, len);
  }
  const uint32_t nev = (uint32_t)__builtin_amdgcn_readfirstlane((int)eb.count[b]);
  const uint2* __restrict__ evp = eb.ev + b * (int64_t)run * kAsmRows;
  // the change list, 64 entries at a time: one per lane (a coalesced load), consumed in order with v_readlane.  (The first version
  // took every entry and its text with scalar loads - s_load_dwordx4 x 5 + s_waitcnt per change - and was latency-bound.)
  uint32_t ebase = 0, consumed = 0;
  uint2 evr = make_uint2(0u, 0xFFFFFFFFu);
  if ((uint32_t)lane < nev) evr = evp[lane];
  for (int jj = 0; jj < cnt; ++jj) {                        // uniform
    // ---- the changes of this record --------------------------------------------------------------------------------------------------
    bool changed = false;
    for (;;) {                                              // uniform
      if (ebase + consumed >= nev) break;
      if (consumed == 64u) {
        ebase += 64u; consumed = 0;
        evr = make_uint2(0u, 0xFFFFFFFFu);
        if (ebase + (uint32_t)lane < nev) evr = evp[ebase + lane];
      }
      const uint32_t ey = (uint32_t)__builtin_amdgcn_readlane((int)evr.y, (int)consumed);
      if ((ey >> 26) != (uint32_t)jj) break;
      const uint32_t ex = (uint32_t)__builtin_amdgcn_readlane((int)evr.x, (int)consumed);
      if ((uint32_t)lane == ((ey >> kEvLenBits) & 63u)) {
        len = ey & kEvLenMask;
        cur_off = ex;
        changed = true;
      }
      ++consumed;
    }
    if (changed) {      // the texts of all lanes that changed in this record: one group of loads (one exposed latency)
      const char* src = ((cur_off & kOverflowBit) ? pool_ovf : pool) + (size_t)(cur_off & ~kOverflowBit) * 16;
#pragma unroll
      for (int q = 0; q < kTextChunks; ++q) txt.x[q] = load_chunk(src, q, len);
    }
    const uint32_t inc = wave_inclusive_scan_dpp(len);
    const uint32_t excl = inc - len;
    const uint32_t total = wave_total(inc);
    if (total == 0) continue;                             // uniform: no FORMAT columns in this record
    char* const grec = arena + readlane64(my_dst, jj);
    const char* cur_src = ((cur_off & kOverflowBit) ? pool_ovf : pool) + (size_t)(cur_off & ~kOverflowBit) * 16;
    uint32_t l0 = 0, base_off = 0;
    while (base_off < total) {                            // uniform
      char* gdst = grec + base_off;
      const uint32_t al = (uint32_t)((uintptr_t)gdst & 15u);
      const bool fits = (uint32_t)lane >= l0 && al + (inc - base_off) <= (uint32_t)kWaveLds;
      const uint64_t fit_mask = __ballot(fits) >> l0;
      const uint32_t len_l0 = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)(l0 < 64u ? l0 : 63u));
      if (!(fit_mask & 1ull) || len_l0 > (uint32_t)kCooperativeEntry) {   // a long text (or one that exceeds the image): the whole wavefront copies it
        const char* big_src = (const char*)(uintptr_t)readlane64((int64_t)(uintptr_t)cur_src, (int)l0);
        const uint32_t big_len = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)l0);
        uint32_t head = (16u - al) & 15u;
        if (head > big_len) head = big_len;
        if ((uint32_t)lane < head) gdst[lane] = big_src[lane];
        const uint32_t nwords = (big_len - head) >> 4;
        uint4* gw = reinterpret_cast<uint4*>(gdst + head);
        for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) {
          const char* sp = big_src + head + ((size_t)wq << 4);
          const uint4 lo = *reinterpret_cast<const uint4*>((uintptr_t)sp & ~(uintptr_t)15);
          const uint4 hi = *reinterpret_cast<const uint4*>(((uintptr_t)sp & ~(uintptr_t)15) + 16);
          const uint32_t sh = (uint32_t)((uintptr_t)sp & 15u);
          const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          uint32_t o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t a = w[(sh >> 2) + q], bb = w[(sh >> 2) + q + 1 < 8 ? (sh >> 2) + q + 1 : 7];
            o[q] = __builtin_amdgcn_alignbyte(bb, a, sh & 3u);
          }
          gw[wq] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        const uint32_t tail_at = head + (nwords << 4);
        if ((uint32_t)lane < big_len - tail_at) gdst[tail_at + lane] = big_src[tail_at + lane];
        base_off = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)l0);
        ++l0;
        continue;
      }
      const uint32_t nfit = fit_mask == ~0ull ? 64u - l0 : (uint32_t)__builtin_ctzll(~fit_mask);
      const uint32_t l1 = l0 + nfit;
      const uint32_t pass_total = (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(l1 - 1)) - base_off;
      const bool mine = (uint32_t)lane >= l0 && (uint32_t)lane < l1;
      SlotCopy cp;
      cp.begin((gdb_lds_char*)lds_buf + al + (excl - base_off), mine ? len : 0u, (gdb_lds_char*)lds_buf + kWaveLds + 16 + 4 * lane);
      cp.chunk(0, txt.x[0]);
#pragma unroll
      for (int q = 1; q < kTextChunks; ++q) if (__any((int)cp.needs(q))) cp.chunk(q, txt.x[q]);
      for (uint32_t q = kTextChunks; __any((int)cp.needs(q)); ++q) cp.chunk(q, load_chunk(cur_src, q, mine ? len : 0u));
      cp.finish();
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      const char* img = lds_buf + al;
      uint32_t head = (16u - al) & 15u;
      if (head > pass_total) head = pass_total;
      const uint32_t nwords = (pass_total - head) >> 4;
      const uint32_t tail_at = head + (nwords << 4);
      if ((uint32_t)lane < head) gdst[lane] = img[lane];
      const uint4* lsrc = reinterpret_cast<const uint4*>(img + head);
      uint4* gw = reinterpret_cast<uint4*>(gdst + head);
      for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) gw[wq] = lsrc[wq];
      if ((uint32_t)lane < pass_total - tail_at) gdst[tail_at + lane] = img[tail_at + lane];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      l0 = l1;
      base_off += pass_total;
    }
  }
}

// ---- BCF2 page assembly ("bu") ------------------------------------------------------------------------------------------------
// The FORMAT block of a BCF record is field-major with a fixed stride per sample, so once the vector length and the integer
// type of every (record, field) are known a sample's values have a fixed place: no scan over variable-length texts.
// k_bcf_field_meta reduces the entries' summaries per (record, 64-sample chunk), k_bcf_layout per record (and sizes the record),
// k_bcf_shared writes l_shared / l_indiv, the shared block and the key + type bytes of every field, k_bcf_write the values.
constexpr int kBcfRun = 256;       // records one wavefront of the BCF kernels takes in a row (same chunk), 64 at a time
  // (every run begins with a full format of its 64 samples and two dependent fetches of per-record data: with runs of 32 that start
  // was most of the kernel)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// maximum over the wavefront of a pair of 16-bit summaries (vector length: maximum; class bits: OR), the same value in every lane.
// A scan-shaped reduction through the DPP network (row shifts, then the row broadcasts of gfx9) instead of six ds_bpermute round trips
// per word: lanes a shift does not reach read 0, the identity of both operations.
__device__ __forceinline__ uint32_t bcf_pair_combine(uint32_t v, uint32_t o) {
  const u16x2 n = __builtin_elementwise_max(__builtin_bit_cast(u16x2, v & 0x3FFF3FFFu), __builtin_bit_cast(u16x2, o & 0x3FFF3FFFu));
  return __builtin_bit_cast(uint32_t, n) | ((v | o) & 0xC000C000u);
}
__device__ __forceinline__ uint32_t bcf_pair_reduce(uint32_t v) {
  v = bcf_pair_combine(v, dpp_row_shr(v, 1));
  v = bcf_pair_combine(v, dpp_row_shr(v, 2));
  v = bcf_pair_combine(v, dpp_row_shr(v, 4));
  v = bcf_pair_combine(v, dpp_row_shr(v, 8));
  v = bcf_pair_combine(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));   // row_bcast:15 into rows 1 and 3
  v = bcf_pair_combine(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));   // row_bcast:31 into rows 2 and 3
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// part[chunk][record][pair word]: a lane's entry rarely changes from one record to the next (a call spans many records), so the
// summary words stay in registers and only the lanes whose entry changed go to memory.  The resolved rows of kMetaBatch records are
// fetched together and then the entry heads those rows name (a load per record made every step wait two memory latencies in a row).
constexpr int kMetaBatch = 4;
__global__ void __launch_bounds__(kAsmRows) k_bcf_field_meta(const ResMatrix resolved, const char* __restrict__ pool, const char* __restrict__ pool_ovf,
                                                           const uint32_t* __restrict__ fmt_mask, const int32_t* __restrict__ order, int64_t P, int nchunks, int F, uint32_t* __restrict__ part) {
  const int64_t unit = xcd_aware_unit<1>(((P + kBcfRun - 1) / kBcfRun) * (int64_t)nchunks);
  if (unit < 0) return;
  const int64_t k0 = (unit / nchunks) * kBcfRun, k1 = min(P, k0 + (int64_t)kBcfRun);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x;
  const int W = (F + 1) >> 1;
  uint32_t held = 0xFFFFFFFFu;        // the entry whose first 16 bytes are in h
  u32x4 h = {0u, 0u, 0u, 0u};
  uint32_t prev_key = 0xFFFFFFFFu, mine = 0;
  int prev_words = -1;
  // records in (block, type) order (`order`, like the text assembly): along a run the samples keep their entries
  int32_t my_k = 0; uint32_t my_mask = 0;
  for (int64_t i0 = k0; i0 < k1; i0 += kMetaBatch) {       // uniform
    if (((i0 - k0) & 63) == 0) { my_k = (i0 + lane < k1) ? order[i0 + lane] : 0; my_mask = (i0 + lane < k1) ? fmt_mask[my_k] : 0u; }
    uint2 t[kMetaBatch]; u32x4 hd[kMetaBatch]; bool ld[kMetaBatch];
#pragma unroll
    for (int b = 0; b < kMetaBatch; ++b) {
      const int64_t i = i0 + b < k1 ? i0 + b : k1 - 1;
      const int64_t k = __builtin_amdgcn_readlane(my_k, (int)((i - k0) & 63));
      t[b] = resolved.load((k * nchunks + ch) * kAsmRows + lane);
    }
    uint32_t will_hold = held;
#pragma unroll
    for (int b = 0; b < kMetaBatch; ++b) {
      ld[b] = t[b].y && t[b].x != will_hold;
      hd[b] = u32x4{0u, 0u, 0u, 0u};
      if (ld[b]) { hd[b] = *reinterpret_cast<const u32x4*>(((t[b].x & kOverflowBit) ? pool_ovf : pool) + (size_t)(t[b].x & ~kOverflowBit) * 16); will_hold = t[b].x; }   // (entries are 16-byte aligned and padded)
    }
#pragma unroll
    for (int b = 0; b < kMetaBatch; ++b) {
      const int64_t i = i0 + b;
      if (i >= k1) break;                                   // uniform
      const int64_t k = __builtin_amdgcn_readlane(my_k, (int)((i - k0) & 63));
      const uint2 d = t[b];
      if (ld[b]) { h = hd[b]; held = d.x; }
      const int words = (__popc((uint32_t)__builtin_amdgcn_readlane((int)my_mask, (int)((i - k0) & 63))) + 1) >> 1;
      uint32_t* const out = part + ((int64_t)ch * P + k) * W;
      // a record whose 64 samples all keep the entries of the record before it (a third of the steps along a type run) has its
      // summary maxima too: the reductions below are skipped
      const uint32_t key = d.y ? d.x : 0xFFFFFFFEu;
      const bool same = words == prev_words && !__any((int)(key != prev_key));
      prev_key = key; prev_words = words;
      if (!same) {
        mine = 0;
        const char* src = ((d.x & kOverflowBit) ? pool_ovf : pool) + (size_t)(d.x & ~kOverflowBit) * 16;
        for (int w0 = 0; w0 < words; w0 += 4) {               // uniform
          u32x4 v = {0u, 0u, 0u, 0u};
          if (d.y) v = w0 == 0 ? h : *reinterpret_cast<const u32x4*>(src + 4 * w0);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (w0 + c >= words) break;
            const uint32_t r = bcf_pair_reduce(v[c]);
            if (lane == w0 + c) mine = r;
          }
        }
      }
      if (lane < words) out[lane] = mine;
    }
  }
}
struct BcfLayout {            // per record
  uint32_t* fmeta;            // [P * F]  bits 0..15 vector length, 16..19 BCF type code, 24..31 bytes of key + type descriptor
  uint32_t* foff;             // [P * F]  offset of the field's key inside the record
  uint32_t* l_indiv;          // [P]
  uint64_t* rec_size;         // [P]
};
__global__ void k_bcf_layout(CombinePlan pl, const uint32_t* __restrict__ part, const uint32_t* __restrict__ fmt_mask, const uint32_t* __restrict__ prefix_len, int64_t P, int nchunks,
                             int F, BcfLayout lay) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= P) return;
  const uint32_t mask = fmt_mask[k];
  const int W = (F + 1) >> 1;
  uint32_t at = 8u + prefix_len[k];              // l_shared, l_indiv, shared block
  const uint32_t indiv_begin = at;
  int q = 0;
  for (int i = 0; i < pl.n_format; ++i) {
    if (!((mask >> i) & 1u)) continue;
    uint32_t sum = 0;
    for (int ch = 0; ch < nchunks; ++ch) sum = bcf_summary_max(sum, (part[((int64_t)ch * P + k) * W + (q >> 1)] >> (16 * (q & 1))) & 0xFFFFu);
    const int t = bcf_field_type(pl, i, sum);
    const uint32_t cnt = bcf_summary_n(sum) ? bcf_summary_n(sum) : 1u;   // (a field in the mask has a value somewhere; 1 keeps the layout sane otherwise)
    const int f = pl.format_field[i];
    const int32_t key = (f == pl.f_DP && pl.f_DP_FORMAT >= 0) ? pl.bcf_dp_id : pl.bcf_id[f];
    const uint32_t hdr = (uint32_t)bcf_enc_int1_bytes(key) + (uint32_t)bcf_enc_size_bytes((int)cnt);
    lay.fmeta[k * F + q] = cnt | ((uint32_t)t << 16) | (hdr << 24);
    lay.foff[k * F + q] = at;
    at += hdr + (uint32_t)pl.bcf_n_sample * cnt * (uint32_t)bcf_type_width(t);
    ++q;
  }
  lay.l_indiv[k] = at - indiv_begin;
  lay.rec_size[k] = at;
}
__global__ void k_bcf_max_record(const uint64_t* rec_size, int64_t P, unsigned long long* max_record) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = k < P ? rec_size[k] : 0ull;
  for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(mine, d); mine = o > mine ? o : mine; }
  if ((threadIdx.x & 63) == 0 && mine) atomicMax(max_record, mine);
}
// l_shared, l_indiv, the shared block (parked by the site pass) and the key / type bytes of every FORMAT field: one thread per record
__global__ void k_bcf_shared(const SiteCtx* __restrict__ sxp, const char* __restrict__ staging, const char* __restrict__ spill_buf, const int32_t* __restrict__ spill_chunk,
                             int64_t k_begin, int64_t k_end, const uint64_t* __restrict__ rec_off, uint64_t page_base, BcfLayout lay, int F, char* __restrict__ arena, uint32_t* err) {
  const SiteCtx& sx = *sxp;
  const int64_t k = k_begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= k_end) return;
  const CombinePlan& pl = sx.pl;
  char* rec = arena + (rec_off[k] - page_base);
  const uint32_t n = sx.so.prefix_len[k];
  ByteSink head(rec);
  bcf_put_u32(head, n);
  bcf_put_u32(head, lay.l_indiv[k]);
  char* dst = rec + 8;
  uint32_t e = 0;
  if (n <= (uint32_t)kSiteCap) {
    const char* src = staging + (size_t)k * kSiteStride;
    for (uint32_t i = 0; i < n; ++i) dst[i] = src[i];
  } else if (spill_chunk[k] >= 0) {
    const char* src = staging + (size_t)k * kSiteStride;
    for (uint32_t i = 0; i < (uint32_t)kSiteCap; ++i) dst[i] = src[i];
    const char* tail = spill_buf + (size_t)k * kSpillTail;
    for (uint32_t i = kSiteCap; i < n; ++i) dst[i] = tail[i - kSiteCap];
  } else {
    ByteSink bs(dst);
    site_emit(sx, k, bs, false, &e);
  }
  const uint32_t mask = sx.so.fmt_mask[k];
  int q = 0;
  for (int i = 0; i < pl.n_format; ++i) {
    if (!((mask >> i) & 1u)) continue;
    const uint32_t m = lay.fmeta[k * F + q];
    ByteSink fs(rec + lay.foff[k * F + q]);
    const int f = pl.format_field[i];
    bcf_enc_int1(fs, (f == pl.f_DP && pl.f_DP_FORMAT >= 0) ? pl.bcf_dp_id : pl.bcf_id[f]);
    bcf_enc_size(fs, (int)(m & 0xFFFFu), (int)((m >> 16) & 0xFu));
    ++q;
  }
  if (e) atomicOr(err, e);
}
// values: one wavefront = a run of records x 64 samples.  A lane keeps its entry in an LDS slot (and reloads it only when the
// resolved pointer changes), converts the elements of field q to the record's type, pads up to the record's vector length
// (collect_and_extend_fields, variant_field_handler.cc:846-866) into an LDS image of the 64 samples' values, and the wavefront
// moves the image to the record with aligned 16-byte stores.
struct __attribute__((packed)) PackedU16 { uint16_t v; };
#ifndef GDB_BCF_ENTRY_CAP
#define GDB_BCF_ENTRY_CAP 96
#endif
constexpr int kBcfEntryCap = GDB_BCF_ENTRY_CAP;     // bytes of an entry kept in the lane's LDS slot (the rest is read from the pool)
constexpr int kBcfImageSample = 32;  // bytes per sample and field that go through the LDS image (longer vectors: direct stores)
// one sample's cnt values of a field of BCF type t, from its entry (n elements at `in`) to `out`.  first / rest: what stands where
// the call has no element (the first position of an empty vector, every other one).  The two pointers are LDS or global by
// instantiation, so the accesses are ds_ / global_ instructions and not flat ones.  4-byte elements are 4-byte aligned in the entries.
template <class Out, class In> __device__ __forceinline__ void bcf_sample_values(Out* out, const In* in, uint32_t n, uint32_t cnt, uint32_t t, uint32_t first, uint32_t rest) {
  if (t == GDB_BT_CHAR) {
    for (uint32_t j = 0; j < cnt; ++j) out[j] = j < n ? in[j] : (char)(j == 0 ? first : rest);
  } else if (t == GDB_BT_INT8) {
    for (uint32_t j = 0; j < cnt; ++j) {
      const int32_t v = (int32_t)(j < n ? reinterpret_cast<const uint32_t*>(in)[j] : (j == 0 ? first : rest));
      out[j] = v == GDB_BCF_INT32_MISSING ? (char)0x80 : v == GDB_BCF_INT32_VECTOR_END ? (char)0x81 : (char)v;
    }
  } else if (t == GDB_BT_INT16) {
    for (uint32_t j = 0; j < cnt; ++j) {
      const int32_t v = (int32_t)(j < n ? reinterpret_cast<const uint32_t*>(in)[j] : (j == 0 ? first : rest));
      reinterpret_cast<PackedU16*>(out + 2u * j)->v = v == GDB_BCF_INT32_MISSING ? (uint16_t)0x8000u : v == GDB_BCF_INT32_VECTOR_END ? (uint16_t)0x8001u : (uint16_t)v;
    }
  } else {                      // int32 and float: the stored bits
    for (uint32_t j = 0; j < cnt; ++j) reinterpret_cast<PackedU32*>(out + 4u * j)->v = j < n ? reinterpret_cast<const uint32_t*>(in)[j] : (j == 0 ? first : rest);
  }
}
// is the layout of the idx-th record (in assembly order) the layout of the one before it?  (exact comparison of mask and rows)
__global__ void k_bcf_same_layout(const int32_t* __restrict__ order, const uint32_t* __restrict__ fmt_mask, const uint32_t* __restrict__ fmeta, int64_t np, int F, uint8_t* __restrict__ same) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= np) return;
  bool eq = idx > 0;
  if (eq) {
    const int64_t k = order[idx], kp = order[idx - 1];
    const uint32_t m = fmt_mask[k];
    eq = m == fmt_mask[kp];
    const int nf = __popc(m);
    for (int q = 0; eq && q < nf; ++q) eq = fmeta[k * F + q] == fmeta[kp * F + q];
  }
  same[idx] = eq ? 1 : 0;
}
// GDB_BCF_PROF (a variant build, tests/tools/build_variant.sh): cycles per section of k_bcf_write's step, summed over the wavefronts
#ifdef GDB_BCF_PROF
#define BCF_PROF_PARAM , unsigned long long* prof
#define BCF_PROF_ARG , bcf_prof_buffer()
#define BCF_STAMP(n) { const uint64_t now_ = __builtin_amdgcn_s_memtime(); prof_acc[n] += now_ - prof_t; prof_t = now_; }
#define BCF_COUNT(n) { ++prof_acc[n]; }
static unsigned long long* bcf_prof_buffer() { static unsigned long long* p = nullptr; if (!p) { (void)hipMalloc((void**)&p, 128); (void)hipMemset(p, 0, 128); } return p; }
#else
#define BCF_PROF_PARAM
#define BCF_PROF_ARG
#define BCF_STAMP(n)
#define BCF_COUNT(n)
#endif
struct __attribute__((packed)) PackedU128 { u32x4 v; };
constexpr int kBcfImageBytes = 3072;     // LDS image of the FORMAT values of one (record, 64-sample chunk), every field a 16-byte aligned region
constexpr int kBcfImageSets = kBcfImageBytes / (16 * kAsmRows);
constexpr int kBcfBatch = 4;             // records whose resolved rows are fetched together
// One wavefront = a run of records x 64 samples.
// FAST PATH (the record's values for 64 samples fit the LDS image): the image holds one region per FORMAT field (sample-major,
// exactly the bytes that go to the record) and PERSISTS from record to record.  What a lane wrote for its sample stays valid while
// its resolved entry and the record's layout (vector length and type of every field) do not change - a call spans many records,
// so most (record, chunk) steps format nothing or a lane or two - and every record ends with the same flush: each lane moves
// 16-byte words of the image to offsets it computed when the layout last changed (unaligned 16-byte stores; the record's base is
// the only thing that differs).  Per-field constants live in lanes (lane q = q-th field of the record) and reach the field loop
// through v_readlane; everything the loop needs from memory arrives one record ahead as vector loads.
// SLOW PATH (long vectors): field by field through a 2 KB image with aligned stores, or straight to the page.
// Resident wavefronts are what this kernel runs on (a step is a chain of LDS round trips): 96-byte entry slots, a 3 KiB image and at
// most 128 registers (amdgpu_waves_per_eu) make it 14 per CU instead of 11 - 18.1 -> 14.3 ms on the c2 window.  (With 129 registers
// the register file allowed 12, which is why smaller slots alone changed nothing in round 2; a 2 KiB image sends c2's records down
// the slow path, 2 instead of 4 rows per batch costs more than the 15th wavefront brings.)
__global__ void __launch_bounds__(kAsmRows) __attribute__((amdgpu_waves_per_eu(4))) k_bcf_write(CombinePlan pl, const ResMatrix resolved, const char* __restrict__ pool, const char* __restrict__ pool_ovf,
                                                      const uint32_t* __restrict__ fmt_mask, const int32_t* __restrict__ order, const uint8_t* __restrict__ same_layout, int64_t np, int nchunks,
                                                      int F, int32_t N, BcfLayout lay, const uint64_t* __restrict__ rec_off, uint64_t page_base, char* __restrict__ arena BCF_PROF_PARAM) {
  __shared__ __attribute__((aligned(16))) char s_entry[kAsmRows * kBcfEntryCap];
#ifdef GDB_BCF_PROF
  uint64_t prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t prof_t = __builtin_amdgcn_s_memtime();
#endif
  __shared__ __attribute__((aligned(16))) char s_image[kBcfImageBytes + 16];
  __shared__ uint2 s_rows[kBcfBatch][kAsmRows];
  __shared__ uint8_t s_changed[kAsmRows];     // the lanes (samples) whose entry changed at this record, in lane order
  const int64_t unit = xcd_aware_unit<1>(((np + kBcfRun - 1) / kBcfRun) * (int64_t)nchunks);
  if (unit < 0) return;
  // order[0 .. np): the page's records in (block, type) order - along a run of one type the layout and most entries stay the same
  const int64_t i0 = (unit / nchunks) * kBcfRun, i1 = min(np, i0 + (int64_t)kBcfRun);
  const int ch = (int)(unit % nchunks);
  const int lane = threadIdx.x;
  const uint32_t nsamp = (uint32_t)min((int32_t)kAsmRows, N - ch * kAsmRows);
  const bool flag = pl.use_missing_values_not_vector_end != 0;
  const bool live = (uint32_t)lane < nsamp;
  uint32_t slot_key = 0xFFFFFFFFu;      // the entry in this lane's LDS slot
  uint32_t img_key = 0xFFFFFFFFu;       // the entry (or "none") this lane's image bytes were formatted from
  bool img_valid = false;               // uniform: the image holds the previous record's layout
  // per FORMAT field of the plan (lane i): bytes per element in the entries, GT or not
  uint32_t my_field = 0;
  if (lane < pl.n_format) my_field = (uint32_t)bcf_field_elem_size(pl, lane) | (pl.format_field[lane] == pl.f_GT ? 0x100u : 0u);
  const int n_format = pl.n_format;
  // per field q of the current layout (lane q): bytes per sample, region offset in the image, offset of the region in the record
  // relative to the first field's key, padding values, element size in the entries
  uint32_t q_per = 0, q_reg = 0, q_grel = 0, q_first = 0, q_rest = 0, q_es = 0;
  uint32_t nsets = 0;
  // per output element e of ONE sample under the current layout (lane e; all fields back to back): its field, index in the field,
  // BCF type, place in the image (sample 0) - what the one-sample-at-a-time format below needs
  uint32_t e_q = 0, e_j = 0, e_t = 0, e_dst = 0;
  uint32_t n_elems = 0;                 // elements per sample (0: more than 64, the lanes-are-samples format is used)
  // the same per GROUP of lanes: the wavefront is cut into 64 >> g_shift groups of (1 << g_shift) >= n_elems lanes and every group
  // formats another changed sample in the same pass; lane l of a group is that sample's l-th output element
  uint32_t g_q = 0, g_j = 0, g_t = 0, g_dst = 0, g_first = 0, g_rest = 0, g_per = 0, g_shift = 0;
  bool g_ok = false;
  uint32_t desc_g[kBcfImageSets], desc_nv[kBcfImageSets];
#pragma unroll
  for (int sidx = 0; sidx < kBcfImageSets; ++sidx) { desc_g[sidx] = 0; desc_nv[sidx] = 0; }
  // per record of the run (lane j = j-th record): index, mask, offset in the page, offset of the first FORMAT key, "same layout as
  // the record before" - one load each for the whole run, v_readlane in the loop.
  int32_t my_k = 0; uint32_t my_mask = 0, my_foff0 = 0, my_same = 0; uint64_t my_roff = 0;
  int cnt_run = 0;
  uint32_t my_meta = 0, my_foff = 0;
  for (int64_t i = i0; i < i1; ++i) {                    // uniform
    const int j = (int)((i - i0) & 63);
    if (j == 0) {                                         // the next 64 records of the run
      const bool in_run = i + lane < i1;
      my_k = in_run ? order[i + lane] : 0;
      my_mask = in_run ? fmt_mask[my_k] : 0u;
      my_roff = in_run ? rec_off[my_k] : 0ull;
      my_foff0 = in_run ? lay.foff[(int64_t)my_k * F] : 0u;
      my_same = in_run ? (uint32_t)same_layout[i + lane] : 0u;
      cnt_run = (int)min((int64_t)64, i1 - i);
      // (the loads are waited for HERE: left pending, the compiler's s_waitcnt vmcnt(0) lands at the top of every step, where it also
      // waits for the previous step's stores)
      asm volatile("" : "+v"(my_k), "+v"(my_mask), "+v"(my_foff0), "+v"(my_same), "+v"(my_roff));
    }
    // the samples' resolved entries of the next kBcfBatch records: all loads in flight at once, parked in LDS.  (A load per record
    // - even issued records ahead - ends in s_waitcnt vmcnt(0) at the loop's back edge: every step then waits a full memory
    // latency, for its load and for the previous step's stores.)
    if ((j % kBcfBatch) == 0) {
      uint2 t[kBcfBatch];
#pragma unroll
      for (int b = 0; b < kBcfBatch; ++b) {
        const int jb = j + b < cnt_run ? j + b : cnt_run - 1;
        const int64_t kk = __builtin_amdgcn_readlane(my_k, jb);
        t[b] = resolved.load((kk * nchunks + ch) * kAsmRows + lane);
      }
#pragma unroll
      for (int b = 0; b < kBcfBatch; ++b) s_rows[b][lane] = t[b];
    }
    const uint2 d = s_rows[j % kBcfBatch][lane];
    const int64_t k = __builtin_amdgcn_readlane(my_k, j);
    const uint32_t mask = (uint32_t)__builtin_amdgcn_readlane((int)my_mask, j);
    char* rec = arena + (readlane64((int64_t)my_roff, j) - (int64_t)page_base);
    const char* src = ((d.x & kOverflowBit) ? pool_ovf : pool) + (size_t)(d.x & ~kOverflowBit) * 16;
    const bool fetch = d.y && d.x != slot_key;
    if (__any((int)fetch)) {                              // uniform: a step in which no sample changes its entry has no wait for memory at all
    if (fetch) {                                          // all 16-byte pieces in flight at once, then into the slot: one exposed latency
      const uint32_t take = d.y < (uint32_t)kBcfEntryCap ? d.y : (uint32_t)kBcfEntryCap;
      u32x4 piece[kBcfEntryCap / 16];
#pragma unroll
      for (int o = 0; o < kBcfEntryCap / 16; ++o) { piece[o] = u32x4{0u, 0u, 0u, 0u}; if ((uint32_t)o * 16u < take) piece[o] = *reinterpret_cast<const u32x4*>(src + o * 16); }
#pragma unroll
      for (int o = 0; o < kBcfEntryCap / 16; ++o) if ((uint32_t)o * 16u < take) *reinterpret_cast<u32x4*>(s_entry + lane * kBcfEntryCap + o * 16) = piece[o];
      slot_key = d.x;
    }
    settle_loads();        // the step's one wait for memory: the entry fetch and the stores of the steps before
    }
    const uint32_t key = d.y ? d.x : 0xFFFFFFFEu;
    const bool whole = !__any((int)(live && d.y > (uint32_t)kBcfEntryCap));   // every live lane's entry is in its slot
    const int nf = __popc(mask);
    BCF_STAMP(0) BCF_COUNT(8)
    // ---- layout: unchanged from the previous record? ------------------------------------------------------------------------------
    const bool same = img_valid && __builtin_amdgcn_readlane((int)my_same, j) != 0;
    bool fits = true;
    if (!same) {
      my_meta = lane < F ? lay.fmeta[k * F + lane] : 0u;       // (a layout change is rare: these two loads are not worth prefetching)
      my_foff = lane < F ? lay.foff[k * F + lane] : 0u;
      // lane q: the plan field behind the q-th set bit of the mask
      int my_i = 0;
      { int qq = 0; for (int i = 0; i < n_format; ++i) if ((mask >> i) & 1u) { if (lane == qq) my_i = i; ++qq; } }
      const uint32_t fld = (uint32_t)__shfl((int)my_field, my_i, 64);
      const uint32_t cnt = my_meta & 0xFFFFu, t = (my_meta >> 16) & 0xFu, hdr = my_meta >> 24;
      const uint32_t w = (uint32_t)bcf_type_width((int)t);
      const bool is_gt = (fld & 0x100u) != 0u;
      q_es = fld & 0xFFu;
      q_per = lane < nf ? cnt * w : 0u;
      if (t == GDB_BT_CHAR) { q_first = flag ? 0u : 7u; q_rest = 0u; }
      else if (t == GDB_BT_FLOAT) { q_first = GDB_BCF_FLOAT_MISSING_BITS; q_rest = flag ? GDB_BCF_FLOAT_MISSING_BITS : GDB_BCF_FLOAT_VECTOR_END_BITS; }
      else { q_first = is_gt ? (flag ? 0u : (uint32_t)GDB_BCF_INT32_VECTOR_END) : (uint32_t)GDB_BCF_INT32_MISSING; q_rest = flag ? (uint32_t)GDB_BCF_INT32_MISSING : (uint32_t)GDB_BCF_INT32_VECTOR_END; }
      const uint32_t size = nsamp * q_per, size_a = (size + 15u) & ~15u;
      const uint32_t incl = wave_inclusive_scan_dpp(size_a);
      q_reg = incl - size_a;
      const uint32_t total = wave_total(incl);
      const uint32_t foff0 = (uint32_t)__builtin_amdgcn_readlane((int)my_foff, 0);
      q_grel = (my_foff - foff0) + hdr + (uint32_t)ch * kAsmRows * q_per;
      fits = total <= (uint32_t)kBcfImageBytes && !__any((int)(lane < nf && (q_per == 0u || q_per > 1024u)));
      if (fits) {
        nsets = (total + 1023u) >> 10;
#pragma unroll
        for (int sidx = 0; sidx < kBcfImageSets; ++sidx) {
          const uint32_t b = 16u * ((uint32_t)lane + 64u * (uint32_t)sidx);
          uint32_t g = 0, nv = 0;
          if ((uint32_t)sidx < nsets)
            for (int q = 0; q < nf; ++q) {               // uniform
              const uint32_t ro = (uint32_t)__builtin_amdgcn_readlane((int)q_reg, q), sz = nsamp * (uint32_t)__builtin_amdgcn_readlane((int)q_per, q);
              const uint32_t gr = (uint32_t)__builtin_amdgcn_readlane((int)q_grel, q);
              if (b >= ro && b < ro + sz) { g = gr + (b - ro); nv = ro + sz - b < 16u ? ro + sz - b : 16u; }
            }
          desc_g[sidx] = g; desc_nv[sidx] = nv;
        }
        // lane e -> (field, index): the fields' element counts are a prefix-sum away
        const uint32_t cnt_incl = wave_inclusive_scan_dpp(lane < nf ? cnt : 0u);
        const uint32_t all_elems = wave_total(cnt_incl);
        n_elems = all_elems <= (uint32_t)kAsmRows ? all_elems : 0u;
        e_q = 0; e_j = 0; e_t = 0; e_dst = 0;
        if (n_elems)
          for (int q = 0; q < nf; ++q) {                 // uniform
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)cnt_incl, q), m = (uint32_t)__builtin_amdgcn_readlane((int)my_meta, q);
            const uint32_t lo = hi - (m & 0xFFFFu);
            if ((uint32_t)lane >= lo && (uint32_t)lane < hi) {
              e_q = (uint32_t)q; e_j = (uint32_t)lane - lo; e_t = (m >> 16) & 0xFu;
              e_dst = (uint32_t)__builtin_amdgcn_readlane((int)q_reg, q) + e_j * (uint32_t)bcf_type_width((int)e_t);
            }
          }
        if (n_elems) {
          g_shift = n_elems > 1u ? 32u - (uint32_t)__builtin_clz(n_elems - 1u) : 0u;      // uniform
          const int el = lane & ((1 << g_shift) - 1);
          g_ok = (uint32_t)el < n_elems;
          g_q = (uint32_t)__shfl((int)e_q, el, 64); g_j = (uint32_t)__shfl((int)e_j, el, 64);
          g_t = (uint32_t)__shfl((int)e_t, el, 64); g_dst = (uint32_t)__shfl((int)e_dst, el, 64);
          g_first = (uint32_t)__shfl((int)q_first, (int)g_q, 64); g_rest = (uint32_t)__shfl((int)q_rest, (int)g_q, 64);
          g_per = (uint32_t)__shfl((int)q_per, (int)g_q, 64);
        }
      }
    }
    BCF_STAMP(1)
    if (fits) {
      // ---- format the lanes whose entry changed (all of them after a layout change) -------------------------------------------------
      const bool need = live && (!same || key != img_key);
      // the summaries (vector length per field) of an entry: 2 bytes per field at its start - up to 8 fields arrive with ONE 16-byte LDS read
      // and are shifted out field by field (a read per field is an LDS round trip per field in every pass below)
      const bool packed = nf <= 8;
      // Who formats what.  The layout has at most 64 elements per sample (n_elems): the samples whose entry changed go through the
      // group passes below, a group of lanes per sample; after a layout change every sample is due, and the lanes format their own
      // sample (lanes-are-samples) - except those whose entry is longer than its LDS slot, which take the group passes too (a lane
      // that walks a long entry in the pool waits a memory latency per element: 15 and more in a row).  More than 64 elements per
      // sample: lanes-are-samples for all, from the pool when some entry is longer than its slot.
      const bool big = d.y > (uint32_t)kBcfEntryCap;
      const bool grp = need && n_elems && (same || big);
      const bool lanes_pass = need && !grp;
      const bool from_slot = n_elems ? true : whole;
      if (__any((int)lanes_pass)) {
        uint64_t lo = 0, hi = 0;
        if (packed && d.y) {
          const u32x4 sm = *reinterpret_cast<const u32x4*>(s_entry + lane * kBcfEntryCap);
          lo = (uint64_t)sm[0] | ((uint64_t)sm[1] << 32); hi = (uint64_t)sm[2] | ((uint64_t)sm[3] << 32);
        }
        uint32_t body = bcf_summary_bytes(nf);
        for (int q = 0; q < nf; ++q) {                   // uniform
          const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)my_meta, q);
          const uint32_t cnt = m & 0xFFFFu, t = (m >> 16) & 0xFu;
          const uint32_t per = (uint32_t)__builtin_amdgcn_readlane((int)q_per, q), ro = (uint32_t)__builtin_amdgcn_readlane((int)q_reg, q);
          const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)q_first, q), rest = (uint32_t)__builtin_amdgcn_readlane((int)q_rest, q);
          const uint32_t es = (uint32_t)__builtin_amdgcn_readlane((int)q_es, q);
          uint32_t n;
          if (packed) { n = bcf_summary_n((uint32_t)lo & 0xFFFFu); lo = (lo >> 16) | (hi << 48); hi >>= 16; }
          else n = d.y ? bcf_summary_n(reinterpret_cast<const uint16_t*>(s_entry + lane * kBcfEntryCap)[q]) : 0u;
          if (es == 4u) body = (body + 3u) & ~3u;
          if (lanes_pass) {
            if (from_slot) bcf_sample_values(s_image + ro + (uint32_t)lane * per, s_entry + lane * kBcfEntryCap + body, n, cnt, t, first, rest);
            else bcf_sample_values(s_image + ro + (uint32_t)lane * per, src + body, n, cnt, t, first, rest);
          }
          body += n * es;
        }
        if (!from_slot) settle_loads();
      }
      BCF_STAMP(2)
      const uint64_t grp_mask = __ballot(grp);
      if (grp_mask) BCF_COUNT(9)
      if (grp_mask) {
        // A group of lanes per sample, lane l of the group = the sample's l-th output element; 64 >> g_shift samples per pass.  The
        // offset of the element's field inside the sample's entry comes from a walk over the entry's summary (4-byte fields are
        // 4-byte aligned), then every lane fetches, converts and stores its own element: from the sample's slot, or from the pool
        // where a long entry reaches beyond its slot (one load per lane, all in flight together).
        if (grp) s_changed[__popcll(grp_mask & ((1ull << lane) - 1ull))] = (uint8_t)(lane | (d.y ? 0x80 : 0));
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const int n_changed = __popcll(grp_mask);
        const int per_pass = 64 >> g_shift;
        // (two instances: the one that may go to the pool has a wait for memory in every pass)
        const auto passes = [&](auto with_pool) {
          for (int p0 = 0; p0 < n_changed; p0 += per_pass) {        // uniform
            const int r = p0 + (lane >> g_shift);
            const bool act = g_ok && r < n_changed;
            const uint32_t cL = act ? (uint32_t)s_changed[r] : (uint32_t)lane;
            const int L = (int)(cL & 63u);
            const bool hasL = (cL & 0x80u) != 0u;
            const char* const slotL = s_entry + L * kBcfEntryCap;
            const char* srcL = nullptr;
            if constexpr (decltype(with_pool)::value) {
              const uint32_t xL = (uint32_t)__shfl((int)d.x, L, 64);
              srcL = ((xL & kOverflowBit) ? pool_ovf : pool) + (size_t)(xL & ~kOverflowBit) * 16;
            }
            uint32_t b = bcf_summary_bytes(nf), body = 0, n = 0;
            if (packed) {
              const u32x4 sm = *reinterpret_cast<const u32x4*>(slotL);
              uint64_t lo = hasL ? ((uint64_t)sm[0] | ((uint64_t)sm[1] << 32)) : 0ull, hi = hasL ? ((uint64_t)sm[2] | ((uint64_t)sm[3] << 32)) : 0ull;
              for (int q = 0; q < nf; ++q) {                         // uniform
                const uint32_t es = (uint32_t)__builtin_amdgcn_readlane((int)q_es, q);
                const uint32_t nq = bcf_summary_n((uint32_t)lo & 0xFFFFu);
                lo = (lo >> 16) | (hi << 48); hi >>= 16;
                if (es == 4u) b = (b + 3u) & ~3u;
                if ((uint32_t)q == g_q) { body = b; n = nq; }
                b += nq * es;
              }
            } else {
              for (int q = 0; q < nf; ++q) {                         // uniform
                const uint32_t es = (uint32_t)__builtin_amdgcn_readlane((int)q_es, q);
                const uint32_t nq = hasL ? bcf_summary_n(reinterpret_cast<const uint16_t*>(slotL)[q]) : 0u;
                if (es == 4u) b = (b + 3u) & ~3u;
                if ((uint32_t)q == g_q) { body = b; n = nq; }
                b += nq * es;
              }
            }
            const bool have = act && g_j < n;
            const bool is_char = g_t == GDB_BT_CHAR;
            const uint32_t at = body + (is_char ? g_j : 4u * g_j);
            uint32_t v = g_j == 0 ? g_first : g_rest;
            bool in_slot = true;
            if constexpr (decltype(with_pool)::value) {
              in_slot = at + (is_char ? 1u : 4u) <= (uint32_t)kBcfEntryCap;
              uint32_t pv = 0;
              if (have && !in_slot) pv = is_char ? (uint32_t)(uint8_t)srcL[at] : *reinterpret_cast<const uint32_t*>(srcL + at);
              settle_loads();
              if (have && !in_slot) v = pv;
            }
            if (have && in_slot) v = is_char ? (uint32_t)(uint8_t)slotL[at] : *reinterpret_cast<const uint32_t*>(slotL + at);
            if (act) {
              char* const out = s_image + g_dst + (uint32_t)L * g_per;
              if (is_char) out[0] = (char)v;
              else if (g_t == GDB_BT_INT8) out[0] = (int32_t)v == GDB_BCF_INT32_MISSING ? (char)0x80 : (int32_t)v == GDB_BCF_INT32_VECTOR_END ? (char)0x81 : (char)v;
              else if (g_t == GDB_BT_INT16)
                *reinterpret_cast<uint16_t*>(out) = (int32_t)v == GDB_BCF_INT32_MISSING ? (uint16_t)0x8000u : (int32_t)v == GDB_BCF_INT32_VECTOR_END ? (uint16_t)0x8001u : (uint16_t)v;
              else *reinterpret_cast<uint32_t*>(out) = v;
            }
          }
        };
        if (__ballot(grp && big) != 0ull) passes(std::true_type{}); else passes(std::false_type{});
      }
      if (need) img_key = key;
      BCF_STAMP(3)
      img_valid = true;
      // ---- flush: every lane its words of the image ----------------------------------------------------------------------------------
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      char* const gbase = rec + (uint32_t)__builtin_amdgcn_readlane((int)my_foff0, j);
#pragma unroll
      for (int sidx = 0; sidx < kBcfImageSets; ++sidx) {
        if ((uint32_t)sidx >= nsets) break;              // uniform
        const char* lw = s_image + 16u * ((uint32_t)lane + 64u * (uint32_t)sidx);
        // (the last word of a field's region holds 1..15 bytes: 8 + 4 + 2 + 1, four predicated stores - a byte loop ran its 15
        // iterations in nearly every step, for the one lane per field that has such a word)
        const uint32_t nv = desc_nv[sidx];
        char* const g = gbase + desc_g[sidx];
        if (nv == 16u) reinterpret_cast<PackedU128*>(g)->v = *reinterpret_cast<const u32x4*>(lw);
        else if (nv) {
          uint32_t o = 0;
          if (nv & 8u) { reinterpret_cast<PackedU64*>(g)->v = *reinterpret_cast<const uint64_t*>(lw); o = 8u; }
          if (nv & 4u) { reinterpret_cast<PackedU32*>(g + o)->v = *reinterpret_cast<const uint32_t*>(lw + o); o += 4u; }
          if (nv & 2u) { reinterpret_cast<PackedU16*>(g + o)->v = *reinterpret_cast<const uint16_t*>(lw + o); o += 2u; }
          if (nv & 1u) g[o] = lw[o];
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      BCF_STAMP(4)
      continue;
    }
    // ---- slow path: field by field --------------------------------------------------------------------------------------------------
    img_valid = false;
    uint32_t body = bcf_summary_bytes(nf);        // this lane's read position inside its entry
    int q = 0;
    for (int i = 0; i < n_format; ++i) {          // uniform
      if (!((mask >> i) & 1u)) continue;
      const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)my_meta, q);
      const uint32_t cnt = m & 0xFFFFu, t = (m >> 16) & 0xFu, hdr = m >> 24;
      const uint32_t w = (uint32_t)bcf_type_width((int)t);
      const uint32_t fld = (uint32_t)__builtin_amdgcn_readlane((int)my_field, i);
      const int es = (int)(fld & 0xFFu);
      const uint32_t n = d.y ? bcf_summary_n(reinterpret_cast<const uint16_t*>(s_entry + lane * kBcfEntryCap)[q]) : 0u;
      const bool is_gt = (fld & 0x100u) != 0u;
      if (es == 4) body = (body + 3u) & ~3u;
      // no element: CHAR '.' then vector ends (all vector ends under the htsjdk flag); FLOAT / INT missing then vector ends
      // (missing everywhere under the flag); GT is all vector ends (no-call alleles under the flag)
      uint32_t first, rest;
      if (t == GDB_BT_CHAR) { first = flag ? 0u : 7u; rest = 0u; }
      else if (t == GDB_BT_FLOAT) { first = GDB_BCF_FLOAT_MISSING_BITS; rest = flag ? GDB_BCF_FLOAT_MISSING_BITS : GDB_BCF_FLOAT_VECTOR_END_BITS; }
      else { first = is_gt ? (flag ? 0u : (uint32_t)GDB_BCF_INT32_VECTOR_END) : (uint32_t)GDB_BCF_INT32_MISSING; rest = flag ? (uint32_t)GDB_BCF_INT32_MISSING : (uint32_t)GDB_BCF_INT32_VECTOR_END; }
      const uint32_t per = cnt * w;                // bytes per sample
      char* const fdst = rec + (uint32_t)__builtin_amdgcn_readlane((int)my_foff, q) + hdr + (size_t)ch * kAsmRows * per;     // the 64 samples' values of this field
      if (per <= (uint32_t)kBcfImageSample) {      // uniform
        const uint32_t al = (uint32_t)((uintptr_t)fdst & 15u);
        if (live) {
          if (whole) bcf_sample_values(s_image + al + (uint32_t)lane * per, s_entry + lane * kBcfEntryCap + body, n, cnt, t, first, rest);
          else bcf_sample_values(s_image + al + (uint32_t)lane * per, src + body, n, cnt, t, first, rest);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        const char* img = s_image + al;
        const uint32_t total = nsamp * per;
        uint32_t head = (16u - al) & 15u;
        if (head > total) head = total;
        const uint32_t nwords = (total - head) >> 4;
        const uint32_t tail_at = head + (nwords << 4);
        if ((uint32_t)lane < head) fdst[lane] = img[lane];
        const u32x4* lsrc = reinterpret_cast<const u32x4*>(img + head);
        u32x4* gw = reinterpret_cast<u32x4*>(fdst + head);
        for (uint32_t wq = lane; wq < nwords; wq += kAsmRows) gw[wq] = lsrc[wq];
        if ((uint32_t)lane < total - tail_at) fdst[tail_at + lane] = img[tail_at + lane];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      } else if (live) {
        bcf_sample_values(fdst + (size_t)lane * per, src + body, n, cnt, t, first, rest);
      }
      body += n * (uint32_t)es;
      ++q;
    }
    BCF_STAMP(5)
  }
#ifdef GDB_BCF_PROF
  if (lane == 0) for (int c = 0; c < 12; ++c) atomicAdd(prof + c, (unsigned long long)prof_acc[c]);
#endif
}

// SURVEY 8(d) "remap elements": per record that needs re-indexing, (calls with a valid genotype-length field) x (merged genotypes) -
// the element count of the reference's O(G) remap loop (variant_field_handler.cc:134-191), diploid genotype count
__global__ void k_remap_elements(SiteOut so, const int32_t* __restrict__ pl_cnt, int64_t P, int pl_bit, unsigned long long* total) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = 0;
  if (k < P && (so.rflags[k] & GDB_RF_REMAPPING_NEEDED) && ((so.fmt_mask[k] >> pl_bit) & 1u)) {
    const unsigned long long a = so.num_alleles[k];
    mine = (unsigned long long)pl_cnt[k] * (a * (a + 1) / 2);
  }
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(total, mine);
}

// sort key of the assembly order: (block of 2^block_log2 consecutive records, record type); the value is the record index
__global__ void k_order_keys(const uint8_t* rtype, int32_t base, int64_t n, int block_log2, uint32_t* keys, int32_t* vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  keys[i] = ((uint32_t)(i >> block_log2) << 8) | rtype[base + i];
  vals[i] = base + (int32_t)i;
}

// ---- S-1: the reference's binary cell stream -> columnar fragment, on the device ---------------------------------------
// Cell = [row i64][col i64][size u64] then the attributes in schema order, each fixed (num x elem) or var ([len i32] +
// len x elem)  (vcf2binary.cc:991-1196, variant_cell.cc:79-117).  One thread per cell walks the ~20 attributes twice:
// pass 1 measures (keep flag, coordinates, element counts of the variable-length plan fields), scans turn the counts into
// column offsets, pass 2 copies the payloads.  Unaligned sources: everything is read bytewise.
constexpr int kMaxSchemaAttrs = 96;
struct CellAttrDesc { int16_t var, elem_size; int32_t num; int32_t field; };   // field: plan field fed by this attribute or -1
struct CellSchemaDev { int32_t nattrs; CellAttrDesc a[kMaxSchemaAttrs]; };
struct CellColumnsDev { char* data[GDB_MAX_FIELDS]; uint32_t* len[GDB_MAX_FIELDS]; const uint32_t* off[GDB_MAX_FIELDS]; uint32_t* frag_off[GDB_MAX_FIELDS]; };

// Unaligned sources.  A wavefront takes 64 consecutive cells (~10 KB of the stream): their bytes are fetched ONCE, coalesced
// (16 bytes per lane), into an LDS tile, and the lanes - one cell each - walk their attributes out of LDS; spans that do not fit
// the tile (cells with very long vectors) are read from global memory with unaligned dword loads instead.  (One thread per cell
// reading global memory bytewise touched 64 different cache lines per load instruction: 54 GB of traffic for 1.45 GB of cells.)
constexpr int kStageTile = 12 * 1024;       // LDS bytes per wavefront
constexpr int kStageBlock = 256;
struct GlobalBytes {
  const uint8_t* p;
  __device__ __forceinline__ uint8_t u8(uint32_t o) const { return p[o]; }
  __device__ __forceinline__ uint32_t u32(uint32_t o) const { return reinterpret_cast<const PackedU32*>(p + o)->v; }
  __device__ __forceinline__ uint64_t u64(uint32_t o) const { return reinterpret_cast<const PackedU64*>(p + o)->v; }
};
typedef __attribute__((address_space(3))) uint8_t gdb_lds_u8;
struct LdsBytes {
  const gdb_lds_u8* p;
  __device__ __forceinline__ uint8_t u8(uint32_t o) const { return p[o]; }
  __device__ __forceinline__ uint32_t u32(uint32_t o) const { return reinterpret_cast<const __attribute__((address_space(3))) PackedU32*>(p + o)->v; }
  __device__ __forceinline__ uint64_t u64(uint32_t o) const { return reinterpret_cast<const __attribute__((address_space(3))) PackedU64*>(p + o)->v; }
};
// fetch the bytes of cells [w0, w1) of a wavefront into its tile; returns false (nothing fetched) when they do not fit
__device__ __forceinline__ bool stage_tile_fetch(const uint8_t* __restrict__ cells, uint64_t span_begin, uint64_t span_end, gdb_lds_u8* tile, int lane, uint32_t& skew) {
  const uint64_t a0 = span_begin & ~(uint64_t)15;
  const uint64_t bytes = span_end - a0;
  skew = (uint32_t)(span_begin - a0);
  if (bytes > (uint64_t)kStageTile) return false;
  for (uint32_t o = (uint32_t)lane * 16u; o < (uint32_t)bytes; o += 64u * 16u)   // (the buffer is padded by 16 bytes: the last load may run past the span)
    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(tile + o) = *reinterpret_cast<const u32x4*>(cells + a0 + o);
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // one wavefront, in-order LDS: a compiler-level fence is all it takes
  return true;
}

template <class Bytes> __device__ __forceinline__ void cell_measure(const Bytes& c, int64_t i, int64_t size, const CellSchemaDev& sch, const int32_t* __restrict__ row_map,
                                                                    int64_t nrows_array, uint32_t* keep, uint32_t* is_marker, int32_t* qrow, int64_t* begin, int64_t* end,
                                                                    const CellColumnsDev& cols, uint32_t* err) {
  const int64_t row = (int64_t)c.u64(0);
  const int32_t q = (row >= 0 && row < nrows_array) ? row_map[row] : -1;
  keep[i] = q >= 0 ? 1u : 0u;
  is_marker[i] = (q < 0 && row >= 0 && row < nrows_array) ? 1u : 0u;   // an array row outside the query: boundary marker
  qrow[i] = q;
  begin[i] = (int64_t)c.u64(8);
  int64_t p = 24;
  for (int ai = 0; ai < sch.nattrs; ++ai) {
    const CellAttrDesc a = sch.a[ai];
    uint32_t cnt = (uint32_t)a.num;
    if (a.var) { if (p + 4 > size) { p = size + 1; break; } cnt = c.u32((uint32_t)p); p += 4; }
    if (ai == 0) { if (p + 8 > size) { p = size + 1; break; } end[i] = (int64_t)c.u64((uint32_t)p); }
    if (a.field >= 0 && a.var) cols.len[a.field][i] = q >= 0 ? cnt : 0u;
    p += (int64_t)cnt * (int64_t)a.elem_size;
    if (p > size) break;
  }
  if (p != size) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);
}
__global__ void __launch_bounds__(kStageBlock) k_cells_measure(const uint8_t* __restrict__ cells, const uint64_t* __restrict__ cell_off, int64_t n, CellSchemaDev sch,
                                                             const int32_t* __restrict__ row_map, int64_t nrows_array, uint32_t* keep, uint32_t* is_marker, int32_t* qrow,
                                                             int64_t* begin, int64_t* end, CellColumnsDev cols, uint32_t* err) {
  __shared__ __attribute__((aligned(16))) uint8_t tiles[(kStageBlock / 64) * kStageTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t w0 = ((int64_t)blockIdx.x * (kStageBlock / 64) + wave) * 64;
  if (w0 >= n) return;
  const int64_t w1 = min(n, w0 + 64);
  const int64_t i = w0 + lane;
  gdb_lds_u8* tile = (gdb_lds_u8*)tiles + wave * kStageTile;
  uint32_t skew;
  const uint64_t span_begin = cell_off[w0];
  const bool tiled = stage_tile_fetch(cells, span_begin, cell_off[w1], tile, lane, skew);   // uniform per wavefront
  if (i >= w1) return;
  const uint64_t o = cell_off[i];
  const int64_t size = (int64_t)(cell_off[i + 1] - o);
  if (tiled) cell_measure(LdsBytes{tile + skew + (uint32_t)(o - span_begin)}, i, size, sch, row_map, nrows_array, keep, is_marker, qrow, begin, end, cols, err);
  else cell_measure(GlobalBytes{cells + o}, i, size, sch, row_map, nrows_array, keep, is_marker, qrow, begin, end, cols, err);
}

template <class Bytes> __device__ __forceinline__ void cell_scatter(const Bytes& c, int64_t i, uint32_t d, const CellSchemaDev& sch, const CellColumnsDev& cols) {
  uint32_t p = 24;
  for (int ai = 0; ai < sch.nattrs; ++ai) {
    const CellAttrDesc a = sch.a[ai];
    uint32_t cnt = (uint32_t)a.num;
    if (a.var) { cnt = c.u32(p); p += 4; }
    const uint32_t bytes = cnt * (uint32_t)a.elem_size;
    if (a.field >= 0) {
      char* dst;
      if (a.var) { const uint32_t o = cols.off[a.field][i]; cols.frag_off[a.field][d] = o; dst = cols.data[a.field] + (size_t)o * a.elem_size; }
      else dst = cols.data[a.field] + (size_t)d * bytes;
      // destinations are element-aligned (columns are arrays of elements): whole-word stores for 4- and 8-byte elements
      if (a.elem_size == 4) { for (uint32_t j = 0; j < cnt; ++j) reinterpret_cast<uint32_t*>(dst)[j] = c.u32(p + 4u * j); }
      else if (a.elem_size == 8) { for (uint32_t j = 0; j < cnt; ++j) reinterpret_cast<uint64_t*>(dst)[j] = c.u64(p + 8u * j); }
      else for (uint32_t b = 0; b < bytes; ++b) dst[b] = (char)c.u8(p + b);
    }
    p += bytes;
  }
}
__global__ void __launch_bounds__(kStageBlock) k_cells_scatter(const uint8_t* __restrict__ cells, const uint64_t* __restrict__ cell_off, int64_t n, CellSchemaDev sch,
                                                             const uint32_t* __restrict__ keep, const uint32_t* __restrict__ dest, const int32_t* __restrict__ qrow,
                                                             const int64_t* __restrict__ begin, const int64_t* __restrict__ end, int32_t* row_out, int64_t* begin_out,
                                                             int64_t* end_out, CellColumnsDev cols) {
  __shared__ __attribute__((aligned(16))) uint8_t tiles[(kStageBlock / 64) * kStageTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t w0 = ((int64_t)blockIdx.x * (kStageBlock / 64) + wave) * 64;
  if (w0 >= n) return;
  const int64_t w1 = min(n, w0 + 64);
  const int64_t i = w0 + lane;
  gdb_lds_u8* tile = (gdb_lds_u8*)tiles + wave * kStageTile;
  uint32_t skew;
  const uint64_t span_begin = cell_off[w0];
  const bool tiled = stage_tile_fetch(cells, span_begin, cell_off[w1], tile, lane, skew);   // uniform per wavefront
  if (i >= w1 || !keep[i]) return;
  const uint32_t d = dest[i];
  row_out[d] = qrow[i]; begin_out[d] = begin[i]; end_out[d] = end[i];
  const uint64_t o = cell_off[i];
  if (tiled) cell_scatter(LdsBytes{tile + skew + (uint32_t)(o - span_begin)}, i, d, sch, cols);
  else cell_scatter(GlobalBytes{cells + o}, i, d, sch, cols);
}
// ---- the walk of the cell sizes, on the device ---------------------------------------------------------------------------------
// Every cell names its own size, so finding the cell boundaries is a pointer chase: 13 ms of one host core per 270 MB, the longest
// item of the input path.  Here every byte position with a plausible cell header is a candidate (k_walk_candidates: row inside the
// array, column >= 0, a size that fits), a candidate's successor is the candidate at position + size, and the true cells are the
// ones reachable from position 0: pointer doubling over the candidate list (log2 #candidates rounds, k_walk_jump).  False
// candidates (payload bytes that look like a header) are harmless: nothing true points at them.  The chain's end says what
// happened: it ended exactly at the buffer's end, the buffer ends inside a cell / a header (more bytes follow: the window is cut
// in front of that cell), or a successor is no candidate (malformed stream: error).
constexpr uint32_t kWalkEnd = 0, kWalkTruncNext = 1, kWalkTruncSelf = 2, kWalkBroken = 3;   // successor codes: M + code
__device__ __forceinline__ uint64_t load_u64_unaligned(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ bool walk_plausible(const uint8_t* __restrict__ cells, uint64_t p, uint64_t nbytes, uint64_t nrows, uint64_t& size) {
  if (p + 32 > nbytes) return false;
  size = load_u64_unaligned(cells + p + 16);                  // the most selective test first: most positions stop here
  if (size < 32 || size >= (1ull << 31)) return false;
  const uint64_t row = load_u64_unaligned(cells + p), col = load_u64_unaligned(cells + p + 8);
  return row < nrows && (int64_t)col >= 0;
}
// 16 byte positions per lane, out of three aligned 16-byte loads (a load per position at a byte stride of one costs the texture
// path 64 different addresses per instruction: 4.1 ms per 270 MB chunk, the longest kernel of the input path): the header
// fields of position j are funnel-shifted (v_alignbyte) out of the twelve words in registers; four lanes make one bitmap word.
__global__ void __launch_bounds__(256) k_walk_candidates(const uint8_t* __restrict__ cells, uint64_t nbytes, uint64_t nrows, uint64_t nwords, uint64_t* __restrict__ bitmap, uint32_t* __restrict__ word_count) {
  const uint64_t lane_base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u;     // (cells is 16-byte aligned and padded with 64 zero bytes)
  uint32_t mask = 0;
  if (lane_base < nbytes) {
    const uint4* src = reinterpret_cast<const uint4*>(cells + lane_base);
    const uint4 a = src[0], b = src[1], c = src[2];
    const uint32_t D[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    // pass 1 (registers only): the positions whose own header is plausible.  Their successors are looked at afterwards, one per
    // lane and round - as 16 divergent blocks, each with its dependent random loads, the kernel took 2.4 ms per 270 MB chunk
    uint64_t succ0 = 0, succ1 = 0, succ2 = 0, succ3 = 0;      // successor positions of up to four such headers
    uint32_t at = 0;                                           // their positions j, one nibble each
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int q = j >> 2;
      const uint32_t sh = (uint32_t)(j & 3);
      // size = bytes [j + 16, j + 24): the most selective test (most positions stop here)
      const uint32_t slo = sh ? __builtin_amdgcn_alignbyte(D[q + 5], D[q + 4], sh) : D[q + 4];
      const uint32_t shi = sh ? __builtin_amdgcn_alignbyte(D[q + 6], D[q + 5], sh) : D[q + 5];
      bool ok = shi == 0u && slo >= 32u && slo < (1u << 31) && lane_base + (uint64_t)j + 32u <= nbytes;
      if (ok) {
        const uint32_t rlo = sh ? __builtin_amdgcn_alignbyte(D[q + 1], D[q], sh) : D[q];
        const uint32_t rhi = sh ? __builtin_amdgcn_alignbyte(D[q + 2], D[q + 1], sh) : D[q + 1];
        const uint32_t chi = sh ? __builtin_amdgcn_alignbyte(D[q + 4], D[q + 3], sh) : D[q + 3];      // bytes [j + 12, j + 16): the column's high word
        ok = (((uint64_t)rhi << 32) | rlo) < nrows && (int32_t)chi >= 0;
        // A header must also POINT at a header (or past the buffer's end): a true cell always does, a look-alike inside a payload
        // of small integers rarely - without this test a c2 chunk has ~10 candidates per cell, and every round of the pointer
        // doubling below pays for them.  (A stream whose chain breaks is still found out: the cell before the break loses its successor.)
        const uint64_t sp = lane_base + (uint64_t)j + slo;
        if (ok && sp + 32u <= nbytes) {
          if (cnt < 4u) {
            if (cnt == 0u) succ0 = sp; else if (cnt == 1u) succ1 = sp; else if (cnt == 2u) succ2 = sp; else succ3 = sp;
            at |= (uint32_t)j << (4u * cnt);
            ++cnt;
            ok = false;                                        // decided in pass 2
          } else { uint64_t size2; ok = walk_plausible(cells, sp, nbytes, nrows, size2); }   // (a fifth one in 16 bytes: at once)
        }
      }
      mask |= (ok ? 1u : 0u) << j;
    }
    // pass 2: the successors' headers, round c = the c-th pending position of every lane
#pragma unroll
    for (uint32_t c = 0; c < 4u; ++c) {
      if (cnt > c) {
        const uint64_t sp = c == 0u ? succ0 : c == 1u ? succ1 : c == 2u ? succ2 : succ3;
        uint64_t size2;
        if (walk_plausible(cells, sp, nbytes, nrows, size2)) mask |= 1u << ((at >> (4u * c)) & 15u);
      }
    }
  }
  unsigned long long v = (unsigned long long)mask << (16u * (threadIdx.x & 3u));
  v |= __shfl_xor(v, 1, 64);
  v |= __shfl_xor(v, 2, 64);
  const uint64_t word = lane_base >> 6;
  if ((threadIdx.x & 3u) == 0u && word < nwords) { bitmap[word] = v; word_count[word] = (uint32_t)__popcll(v); }
}
__global__ void k_walk_successors(const uint8_t* __restrict__ cells, uint64_t nbytes, uint64_t nwords, const uint64_t* __restrict__ bitmap, const uint32_t* __restrict__ word_rank,
                                  uint32_t M, uint64_t* __restrict__ pos, uint32_t* __restrict__ succ) {
  const uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;     // one bitmap word per thread: its candidates, in order
  if (wi >= nwords) return;
  uint32_t i = word_rank[wi];
  for (uint64_t w = bitmap[wi]; w; w &= w - 1ull, ++i) {
    const uint64_t p = (wi << 6) + (uint64_t)__builtin_ctzll(w);
    const uint64_t s = p + load_u64_unaligned(cells + p + 16);
    pos[i] = p;
    uint32_t nx;
    if (s > nbytes) nx = M + kWalkTruncSelf;
    else if (s == nbytes) nx = M + kWalkEnd;
    else if (s + 32 > nbytes) nx = M + kWalkTruncNext;
    else {
      const uint64_t ws = bitmap[s >> 6];
      nx = ((ws >> (s & 63)) & 1ull) ? word_rank[s >> 6] + (uint32_t)__popcll(ws & ((1ull << (s & 63)) - 1ull)) : M + kWalkBroken;
    }
    succ[i] = nx;
  }
}
__global__ void k_walk_init(uint32_t M, const uint64_t* __restrict__ pos, uint8_t* __restrict__ reach, uint32_t* __restrict__ jump, uint32_t* err) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) reach[i] = (i == 0 && pos[0] == 0) ? 1 : 0;
  else if (i < M + 4) { reach[i] = 0; jump[i] = i; }                       // the four end codes point at themselves
  if (i == 0 && pos[0] != 0) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);   // the stream does not begin with a cell
}
// one round of pointer doubling: whatever is reached marks what it points at, then every pointer jumps twice as far
__global__ void k_walk_jump(uint32_t M, uint8_t* __restrict__ reach, const uint32_t* __restrict__ jump_in, uint32_t* __restrict__ jump_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const uint32_t j = jump_in[i];
  if (reach[i]) reach[j] = 1;
  jump_out[i] = jump_in[j];
}
__global__ void k_walk_flags(uint32_t M, const uint8_t* __restrict__ reach, const uint32_t* __restrict__ succ, uint32_t* __restrict__ flag) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= M) flag[i] = (i < M && reach[i] && succ[i] != M + kWalkTruncSelf) ? 1u : 0u;     // a cell the buffer cuts through is not taken
}
// walk_out: [0] number of cells, [1] end offset of the last one, [2] how the chain ended
__global__ void k_walk_compact(uint32_t M, const uint8_t* __restrict__ cells, const uint8_t* __restrict__ reach, const uint32_t* __restrict__ succ, const uint32_t* __restrict__ flag,
                               const uint32_t* __restrict__ dest, const uint64_t* __restrict__ pos, uint64_t* __restrict__ cell_off, uint64_t* __restrict__ walk_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  if (flag[i]) cell_off[dest[i]] = pos[i];
  if (reach[i] && succ[i] >= M) {                                         // the one reached cell that ends the chain
    const uint32_t code = succ[i] - M;
    const uint64_t end = code == kWalkTruncSelf ? pos[i] : pos[i] + load_u64_unaligned(cells + pos[i] + 16);
    cell_off[dest[M]] = end;
    walk_out[0] = dest[M]; walk_out[1] = end; walk_out[2] = code;
  }
}
// After k_cells_measure: begin columns must not decrease; with more bytes to come the last begin column seen is left for the
// next call (it may continue there).  cut_out: [0] cells taken, [1] bytes taken, [2] next begin column, [3] "one column fills the
// buffer", [4] kept cells, [5] markers, [6] first begin, [7] last begin, [8] bytes of the kept cells, [9 ...] elements of every
// variable-length column.
__global__ void k_walk_order(const int64_t* __restrict__ begin, int64_t n, uint32_t* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 1 && i < n && begin[i] < begin[i - 1]) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);
}
struct WalkCutCols { const uint32_t* off[GDB_MAX_FIELDS]; int32_t nf; };
__global__ void k_walk_cut(const int64_t* __restrict__ begin, const uint64_t* __restrict__ cell_off, const uint32_t* __restrict__ keep_dest, const uint32_t* __restrict__ mark_dest,
                           int64_t n, int whole_columns_only, WalkCutCols vc, int64_t* __restrict__ cut_out) {
  if (blockIdx.x || threadIdx.x) return;
  int64_t take = n, next_begin = INT64_MAX, single = 0;
  if (whole_columns_only) {
    if (n == 0) single = 1;
    else {
      const int64_t last = begin[n - 1];
      int64_t lo = 0, hi = n;
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (begin[mid] < last) lo = mid + 1; else hi = mid; }
      if (lo > 0) { take = lo; next_begin = last; } else { take = 0; single = 1; next_begin = last; }
    }
  }
  cut_out[0] = take; cut_out[1] = (int64_t)cell_off[take]; cut_out[2] = next_begin; cut_out[3] = single;
  cut_out[4] = keep_dest[take]; cut_out[5] = mark_dest[take];
  cut_out[6] = take > 0 ? begin[0] : 0; cut_out[7] = take > 0 ? begin[take - 1] : 0;
  for (int f = 0; f < vc.nf; ++f) cut_out[9 + f] = vc.off[f] ? (int64_t)vc.off[f][take] : 0;
}
__global__ void k_walk_kept_bytes(const uint32_t* __restrict__ keep, const uint64_t* __restrict__ cell_off, const int64_t* __restrict__ cut_out, unsigned long long* total) {
  __shared__ unsigned long long part[kBlock / 64];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = (i < cut_out[0] && keep[i]) ? (unsigned long long)(cell_off[i + 1] - cell_off[i]) : 0ull;
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_xor(mine, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) { unsigned long long sum = 0; for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += part[w]; if (sum) atomicAdd(total, sum); }
}

__global__ void k_cells_markers(const uint32_t* is_marker, const uint32_t* mdest, const int64_t* begin, int64_t n, int64_t* marker_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && is_marker[i]) marker_out[mdest[i]] = begin[i];
}
__global__ void k_cells_order_check(const int32_t* row, const int64_t* begin, int64_t n, uint32_t* err) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d < 1 || d >= n) return;
  if (begin[d] < begin[d - 1] || (begin[d] == begin[d - 1] && row[d] <= row[d - 1])) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);
}

// ---- carry-over between column windows of an array that is streamed through HBM ---------------------------------------
// The device analogue of VariantQueryProcessorScanState (query_variants.h:126-191): what survives a window is, per sample, its
// last cell if that cell's interval reaches the first column of the next window (an earlier cell of the same sample is either
// over or has been overridden by its successor, query_variants.cc:512-543).  At most one cell per sample.
__global__ void k_last_in_row(const int32_t* __restrict__ row, int64_t C, long long* last) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) atomicMax(&last[row[c]], (long long)c);
}
__global__ void k_carry_select(const long long* last, const int64_t* __restrict__ end, int32_t N, int64_t carry_from, uint64_t* keys) {
  const int32_t r = (int32_t)(blockIdx.x * blockDim.x + threadIdx.x);
  if (r >= N) return;
  const long long c = last[r];
  keys[r] = (c >= 0 && end[c] >= carry_from) ? (uint64_t)c : ~0ull;
}
__global__ void k_count_valid_keys(const uint64_t* sorted, int32_t N, int64_t* out) {   // keys are sorted: first index holding ~0
  if (blockIdx.x || threadIdx.x) return;
  int32_t lo = 0, hi = N;
  while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (sorted[mid] != ~0ull) lo = mid + 1; else hi = mid; }
  *out = lo;
}
__global__ void k_gather_coords(const uint64_t* list, int64_t K, const int32_t* row, const int64_t* begin, const int64_t* end, int32_t* row_out, int64_t* begin_out, int64_t* end_out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int64_t c = (int64_t)list[k];
  row_out[k] = row[c]; begin_out[k] = begin[c]; end_out[k] = end[c];
}
__global__ void k_gather_fixed(const uint64_t* list, int64_t K, const char* src, uint32_t bytes_per_cell, char* dst) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const char* s = src + (size_t)list[k] * bytes_per_cell;
  char* d = dst + (size_t)k * bytes_per_cell;
  for (uint32_t b = 0; b < bytes_per_cell; ++b) d[b] = s[b];
}
__global__ void k_gather_var_len(const uint64_t* list, int64_t K, const uint32_t* off, uint32_t* len) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > K) return;
  len[k] = k < K ? off[list[k] + 1] - off[list[k]] : 0u;
}
__global__ void k_gather_var(const uint64_t* list, int64_t K, const uint32_t* off, const char* src, uint32_t elem_size, const uint32_t* new_off, char* dst) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  const uint32_t a = off[list[k]], n = (off[list[k] + 1] - a) * elem_size;
  const char* s = src + (size_t)a * elem_size;
  char* d = dst + (size_t)new_off[k] * elem_size;
  for (uint32_t b = 0; b < n; ++b) d[b] = s[b];
}

// The contents of a part read from a fragment file are checked where they arrive (the header of the file is validated at open, but a
// damaged column must not send a later kernel out of bounds): rows inside the query's rows, END >= begin, cells in (begin, row) order
__global__ void k_fragment_part_check(const int32_t* __restrict__ row, const int64_t* __restrict__ begin, const int64_t* __restrict__ end, int64_t n, int32_t nrows, uint32_t* err) {
  const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= n) return;
  bool bad = row[d] < 0 || row[d] >= nrows || end[d] < begin[d];
  if (d > 0 && (begin[d] < begin[d - 1] || (begin[d] == begin[d - 1] && row[d] <= row[d - 1]))) bad = true;
  if (bad) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);
}
// ... and the n + 1 offsets of a variable-length column (already rebased to the part): from 0 to `total` elements, never decreasing
__global__ void k_fragment_offsets_check(const uint32_t* __restrict__ off, int64_t n, uint32_t total, uint32_t* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  const bool bad = (i == 0 && off[0] != 0u) || (i == n && off[n] != total) || (i < n && off[i + 1] < off[i]);
  if (bad) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);
}
__global__ void k_copy_offsets(const uint32_t* src, int64_t n, uint32_t base, uint32_t* dst) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i] + base;
}
// record offsets (the page splitter's table) and the largest record (one atomic per workgroup): the host only needs the two
// totals unless the interval has to be paged
__global__ void k_gather_record_offsets(const uint64_t* chunk_off, int nchunks, int64_t P, uint64_t* rec_off, unsigned long long* max_record) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long mine = 0;
  if (k <= P) {
    const uint64_t at = chunk_off[k * nchunks];
    rec_off[k] = at;
    if (k < P) mine = chunk_off[(k + 1) * nchunks] - at;
  }
  for (int d = 32; d > 0; d >>= 1) { const unsigned long long o = __shfl_xor(mine, d); mine = o > mine ? o : mine; }
  __shared__ unsigned long long wave_max[kBlock / 64];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) mine = wave_max[w] > mine ? wave_max[w] : mine;
    if (mine) atomicMax(max_record, mine);
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
struct DevicePipeline::Impl {
  HostPlan hp;
  int device = 0;
  hipStream_t stream = nullptr;        // the stream in use: stream_compute, or stream_stage between begin_staging_from() and finish_staging()
  hipStream_t stream_page = nullptr;   // the page kernel on a high-priority stream of its own (several windows in flight: it is dispatched ahead of the other lanes' kernels and runs at its stand-alone duration)
  bool page_priority = false;
  hipEvent_t ev_page_fork = nullptr, ev_page_join = nullptr;
  hipStream_t stream_compute = nullptr, stream_stage = nullptr;
  // A window staged ahead (overlapped staging) shares the GPU with the other pipeline's page assembly, whose kernels fill every CU
  // for milliseconds: the short staging kernels (1 ms per 270 MB chunk) run on a stream of higher priority so that the staging
  // thread's many small round trips do not queue up behind them.  Measured (c3 shape, 5 Mb, alternating on one box): the staging
  // thread waits less (0.88 -> 0.69 s) and the page assembly takes as much longer (1.77 -> 1.68 M positions/s of device time):
  // 1.27 vs 1.28 M positions/s end to end - no gain, so it is off unless GDBAMD_STAGE_PRIORITY=1.
  void use_stage_stream() {
    static const bool on = getenv("GDBAMD_STAGE_PRIORITY") && atoi(getenv("GDBAMD_STAGE_PRIORITY")) != 0;
    if (!on) return;
    if (!stream_stage) {
      int least = 0, greatest = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_CHECK(hipStreamCreateWithPriority(&stream_stage, hipStreamDefault, greatest));
    }
    HIP_CHECK(hipStreamSynchronize(stream_compute));     // (idle by now: the pipeline's last page was handed over before its buffers are reused)
    stream = stream_stage;
  }
  void use_compute_stream() { stream = stream_compute; }
  // staged fragment
  FragmentView fr;
  bool owns_fragment = false;
  std::vector<void*> owned;
  // small tables
  DevBuf<char> names_text, contig_names, ref_bases;
  DevBuf<int32_t> field_name_off, field_name_len, filter_name_off, filter_name_len, filter_bcf_id;
  DevBuf<uint32_t> bcf_part, bcf_fmeta, bcf_foff, bcf_lindiv; DevBuf<uint64_t> bcf_rec_size; DevBuf<uint8_t> bcf_same;   // BCF page assembly
  DevBuf<GdbContig> contigs;
  int64_t ref_begin = 0, ref_len = 0;
  // per-cell
  DevBuf<uint64_t> vmask; DevBuf<uint32_t> cflags; DevBuf<int32_t> dpval, k_lo, k_hi; DevBuf<int64_t> eff_end;
  DevBuf<int32_t> row_keys, row_keys_sorted; DevBuf<int64_t> cell_ids, perm, rm_begin, row_ptr, span, span_max, cwin;
  int64_t max_span = 0;
  DevBuf<uint64_t> ev_keys, ev_keys_sorted; DevBuf<int64_t> ev_delta, ev_incl; DevBuf<int32_t> run_end, run_excl;
  DevBuf<int64_t> bpos, bnrec, rbase; DevBuf<int32_t> bcov, bdel;
  DevBuf<int64_t> rstart, rend;
  DevBuf<int32_t> diff, first_record_at; DevBuf<uint64_t> diff_packed; DevBuf<int64_t> heavy_count, hoff;
  DevBuf<uint64_t> inc_keys, inc_keys_sorted; DevBuf<int64_t> inc_vals, inc_vals_sorted, hbase;
  DevBuf<uint32_t> lut_len, i2m_off; DevBuf<int8_t> i2m, gt_override; DevBuf<uint8_t> iflags;
  DevBuf<uint8_t> num_alleles, rflags; DevBuf<uint32_t> fmt_mask, prefix_len; DevBuf<char> site_staging, spill_buf; DevBuf<int32_t> spill_chunk;
  DevBuf<uint64_t> chunk_size, chunk_off, rec_off; DevBuf<unsigned long long> max_record;
  std::unique_ptr<BgzfDeviceCompressor> bgzf;   // output formats "z" / "b"
  // "z" / "b": the compression of a page is queued right behind the kernels that assemble it (finish_page only collects the size),
  // so the device works on page k + 1 - assembly and compression - while the host hands out page k.  (Round 6 measured the compression on a
  // stream of its own, beside the assembly of the next page: slower - 5.2-5.4 against 5.7-5.9 M positions/s for "z" - the two compete for the
  // CUs instead of taking turns: profiles/r6_ab_bgzf_own_stream.txt.)
  void queue_compression(int ai, char* arena, uint64_t page_bytes) {
    if (!hp.bgzf || page_bytes == 0) return;
    if (!bgzf) { bgzf.reset(new BgzfDeviceCompressor); bgzf->set_text(!hp.plan.bcf_mode); bgzf->set_bcf2(hp.plan.bcf_mode != 0); }   // ("z": pages of VCF text through the anchored kernel; "b": BCF2 records through the byte-level one)
    bgzf->enqueue(ai, arena, page_bytes, arena, (void*)stream);
  }
  DevBuf<unsigned int> slot_bump;               // pass 0's bump allocator of the overflow text pool: next unit, texts it could not place
  bool long_texts_seen = false;                 // an interval of this pipeline had texts longer than an inline slot
  uint64_t pool_ovf_need = 0;                   // most 16-byte units of overflow texts an interval has needed so far
  DevBuf<char> arena[2], temp;       // two output arenas: a consumer drains one while the next page is assembled into the other
  DevBuf<uint32_t> err; DevBuf<int32_t> counters;
  DevBuf<uint32_t> site_key, site_key_sorted; DevBuf<int32_t> site_ord_in, site_ord;
  int ctx_slot = -1;                 // this pipeline's element of c_ex
  DevBuf<char> calls_names, calls_text; DevBuf<int32_t> calls_name_off; DevBuf<uint64_t> calls_len, calls_off; bool calls_names_ready = false; DevBuf<int64_t> calls_array_row; size_t calls_array_row_n = 0;   // --print-calls
  // persistent events (no create / destroy per interval) and one pinned block for every scalar that comes back to the host
  hipEvent_t ev_prep[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_page[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // per arena: start, before / after the page assembly, page done
  hipEvent_t arena_release[2] = {nullptr, nullptr};   // consumer's "this arena has been read" events (not owned)
  struct HostBlock { uint64_t q[80]; uint32_t page_err[2]; } *hb = nullptr;   // hipHostMalloc
  DevBuf<SiteCtx> d_sx;
  DevBuf<uint64_t> med_keys, med_keys_sorted; DevBuf<uint32_t> med_idx, med_idx_sorted;
  DevBuf<int32_t> big_index, big_list; DevBuf<uint32_t> big_value; DevBuf<uint8_t> big_ok;
  DevBuf<unsigned long long> remap_total;
  DevBuf<int32_t> huge_index, huge_list; DevBuf<HugeSiteOut> huge_out; DevBuf<SiteCtx> d_sx0;
  DevBuf<uint32_t> slot_cell; DevBuf<float> tie_buf; DevBuf<unsigned long long> tie_used; DevBuf<uint32_t> scalar_pre;
  // entry text table
  DevBuf<unsigned long long> type_hkeys; DevBuf<int32_t> type_hrep, type_rep; DevBuf<uint8_t> type_hid, rtype;
  DevBuf<WalkCell> walk; DevBuf<int64_t> walk_inv;
  DevBuf<int32_t> jhint;
  DevBuf<unsigned long long> dbg_counters;
  DevBuf<uint32_t> pl_cnt, pl_ofs; DevBuf<PieceEntry> plist;     // piece lists (GDBAMD_ASM_PATH=3)
  uint32_t walk_epoch = 0;           // stamp of the interval whose cells the walk list describes (24 bits; 0: none yet)
  // cell-stream staging (append_cells)
  DevBuf<uint8_t> raw_cells; DevBuf<uint64_t> raw_off; DevBuf<int32_t> raw_row_map, raw_qrow; DevBuf<uint32_t> raw_keep, raw_dest, raw_len, raw_voff, raw_mark, raw_mdest;
  DevBuf<int64_t> raw_begin, raw_end;
  // the walk of the cell sizes on the device
  DevBuf<uint64_t> walk_bitmap, walk_pos, walk_out; DevBuf<uint32_t> walk_wcount, walk_wrank, walk_succ, walk_jump_a, walk_jump_b, walk_flag, walk_dest; DevBuf<uint8_t> walk_reach;
  DevBuf<int64_t> walk_cut; DevBuf<unsigned long long> walk_kept_bytes;
  DevBuf<uint8_t> inflate_in, inflate_out, inflate_scratch; DevBuf<uint64_t> inflate_off; DevBuf<uint32_t> inflate_want;   // DEFLATE tiles of a compressed fragment file
  DevBuf<long long> carry_last; DevBuf<uint64_t> carry_keys, carry_sorted; int64_t carried_cells = 0;
  DevBuf<uint2> chk_resolved; DevBuf<uint64_t> chk_sizes; DevBuf<unsigned long long> chk_count;   // GDBAMD_SIZE3_CHECK
  DevBuf<uint2> resolved;          // (pool offset, length) of every (record, sample): whole interval, or one page when that exceeds the budget
  DevBuf<uint32_t> res_off; DevBuf<uint8_t> res_len8;   // the same matrix in its compact layout (ResMatrix): offsets and byte lengths as two planes
  DevBuf<unsigned int> slot_over;  // [kBumpShards] "an entry text is longer than 255 bytes" (pass 0 of the slot kernels)
  DevBuf<uint32_t> type_occ;
  DevBuf<uint32_t> untabled, ubase; DevBuf<int32_t> urec, iota, order; DevBuf<uint32_t> order_keys, order_keys_sorted;
  DevBuf<uint64_t> tmask; DevBuf<uint32_t> nslots, tbase, slot_len, slot_units, slot_off; DevBuf<uint2> slot_desc; DevBuf<char> pool, pool_ovf;
  bool classified = false;
  uint64_t fragment_generation = 0;   // counts the fragments this pipeline has held: a pipeline that adopted this one's fragment (CombineEngine lanes) re-adopts when it changes
  struct Part { FragmentView v; std::vector<size_t> data_bytes; std::vector<void*> bufs; };
  std::vector<Part> parts;
  std::vector<int> col_elem_size; std::vector<bool> col_var; std::vector<int> col_fixed_num;
  std::unique_ptr<DevicePipeline::FragmentFile> ff;   // open columnar fragment file (windowed reads)
  struct IntervalState {
    bool active = false;
    int64_t P = 0, kp = 0;
    int nchunks = 0;
    uint64_t max_record_bytes = 0, total_bytes = 0;
    float write_kernel_ms = 0;
    std::vector<uint64_t> rec_off;
    IntervalStats stats;
    SiteCtx sx; EntryCtx ex; RowIndex ri; SiteOut so; RecordTable rec; AsmCtx ac; PieceCtx pc2;
    bool resolved_whole = false, piece_path = false, res_compact = false; int asm_path = 0, NT = 0; int64_t P_rows = 0;
    bool bcf = false; int bcf_F = 0; BcfLayout lay{nullptr, nullptr, nullptr, nullptr};
    bool events = false; int evrun = 0; EventBuf eb{nullptr, nullptr, nullptr, 0};
  } iv;

  void* temp_storage(size_t bytes) { temp.ensure(bytes + 256); return temp.p; }
  template <class K, class V> void sort_pairs(const K* kin, K* kout, const V* vin, V* vout, size_t n, int end_bit) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, end_bit, stream));
    void* t = temp_storage(bytes);
    HIP_CHECK(rocprim::radix_sort_pairs(t, bytes, kin, kout, vin, vout, n, 0, end_bit, stream));
  }
  template <class K> void sort_keys(const K* kin, K* kout, size_t n, int end_bit) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::radix_sort_keys(nullptr, bytes, kin, kout, n, 0, end_bit, stream));
    void* t = temp_storage(bytes);
    HIP_CHECK(rocprim::radix_sort_keys(t, bytes, kin, kout, n, 0, end_bit, stream));
  }
  template <class T, class Op> void incl_scan(const T* in, T* out, size_t n, Op op) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, in, out, n, op, stream));
    void* t = temp_storage(bytes);
    HIP_CHECK(rocprim::inclusive_scan(t, bytes, in, out, n, op, stream));
  }
  template <class T> void excl_scan(const T* in, T* out, size_t n) {
    size_t bytes = 0;
    HIP_CHECK(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), stream));
    void* t = temp_storage(bytes);
    HIP_CHECK(rocprim::exclusive_scan(t, bytes, in, out, T(0), n, rocprim::plus<T>(), stream));
  }
  // a + b with one stream synchronisation (the usual "last exclusive-scan element + last count" total)
  // (all read-backs land in the pinned block first: a copy to pageable memory would go through the runtime's staging path)
  template <class A, class B> int64_t read_back_sum(const A* pa, const B* pb) {
    HIP_CHECK(hipMemcpyAsync(&hb->q[0], pa, sizeof(A), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(&hb->q[1], pb, sizeof(B), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    A a; B b;
    memcpy(&a, &hb->q[0], sizeof(A)); memcpy(&b, &hb->q[1], sizeof(B));
    return (int64_t)a + (int64_t)b;
  }
  // several scalars, one stream synchronisation
  struct Pending { void* dst; const void* src; size_t bytes; };
  void read_back_many(std::initializer_list<Pending> items) {
    size_t at = 0;
    for (const Pending& it : items) {
      if (at + it.bytes > sizeof(hb->q)) throw GenomicsDBDeviceException("read_back_many: pinned block too small");
      HIP_CHECK(hipMemcpyAsync((char*)hb->q + at, it.src, it.bytes, hipMemcpyDeviceToHost, stream));
      at += (it.bytes + 7) & ~(size_t)7;
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    at = 0;
    for (const Pending& it : items) { memcpy(it.dst, (const char*)hb->q + at, it.bytes); at += (it.bytes + 7) & ~(size_t)7; }
  }
  template <class T> T read_back(const T* p) {
    T v;
    HIP_CHECK(hipMemcpyAsync(&hb->q[0], p, sizeof(T), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    memcpy(&v, &hb->q[0], sizeof(T));
    return v;
  }
  // records [k0, k0+n) in (type, position) order -> order[]
  int64_t order_k0 = -1, order_n = -1;   // what `order` currently holds
  void order_by_type(int64_t k0, int64_t n) {
    if (k0 == order_k0 && n == order_n) return;   // a page that is the whole interval reuses the sizing pass's order
    order_k0 = k0; order_n = n;
    // Types are grouped inside blocks of consecutive records, not over the whole range: a wavefront still walks ~100
    // same-type records, but the page stores and matrix reads in flight stay within a few tens of MB (TLB reach, DRAM pages).
    const int block_log2 = order_block_log2();
    iota.ensure(n); order.ensure(n); order_keys.ensure(n); order_keys_sorted.ensure(n);
    hipLaunchKernelGGL(k_order_keys, dim3(blocks_for(n)), dim3(kBlock), 0, stream, (const uint8_t*)rtype.p, (int32_t)k0, n, block_log2, order_keys.p, iota.p);
    sort_pairs(order_keys.p, order_keys_sorted.p, iota.p, order.p, (size_t)n, std::min(32, 8 + bits_for((uint64_t)(n >> block_log2))));
  }
  // the change-list flavour keeps the interval's order for all its pages (pages that do not fall on order blocks re-sort `order`)
  DevBuf<int32_t> order_iv; bool order_iv_valid = false;
  DevBuf<uint2> ev_init, ev_buf; DevBuf<uint32_t> ev_count;
  BlockPool blocks;
  void free_owned() { for (void* p : owned) blocks.put(p); owned.clear(); }
  void free_parts() { for (auto& p : parts) for (void* b : p.bufs) blocks.put(b); parts.clear(); }
};

int DevicePipeline::device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

DevicePipeline::DevicePipeline(const HostPlan& hp, int device) : m_(new Impl) {
  m_->hp = hp;
  m_->device = device;
  if (device_count() <= 0) { delete m_; throw GenomicsDBDeviceException("no HIP device visible: the variant-combine path has no CPU fallback"); }
  HIP_CHECK(hipSetDevice(device));
  HIP_CHECK(hipStreamCreate(&m_->stream_compute));
  m_->stream = m_->stream_compute;
  memset(&m_->fr, 0, sizeof(m_->fr));
  auto up = [&](auto& buf, const auto* src, size_t n) {
    buf.ensure(std::max<size_t>(n, 1));
    if (n) HIP_CHECK(hipMemcpy(buf.p, src, n * sizeof(*src), hipMemcpyHostToDevice));
  };
  up(m_->names_text, hp.names_text.data(), hp.names_text.size());
  up(m_->contig_names, hp.contig_names.data(), hp.contig_names.size());
  up(m_->field_name_off, hp.field_name_off.data(), hp.field_name_off.size());
  up(m_->field_name_len, hp.field_name_len.data(), hp.field_name_len.size());
  up(m_->filter_name_off, hp.filter_name_off.data(), hp.filter_name_off.size());
  up(m_->filter_name_len, hp.filter_name_len.data(), hp.filter_name_len.size());
  up(m_->filter_bcf_id, hp.filter_bcf_id.data(), hp.filter_bcf_id.size());
  up(m_->contigs, hp.contigs.data(), hp.contigs.size());
  m_->err.ensure(4);
  m_->counters.ensure(16 + kCountSpread);
  for (auto& e : m_->ev_prep) HIP_CHECK(hipEventCreate(&e));
  for (auto& a : m_->ev_page) for (auto& e : a) HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipHostMalloc((void**)&m_->hb, sizeof(*m_->hb), hipHostMallocDefault));
  memset(m_->hb, 0, sizeof(*m_->hb));
  {
    std::lock_guard<std::mutex> g(g_ctx_slot_mutex);
    for (int i = 0; i < kCtxSlots && m_->ctx_slot < 0; ++i) if (!g_ctx_slot_used[i]) { g_ctx_slot_used[i] = true; m_->ctx_slot = i; }
  }
  if (m_->ctx_slot < 0) { this->~DevicePipeline(); throw GenomicsDBDeviceException("more than " + std::to_string(kCtxSlots) + " device pipelines alive in one process (GDBAMD_MAX_PIPELINES_PER_PROCESS: one per engine or query stream, two for an engine that streams an array in windows)"); }
}

DevicePipeline::~DevicePipeline() {
  if (!m_) return;
  if (m_->stream) (void)hipStreamSynchronize(m_->stream);
  m_->free_owned();
  m_->free_parts();
  m_->blocks.release_all();
  for (auto& e : m_->ev_prep) if (e) (void)hipEventDestroy(e);
  for (auto& a : m_->ev_page) for (auto& e : a) if (e) (void)hipEventDestroy(e);
  if (m_->hb) (void)hipHostFree(m_->hb);
  if (m_->ctx_slot >= 0) { std::lock_guard<std::mutex> g(g_ctx_slot_mutex); g_ctx_slot_used[m_->ctx_slot] = false; }
  if (m_->stream_compute) (void)hipStreamDestroy(m_->stream_compute);
  if (m_->stream_stage) (void)hipStreamDestroy(m_->stream_stage);
  if (m_->stream_page) { (void)hipStreamDestroy(m_->stream_page); (void)hipEventDestroy(m_->ev_page_fork); (void)hipEventDestroy(m_->ev_page_join); }
  delete m_;
  m_ = nullptr;
}

void DevicePipeline::stage_fragment(const HostFragment& hf) {
  HIP_CHECK(hipSetDevice(m_->device));
  m_->free_owned();
  FragmentView v;
  memset(&v, 0, sizeof(v));
  v.ncells = hf.ncells();
  auto up = [&](const void* src, size_t bytes) -> void* {
    void* d = m_->blocks.get(bytes);
    m_->owned.push_back(d);
    if (bytes) HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    return d;
  };
  v.row = (const int32_t*)up(hf.row.data(), hf.row.size() * 4);
  v.begin = (const int64_t*)up(hf.begin.data(), hf.begin.size() * 8);
  v.end = (const int64_t*)up(hf.end.data(), hf.end.size() * 8);
  for (size_t f = 0; f < hf.cols.size(); ++f) {
    v.col[f].data = up(hf.cols[f].data.data(), hf.cols[f].data.size());
    v.col[f].off = hf.cols[f].var ? (const uint32_t*)up(hf.cols[f].off.data(), hf.cols[f].off.size() * 4) : nullptr;
  }
  v.nmarkers = (int64_t)hf.marker_begin.size();
  v.marker_begin = (const int64_t*)up(hf.marker_begin.data(), hf.marker_begin.size() * 8);
  m_->fr = v;
  m_->owns_fragment = true;
  m_->classified = false; ++m_->fragment_generation;
}

void DevicePipeline::begin_staging(int64_t carry_from) { begin_staging_from(*this, carry_from); }

// The same with the carried intervals taken from the fragment ANOTHER pipeline (of the same plan, on the same device) has staged:
// that pipeline keeps computing on its fragment - nothing of it is written - while this one stages the next column window.
void DevicePipeline::begin_staging_from(DevicePipeline& source, int64_t carry_from) {
  Impl& S = *m_;
  Impl& SRC = *source.m_;
  S.free_parts();
  S.carried_cells = 0;
  HIP_CHECK(hipSetDevice(S.device));
  S.use_stage_stream();
  if (carry_from == INT64_MIN || SRC.fr.ncells == 0) return;
  // ---- the staged fragment's live intervals at carry_from become the first part of the next fragment ---------------------
  HIP_CHECK(hipSetDevice(S.device));
  hipStream_t st = S.stream;
  const int nf = S.hp.plan.nfields;
  const int32_t N = S.hp.plan.num_query_rows;
  if ((int)SRC.col_elem_size.size() < nf || !SRC.owns_fragment) throw GenomicsDBDeviceException("carry-over needs a fragment staged by a pipeline");
  if (&SRC != &S) { S.col_elem_size = SRC.col_elem_size; S.col_var = SRC.col_var; S.col_fixed_num = SRC.col_fixed_num; }
  const FragmentView fr = SRC.fr;
  const int64_t C = fr.ncells;
  S.carry_last.ensure((size_t)N + 1); S.carry_keys.ensure((size_t)N + 1); S.carry_sorted.ensure((size_t)N + 1); S.cwin.ensure(4);
  HIP_CHECK(hipMemsetAsync(S.carry_last.p, 0xFF, (size_t)N * sizeof(long long), st));
  hipLaunchKernelGGL(k_last_in_row, dim3(blocks_for(C)), dim3(kBlock), 0, st, fr.row, C, S.carry_last.p);
  hipLaunchKernelGGL(k_carry_select, dim3(blocks_for(N)), dim3(kBlock), 0, st, (const long long*)S.carry_last.p, fr.end, N, carry_from, S.carry_keys.p);
  S.sort_keys(S.carry_keys.p, S.carry_sorted.p, (size_t)N, 64);
  hipLaunchKernelGGL(k_count_valid_keys, dim3(1), dim3(64), 0, st, (const uint64_t*)S.carry_sorted.p, N, S.cwin.p);
  const int64_t K = S.read_back(S.cwin.p);
  if (K == 0) return;
  Impl::Part part;
  memset(&part.v, 0, sizeof(part.v));
  part.v.ncells = K;
  auto alloc = [&](size_t bytes) -> void* { void* d = S.blocks.get(bytes); part.bufs.push_back(d); return d; };
  const uint64_t* list = S.carry_sorted.p;
  int32_t* row = (int32_t*)alloc((size_t)K * 4); int64_t* begin = (int64_t*)alloc((size_t)K * 8); int64_t* end = (int64_t*)alloc((size_t)K * 8);
  hipLaunchKernelGGL(k_gather_coords, dim3(blocks_for(K)), dim3(kBlock), 0, st, list, K, fr.row, fr.begin, fr.end, row, begin, end);
  part.v.row = row; part.v.begin = begin; part.v.end = end;
  // variable-length columns: lengths -> offsets -> totals (one read-back for all fields)
  std::vector<uint32_t*> new_off((size_t)nf, nullptr);
  std::vector<uint32_t> totals((size_t)nf, 0);
  S.raw_len.ensure((size_t)K + 2);
  for (int f = 0; f < nf; ++f) if (S.col_var[(size_t)f]) {
    new_off[(size_t)f] = (uint32_t*)alloc(((size_t)K + 1) * 4);
    hipLaunchKernelGGL(k_gather_var_len, dim3(blocks_for(K + 1)), dim3(kBlock), 0, st, list, K, fr.col[f].off, S.raw_len.p);
    S.excl_scan((const uint32_t*)S.raw_len.p, new_off[(size_t)f], (size_t)K + 1);
  }
  for (int f = 0; f < nf; ++f) if (S.col_var[(size_t)f]) totals[(size_t)f] = S.read_back(new_off[(size_t)f] + K);
  part.data_bytes.assign((size_t)nf, 0);
  for (int f = 0; f < nf; ++f) {
    const uint32_t es = (uint32_t)S.col_elem_size[(size_t)f];
    if (S.col_var[(size_t)f]) {
      const size_t bytes = (size_t)totals[(size_t)f] * es;
      char* data = (char*)alloc(bytes);
      hipLaunchKernelGGL(k_gather_var, dim3(blocks_for(K)), dim3(kBlock), 0, st, list, K, fr.col[f].off, (const char*)fr.col[f].data, es, (const uint32_t*)new_off[(size_t)f], data);
      part.v.col[f].data = data; part.v.col[f].off = new_off[(size_t)f];
      part.data_bytes[(size_t)f] = bytes;
    } else {
      const uint32_t bpc = (uint32_t)S.col_fixed_num[(size_t)f] * es;
      char* data = (char*)alloc((size_t)K * bpc);
      hipLaunchKernelGGL(k_gather_fixed, dim3(blocks_for(K)), dim3(kBlock), 0, st, list, K, (const char*)fr.col[f].data, bpc, data);
      part.v.col[f].data = data;
      part.data_bytes[(size_t)f] = (size_t)K * bpc;
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  S.parts.push_back(part);
  S.carried_cells = K;
}

void DevicePipeline::append_fragment(const HostFragment& hf) {
  HIP_CHECK(hipSetDevice(m_->device));
  if (hf.ncells() == 0) return;
  Impl::Part part;
  memset(&part.v, 0, sizeof(part.v));
  part.v.ncells = hf.ncells();
  auto up = [&](const void* src, size_t bytes) -> void* {
    void* d = m_->blocks.get(bytes);
    part.bufs.push_back(d);
    if (bytes) HIP_CHECK(hipMemcpy(d, src, bytes, hipMemcpyHostToDevice));
    return d;
  };
  part.v.row = (const int32_t*)up(hf.row.data(), hf.row.size() * 4);
  part.v.begin = (const int64_t*)up(hf.begin.data(), hf.begin.size() * 8);
  part.v.end = (const int64_t*)up(hf.end.data(), hf.end.size() * 8);
  m_->col_elem_size.resize(hf.cols.size()); m_->col_var.resize(hf.cols.size()); m_->col_fixed_num.resize(hf.cols.size());
  for (size_t f = 0; f < hf.cols.size(); ++f) {
    part.v.col[f].data = up(hf.cols[f].data.data(), hf.cols[f].data.size());
    part.v.col[f].off = hf.cols[f].var ? (const uint32_t*)up(hf.cols[f].off.data(), hf.cols[f].off.size() * 4) : nullptr;
    part.data_bytes.push_back(hf.cols[f].data.size());
    m_->col_elem_size[f] = hf.cols[f].elem_size; m_->col_var[f] = hf.cols[f].var; m_->col_fixed_num[f] = hf.cols[f].fixed_num;
  }
  part.v.nmarkers = (int64_t)hf.marker_begin.size();
  part.v.marker_begin = (const int64_t*)up(hf.marker_begin.data(), hf.marker_begin.size() * 8);
  m_->parts.push_back(part);
}

int64_t DevicePipeline::carried_cells() const { return m_->carried_cells; }

// Walk of the cell sizes: the one sequential dependency of the format (every cell names its own size).  Stops at `nbytes` or,
// with whole_columns_only, in front of the last begin column seen (it may continue behind the buffer).  offs gets the offset of
// every cell taken plus the end offset.
DevicePipeline::CellWalk DevicePipeline::walk_cells(const uint8_t* cells, uint64_t nbytes, const std::vector<int32_t>& row_map, bool whole_columns_only, std::vector<uint64_t>& offs) {
  CellWalk w;
  offs.clear();
  offs.reserve((size_t)(nbytes / 128) + 16);
  uint64_t off = 0, last_col_off = 0;
  size_t last_col_idx = 0;
  int64_t last_col = INT64_MIN, kept_at_col = 0, marks_at_col = 0; uint64_t bytes_at_col = 0;
  while (off < nbytes) {
    if (off + 32 > nbytes) { if (whole_columns_only) break; throw std::runtime_error("truncated cell stream"); }
    int64_t row, col; uint64_t sz;
    memcpy(&row, cells + off, 8); memcpy(&col, cells + off + 8, 8); memcpy(&sz, cells + off + 16, 8);
    if (sz < 32) throw std::runtime_error("malformed cell stream (cell size below the fixed part)");
    if (off + sz > nbytes) { if (whole_columns_only) break; throw std::runtime_error("truncated cell stream"); }
    if (col != last_col) {
      if (col < last_col) throw std::runtime_error("cells are not in column-major (col,row) order");
      last_col = col; last_col_off = off; last_col_idx = offs.size(); kept_at_col = w.nkept; marks_at_col = w.nmark; bytes_at_col = w.reference_cell_bytes;
    }
    offs.push_back(off);
    if (row >= 0 && (size_t)row < row_map.size()) { if (row_map[(size_t)row] >= 0) { ++w.nkept; w.reference_cell_bytes += sz; } else ++w.nmark; }
    off += sz;
  }
  if (whole_columns_only && offs.empty() && nbytes > 0) { w.single_column = true; }   // not even one whole cell: offer more bytes
  if (whole_columns_only && !offs.empty()) {
    // more bytes follow this buffer: the last begin column seen may continue there, so it is left for the next call
    if (last_col_idx > 0) { offs.resize(last_col_idx); off = last_col_off; w.nkept = kept_at_col; w.nmark = marks_at_col; w.reference_cell_bytes = bytes_at_col; }
    else { offs.clear(); off = 0; w.nkept = w.nmark = 0; w.reference_cell_bytes = 0; w.single_column = true; }   // one column fills the buffer: offer more bytes
    w.next_begin = last_col;
  }
  w.bytes_taken = off;
  if (!offs.empty()) { int64_t c0; memcpy(&c0, cells + offs[0] + 8, 8); w.first_begin = c0; int64_t c1; memcpy(&c1, cells + offs.back() + 8, 8); w.last_begin = c1; }
  offs.push_back(off);
  return w;
}

DevicePipeline::CellStreamInfo DevicePipeline::append_cells(const uint8_t* cells, uint64_t nbytes, const VariantArraySchemaLite& schema,
                                                          const std::vector<int>& attr_to_field, const std::vector<int32_t>& row_map, const std::vector<uint64_t>* walked,
                                                          const CellWalk* walk_info, bool whole_columns_only, CellWalk* walk_out) {
  Impl& S = *m_;
  CellStreamInfo info;
  HIP_CHECK(hipSetDevice(S.device));
  hipStream_t st = S.stream;
  if (walk_out) *walk_out = CellWalk();
  if (nbytes == 0) return info;
  if (schema.attrs.size() > (size_t)kMaxSchemaAttrs) throw UnsupportedOnDeviceException("more than 96 attributes in the array schema");
  static const bool trace = getenv("GDBAMD_STREAM_TRACE") != nullptr;
  const auto tw0 = std::chrono::steady_clock::now();
  const bool device_walk = walked == nullptr;
  const int nf = S.hp.plan.nfields;
  CellSchemaDev sch;
  memset(&sch, 0, sizeof(sch));
  sch.nattrs = (int32_t)schema.attrs.size();
  S.col_elem_size.assign((size_t)nf, 4); S.col_var.assign((size_t)nf, false); S.col_fixed_num.assign((size_t)nf, 1);
  for (size_t ai = 0; ai < schema.attrs.size(); ++ai) {
    const auto& a = schema.attrs[ai];
    sch.a[ai].var = a.var ? 1 : 0; sch.a[ai].elem_size = (int16_t)a.elem_size; sch.a[ai].num = a.num; sch.a[ai].field = attr_to_field[ai];
    const int f = attr_to_field[ai];
    if (f >= 0) { S.col_elem_size[(size_t)f] = a.elem_size; S.col_var[(size_t)f] = a.var; S.col_fixed_num[(size_t)f] = a.num; }
  }
  // ---- device: raw bytes, cell offsets, row map -------------------------------------------------------------------------
  HIP_CHECK(hipMemsetAsync(S.err.p, 0, sizeof(uint32_t), st));
  int64_t n = 0;
  if (!device_walk) {
    nbytes = walked->back();
    n = (int64_t)walked->size() - 1;
    if (walk_info->nkept == 0 && walk_info->nmark == 0) return info;
    S.raw_cells.ensure(nbytes + 64); S.raw_off.ensure((size_t)n + 1);
    HIP_CHECK(hipMemcpyAsync(S.raw_cells.p, cells, nbytes, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(S.raw_off.p, walked->data(), (size_t)(n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  } else {
    S.raw_cells.ensure(nbytes + 64);
    const auto th0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipMemcpyAsync(S.raw_cells.p, cells, nbytes, hipMemcpyHostToDevice, st));
    if (trace) { HIP_CHECK(hipStreamSynchronize(st)); fprintf(stderr, "[gdbamd stage] host -> device %.1f MB in %.4f s\n", nbytes / 1e6, std::chrono::duration<double>(std::chrono::steady_clock::now() - th0).count()); }
    HIP_CHECK(hipMemsetAsync(S.raw_cells.p + nbytes, 0, 64, st));
    const uint64_t nwords = (nbytes + 63) >> 6;
    S.walk_bitmap.ensure(nwords + 1); S.walk_wcount.ensure(nwords + 1); S.walk_wrank.ensure(nwords + 1); S.walk_out.ensure(8);
    HIP_CHECK(hipMemsetAsync(S.walk_out.p, 0, 4 * sizeof(uint64_t), st));
    const unsigned pos_blocks = (unsigned)((nwords * 4u + 255u) / 256u);      // 16 positions per lane, 4 lanes per bitmap word
    hipLaunchKernelGGL(k_walk_candidates, dim3(pos_blocks), dim3(256), 0, st, (const uint8_t*)S.raw_cells.p, nbytes, (uint64_t)1 << 40, nwords, S.walk_bitmap.p, S.walk_wcount.p);
    HIP_CHECK(hipMemsetAsync(S.walk_wcount.p + nwords, 0, sizeof(uint32_t), st));
    S.excl_scan((const uint32_t*)S.walk_wcount.p, S.walk_wrank.p, (size_t)nwords + 1);
    const uint32_t M = S.read_back(S.walk_wrank.p + nwords);
    if (M == 0) {
      if (whole_columns_only && nbytes < 32) { if (walk_out) walk_out->single_column = true; return info; }   // not even a header: offer more bytes
      throw std::runtime_error(nbytes < 32 ? "truncated cell stream" : "malformed cell stream (no cell header at its beginning)");
    }
    if (M >= 0xFFFFFFF0u) throw GenomicsDBDeviceException("more than 2^32 cell-header candidates in one part: stage in smaller parts");
    S.walk_pos.ensure(M); S.walk_succ.ensure((size_t)M + 4); S.walk_jump_a.ensure((size_t)M + 4); S.walk_jump_b.ensure((size_t)M + 4); S.walk_reach.ensure((size_t)M + 4);
    S.walk_flag.ensure((size_t)M + 1); S.walk_dest.ensure((size_t)M + 1); S.raw_off.ensure((size_t)M + 1);
    hipLaunchKernelGGL(k_walk_successors, dim3(blocks_for((int64_t)nwords)), dim3(kBlock), 0, st, (const uint8_t*)S.raw_cells.p, nbytes, nwords, (const uint64_t*)S.walk_bitmap.p, (const uint32_t*)S.walk_wrank.p, M,
                       S.walk_pos.p, S.walk_jump_a.p);
    HIP_CHECK(hipMemcpyAsync(S.walk_succ.p, S.walk_jump_a.p, (size_t)M * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_walk_init, dim3(blocks_for((int64_t)M + 4)), dim3(kBlock), 0, st, M, (const uint64_t*)S.walk_pos.p, S.walk_reach.p, S.walk_jump_a.p, S.err.p);
    HIP_CHECK(hipMemcpyAsync(S.walk_jump_b.p + M, S.walk_jump_a.p + M, 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    uint32_t* ja = S.walk_jump_a.p; uint32_t* jb = S.walk_jump_b.p;
    int rounds = 1;
    while ((1ull << rounds) < (uint64_t)M + 1) ++rounds;
    for (int r = 0; r <= rounds; ++r) {      // (a "every candidate's successor is its neighbour" shortcut does not apply: c2's payloads hold header look-alikes)
      hipLaunchKernelGGL(k_walk_jump, dim3(blocks_for((int64_t)M)), dim3(kBlock), 0, st, M, S.walk_reach.p, (const uint32_t*)ja, jb);
      std::swap(ja, jb);
    }
    hipLaunchKernelGGL(k_walk_flags, dim3(blocks_for((int64_t)M + 1)), dim3(kBlock), 0, st, M, (const uint8_t*)S.walk_reach.p, (const uint32_t*)S.walk_succ.p, S.walk_flag.p);
    S.excl_scan((const uint32_t*)S.walk_flag.p, S.walk_dest.p, (size_t)M + 1);
    hipLaunchKernelGGL(k_walk_compact, dim3(blocks_for((int64_t)M)), dim3(kBlock), 0, st, M, (const uint8_t*)S.raw_cells.p, (const uint8_t*)S.walk_reach.p, (const uint32_t*)S.walk_succ.p,
                       (const uint32_t*)S.walk_flag.p, (const uint32_t*)S.walk_dest.p, (const uint64_t*)S.walk_pos.p, S.raw_off.p, S.walk_out.p);
    uint64_t wo[3]; uint32_t eb0 = 0;
    S.read_back_many({{wo, S.walk_out.p, sizeof(wo)}, {&eb0, S.err.p, sizeof(uint32_t)}});
    if (eb0) throw std::runtime_error("malformed cell stream (it does not begin with a cell)");
    if (wo[2] == kWalkBroken) throw std::runtime_error("malformed cell stream (a cell's size does not lead to the next cell)");
    if (wo[2] != kWalkEnd && !whole_columns_only) throw std::runtime_error("truncated cell stream");
    n = (int64_t)wo[0];
    if (n == 0) { if (walk_out && whole_columns_only) walk_out->single_column = true; return info; }   // not even one whole cell: offer more bytes
  }
  const auto tw1 = std::chrono::steady_clock::now();
  S.raw_row_map.ensure(std::max<size_t>(row_map.size(), 1));
  if (!row_map.empty()) HIP_CHECK(hipMemcpyAsync(S.raw_row_map.p, row_map.data(), row_map.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
  S.raw_keep.ensure((size_t)n + 1); S.raw_dest.ensure((size_t)n + 1); S.raw_mark.ensure((size_t)n + 1); S.raw_mdest.ensure((size_t)n + 1); S.raw_qrow.ensure((size_t)n); S.raw_begin.ensure((size_t)n); S.raw_end.ensure((size_t)n);
  CellColumnsDev cols;
  memset(&cols, 0, sizeof(cols));
  int nvar = 0;
  for (int f = 0; f < nf; ++f) if (S.col_var[(size_t)f]) ++nvar;
  S.raw_len.ensure((size_t)std::max(nvar, 1) * ((size_t)n + 1)); S.raw_voff.ensure((size_t)std::max(nvar, 1) * ((size_t)n + 1));
  {
    int v = 0;
    for (int f = 0; f < nf; ++f) if (S.col_var[(size_t)f]) { cols.len[f] = S.raw_len.p + (size_t)v * ((size_t)n + 1); cols.off[f] = S.raw_voff.p + (size_t)v * ((size_t)n + 1); ++v; }
  }
  hipLaunchKernelGGL(k_cells_measure, dim3(blocks_for(n, kStageBlock)), dim3(kStageBlock), 0, st, (const uint8_t*)S.raw_cells.p, (const uint64_t*)S.raw_off.p, n, sch,
                     (const int32_t*)S.raw_row_map.p, (int64_t)row_map.size(), S.raw_keep.p, S.raw_mark.p, S.raw_qrow.p, S.raw_begin.p, S.raw_end.p, cols, S.err.p);
  HIP_CHECK(hipMemsetAsync(S.raw_keep.p + n, 0, sizeof(uint32_t), st));
  S.excl_scan(S.raw_keep.p, S.raw_dest.p, (size_t)n + 1);
  HIP_CHECK(hipMemsetAsync(S.raw_mark.p + n, 0, sizeof(uint32_t), st));
  S.excl_scan(S.raw_mark.p, S.raw_mdest.p, (size_t)n + 1);
  for (int f = 0; f < nf; ++f) if (S.col_var[(size_t)f]) {
    HIP_CHECK(hipMemsetAsync(cols.len[f] + n, 0, sizeof(uint32_t), st));
    S.excl_scan((const uint32_t*)cols.len[f], const_cast<uint32_t*>(cols.off[f]), (size_t)n + 1);
  }
  // what is taken: everything, or (more bytes to come) everything in front of the last begin column seen
  CellWalk wk;
  std::vector<uint32_t> totals((size_t)nf, 0);
  {
    S.walk_cut.ensure(9 + GDB_MAX_FIELDS); S.walk_kept_bytes.ensure(1);
    HIP_CHECK(hipMemsetAsync(S.walk_kept_bytes.p, 0, sizeof(unsigned long long), st));
    WalkCutCols vc;
    memset(&vc, 0, sizeof(vc));
    vc.nf = nf;
    for (int f = 0; f < nf; ++f) vc.off[f] = S.col_var[(size_t)f] ? cols.off[f] : nullptr;
    if (device_walk) hipLaunchKernelGGL(k_walk_order, dim3(blocks_for(n)), dim3(kBlock), 0, st, (const int64_t*)S.raw_begin.p, n, S.err.p);
    hipLaunchKernelGGL(k_walk_cut, dim3(1), dim3(64), 0, st, (const int64_t*)S.raw_begin.p, (const uint64_t*)S.raw_off.p, (const uint32_t*)S.raw_dest.p, (const uint32_t*)S.raw_mdest.p, n,
                       (device_walk && whole_columns_only) ? 1 : 0, vc, S.walk_cut.p);
    hipLaunchKernelGGL(k_walk_kept_bytes, dim3(blocks_for(n)), dim3(kBlock), 0, st, (const uint32_t*)S.raw_keep.p, (const uint64_t*)S.raw_off.p, (const int64_t*)S.walk_cut.p, S.walk_kept_bytes.p);
    int64_t cut[9 + GDB_MAX_FIELDS];
    unsigned long long kept_bytes = 0;
    uint32_t eb = 0;
    S.read_back_many({{cut, S.walk_cut.p, (size_t)(9 + nf) * sizeof(int64_t)}, {&kept_bytes, S.walk_kept_bytes.p, sizeof(kept_bytes)}, {&eb, S.err.p, sizeof(uint32_t)}});
    if (eb) throw std::runtime_error(device_walk ? "malformed cell stream (cell sizes do not match the schema, or the cells are not in column-major (col,row) order)"
                                                 : "cell size mismatch while parsing the cell stream");
    n = cut[0];
    wk.bytes_taken = (uint64_t)cut[1]; wk.next_begin = cut[2]; wk.single_column = cut[3] != 0;
    wk.nkept = cut[4]; wk.nmark = cut[5];
    if (n > 0) { wk.first_begin = cut[6]; wk.last_begin = cut[7]; }
    wk.reference_cell_bytes = kept_bytes;
    for (int f = 0; f < nf; ++f) totals[(size_t)f] = (uint32_t)cut[9 + f];
    if (!device_walk) { wk.next_begin = walk_info->next_begin; wk.single_column = walk_info->single_column; }
  }
  if (walk_out) *walk_out = wk;
  const auto tw2 = std::chrono::steady_clock::now();
  const int64_t nkept = wk.nkept, nmark = wk.nmark;
  info.reference_cell_bytes = wk.reference_cell_bytes;
  info.ncells = nkept;
  nbytes = wk.bytes_taken;
  if (nkept == 0 && nmark == 0) return info;
  if (nkept >= (1ll << 32)) throw GenomicsDBDeviceException("more than 2^32 cells in one part: stage in smaller parts");
  uint32_t eb = 0;
  // ---- the part's columns ----------------------------------------------------------------------------------------------
  Impl::Part part;
  memset(&part.v, 0, sizeof(part.v));
  part.v.ncells = nkept;
  if (nkept == 0) {   // markers only (no cell of a queried row in this part)
    void* d = S.blocks.get((size_t)nmark * 8);
    part.bufs.push_back(d);
    hipLaunchKernelGGL(k_cells_markers, dim3(blocks_for(n)), dim3(kBlock), 0, st, (const uint32_t*)S.raw_mark.p, (const uint32_t*)S.raw_mdest.p, (const int64_t*)S.raw_begin.p, n, (int64_t*)d);
    HIP_CHECK(hipStreamSynchronize(st));
    part.v.nmarkers = nmark; part.v.marker_begin = (const int64_t*)d;
    part.data_bytes.assign((size_t)nf, 0);
    S.parts.push_back(part);
    return info;
  }
  auto alloc = [&](size_t bytes) -> void* { void* d = S.blocks.get(bytes); part.bufs.push_back(d); return d; };
  int32_t* row = (int32_t*)alloc((size_t)nkept * 4); int64_t* begin = (int64_t*)alloc((size_t)nkept * 8); int64_t* end = (int64_t*)alloc((size_t)nkept * 8);
  part.v.row = row; part.v.begin = begin; part.v.end = end;
  if (nmark > 0) {
    int64_t* mk = (int64_t*)alloc((size_t)nmark * 8);
    hipLaunchKernelGGL(k_cells_markers, dim3(blocks_for(n)), dim3(kBlock), 0, st, (const uint32_t*)S.raw_mark.p, (const uint32_t*)S.raw_mdest.p, (const int64_t*)S.raw_begin.p, n, mk);
    part.v.nmarkers = nmark; part.v.marker_begin = mk;
  }
  for (int f = 0; f < nf; ++f) {
    const size_t es = (size_t)S.col_elem_size[(size_t)f];
    size_t bytes;
    if (S.col_var[(size_t)f]) {
      bytes = (size_t)totals[(size_t)f] * es;
      uint32_t* fo = (uint32_t*)alloc(((size_t)nkept + 1) * 4);
      cols.frag_off[f] = fo;
      HIP_CHECK(hipMemcpyAsync(fo + nkept, &totals[(size_t)f], sizeof(uint32_t), hipMemcpyHostToDevice, st));
      part.v.col[f].off = fo;
    } else bytes = (size_t)nkept * (size_t)S.col_fixed_num[(size_t)f] * es;
    cols.data[f] = (char*)alloc(bytes);
    part.v.col[f].data = cols.data[f];
    part.data_bytes.push_back(bytes);
  }
  hipLaunchKernelGGL(k_cells_scatter, dim3(blocks_for(n, kStageBlock)), dim3(kStageBlock), 0, st, (const uint8_t*)S.raw_cells.p, (const uint64_t*)S.raw_off.p, n, sch, (const uint32_t*)S.raw_keep.p,
                     (const uint32_t*)S.raw_dest.p, (const int32_t*)S.raw_qrow.p, (const int64_t*)S.raw_begin.p, (const int64_t*)S.raw_end.p, row, begin, end, cols);
  hipLaunchKernelGGL(k_cells_order_check, dim3(blocks_for(nkept)), dim3(kBlock), 0, st, (const int32_t*)row, (const int64_t*)begin, nkept, S.err.p);
  // first begin (the stream is sorted) and the largest END: what the FASTA window of the engine needs
  S.span_max.ensure(2);
  {
    size_t bytes = 0;
    HIP_CHECK(rocprim::reduce(nullptr, bytes, end, S.span_max.p, (int64_t)INT64_MIN, (size_t)nkept, rocprim::maximum<int64_t>(), st));
    void* t = S.temp_storage(bytes);
    HIP_CHECK(rocprim::reduce(t, bytes, end, S.span_max.p, (int64_t)INT64_MIN, (size_t)nkept, rocprim::maximum<int64_t>(), st));
  }
  S.read_back_many({{&info.min_begin, begin, sizeof(int64_t)}, {&info.max_end, S.span_max.p, sizeof(int64_t)}, {&eb, S.err.p, sizeof(uint32_t)}});
  if (eb) { for (void* b : part.bufs) S.blocks.put(b); throw std::runtime_error("cells are not in column-major (col,row) order"); }
  S.parts.push_back(part);
  if (trace) {
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    fprintf(stderr, "[gdbamd stage] %.1f MB, %lld cells: size walk %.4f s, copy + measure + scans %.4f s, allocate + scatter %.4f s\n", nbytes / 1e6, (long long)n, secs(tw0, tw1),
            secs(tw1, tw2), secs(tw2, std::chrono::steady_clock::now()));
  }
  return info;
}

void DevicePipeline::finish_staging() {
  Impl& S = *m_;
  HIP_CHECK(hipSetDevice(S.device));
  S.free_owned();
  FragmentView v;
  memset(&v, 0, sizeof(v));
  int64_t C = 0;
  for (auto& p : S.parts) C += p.v.ncells;
  v.ncells = C;
  auto alloc = [&](size_t bytes) -> void* { void* d = S.blocks.get(bytes); S.owned.push_back(d); return d; };
  int32_t* row = (int32_t*)alloc((size_t)C * 4); int64_t* begin = (int64_t*)alloc((size_t)C * 8); int64_t* end = (int64_t*)alloc((size_t)C * 8);
  int64_t at = 0;
  for (auto& p : S.parts) {
    // (copies on the pipeline's own stream: a synchronous hipMemcpy runs on the null stream, which waits for - and holds up -
    // every other pipeline's stream)
    HIP_CHECK(hipMemcpyAsync(row + at, p.v.row, (size_t)p.v.ncells * 4, hipMemcpyDeviceToDevice, S.stream));
    HIP_CHECK(hipMemcpyAsync(begin + at, p.v.begin, (size_t)p.v.ncells * 8, hipMemcpyDeviceToDevice, S.stream));
    HIP_CHECK(hipMemcpyAsync(end + at, p.v.end, (size_t)p.v.ncells * 8, hipMemcpyDeviceToDevice, S.stream));
    at += p.v.ncells;
  }
  v.row = row; v.begin = begin; v.end = end;
  const size_t nf = S.col_elem_size.size();
  for (size_t f = 0; f < nf; ++f) {
    size_t total = 0;
    for (auto& p : S.parts) total += p.data_bytes[f];
    char* data = (char*)alloc(total);
    uint32_t* off = S.col_var[f] ? (uint32_t*)alloc((size_t)(C + 1) * 4) : nullptr;
    size_t byte_at = 0;
    int64_t cell_at = 0;
    for (auto& p : S.parts) {
      if (p.data_bytes[f]) HIP_CHECK(hipMemcpyAsync(data + byte_at, p.v.col[f].data, p.data_bytes[f], hipMemcpyDeviceToDevice, S.stream));
      if (off && p.v.ncells > 0) {
        const uint64_t base_elems = byte_at / (size_t)S.col_elem_size[f];
        if (base_elems + p.data_bytes[f] / (size_t)S.col_elem_size[f] >= (1ull << 32)) throw GenomicsDBDeviceException("variable-length column exceeds 2^32 elements: stage a narrower column interval");
        hipLaunchKernelGGL(k_copy_offsets, dim3(blocks_for(p.v.ncells + 1)), dim3(kBlock), 0, S.stream, p.v.col[f].off, p.v.ncells + 1, (uint32_t)base_elems, off + cell_at);
      }
      byte_at += p.data_bytes[f];
      cell_at += p.v.ncells;
    }
    v.col[f].data = data;
    v.col[f].off = off;
  }
  {
    int64_t M = 0;
    for (auto& p : S.parts) M += p.v.nmarkers;
    int64_t* mk = (int64_t*)alloc((size_t)M * 8);
    int64_t at_m = 0;
    for (auto& p : S.parts) { if (p.v.nmarkers) HIP_CHECK(hipMemcpyAsync(mk + at_m, p.v.marker_begin, (size_t)p.v.nmarkers * 8, hipMemcpyDeviceToDevice, S.stream)); at_m += p.v.nmarkers; }
    v.nmarkers = M; v.marker_begin = mk;
  }
  HIP_CHECK(hipStreamSynchronize(S.stream));
  S.use_compute_stream();
  S.free_parts();
  S.fr = v;
  S.owns_fragment = true;
  S.classified = false; ++S.fragment_generation;
}

// ---- columnar fragment file: the staged fragment as it lies in HBM, so that opening an array is file -> HBM copies ------
// (the build's own fragment format; the Intel TileDB fork's on-disk format is not available.)
// Layout v2, little endian: "GDBAMDF2", u32 version, u32 nfields, i64 ncells, i64 nmarkers, i32 num_rows, i32 pad,
// u64 reference_cell_bytes, i64 min_begin, i64 max_end, u64 schema_hash, u64 source_bytes, i64 source_mtime; per field: u8 var,
// u8 elem_size, u16 name_len, i32 fixed_num, u64 data_bytes, name; then, each 64-byte aligned: row[], begin[], end[],
// marker_begin[], per field off[] (var only), data[].  Cells are in (column, row) order, so a column window of the array is a
// contiguous range of every section: the file is read window by window (open_fragment_file / append_fragment_cells) when the
// array does not fit the staging budget.
namespace {
// ---- DEFLATE tiles of the fragment file, inflated on the device -------------------------------------------------------------------
// The data sections of a compressed fragment file (version 3) are cut into tiles of kFragTile uncompressed bytes, each a raw
// DEFLATE stream (RFC 1951) of its own - the analogue of the gzip'd attribute tiles of the reference's TileDB arrays
// (genomicsdb_iterators.cc:334-423 reads them tile by tile).  Compressed bytes cross PCIe, one thread inflates one tile:
// a window has tens of thousands of tiles, so the serial bit-by-bit nature of a DEFLATE stream is spread over as many lanes.
// The writer (save_fragment, zlib with strategy Z_FIXED) emits stored and FIXED-Huffman blocks only: the decoder then needs no
// per-stream code tables - a literal / length code is 7 to 9 bits and is classified arithmetically (RFC 1951, 3.2.6).  Blocks with
// dynamic codes (a file compressed by something else) are decoded too, with tables in memory; a malformed or overlong stream raises
// an error bit.
constexpr uint32_t kFragTile = 8192;
__device__ const uint16_t kInfLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__device__ const uint8_t kInfLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__device__ const uint16_t kInfDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__device__ const uint8_t kInfDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
struct InflateBits {
  const uint8_t* p; const uint8_t* end;
  uint64_t buf; int n;
  // After refill() at least 33 bits are buffered while input remains (no consumer takes more than 19 between two refills).  Eight
  // bytes at a time while they last: the stream is read front to back by one lane, one unaligned 8-byte load per 3 to 7 bytes used.
  __device__ __forceinline__ void refill() {
    if (n > 32) return;
    if (p + 8 <= end) {
      uint64_t w;
      __builtin_memcpy(&w, p, 8);
      buf |= w << n;
      const int take = (63 - n) >> 3;
      p += take; n += take * 8;
      buf &= (1ull << n) - 1ull;                 // (n <= 63: the bits of the byte that was not taken are read again next time)
    } else while (n <= 56 && p < end) { buf |= (uint64_t)(*p++) << n; n += 8; }
  }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1ull)); }
  __device__ __forceinline__ void drop(int k) { buf >>= k; n -= k; }
  __device__ __forceinline__ uint32_t get(int k) { const uint32_t v = peek(k); drop(k); return v; }   // (k <= 16; refill() first)
};
// Blocks with DYNAMIC codes (what a default zlib / gzip writer produces): canonical Huffman decoding bit by bit over a count-per-length
// and a symbols-in-code-order table (RFC 1951, 3.2.2 and 3.2.7), the tables of a lane in its own slice of a global scratch buffer.
// Slower than the fixed codes (up to 15 steps per symbol, tables in memory) - the files this build writes do not need it.
struct InflateCode { uint16_t* count; uint16_t* symbol; };            // count[16], symbol[n]
constexpr int kInflateScratch = 320 + 2 * (16 + 288) + 2 * (16 + 32);  // bytes per lane: code lengths, literal / length code, distance code
__device__ __forceinline__ int inflate_decode(InflateBits& b, const InflateCode& h) {
  int code = 0, first = 0, index = 0;
  b.refill();
  for (int len = 1; len <= 15; ++len) {
    if (b.n < 1) return -1;
    code |= (int)b.get(1);
    const int count = h.count[len];
    if (code - count < first) return h.symbol[index + (code - first)];
    index += count; first += count;
    first <<= 1; code <<= 1;
    if (len == 8) b.refill();
  }
  return -1;
}
// code lengths -> tables; returns the number of unused codes (0: complete; < 0: over-subscribed)
__device__ __forceinline__ int inflate_construct(const InflateCode& h, const uint8_t* length, int n) {
  for (int len = 0; len <= 15; ++len) h.count[len] = 0;
  for (int sym = 0; sym < n; ++sym) h.count[length[sym]]++;
  if (h.count[0] == n) return 0;
  int left = 1;
  for (int len = 1; len <= 15; ++len) { left <<= 1; left -= h.count[len]; if (left < 0) return left; }
  uint16_t offs[16];
  offs[1] = 0;
  for (int len = 1; len < 15; ++len) offs[len + 1] = (uint16_t)(offs[len] + h.count[len]);
  for (int sym = 0; sym < n; ++sym) if (length[sym] != 0) h.symbol[offs[length[sym]]++] = (uint16_t)sym;
  return left;
}
// The copy of a match: 8 bytes per load / store while the distance allows it (a copy with dist >= 8 never reads a byte its own
// current 8-byte store writes; bytes written by EARLIER stores of the same lane are seen - the vector memory pipeline keeps one
// lane's accesses to an address in order, which the byte-by-byte loop of round 2 relied on as well).  The byte loop made every
// copied byte a global load behind a global store - one memory round trip per byte, the largest item of the kernel's time.
__device__ __forceinline__ void inflate_copy_match(uint8_t* o, uint32_t at, uint32_t len, uint32_t dist) {
  uint32_t i = 0;
  if (dist >= 8u)
    for (; i + 8u <= len; i += 8u) { uint64_t w; __builtin_memcpy(&w, o + at - dist + i, 8); __builtin_memcpy(o + at + i, &w, 8); }
  for (; i < len; ++i) o[at + i] = o[at + i - dist];
}
// job t: compressed bytes [in_off[t], in_off[t + 1]) of `comp` -> want[t] bytes at out + t * kFragTile
__global__ void k_inflate_tiles(const uint8_t* __restrict__ comp, const uint64_t* __restrict__ in_off, const uint32_t* __restrict__ want_bytes, int64_t ntiles, uint8_t* __restrict__ out,
                                uint8_t* __restrict__ scratch, uint32_t* err) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  uint8_t* const lengths = scratch + (uint64_t)t * kInflateScratch;                                       // [320]
  const InflateCode lencode{reinterpret_cast<uint16_t*>(lengths + 320), reinterpret_cast<uint16_t*>(lengths + 320) + 16};
  const InflateCode distcode{lencode.symbol + 288, lencode.symbol + 288 + 16};
  uint8_t* const o = out + (uint64_t)t * kFragTile;
  const uint32_t want = want_bytes[t];
  InflateBits b{comp + in_off[t], comp + in_off[t + 1], 0ull, 0};
  uint32_t at = 0;
  bool bad = false, last = false;
  while (!last && !bad) {
    b.refill();
    if (b.n < 3) { bad = true; break; }
    last = b.get(1) != 0;
    const uint32_t type = b.get(2);
    if (type == 0) {                                   // stored: LEN, NLEN, bytes
      b.drop(b.n & 7);
      b.refill();
      if (b.n < 32) { bad = true; break; }
      const uint32_t len = b.get(16), nlen = b.get(16);
      if ((len ^ nlen) != 0xFFFFu || at + len > want) { bad = true; break; }
      for (uint32_t i = 0; i < len; ++i) { b.refill(); if (b.n < 8) { bad = true; break; } o[at++] = (uint8_t)b.get(8); }
    } else if (type == 1) {                            // fixed Huffman codes
      for (;;) {
        b.refill();
        if (b.n < 7) { bad = true; break; }
        uint32_t sym;
        const uint32_t c7 = __brev(b.peek(7)) >> 25;   // Huffman codes are packed most-significant bit first
        if (c7 <= 0x17u) { sym = 256u + c7; b.drop(7); }
        else {
          const uint32_t c8 = __brev(b.peek(8)) >> 24;
          if (c8 >= 0x30u && c8 <= 0xBFu) { sym = c8 - 0x30u; b.drop(8); }
          else if (c8 >= 0xC0u && c8 <= 0xC7u) { sym = 280u + (c8 - 0xC0u); b.drop(8); }
          else {
            const uint32_t c9 = __brev(b.peek(9)) >> 23;
            if (c9 < 0x190u) { bad = true; break; }
            sym = 144u + (c9 - 0x190u); b.drop(9);
          }
        }
        if (b.n < 0) { bad = true; break; }
        if (sym < 256u) { if (at >= want) { bad = true; break; } o[at++] = (uint8_t)sym; continue; }
        if (sym == 256u) break;
        if (sym > 285u) { bad = true; break; }
        b.refill();
        const uint32_t li = sym - 257u;
        const uint32_t len = kInfLenBase[li] + b.get(kInfLenExtra[li]);
        const uint32_t dc = __brev(b.get(5)) >> 27;
        if (dc >= 30u) { bad = true; break; }
        b.refill();
        const uint32_t dist = kInfDistBase[dc] + b.get(kInfDistExtra[dc]);
        if (b.n < 0 || dist > at || at + len > want) { bad = true; break; }
        inflate_copy_match(o, at, len, dist);
        at += len;
      }
    } else if (type == 2) {                            // dynamic codes: read the code lengths, build both tables, decode
      b.refill();
      if (b.n < 14) { bad = true; break; }
      const int nlen = (int)b.get(5) + 257, ndist = (int)b.get(5) + 1, ncode = (int)b.get(4) + 4;
      if (nlen > 286 || ndist > 30) { bad = true; break; }
      const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      for (int i = 0; i < 19; ++i) lengths[i] = 0;
      for (int i = 0; i < ncode; ++i) { b.refill(); if (b.n < 3) { bad = true; break; } lengths[order[i]] = (uint8_t)b.get(3); }
      if (bad) break;
      if (inflate_construct(lencode, lengths, 19) != 0) { bad = true; break; }       // the code-length code must be complete
      int index = 0;
      while (index < nlen + ndist && !bad) {
        int sym = inflate_decode(b, lencode);
        if (sym < 0) { bad = true; break; }
        if (sym < 16) lengths[index++] = (uint8_t)sym;
        else {
          int rep_len = 0;
          b.refill();
          if (sym == 16) { if (index == 0) { bad = true; break; } rep_len = lengths[index - 1]; sym = 3 + (int)b.get(2); }
          else if (sym == 17) sym = 3 + (int)b.get(3);
          else sym = 11 + (int)b.get(7);
          if (b.n < 0 || index + sym > nlen + ndist) { bad = true; break; }
          while (sym--) lengths[index++] = (uint8_t)rep_len;
        }
      }
      if (bad) break;
      if (lengths[256] == 0) { bad = true; break; }                                 // no end-of-block code
      // (the distance lengths are moved aside first: building the literal / length tables must not clobber them)
      uint8_t dlen[30];
      for (int i = 0; i < ndist; ++i) dlen[i] = lengths[nlen + i];
      int rc = inflate_construct(lencode, lengths, nlen);
      if (rc < 0 || (rc > 0 && nlen - lencode.count[0] != 1)) { bad = true; break; }
      rc = inflate_construct(distcode, dlen, ndist);
      if (rc < 0 || (rc > 0 && ndist - distcode.count[0] != 1)) { bad = true; break; }
      for (;;) {
        const int sym = inflate_decode(b, lencode);
        if (sym < 0) { bad = true; break; }
        if (sym < 256) { if (at >= want) { bad = true; break; } o[at++] = (uint8_t)sym; continue; }
        if (sym == 256) break;
        if (sym > 285) { bad = true; break; }
        b.refill();
        const uint32_t li = (uint32_t)sym - 257u;
        const uint32_t len = kInfLenBase[li] + b.get(kInfLenExtra[li]);
        const int dc = inflate_decode(b, distcode);
        if (dc < 0 || dc >= 30) { bad = true; break; }
        b.refill();
        const uint32_t dist = kInfDistBase[dc] + b.get(kInfDistExtra[dc]);
        if (b.n < 0 || dist > at || at + len > want) { bad = true; break; }
        inflate_copy_match(o, at, len, dist);
        at += len;
      }
    } else bad = true;                                 // the reserved block type
  }
  if (bad || at != want) atomicOr(err, (uint32_t)GDB_ERR_CELL_STREAM);
}

const char kFragMagic[8] = {'G', 'D', 'B', 'A', 'M', 'D', 'F', '2'};
struct FragFieldHdr { uint8_t var, elem_size; uint16_t name_len; int32_t fixed_num; uint64_t data_bytes; };
void put_bytes(std::vector<uint8_t>& o, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; o.insert(o.end(), b, b + n); }
void pad64(FILE* f, uint64_t& at) { static const char z[64] = {0}; const size_t r = (size_t)((64 - (at & 63)) & 63); if (r) { fwrite(z, 1, r, f); at += r; } }
uint64_t align64(uint64_t at) { return (at + 63) & ~(uint64_t)63; }
}  // namespace

struct DevicePipeline::FragmentFile {
  int fd = -1;
  std::string path;
  uint64_t file_size = 0;
  FragmentFileMeta meta;
  int64_t C = 0, M = 0;
  uint64_t marker_at = 0;
  // a section of the file: `bytes` of a column.  Version 3 stores every section but the markers as DEFLATE tiles of kFragTile bytes:
  // [payload][u64 tile offset x (ntiles + 1)][u64 ntiles], `stored` bytes in all
  struct Section { uint64_t at = 0, bytes = 0, stored = 0, ntiles = 0, index_at = 0; bool compressed = false; };
  Section sec_row, sec_begin, sec_end;
  struct Field { bool var = false; int elem_size = 4, fixed_num = 1; uint64_t data_bytes = 0; int plan_field = -1; Section off, data; };
  std::vector<Field> fields;
  bool compressed = false;
  uint64_t raw_bytes = 0;       // of all column sections as they are in device memory (== their file bytes when not compressed)
  // the last tile a host-side lookup inflated (the binary searches over `begin` and the offset look-ups stay on the host)
  const Section* cached_sec = nullptr; uint64_t cached_tile = ~0ull; std::vector<uint8_t> cached_bytes;
  std::vector<int64_t> markers;     // whole array on the host (boundary markers are rare: cells of rows outside the query)
  size_t marker_cursor = 0;
  // two pinned bounce buffers for file -> HBM copies
  char* bounce[2] = {nullptr, nullptr};
  hipEvent_t bounce_free[2] = {nullptr, nullptr};
  static constexpr size_t kBounce = (size_t)64 << 20;
  ~FragmentFile() {
    if (fd >= 0) ::close(fd);
    for (auto& b : bounce) if (b) (void)hipHostFree(b);
    for (auto& e : bounce_free) if (e) (void)hipEventDestroy(e);
  }
  void read_at(void* dst, uint64_t at, size_t n) const {
    char* d = (char*)dst;
    while (n) {
      const ssize_t k = ::pread(fd, d, n, (off_t)at);
      if (k <= 0) throw std::runtime_error(path + ": truncated fragment file");
      d += k; at += (uint64_t)k; n -= (size_t)k;
    }
  }
  // bytes [off, off + n) of a section into host memory; a compressed section: zlib inflates the tile(s) that hold them
  void host_read(const Section& sec, uint64_t off, void* dst, size_t n) {
    if (off + n > sec.bytes) throw std::runtime_error(path + ": read beyond a section of the fragment file");
    if (!sec.compressed) { read_at(dst, sec.at + off, n); return; }
    char* d = (char*)dst;
    while (n) {
      const uint64_t t = off / kFragTile, in_tile = off % kFragTile;
      if (cached_sec != &sec || cached_tile != t) {
        uint64_t o[2];
        read_at(o, sec.index_at + 8 * t, 16);
        if (o[1] < o[0] || sec.at + o[1] > sec.index_at) throw std::runtime_error(path + ": corrupt tile index");
        std::vector<uint8_t> z((size_t)(o[1] - o[0]));
        read_at(z.data(), sec.at + o[0], z.size());
        const size_t want = (size_t)std::min<uint64_t>(kFragTile, sec.bytes - t * kFragTile);
        cached_bytes.resize(want);
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("inflateInit2 failed");
        zs.next_in = z.data(); zs.avail_in = (uInt)z.size(); zs.next_out = cached_bytes.data(); zs.avail_out = (uInt)want;
        const int rc = inflate(&zs, Z_FINISH);
        const size_t got = want - zs.avail_out;
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || got != want) { cached_sec = nullptr; throw std::runtime_error(path + ": a compressed tile does not inflate to its size"); }
        cached_sec = &sec; cached_tile = t;
      }
      const size_t k = (size_t)std::min<uint64_t>(n, cached_bytes.size() - in_tile);
      memcpy(d, cached_bytes.data() + in_tile, k);
      d += k; off += k; n -= k;
    }
  }
  int64_t begin_of(int64_t c) { int64_t v; host_read(sec_begin, (uint64_t)c * 8, &v, 8); return v; }
  // file bytes [at, at + n) -> device memory, through the pinned buffers (the read of chunk i + 1 overlaps the copy of chunk i)
  void to_device(void* dev, uint64_t at, uint64_t n, hipStream_t st) {
    int which = 0;
    for (uint64_t done = 0; done < n;) {
      const size_t k = (size_t)std::min<uint64_t>(kBounce, n - done);
      if (!bounce[which]) { HIP_CHECK(hipHostMalloc((void**)&bounce[which], kBounce, hipHostMallocDefault)); HIP_CHECK(hipEventCreateWithFlags(&bounce_free[which], hipEventDisableTiming)); }
      else HIP_CHECK(hipEventSynchronize(bounce_free[which]));
      read_at(bounce[which], at + done, k);
      HIP_CHECK(hipMemcpyAsync((char*)dev + done, bounce[which], k, hipMemcpyHostToDevice, st));
      HIP_CHECK(hipEventRecord(bounce_free[which], st));
      done += k; which ^= 1;
    }
  }
};

// one tile as a raw DEFLATE stream of stored / fixed-Huffman blocks (zlib, strategy Z_FIXED): what k_inflate_tiles reads
static void deflate_tile_fixed(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
  z_stream zs;
  memset(&zs, 0, sizeof(zs));
  if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_FIXED) != Z_OK) throw std::runtime_error("deflateInit2 failed");
  out.resize(deflateBound(&zs, (uLong)n) + 16);
  zs.next_in = const_cast<Bytef*>(src); zs.avail_in = (uInt)n;
  zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
  const int rc = deflate(&zs, Z_FINISH);
  const size_t produced = out.size() - zs.avail_out;
  deflateEnd(&zs);
  if (rc != Z_STREAM_END) throw std::runtime_error("deflate failed");
  out.resize(produced);
}

void DevicePipeline::save_fragment(const std::string& path, const FragmentFileMeta& meta, bool compress) {
  Impl& S = *m_;
  HIP_CHECK(hipSetDevice(S.device));
  const FragmentView& fr = S.fr;
  const int nf = S.hp.plan.nfields;
  if ((int)S.col_elem_size.size() < nf) throw GenomicsDBDeviceException("save_fragment: no fragment staged from a cell stream");
  const int64_t C = fr.ncells;
  std::vector<uint64_t> data_bytes((size_t)nf, 0);
  for (int f = 0; f < nf; ++f) {
    if (S.col_var[(size_t)f]) {
      uint32_t last = 0;
      if (C > 0) HIP_CHECK(hipMemcpy(&last, fr.col[f].off + C, sizeof(uint32_t), hipMemcpyDeviceToHost));
      data_bytes[(size_t)f] = (uint64_t)last * (uint64_t)S.col_elem_size[(size_t)f];
    } else data_bytes[(size_t)f] = (uint64_t)C * (uint64_t)S.col_fixed_num[(size_t)f] * (uint64_t)S.col_elem_size[(size_t)f];
  }
  std::vector<uint8_t> hdr;
  put_bytes(hdr, kFragMagic, 8);
  const uint32_t version = compress ? 3u : 2u, nfields = (uint32_t)nf;
  const int32_t num_rows = S.hp.plan.num_query_rows, pad = 0;
  put_bytes(hdr, &version, 4); put_bytes(hdr, &nfields, 4); put_bytes(hdr, &C, 8); put_bytes(hdr, &fr.nmarkers, 8); put_bytes(hdr, &num_rows, 4); put_bytes(hdr, &pad, 4);
  put_bytes(hdr, &meta.reference_cell_bytes, 8); put_bytes(hdr, &meta.min_begin, 8); put_bytes(hdr, &meta.max_end, 8);
  put_bytes(hdr, &meta.schema_hash, 8); put_bytes(hdr, &meta.source_bytes, 8); put_bytes(hdr, &meta.source_mtime, 8);
  for (int f = 0; f < nf; ++f) {
    const std::string& name = S.hp.field_names[(size_t)f];
    FragFieldHdr h{(uint8_t)(S.col_var[(size_t)f] ? 1 : 0), (uint8_t)S.col_elem_size[(size_t)f], (uint16_t)name.size(), (int32_t)S.col_fixed_num[(size_t)f], data_bytes[(size_t)f]};
    put_bytes(hdr, &h, sizeof(h)); put_bytes(hdr, name.data(), name.size());
  }
  const size_t stored_at = hdr.size();              // version 3: bytes every section occupies in the file (patched in at the end):
  std::vector<uint64_t> stored(3 + 2 * (size_t)nf, 0);   // row, begin, end, then (offsets, data) per field
  if (compress) put_bytes(hdr, stored.data(), stored.size() * 8);
  const std::string tmp_path = path + ".tmp";
  FILE* fp = fopen(tmp_path.c_str(), "wb");
  if (!fp) throw std::runtime_error("cannot create " + tmp_path);
  uint64_t at = 0;
  fwrite(hdr.data(), 1, hdr.size(), fp); at += hdr.size();
  std::vector<uint8_t> host;
  auto dump = [&](const void* dev, uint64_t bytes) {
    pad64(fp, at);
    for (uint64_t done = 0; done < bytes;) {     // in pieces: a column of a large array does not have to fit host memory twice
      const size_t k = (size_t)std::min<uint64_t>((uint64_t)256 << 20, bytes - done);
      host.resize(k);
      HIP_CHECK(hipMemcpy(host.data(), (const char*)dev + done, k, hipMemcpyDeviceToHost));
      if (fwrite(host.data(), 1, k, fp) != k) { fclose(fp); throw std::runtime_error("short write to " + tmp_path); }
      done += k;
    }
    at += bytes;
  };
  // a section as DEFLATE tiles: payload, then the tile offsets and the tile count
  auto dump_tiles = [&](const void* dev, uint64_t bytes) -> uint64_t {
    pad64(fp, at);
    const uint64_t ntiles = (bytes + kFragTile - 1) / kFragTile;
    std::vector<uint64_t> tile_off(1, 0);
    const uint64_t piece = (uint64_t)4096 * kFragTile;           // 128 MB of tiles at a time
    const unsigned nthreads = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    for (uint64_t done = 0; done < bytes; done += piece) {
      const size_t k = (size_t)std::min<uint64_t>(piece, bytes - done);
      host.resize(k);
      HIP_CHECK(hipMemcpy(host.data(), (const char*)dev + done, k, hipMemcpyDeviceToHost));
      const size_t nt = (k + kFragTile - 1) / kFragTile;
      std::vector<std::vector<uint8_t>> z(nt);
      std::vector<std::thread> pool;
      std::string failure;
      std::mutex fm;
      for (unsigned w = 0; w < nthreads; ++w)
        pool.emplace_back([&, w]() {
          try { for (size_t t = w; t < nt; t += nthreads) deflate_tile_fixed(host.data() + t * kFragTile, std::min<size_t>(kFragTile, k - t * kFragTile), z[t]); }
          catch (const std::exception& e) { std::lock_guard<std::mutex> g(fm); failure = e.what(); }
        });
      for (auto& th : pool) th.join();
      if (!failure.empty()) { fclose(fp); throw std::runtime_error(failure); }
      for (size_t t = 0; t < nt; ++t) {
        if (fwrite(z[t].data(), 1, z[t].size(), fp) != z[t].size()) { fclose(fp); throw std::runtime_error("short write to " + tmp_path); }
        tile_off.push_back(tile_off.back() + z[t].size());
      }
    }
    fwrite(tile_off.data(), 8, tile_off.size(), fp);
    fwrite(&ntiles, 8, 1, fp);
    const uint64_t total = tile_off.back() + 8 * tile_off.size() + 8;
    at += total;
    return total;
  };
  auto put_section = [&](const void* dev, uint64_t bytes, size_t stored_idx) { if (compress) stored[stored_idx] = dump_tiles(dev, bytes); else dump(dev, bytes); };
  put_section(fr.row, (uint64_t)C * 4, 0); put_section(fr.begin, (uint64_t)C * 8, 1); put_section(fr.end, (uint64_t)C * 8, 2);
  dump(fr.marker_begin, (uint64_t)fr.nmarkers * 8);
  for (int f = 0; f < nf; ++f) {
    if (S.col_var[(size_t)f]) put_section(fr.col[f].off, (uint64_t)(C + 1) * 4, 3 + 2 * (size_t)f);
    put_section(fr.col[f].data, data_bytes[(size_t)f], 4 + 2 * (size_t)f);
  }
  if (compress) {
    if (fseek(fp, (long)stored_at, SEEK_SET) != 0 || fwrite(stored.data(), 8, stored.size(), fp) != stored.size()) { fclose(fp); throw std::runtime_error("cannot write " + tmp_path); }
  }
  if (fclose(fp) != 0 || rename(tmp_path.c_str(), path.c_str()) != 0) throw std::runtime_error("cannot write " + path);
}

// Opens and VALIDATES a fragment file: every size the header names is checked against the file and every column against the
// layout the query's schema expects (expected[f] for plan field f), so that a foreign or stale file is refused here instead of
// faulting in a kernel.  Nothing of the pipeline changes when this throws.
FragmentFileMeta DevicePipeline::open_fragment_file(const std::string& path, const std::vector<ColumnLayout>& expected, uint64_t expected_schema_hash) {
  Impl& S = *m_;
  std::unique_ptr<FragmentFile> ff(new FragmentFile);
  ff->path = path;
  ff->fd = ::open(path.c_str(), O_RDONLY);
  if (ff->fd < 0) throw std::runtime_error("cannot open " + path);
  struct stat sb;
  if (fstat(ff->fd, &sb) != 0) throw std::runtime_error("cannot stat " + path);
  ff->file_size = (uint64_t)sb.st_size;
  auto fail = [&](const std::string& why) { throw std::runtime_error(path + ": " + why); };
  uint64_t at = 0;
  auto rd = [&](void* p, size_t n) { if (at + n > ff->file_size) fail("truncated fragment file"); ff->read_at(p, at, n); at += n; };
  char magic[8];
  rd(magic, 8);
  if (memcmp(magic, kFragMagic, 8) != 0) fail("not a genomicsdb_amd fragment file (version 2)");
  uint32_t version, nfields; int64_t C, M; int32_t num_rows, pad;
  FragmentFileMeta meta;
  rd(&version, 4); rd(&nfields, 4); rd(&C, 8); rd(&M, 8); rd(&num_rows, 4); rd(&pad, 4);
  rd(&meta.reference_cell_bytes, 8); rd(&meta.min_begin, 8); rd(&meta.max_end, 8); rd(&meta.schema_hash, 8); rd(&meta.source_bytes, 8); rd(&meta.source_mtime, 8);
  if (version != 2 && version != 3) fail("unsupported fragment file version");
  ff->compressed = version == 3;
  if (nfields == 0 || nfields > 4096) fail("implausible number of fields");
  // a raw file stores 20 bytes of coordinates per cell and 8 per marker; a compressed one (version 3) may store up to DEFLATE's
  // maximum ratio (1032 : 1) less - regular data does get well below 20 stored bytes per cell; truncation is caught section by
  // section through the tile index below
  const uint64_t ratio = ff->compressed ? 1032 : 1;
  if (C < 0 || M < 0 || (uint64_t)C / ratio > ff->file_size / 20 + 1 || (uint64_t)M / ratio > ff->file_size / 8 + 1) fail("implausible cell / marker count");
  if (num_rows != S.hp.plan.num_query_rows) fail("fragment file was written for another set of query rows");
  if (expected_schema_hash && meta.schema_hash != expected_schema_hash) fail("fragment file was written under another vid / callset mapping (stale)");
  struct FileField { FragFieldHdr h; std::string name; };
  std::vector<FileField> file_fields(nfields);
  for (auto& x : file_fields) { rd(&x.h, sizeof(x.h)); x.name.resize(x.h.name_len); if (x.h.name_len) rd(&x.name[0], x.h.name_len); }
  std::vector<uint64_t> stored(3 + 2 * (size_t)nfields, 0);
  if (ff->compressed) rd(stored.data(), stored.size() * 8);
  const int nf = S.hp.plan.nfields;
  if ((int)expected.size() != nf) fail("internal: expected column layouts do not match the plan");
  std::vector<int> file_to_plan(nfields, -1);
  for (int f = 0; f < nf; ++f) {
    int found = -1;
    for (uint32_t i = 0; i < nfields; ++i) if (file_fields[i].name == S.hp.field_names[(size_t)f]) found = (int)i;
    if (found < 0) fail("attribute " + S.hp.field_names[(size_t)f] + " of the query is not in the fragment file");
    const FragFieldHdr& h = file_fields[(size_t)found].h;
    const ColumnLayout& want = expected[(size_t)f];
    if ((h.var != 0) != want.var || (int)h.elem_size != want.elem_size || (!want.var && h.fixed_num != want.fixed_num))
      fail("attribute " + S.hp.field_names[(size_t)f] + " has another layout in the fragment file than in the array schema");
    file_to_plan[(size_t)found] = f;
  }
  // section offsets, each checked against the file size
  auto section = [&](uint64_t bytes) -> uint64_t { at = align64(at); const uint64_t here = at; if (bytes > ff->file_size || here > ff->file_size - bytes) fail("truncated fragment file"); at += bytes; return here; };
  ff->C = C; ff->M = M;
  // a column section: raw, or DEFLATE tiles whose index behind the payload has to agree with the section size and the tile count
  auto make_section = [&](uint64_t bytes, uint64_t stored_bytes) -> FragmentFile::Section {
    FragmentFile::Section sec;
    sec.bytes = bytes;
    ff->raw_bytes += bytes;
    if (!ff->compressed) { sec.at = section(bytes); return sec; }
    sec.compressed = true; sec.stored = stored_bytes;
    sec.ntiles = (bytes + kFragTile - 1) / kFragTile;
    if (sec.stored < 16 + 8 * sec.ntiles) fail("compressed section shorter than its tile index");
    sec.at = section(sec.stored);
    uint64_t nt_file = 0;
    ff->read_at(&nt_file, sec.at + sec.stored - 8, 8);
    if (nt_file != sec.ntiles) fail("tile count of a compressed section does not match its size");
    sec.index_at = sec.at + sec.stored - 8 - 8 * (sec.ntiles + 1);
    uint64_t first_off = 1, last_off = 0;
    ff->read_at(&first_off, sec.index_at, 8); ff->read_at(&last_off, sec.index_at + 8 * sec.ntiles, 8);
    if (first_off != 0 || sec.at + last_off != sec.index_at) fail("tile index of a compressed section does not match its payload");
    return sec;
  };
  ff->sec_row = make_section((uint64_t)C * 4, stored[0]); ff->sec_begin = make_section((uint64_t)C * 8, stored[1]); ff->sec_end = make_section((uint64_t)C * 8, stored[2]);
  ff->marker_at = section((uint64_t)M * 8);
  ff->fields.resize(nfields);
  for (uint32_t i = 0; i < nfields; ++i) {
    FragmentFile::Field& fd = ff->fields[i];
    const FragFieldHdr& h = file_fields[i].h;
    fd.var = h.var != 0; fd.elem_size = h.elem_size; fd.fixed_num = h.fixed_num; fd.data_bytes = h.data_bytes; fd.plan_field = file_to_plan[i];
    if (fd.elem_size != 1 && fd.elem_size != 4 && fd.elem_size != 8) fail("unsupported element size");
    if (fd.var) fd.off = make_section((uint64_t)(C + 1) * 4, stored[3 + 2 * (size_t)i]);
    else if (fd.data_bytes != (uint64_t)C * (uint64_t)fd.fixed_num * (uint64_t)fd.elem_size) fail("fixed-length column size does not match the cell count");
    fd.data = make_section(fd.data_bytes, stored[4 + 2 * (size_t)i]);
    if (fd.var && C > 0) {   // the last offset has to name exactly the bytes of the data section
      uint32_t first = 0, last = 0;
      ff->host_read(fd.off, 0, &first, 4); ff->host_read(fd.off, (uint64_t)C * 4, &last, 4);
      if (first != 0 || (uint64_t)last * (uint64_t)fd.elem_size != fd.data_bytes) fail("offsets of a variable-length column do not match its data section");
    }
  }
  if (M > 0) { ff->markers.resize((size_t)M); ff->read_at(ff->markers.data(), ff->marker_at, (size_t)M * 8); }
  meta.ncells = C;
  ff->meta = meta;
  S.ff = std::move(ff);
  return meta;
}

void DevicePipeline::close_fragment_file() { m_->ff.reset(); }

// first cell of the open file whose begin column is >= column
int64_t DevicePipeline::fragment_file_lower_bound(int64_t column) {
  FragmentFile& F = *m_->ff;
  int64_t lo = 0, hi = F.C;
  while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (F.begin_of(mid) < column) lo = mid + 1; else hi = mid; }
  return lo;
}

// Appends the cells [c0, c1) of the open file to the staging area as one part, c1 chosen so that the part holds whole begin
// columns and about budget_bytes of column data (at least one column).  Requires begin_staging() before and finish_staging() after.
DevicePipeline::FragmentWindow DevicePipeline::append_fragment_cells(int64_t c0, uint64_t budget_bytes) {
  Impl& S = *m_;
  if (!S.ff) throw GenomicsDBDeviceException("append_fragment_cells: no fragment file is open");
  FragmentFile& F = *S.ff;
  HIP_CHECK(hipSetDevice(S.device));
  hipStream_t st = S.stream;
  FragmentWindow w;
  w.c0 = c0; w.c1 = c0;
  if (c0 >= F.C) return w;
  // ---- the cut: whole columns, about budget_bytes ----------------------------------------------------------------------
  const double bytes_per_cell = std::max(1.0, (double)F.raw_bytes / (double)std::max<int64_t>(1, F.C));   // (as inflated: the budget is device memory)
  int64_t want = std::max<int64_t>(1, (int64_t)((double)budget_bytes / bytes_per_cell));
  int64_t c1;
  for (;;) {
    c1 = std::min(F.C, c0 + want);
    if (c1 >= F.C) { c1 = F.C; break; }
    // move the cut back to the first cell of the column that holds cell c1 (that column goes to the next window)
    const int64_t col = F.begin_of(c1);
    int64_t lo = c0, hi = c1;      // first cell in [c0, c1] with begin == col
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (F.begin_of(mid) < col) lo = mid + 1; else hi = mid; }
    if (lo > c0) { c1 = lo; break; }
    want *= 2;                     // one column is wider than the budget: take more
  }
  const int64_t n = c1 - c0;
  w.c1 = c1; w.ncells = n;
  w.first_begin = F.begin_of(c0); w.last_begin = F.begin_of(c1 - 1);
  w.next_begin = c1 < F.C ? F.begin_of(c1) : INT64_MAX;
  w.reference_cell_bytes = F.C > 0 ? (uint64_t)((double)F.meta.reference_cell_bytes * ((double)n / (double)F.C)) : 0;
  // ---- the part ----------------------------------------------------------------------------------------------------------
  Impl::Part part;
  memset(&part.v, 0, sizeof(part.v));
  part.v.ncells = n;
  // every way out of this function other than the push at its end (corrupt offsets, a corrupt tile index, a read beyond a section,
  // a failing HIP call) gives the part's device buffers back: a damaged file must not cost HBM for the life of the process
  struct PartGuard {
    Impl::Part& p; BlockPool& pool; bool keep = false;
    ~PartGuard() { if (!keep) { for (void* b : p.bufs) pool.put(b); p.bufs.clear(); } }
  } guard{part, S.blocks};
  auto alloc = [&](size_t bytes) -> void* { void* d = S.blocks.get(bytes); part.bufs.push_back(d); return d; };
  void* row = alloc((size_t)n * 4); void* begin = alloc((size_t)n * 8); void* end = alloc((size_t)n * 8);
  part.v.row = (const int32_t*)row; part.v.begin = (const int64_t*)begin; part.v.end = (const int64_t*)end;
  const int nf = S.hp.plan.nfields;
  S.col_elem_size.assign((size_t)nf, 4); S.col_var.assign((size_t)nf, false); S.col_fixed_num.assign((size_t)nf, 1);
  part.data_bytes.assign((size_t)nf, 0);
  // bytes [a, a + nbytes) of a field's data -> device.  Compressed sections: the tiles that cover the range cross PCIe as they are in
  // the file; ALL tiles of the window (every field) are inflated by ONE launch of k_inflate_tiles (one thread per tile: the more
  // tiles in flight, the better the serial decoding of each is hidden), then the ranges are copied out of the inflated tiles.
  struct InflateCopy { void* dev; uint64_t from, bytes; };
  struct Rebase { uint32_t* off; uint32_t delta, total; };
  std::vector<Rebase> rebase;
  std::vector<uint64_t> job_in(1, 0);
  std::vector<uint32_t> job_want;
  std::vector<InflateCopy> copies;
  std::vector<std::pair<uint64_t, uint64_t>> file_ranges;      // (file offset, bytes) of the compressed bytes, in job order
  auto section_to_device = [&](const FragmentFile::Section& sec, void* dev, uint64_t a, uint64_t nbytes) {
    if (a + nbytes > sec.bytes) throw std::runtime_error(F.path + ": read beyond a section of the fragment file");
    if (!sec.compressed) { F.to_device(dev, sec.at + a, nbytes, st); return; }
    if (nbytes == 0) return;
    const uint64_t t0 = a / kFragTile, t1 = (a + nbytes - 1) / kFragTile, nt = t1 - t0 + 1;
    std::vector<uint64_t> offs((size_t)nt + 1);
    F.read_at(offs.data(), sec.index_at + 8 * t0, (size_t)(nt + 1) * 8);
    for (uint64_t i = 0; i < nt; ++i) if (offs[i + 1] < offs[i] || sec.at + offs[i + 1] > sec.index_at) throw std::runtime_error(F.path + ": corrupt tile index");
    const uint64_t first_job = job_want.size();
    for (uint64_t i = 0; i < nt; ++i) {
      job_in.push_back(job_in.back() + (offs[i + 1] - offs[i]));
      job_want.push_back((uint32_t)std::min<uint64_t>(kFragTile, sec.bytes - (t0 + i) * kFragTile));
    }
    file_ranges.emplace_back(sec.at + offs[0], offs[nt] - offs[0]);
    copies.push_back(InflateCopy{dev, first_job * kFragTile + (a - t0 * kFragTile), nbytes});
  };
  section_to_device(F.sec_row, row, (uint64_t)c0 * 4, (uint64_t)n * 4);
  section_to_device(F.sec_begin, begin, (uint64_t)c0 * 8, (uint64_t)n * 8);
  section_to_device(F.sec_end, end, (uint64_t)c0 * 8, (uint64_t)n * 8);
  auto inflate_all = [&]() {
    const uint64_t njobs = job_want.size();
    if (njobs == 0) return;
    S.inflate_in.ensure(job_in.back() + 16); S.inflate_off.ensure((size_t)njobs + 1); S.inflate_want.ensure((size_t)njobs); S.inflate_out.ensure(njobs * kFragTile);
    S.inflate_scratch.ensure(njobs * (size_t)kInflateScratch);
    uint64_t at_in = 0;
    for (auto& fr : file_ranges) { F.to_device(S.inflate_in.p + at_in, fr.first, fr.second, st); at_in += fr.second; }
    HIP_CHECK(hipMemcpyAsync(S.inflate_off.p, job_in.data(), (size_t)(njobs + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_CHECK(hipMemcpyAsync(S.inflate_want.p, job_want.data(), (size_t)njobs * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_inflate_tiles, dim3(blocks_for((int64_t)njobs, 64)), dim3(64), 0, st, (const uint8_t*)S.inflate_in.p, (const uint64_t*)S.inflate_off.p, (const uint32_t*)S.inflate_want.p,
                       (int64_t)njobs, S.inflate_out.p, S.inflate_scratch.p, S.err.p);
    for (auto& c : copies) HIP_CHECK(hipMemcpyAsync(c.dev, S.inflate_out.p + c.from, c.bytes, hipMemcpyDeviceToDevice, st));
    HIP_CHECK(hipStreamSynchronize(st));                          // (the host vectors above are read by the copies)
    if (S.read_back(S.err.p) != 0u) {
      throw std::runtime_error(F.path + ": a compressed tile does not inflate to its size (corrupt file)");
    }
  };
  auto rebase_offsets = [&]() {
    for (auto& r : rebase) {
      hipLaunchKernelGGL(k_copy_offsets, dim3(blocks_for(n + 1)), dim3(kBlock), 0, st, (const uint32_t*)r.off, n + 1, r.delta, r.off);
      hipLaunchKernelGGL(k_fragment_offsets_check, dim3(blocks_for(n + 1)), dim3(kBlock), 0, st, (const uint32_t*)r.off, n, r.total, S.err.p);
    }
  };
  HIP_CHECK(hipMemsetAsync(S.err.p, 0, sizeof(uint32_t), st));
  for (const FragmentFile::Field& fd : F.fields) {
    const int f = fd.plan_field;
    if (f < 0) continue;
    S.col_elem_size[(size_t)f] = fd.elem_size; S.col_var[(size_t)f] = fd.var; S.col_fixed_num[(size_t)f] = fd.fixed_num;
    if (fd.var) {
      uint32_t o0 = 0, o1 = 0;
      F.host_read(fd.off, (uint64_t)c0 * 4, &o0, 4); F.host_read(fd.off, (uint64_t)c1 * 4, &o1, 4);
      if (o1 < o0 || (uint64_t)o1 * (uint64_t)fd.elem_size > fd.data_bytes) throw std::runtime_error(F.path + ": corrupt offsets in a variable-length column");
      uint32_t* off = (uint32_t*)alloc(((size_t)n + 1) * 4);
      section_to_device(fd.off, off, (uint64_t)c0 * 4, ((uint64_t)n + 1) * 4);
      rebase.push_back(Rebase{off, (uint32_t)(0u - o0), o1 - o0});   // (rebased to the part once the tiles are inflated)
      const uint64_t bytes = (uint64_t)(o1 - o0) * (uint64_t)fd.elem_size;
      void* data = alloc((size_t)bytes);
      section_to_device(fd.data, data, (uint64_t)o0 * (uint64_t)fd.elem_size, bytes);
      part.v.col[f].off = off; part.v.col[f].data = data;
      part.data_bytes[(size_t)f] = (size_t)bytes;
    } else {
      const uint64_t bpc = (uint64_t)fd.fixed_num * (uint64_t)fd.elem_size;
      void* data = alloc((size_t)((uint64_t)n * bpc));
      section_to_device(fd.data, data, (uint64_t)c0 * bpc, (uint64_t)n * bpc);
      part.v.col[f].data = data;
      part.data_bytes[(size_t)f] = (size_t)((uint64_t)n * bpc);
    }
  }
  inflate_all();
  rebase_offsets();
  hipLaunchKernelGGL(k_fragment_part_check, dim3(blocks_for(n)), dim3(kBlock), 0, st, (const int32_t*)row, (const int64_t*)begin, (const int64_t*)end, n, (int32_t)S.hp.plan.num_query_rows, S.err.p);
  // boundary markers whose column falls into this window
  {
    size_t m0 = F.marker_cursor;
    if (m0 > 0 && (m0 > F.markers.size() || F.markers[m0 - 1] >= w.first_begin)) m0 = 0;           // a seek backwards: search again
    while (m0 < F.markers.size() && F.markers[m0] < w.first_begin) ++m0;
    size_t m1 = m0;
    while (m1 < F.markers.size() && F.markers[m1] < w.next_begin) ++m1;
    F.marker_cursor = m1;
    if (m1 > m0) {
      int64_t* mk = (int64_t*)alloc((m1 - m0) * 8);
      HIP_CHECK(hipMemcpyAsync(mk, F.markers.data() + m0, (m1 - m0) * 8, hipMemcpyHostToDevice, st));
      part.v.nmarkers = (int64_t)(m1 - m0); part.v.marker_begin = mk;
    }
  }
  HIP_CHECK(hipStreamSynchronize(st));
  if (S.read_back(S.err.p) != 0u) {
    throw std::runtime_error(F.path + ": the columns of the fragment file are damaged (cells out of order, rows outside the query's rows or offsets that do not match the data)");
  }
  S.parts.push_back(part);
  guard.keep = true;
  return w;
}

FragmentFileMeta DevicePipeline::load_fragment(const std::string& path, const std::vector<ColumnLayout>& expected, uint64_t expected_schema_hash) {
  const FragmentFileMeta meta = open_fragment_file(path, expected, expected_schema_hash);
  begin_staging();
  append_fragment_cells(0, UINT64_MAX / 4);
  finish_staging();
  close_fragment_file();
  return meta;
}

void DevicePipeline::adopt_fragment(const FragmentView& v) {
  m_->free_owned();
  m_->fr = v;
  m_->owns_fragment = false;
  m_->classified = false; ++m_->fragment_generation;
}

void DevicePipeline::set_reference_window(int64_t begin, const std::string& bases) {
  HIP_CHECK(hipSetDevice(m_->device));
  m_->ref_bases.ensure(std::max<size_t>(bases.size(), 1));
  if (!bases.empty()) {
    HIP_CHECK(hipMemcpyAsync(m_->ref_bases.p, bases.data(), bases.size(), hipMemcpyHostToDevice, m_->stream));
    HIP_CHECK(hipStreamSynchronize(m_->stream));
  }
  m_->ref_begin = begin;
  m_->ref_len = (int64_t)bases.size();
}

// Every cell begin closes the current interval of the sweep (query_variants.cc:478-505), so a query interval can be cut right
// before any cell begin without changing a single output byte.  Returns the last column of the first piece of [qb, qe] that
// is at most max_columns wide where the data allows it (wider only if no cell begins in between).
int64_t DevicePipeline::split_point(int64_t qb, int64_t qe, int64_t max_columns) {
  Impl& S = *m_;
  if (max_columns <= 0 || qe - qb < max_columns || S.fr.ncells == 0) return qe;
  HIP_CHECK(hipSetDevice(S.device));
  S.cwin.ensure(4);
  const int64_t pos = qb + max_columns;
  hipLaunchKernelGGL(k_first_begin_at_or_after, dim3(1), dim3(64), 0, S.stream, S.fr.begin, S.fr.ncells, pos, S.cwin.p);
  int64_t p[2] = {INT64_MAX, INT64_MAX};
  if (S.fr.nmarkers > 0) hipLaunchKernelGGL(k_first_begin_at_or_after, dim3(1), dim3(64), 0, S.stream, S.fr.marker_begin, S.fr.nmarkers, pos, S.cwin.p + 1);
  HIP_CHECK(hipMemcpyAsync(p, S.cwin.p, (S.fr.nmarkers > 0 ? 2 : 1) * sizeof(int64_t), hipMemcpyDeviceToHost, S.stream));
  HIP_CHECK(hipStreamSynchronize(S.stream));
  const int64_t cut = std::min(p[0], p[1]);
  return (cut == INT64_MAX || cut > qe) ? qe : cut - 1;
}

// ColumnHistogramOperator on the device (variant_operations.cc:732-767): every staged begin-cell counts for the bin of its begin column,
// begin <= hist_begin in bin 0, begin >= hist_end in the last one.  One atomic per run of equal bins inside a wavefront (the cells
// are sorted by begin: neighbouring lanes nearly always share their bin).
// eff_end != nullptr: only the cells of the query interval [qb, qe] count - the ones the reference's iterate_over_cells hands to the
// operator (calls_select: cells that begin inside, and with_intersecting the intervals that began in front of qb and reach it)
__global__ void k_column_histogram(const int64_t* __restrict__ begin, int64_t C, uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, uint64_t nbins, unsigned long long* __restrict__ counts,
                                   const int64_t* __restrict__ eff_end, int64_t qb, int64_t qe, int with_intersecting) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint64_t bin = ~0ull;
  if (c < C) {
    const int64_t bc = begin[c];
    const bool take = !eff_end || (bc >= qb && bc <= qe) || (with_intersecting && bc < qb && eff_end[c] >= qb);
    const uint64_t b = (uint64_t)bc;
    if (take) bin = b <= hist_begin ? 0ull : b >= hist_end ? nbins - 1 : (b - hist_begin) / bin_size;
  }
  const uint64_t prev = __shfl_up(bin, 1, 64);
  const bool head = lane == 0 || prev != bin;
  const uint64_t heads = __ballot(head);
  if (head && bin != ~0ull) {
    const uint64_t above = lane == 63 ? 0ull : heads >> (lane + 1);
    const int run = above ? __builtin_ctzll(above) + 1 : 64 - lane;
    atomicAdd(&counts[bin], (unsigned long long)run);
  }
}
void DevicePipeline::column_histogram(uint64_t hist_begin, uint64_t hist_end, uint64_t bin_size, uint64_t* counts, uint64_t nbins, bool accumulate, const int64_t* interval, bool with_intersecting) {
  Impl& S = *m_;
  if (bin_size == 0 || hist_end < hist_begin || nbins != (hist_end - hist_begin) / bin_size + 1) throw GenomicsDBDeviceException("column_histogram: #bins must be (end - begin) / bin_size + 1");
  HIP_CHECK(hipSetDevice(S.device));
  if (interval) {     // the cells of one query interval (or of the piece of it the staged window serves): every staged cell is looked at
    if (S.fr.ncells == 0) return;
    classify_fragment();                                        // (eff_end)
    DevBuf<unsigned long long> di;
    di.ensure(nbins);
    if (accumulate) HIP_CHECK(hipMemcpyAsync(di.p, counts, nbins * sizeof(uint64_t), hipMemcpyHostToDevice, S.stream));
    else HIP_CHECK(hipMemsetAsync(di.p, 0, nbins * sizeof(uint64_t), S.stream));
    hipLaunchKernelGGL(k_column_histogram, dim3(blocks_for(S.fr.ncells)), dim3(kBlock), 0, S.stream, S.fr.begin, S.fr.ncells, hist_begin, hist_end, bin_size, nbins, di.p,
                       (const int64_t*)S.eff_end.p, interval[0], interval[1], with_intersecting ? 1 : 0);
    HIP_CHECK(hipMemcpyAsync(counts, di.p, nbins * sizeof(uint64_t), hipMemcpyDeviceToHost, S.stream));
    HIP_CHECK(hipStreamSynchronize(S.stream));
    return;
  }
  DevBuf<unsigned long long> d;
  d.ensure(nbins);
  if (accumulate) HIP_CHECK(hipMemcpyAsync(d.p, counts, nbins * sizeof(uint64_t), hipMemcpyHostToDevice, S.stream));
  else HIP_CHECK(hipMemsetAsync(d.p, 0, nbins * sizeof(uint64_t), S.stream));
  // (the cells carried over from the previous column window open the fragment: they were counted with their own window)
  const int64_t first = std::min<int64_t>(S.carried_cells, S.fr.ncells), n = S.fr.ncells - first;
  if (n > 0)
    hipLaunchKernelGGL(k_column_histogram, dim3(blocks_for(n)), dim3(kBlock), 0, S.stream, S.fr.begin + first, n, hist_begin, hist_end, bin_size, nbins, d.p,
                       (const int64_t*)nullptr, (int64_t)0, (int64_t)0, 0);
  HIP_CHECK(hipMemcpyAsync(counts, d.p, nbins * sizeof(uint64_t), hipMemcpyDeviceToHost, S.stream));
  HIP_CHECK(hipStreamSynchronize(S.stream));
}

// S0 classify + S1 row index + S2 effective END: independent of the query interval, once per staged fragment
void DevicePipeline::classify_fragment() {
  Impl& S = *m_;
  hipStream_t st = S.stream;
  const CombinePlan& pl = S.hp.plan;
  const FragmentView& fr = S.fr;
  const int64_t C = fr.ncells;
  const int32_t N = pl.num_query_rows;
  S.vmask.ensure(C); S.cflags.ensure(C); S.dpval.ensure(C); S.k_lo.ensure(C); S.k_hi.ensure(C); S.eff_end.ensure(C);
  CellMeta cm{S.vmask.p, S.cflags.p, S.dpval.p, S.eff_end.p, S.k_lo.p, S.k_hi.p};
  S.perm.ensure(C); S.rm_begin.ensure(C); S.row_ptr.ensure((size_t)N + 2);
  if (!S.classified) {
    STAGE("k_classify");
    hipLaunchKernelGGL(k_classify, dim3(blocks_for(C)), dim3(kBlock), 0, st, fr, pl, cm, S.err.p);
    S.row_keys.ensure(C); S.row_keys_sorted.ensure(C); S.cell_ids.ensure(C);
    STAGE("k_iota_rows");
    hipLaunchKernelGGL(k_iota_rows, dim3(blocks_for(C)), dim3(kBlock), 0, st, fr, S.row_keys.p, S.cell_ids.p);
    S.sort_pairs(S.row_keys.p, S.row_keys_sorted.p, S.cell_ids.p, S.perm.p, (size_t)C, std::min(32, bits_for((uint64_t)N)));
    STAGE("k_row_ptr");
    hipLaunchKernelGGL(k_row_ptr, dim3(blocks_for((int64_t)N + 1)), dim3(kBlock), 0, st, S.row_keys_sorted.p, C, N, S.row_ptr.p);
    // S2 effective END of every cell (next cell of the same sample overrides) + the longest live span
    S.span.ensure(C); S.span_max.ensure(2);
    STAGE("k_eff_end");
    hipLaunchKernelGGL(k_eff_end, dim3(blocks_for(C)), dim3(kBlock), 0, st, fr, cm, S.perm.p, S.rm_begin.p, S.span.p, S.err.p);
    {
      size_t bytes = 0;
      HIP_CHECK(rocprim::reduce(nullptr, bytes, S.span.p, S.span_max.p, (int64_t)0, (size_t)C, rocprim::maximum<int64_t>(), st));
      void* t = S.temp_storage(bytes);
      HIP_CHECK(rocprim::reduce(t, bytes, S.span.p, S.span_max.p, (int64_t)0, (size_t)C, rocprim::maximum<int64_t>(), st));
    }
    S.walk.ensure(C + N + 2); S.walk_inv.ensure(C);
    STAGE("k_walk_static");
    hipLaunchKernelGGL(k_walk_static, dim3(blocks_for(C + N)), dim3(kBlock), 0, st, S.perm.p, fr.row, S.row_ptr.p, N, S.eff_end.p, S.cflags.p, C, S.walk.p, S.walk_inv.p);
    S.walk_epoch = 0;
    S.max_span = S.read_back(S.span_max.p);
    S.classified = true;
  }
}

// ---- gt_mpi_gather --print-calls: the cells of a query column interval as JSON objects (core/gdb_calls.hpp) --------------------------
// Which cells: the reference's SingleCellTileDBIterator (genomicsdb_iterators.cc:181-301, 425-510) first looks for the intervals that
// began before the query interval and intersect its begin - per row the first cell at or behind it, if that is an END copy -, hands
// them out in column-major order of their begins, then traverses [qb, qe].  The staged fragment holds begin cells only, sorted by
// (begin, row), and every cell's effective END (where the loader's truncated END copy lies): the cells are those with
// begin < qb <= eff_end, printed with eff_end as their END, followed by those with qb <= begin <= qe - already in the right order.
// One thread per cell; the emitter runs twice (length, text) around a scan, like the other emitters of this file.
template <bool WRITE> __global__ void k_calls(FragmentView fr, CombinePlan pl, QueryWindow qw, CallsNames names, const int64_t* __restrict__ eff_end, int64_t c_base, int64_t n,
                                              int64_t qb, int64_t qe, int mode, int indent, int with_intersecting, uint64_t* __restrict__ len_or_off, char* __restrict__ out, uint32_t* err) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = c_base + i;
  int64_t end = 0;
  const bool take = calls_select(fr, eff_end, c, qb, qe, with_intersecting != 0, end);
  uint32_t e = 0;
  auto emit = [&](auto& s) {
    if (mode == 0) { s.put(','); s.put('\n'); calls_emit_cell(s, fr, pl, qw, names, c, end, indent); }   // ",\n" in front of every cell (the caller drops the first)
    else if (mode == 1) calls_emit_csv(s, fr, pl, names, c, end);
    else calls_emit_allele_lines(s, fr, pl, c, indent, &e);                                             // (indent: the GT step)
  };
  if (!WRITE) {
    uint64_t l = 0;
    if (take) { CountSink s; emit(s); l = s.n; }
    len_or_off[i] = l;
  } else if (take) {
    ByteSink s(out + len_or_off[i]);
    emit(s);
  }
  if (e) atomicOr(err, e);
}

void DevicePipeline::set_array_rows(const std::vector<int64_t>& query_row_to_array_row) {
  Impl& S = *m_;
  HIP_CHECK(hipSetDevice(S.device));
  S.calls_array_row_n = query_row_to_array_row.size();
  if (!S.calls_array_row_n) return;
  S.calls_array_row.ensure(S.calls_array_row_n);
  HIP_CHECK(hipMemcpy(S.calls_array_row.p, query_row_to_array_row.data(), S.calls_array_row_n * sizeof(int64_t), hipMemcpyHostToDevice));
}
std::string DevicePipeline::calls_json(int64_t qb, int64_t qe, int indent, bool with_intersecting, int64_t* ncells) {
  std::string out = cells_text(qb, qe, 0, indent, with_intersecting);
  if (ncells) { int64_t k = 0; const std::string head = ",\n" + std::string((size_t)indent, ' ') + "{\n"; for (size_t q = out.find(head); q != std::string::npos; q = out.find(head, q + 1)) ++k; *ncells = k; }
  return out.empty() ? out : out.substr(2);   // (without the first separator)
}
std::string DevicePipeline::cells_text(int64_t qb, int64_t qe, int mode, int indent, bool with_intersecting) {
  Impl& S = *m_;
  HIP_CHECK(hipSetDevice(S.device));
  hipStream_t st = S.stream;
  const FragmentView& fr = S.fr;
  const int64_t C = fr.ncells;
  if (C == 0 || S.hp.plan.num_query_rows == 0) return std::string();
  for (int f = 0; f < S.hp.plan.nfields; ++f)
    if (S.hp.plan.field[f].ndim == 2) throw GenomicsDBDeviceException("print-calls: 2-dimensional field " + S.hp.field_names[(size_t)f] + " is not printed on the device");
  HIP_CHECK(hipMemsetAsync(S.err.p, 0, sizeof(uint32_t), st));
  classify_fragment();
  if (!S.calls_names_ready) {
    std::string text; std::vector<int32_t> off;
    for (const auto& nm : S.hp.field_names) { off.push_back((int32_t)text.size()); text += nm; }
    off.push_back((int32_t)text.size());
    S.calls_names.ensure(text.size() + 1); S.calls_name_off.ensure(off.size());
    HIP_CHECK(hipMemcpy(S.calls_names.p, text.data(), text.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(S.calls_name_off.p, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    S.calls_names_ready = true;
  }
  S.cwin.ensure(4);
  hipLaunchKernelGGL(k_cell_window, dim3(1), dim3(64), 0, st, fr.begin, C, qb > INT64_MIN + S.max_span ? qb - S.max_span : INT64_MIN, qe, S.cwin.p);
  int64_t cw[2] = {0, 0};
  S.read_back_many({{cw, S.cwin.p, 2 * sizeof(int64_t)}});
  const int64_t c_base = cw[0], CW = cw[1] - cw[0];
  if (CW <= 0) return std::string();
  QueryWindow qw;
  memset(&qw, 0, sizeof(qw));
  qw.qb = qb; qw.qe = qe;
  qw.contigs = S.contigs.p; qw.ncontigs = (int32_t)S.hp.contigs.size(); qw.contig_names = S.contig_names.p;
  CallsNames names{S.calls_names.p, S.calls_name_off.p, S.calls_array_row_n ? S.calls_array_row.p : nullptr};
  S.calls_len.ensure((size_t)CW + 1); S.calls_off.ensure((size_t)CW + 1);
  hipLaunchKernelGGL(k_calls<false>, dim3(blocks_for(CW)), dim3(kBlock), 0, st, fr, S.hp.plan, qw, names, (const int64_t*)S.eff_end.p, c_base, CW, qb, qe, mode, indent, with_intersecting ? 1 : 0, S.calls_len.p, (char*)nullptr, S.err.p);
  HIP_CHECK(hipMemsetAsync(S.calls_len.p + CW, 0, sizeof(uint64_t), st));
  S.excl_scan(S.calls_len.p, S.calls_off.p, (size_t)CW + 1);
  const uint64_t total = S.read_back(S.calls_off.p + CW);
  if (total == 0) return std::string();
  S.calls_text.ensure((size_t)total + 16);
  hipLaunchKernelGGL(k_calls<true>, dim3(blocks_for(CW)), dim3(kBlock), 0, st, fr, S.hp.plan, qw, names, (const int64_t*)S.eff_end.p, c_base, CW, qb, qe, mode, indent, with_intersecting ? 1 : 0, S.calls_off.p, S.calls_text.p, S.err.p);
  std::string out((size_t)total, '\0');
  HIP_CHECK(hipMemcpyAsync(&out[0], S.calls_text.p, (size_t)total, hipMemcpyDeviceToHost, st));
  uint32_t eb = 0;
  S.read_back_many({{&eb, S.err.p, sizeof(uint32_t)}});
  if (eb) throw GenomicsDBDeviceException(err_bits_text(eb));
  return out;
}

// the sizing / resolution pass of `n` records in `order` (piece-wise kernel by default; GDBAMD_SIZE3=0: the record-by-record one).
// size_slots: elements of chunk_size (the check mode compares them all)
static void launch_assemble_size(DevicePipeline::Impl& S, hipStream_t st, dim3 grid, const AsmCtx& ac, const int32_t* order, int64_t n, int32_t N, int nchunks, int run,
                                 uint64_t* chunk_size, size_t size_slots, ResMatrix resolved, int64_t resolved_base, int64_t res_rows) {
  const int rounds = size3_rounds();
  if (rounds == 0 || run > 64 * 1024) { hipLaunchKernelGGL(k_assemble_size, grid, dim3(kAsmRows), 0, st, ac, order, n, N, nchunks, run, chunk_size, resolved, resolved_base, res_rows); return; }
  if (rounds >= 16) hipLaunchKernelGGL((k_assemble_size3<16>), grid, dim3(kAsmRows), 0, st, ac, order, n, N, nchunks, run, chunk_size, resolved, resolved_base, res_rows);
  else hipLaunchKernelGGL((k_assemble_size3<8>), grid, dim3(kAsmRows), 0, st, ac, order, n, N, nchunks, run, chunk_size, resolved, resolved_base, res_rows);
  if (!size3_check()) return;
  const size_t nres = resolved.wide ? (size_t)n * (size_t)nchunks * kAsmRows : 0;     // (the check runs with the wide layout: res_compact_wanted())
  S.chk_count.ensure(1);
  HIP_CHECK(hipMemsetAsync(S.chk_count.p, 0, sizeof(unsigned long long), st));
  if (nres) S.chk_resolved.ensure(nres);
  if (chunk_size) { S.chk_sizes.ensure(size_slots); HIP_CHECK(hipMemcpyAsync(S.chk_sizes.p, chunk_size, size_slots * sizeof(uint64_t), hipMemcpyDeviceToDevice, st)); }   // (records outside `order` keep what they had)
  hipLaunchKernelGGL(k_assemble_size, grid, dim3(kAsmRows), 0, st, ac, order, n, N, nchunks, run, chunk_size ? S.chk_sizes.p : (uint64_t*)nullptr,
                     ResMatrix{nres ? S.chk_resolved.p : (uint2*)nullptr, nullptr, nullptr}, resolved_base, res_rows);
  // (the rows of the records in `order` only: with a base, row (k - base); the launch sites pass records [base, base + n))
  if (nres) hipLaunchKernelGGL(k_compare_words, dim3(blocks_for((int64_t)nres * 2)), dim3(kBlock), 0, st, (const uint32_t*)resolved.wide, (const uint32_t*)S.chk_resolved.p, (int64_t)nres * 2, S.chk_count.p);
  if (chunk_size) hipLaunchKernelGGL(k_compare_words, dim3(blocks_for((int64_t)size_slots * 2)), dim3(kBlock), 0, st, (const uint32_t*)chunk_size, (const uint32_t*)S.chk_sizes.p, (int64_t)size_slots * 2, S.chk_count.p);
  const unsigned long long bad = S.read_back(S.chk_count.p);
  if (bad) throw GenomicsDBDeviceException("GDBAMD_SIZE3_CHECK: the piece-wise sizing pass differs from the record-by-record one in " + std::to_string(bad) + " words");
}

void DevicePipeline::prepare_interval(int64_t qb, int64_t qe) {
  Impl& S = *m_;
  S.iv = Impl::IntervalState();
  S.order_k0 = S.order_n = -1;
  S.order_iv_valid = false;
  HIP_CHECK(hipSetDevice(S.device));
  hipStream_t st = S.stream;
  const CombinePlan& pl = S.hp.plan;
  const FragmentView& fr = S.fr;
  const int64_t C = fr.ncells;
  const int32_t N = pl.num_query_rows;
  IntervalStats& stats = S.iv.stats;
  stats.num_cells = C;
  if (C == 0 || N == 0) return;
  hipEvent_t* ev = S.ev_prep;
  HIP_CHECK(hipMemsetAsync(S.err.p, 0, sizeof(uint32_t), st));
  HIP_CHECK(hipMemsetAsync(S.counters.p, 0, (16 + kCountSpread) * sizeof(int32_t), st));
  HIP_CHECK(hipEventRecord(ev[0], st));
  classify_fragment();
  CellMeta cm{S.vmask.p, S.cflags.p, S.dpval.p, S.eff_end.p, S.k_lo.p, S.k_hi.p};
  // ---- cells that can reach the window -----------------------------------------------------------------------------
  S.cwin.ensure(4);
  STAGE("k_cell_window");
  hipLaunchKernelGGL(k_cell_window, dim3(1), dim3(64), 0, st, fr.begin, C, qb > INT64_MIN + S.max_span ? qb - S.max_span : INT64_MIN, qe, S.cwin.p);
  if (fr.nmarkers > 0) hipLaunchKernelGGL(k_cell_window, dim3(1), dim3(64), 0, st, fr.marker_begin, fr.nmarkers, qb, qe, S.cwin.p + 2);   // boundary markers in [qb, qe]
  int64_t cw[4] = {0, 0, 0, 0};
  S.read_back_many({{cw, S.cwin.p, (fr.nmarkers > 0 ? 4 : 2) * sizeof(int64_t)}});
  const int64_t c_base = cw[0], c_end = cw[1];
  const int64_t CW = c_end - c_base;
  if (CW <= 0) return;
  const int64_t m_base = cw[2], MW = std::max<int64_t>(0, cw[3] - cw[2]);
  // ---- S3 events -> boundaries -> records --------------------------------------------------------------------------
  const int64_t NE = 2 * CW + 2 * MW;
  S.ev_keys.ensure(NE); S.ev_keys_sorted.ensure(NE); S.ev_delta.ensure(NE); S.ev_incl.ensure(NE); S.run_end.ensure(NE); S.run_excl.ensure(NE + 1);
  STAGE("k_event_keys");
  hipLaunchKernelGGL(k_event_keys, dim3(blocks_for(CW)), dim3(kBlock), 0, st, fr, cm, c_base, CW, qb, qe, S.ev_keys.p);
  if (MW > 0) hipLaunchKernelGGL(k_marker_keys, dim3(blocks_for(MW)), dim3(kBlock), 0, st, fr, m_base, MW, qb, qe, S.ev_keys.p + 2 * CW);
  {
    const uint64_t span = (uint64_t)(qe - qb) + 2u;
    const int eb = span >= (1ull << 60) ? 64 : bits_for(span << 2);
    S.sort_keys(S.ev_keys.p, S.ev_keys_sorted.p, (size_t)NE, eb);
  }
  STAGE("k_event_delta");
  hipLaunchKernelGGL(k_event_delta, dim3(blocks_for(NE)), dim3(kBlock), 0, st, S.ev_keys_sorted.p, NE, S.ev_delta.p, S.run_end.p);
  S.incl_scan(S.ev_delta.p, S.ev_incl.p, (size_t)NE, PackedAdd());
  S.excl_scan(S.run_end.p, S.run_excl.p, (size_t)NE);
  const int64_t U = S.read_back_sum(S.run_excl.p + (NE - 1), S.run_end.p + (NE - 1));
  int64_t P = 0;
  if (U > 0) {
    S.bpos.ensure(U + 1); S.bcov.ensure(U + 1); S.bdel.ensure(U + 1); S.bnrec.ensure(U + 1); S.rbase.ensure(U + 1);
    Boundaries bd{S.bpos.p, S.bcov.p, S.bdel.p, S.bnrec.p};
    STAGE("k_boundary_write");
    hipLaunchKernelGGL(k_boundary_write, dim3(blocks_for(NE)), dim3(kBlock), 0, st, S.ev_keys_sorted.p, S.ev_incl.p, S.run_end.p, S.run_excl.p, NE, bd, qb);
    STAGE("k_boundary_nrec");
    hipLaunchKernelGGL(k_boundary_nrec, dim3(blocks_for(U)), dim3(kBlock), 0, st, bd, U);
    S.excl_scan(S.bnrec.p, S.rbase.p, (size_t)U);
    P = S.read_back_sum(S.rbase.p + (U - 1), S.bnrec.p + (U - 1));
    if (P > 0) {
      S.rstart.ensure(P); S.rend.ensure(P);
      STAGE("k_record_expand");
      hipLaunchKernelGGL(k_record_expand, dim3(blocks_for(P)), dim3(kBlock), 0, st, bd, S.rbase.p, U, P, S.rstart.p, S.rend.p);
    }
  }
  stats.num_records = P;
  if (P == 0) {
    stats.err_bits = S.read_back(S.err.p);
    if (stats.err_bits) throw GenomicsDBDeviceException(err_bits_text(stats.err_bits));
    return;
  }
  if (P >= (1ll << 31)) throw GenomicsDBDeviceException("more than 2^31 records in one interval: split the query interval");
  RecordTable rec{P, S.rstart.p, S.rend.p};
  // ---- S4/S5 record ranges, difference arrays, scans --------------------------------------------------------------
  const int nf = pl.n_format;
  const int64_t stride = P + 1;
  const size_t ndiff = (size_t)(nf + 2) * (size_t)stride;
  S.diff.ensure(ndiff);
  // counters packed into 64-bit words for the atomics (width = bits of the sample count; fmt fields + <NON_REF> count, DP last)
  DiffPacked pk;
  pk.width = 1; while (pk.width < 32 && (1ll << pk.width) <= (int64_t)N) ++pk.width;
  pk.per_word = 64 / pk.width;
  pk.nwords = (nf + 1 + pk.per_word - 1) / pk.per_word + 1;
  pk.stride = stride;
  S.diff_packed.ensure((size_t)pk.nwords * stride);
  pk.w = S.diff_packed.p;
  HIP_CHECK(hipMemsetAsync(pk.w, 0, (size_t)pk.nwords * stride * sizeof(uint64_t), st));
  int32_t* d_fmt = S.diff.p;
  int32_t* d_dp = S.diff.p + (size_t)nf * stride;
  int32_t* d_nr = d_dp + stride;
  DiffArrays da{d_fmt, d_dp, d_nr, stride};
  S.heavy_count.ensure(CW + 1); S.hoff.ensure(CW + 2);
  // position -> first record table (windows up to 2^27 columns; wider ones keep the binary searches)
  const int32_t* first_record_at = nullptr;
  if ((uint64_t)(qe - qb) + 2u <= (1ull << 27)) {
    const int64_t W = qe - qb + 2;
    S.first_record_at.ensure(W);
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks_for(W)), dim3(kBlock), 0, st, S.first_record_at.p, W, (int32_t)P);
    hipLaunchKernelGGL(k_mark_record_starts, dim3(blocks_for(P)), dim3(kBlock), 0, st, S.rstart.p, P, qb, S.first_record_at.p);
    {
      auto rin = rocprim::make_reverse_iterator(S.first_record_at.p + W);
      size_t bytes = 0;
      HIP_CHECK(rocprim::inclusive_scan(nullptr, bytes, rin, rin, (size_t)W, rocprim::minimum<int32_t>(), st));
      void* t = S.temp_storage(bytes);
      HIP_CHECK(rocprim::inclusive_scan(t, bytes, rin, rin, (size_t)W, rocprim::minimum<int32_t>(), st));
    }
    first_record_at = S.first_record_at.p;
  }
  STAGE("k_cell_ranges");
  hipLaunchKernelGGL(k_cell_ranges, dim3(blocks_for(CW)), dim3(kBlock), 0, st, fr, pl, cm, rec, c_base, CW, qb, qe, S.heavy_count.p, S.counters.p, first_record_at, pk);
  // every difference array sums to zero over its P+1 elements, so ONE scan over the concatenation equals separate scans
  S.incl_scan(pk.w, pk.w, (size_t)pk.nwords * stride, rocprim::plus<uint64_t>());
  hipLaunchKernelGGL(k_unpack_counts, dim3(blocks_for(stride)), dim3(kBlock), 0, st, pk, nf, stride, da);
  S.excl_scan(S.heavy_count.p, S.hoff.p, (size_t)CW);
  int64_t T_a = 0, T_b = 0;
  int32_t n_in_window = 0, spread[kCountSpread], sweep_counts[2] = {0, 0};
  S.read_back_many({{&T_a, S.hoff.p + (CW - 1), sizeof(int64_t)}, {&T_b, S.heavy_count.p + (CW - 1), sizeof(int64_t)}, {spread, S.counters.p + 16, sizeof(spread)},
                    {sweep_counts, S.counters.p + 4, sizeof(sweep_counts)}});
  for (int i = 0; i < kCountSpread; ++i) n_in_window += spread[i];
  stats.gt_profile[GT_NUM_CELLS] = (uint64_t)CW;
  stats.gt_profile[GT_NUM_CELLS_IN_LEFT_SWEEP] = (uint64_t)sweep_counts[0];
  stats.gt_profile[GT_NUM_VALID_CELLS_IN_QUERY] = (uint64_t)n_in_window;
  stats.gt_profile[GT_NUM_ATTR_CELLS_ACCESSED] = (uint64_t)n_in_window * (uint64_t)pl.nfields;
  stats.gt_profile[GT_NUM_PQ_FLUSHES_DUE_TO_OVERLAPPING_CELLS] = (uint64_t)sweep_counts[1];
  stats.gt_profile[GT_NUM_OPERATOR_INVOCATIONS] = (uint64_t)P;
  const int64_t T = T_a + T_b;
  stats.num_heavy_incidences = T;
  stats.num_cells_in_window = n_in_window;
  // ---- S6 incidences sorted by (record,row) --------------------------------------------------------------------------
  S.inc_keys.ensure(T + 1); S.inc_keys_sorted.ensure(T + 1); S.inc_vals.ensure(T + 1); S.inc_vals_sorted.ensure(T + 1);
  S.hbase.ensure(P + 2); S.lut_len.ensure(T + 2); S.i2m_off.ensure(T + 2); S.iflags.ensure(T + 1); S.gt_override.ensure((size_t)GDB_MAX_PLOIDY * T + 2);
  uint32_t lut_total = 0;
  if (T > 0) {
    STAGE("k_incidence_fill");
    hipLaunchKernelGGL(k_incidence_fill, dim3(blocks_for(CW)), dim3(kBlock), 0, st, fr, cm, S.hoff.p, c_base, CW, (int64_t)N, S.inc_keys.p, S.inc_vals.p);
    S.sort_pairs(S.inc_keys.p, S.inc_keys_sorted.p, S.inc_vals.p, S.inc_vals_sorted.p, (size_t)T, bits_for((uint64_t)P * (uint64_t)N));
    STAGE("k_lut_len");
    hipLaunchKernelGGL(k_lut_len, dim3(blocks_for(T)), dim3(kBlock), 0, st, S.inc_vals_sorted.p, S.cflags.p, T, S.lut_len.p);
    HIP_CHECK(hipMemsetAsync(S.lut_len.p + T, 0, sizeof(uint32_t), st));
    S.excl_scan(S.lut_len.p, S.i2m_off.p, (size_t)T + 1);   // i2m_off[T] = total, no host round trip
    if ((uint64_t)T * GDB_MAX_INPUT_ALLELES >= (1ull << 32)) throw GenomicsDBDeviceException("allele LUT storage exceeds 2^32 entries: split the query interval");
    lut_total = (uint32_t)((uint64_t)T * GDB_MAX_INPUT_ALLELES);   // capacity bound (a cell has at most GDB_MAX_INPUT_ALLELES alleles)
  } else {
    HIP_CHECK(hipMemsetAsync(S.i2m_off.p, 0, 2 * sizeof(uint32_t), st));
  }
  STAGE("k_heavy_base");
  hipLaunchKernelGGL(k_heavy_base, dim3(blocks_for(P + 1)), dim3(kBlock), 0, st, S.inc_keys_sorted.p, T, (int64_t)N, P, S.hbase.p);
  S.i2m.ensure((size_t)lut_total + 16);
  HeavyLists hl{S.hbase.p, S.inc_vals_sorted.p, S.i2m_off.p, S.i2m.p, S.iflags.p, S.gt_override.p};
  HIP_CHECK(hipEventRecord(ev[1], st));
  // ---- S7 site pass 0: allele merge, LUTs, prefix sizes -------------------------------------------------------------
  S.num_alleles.ensure(P); S.rflags.ensure(P); S.fmt_mask.ensure(P); S.prefix_len.ensure(P);
  SiteOut so{S.num_alleles.p, S.rflags.p, S.fmt_mask.p, S.prefix_len.p};
  QueryWindow qw;
  memset(&qw, 0, sizeof(qw));
  qw.qb = qb; qw.qe = qe;
  qw.contigs = S.contigs.p; qw.ncontigs = (int32_t)S.hp.contigs.size(); qw.contig_names = S.contig_names.p;
  qw.ref_bases = S.ref_len ? S.ref_bases.p : nullptr; qw.ref_begin = S.ref_begin; qw.ref_len = S.ref_len;
  NameTables nt{S.names_text.p, S.field_name_off.p, S.field_name_len.p, S.filter_name_off.p, S.filter_name_len.p, (int32_t)S.hp.filter_name_off.size(), S.filter_bcf_id.p};
  PresenceCounts pc{d_fmt, d_dp, d_nr, stride};
  MedianOrder med;
  memset(&med, 0, sizeof(med));
  for (int f = 0; f < GDB_MAX_FIELDS; ++f) med.slot[f] = -1;
  if (T > (int64_t)kSortedMedianThreshold * P || getenv("GDBAMD_SORTED_MEDIAN")) {
    std::vector<std::pair<int, int>> fields;   // (plan field, keep_spanning)
    for (int i = 0; i < pl.n_info; ++i) if (pl.field[pl.info_field[i]].combine_op == GDB_OP_MEDIAN && pl.field[pl.info_field[i]].length != GDB_VL_A) fields.push_back(std::make_pair(pl.info_field[i], 0));   // (A-length: element 0 of the REMAPPED vector, known only behind the allele merge - scalar_at)
    if (pl.qual_combine_op == GDB_OP_MEDIAN && pl.f_QUAL >= 0) fields.push_back(std::make_pair(pl.f_QUAL, 1));
    if (!fields.empty() && T > 0 && P < (1ll << 31)) {
      S.med_keys.ensure((size_t)T); S.med_idx.ensure((size_t)T);
      S.med_keys_sorted.ensure(fields.size() * (size_t)T); S.med_idx_sorted.ensure(fields.size() * (size_t)T);
      for (size_t s = 0; s < fields.size(); ++s) {
        hipLaunchKernelGGL(k_median_keys, dim3(blocks_for(T)), dim3(kBlock), 0, st, fr, pl, cm, rec, (const uint64_t*)S.inc_keys_sorted.p, (const int64_t*)S.inc_vals_sorted.p, T,
                           (int64_t)N, fields[s].first, fields[s].second, S.med_keys.p, S.med_idx.p);
        S.sort_pairs(S.med_keys.p, S.med_keys_sorted.p + s * (size_t)T, S.med_idx.p, S.med_idx_sorted.p + s * (size_t)T, (size_t)T, std::min(64, 33 + bits_for((uint64_t)P)));
        med.slot[fields[s].first] = (int8_t)s;
      }
      med.keys = S.med_keys_sorted.p; med.inc = S.med_idx_sorted.p; med.stride = T; med.enabled = 1;
    }
  }
  // records with very many variant calls get their medians from one workgroup each (no host round trip: the list is built
  // and consumed on the device; counters[2] = how many)
  BigMedians big;
  memset(&big, 0, sizeof(big));
  for (int f = 0; f < GDB_MAX_FIELDS; ++f) big.slot[f] = -1;
  if (!med.enabled && T > 0) {
    std::vector<std::pair<int, int>> fields;
    for (int i = 0; i < pl.n_info; ++i) if (pl.field[pl.info_field[i]].combine_op == GDB_OP_MEDIAN && pl.field[pl.info_field[i]].length != GDB_VL_A) fields.push_back(std::make_pair(pl.info_field[i], 0));   // (A-length: element 0 of the REMAPPED vector, known only behind the allele merge - scalar_at)
    if (pl.qual_combine_op == GDB_OP_MEDIAN && pl.f_QUAL >= 0) fields.push_back(std::make_pair(pl.f_QUAL, 1));
    if (!fields.empty()) {
      S.big_index.ensure((size_t)P); S.big_list.ensure(kMaxBigRecords); S.big_value.ensure(fields.size() * (size_t)kMaxBigRecords); S.big_ok.ensure(fields.size() * (size_t)kMaxBigRecords);
      hipLaunchKernelGGL(k_big_records, dim3(blocks_for(P)), dim3(kBlock), 0, st, (const int64_t*)S.hbase.p, P, S.big_index.p, S.big_list.p, S.counters.p + 2);
      for (size_t s = 0; s < fields.size(); ++s) {
        hipLaunchKernelGGL(k_big_medians, dim3(64), dim3(kBlock), 0, st, fr, pl, cm, rec, hl, (const int32_t*)S.big_list.p, (const int32_t*)(S.counters.p + 2), fields[s].first, (int)s,
                           fields[s].second, (int64_t)kMaxBigRecords, S.big_value.p, S.big_ok.p);
        big.slot[fields[s].first] = (int8_t)s;
      }
      big.index = S.big_index.p; big.value = S.big_value.p; big.ok = S.big_ok.p; big.stride = kMaxBigRecords; big.enabled = 1;
    }
  }
  // scratch of the tied-zero medians (gdb_core.hpp: TieScratch): 2 passes x float median fields x incidences
  TieScratch tie{nullptr, nullptr, 0};
  {
    int nf = (pl.qual_combine_op == GDB_OP_MEDIAN && pl.f_QUAL >= 0) ? 1 : 0;
    for (int i = 0; i < pl.n_info; ++i) if (pl.field[pl.info_field[i]].combine_op == GDB_OP_MEDIAN && pl.field[pl.info_field[i]].elem == GDB_ET_FLOAT) ++nf;
    if (nf && T > 0) {
      tie.capacity = 5ull * (uint64_t)nf * (uint64_t)T;       // (3 per value for a record whose workgroup selects, 1 when a thread does; both at most once per field)
      S.tie_buf.ensure((size_t)tie.capacity + 4); S.tie_used.ensure(1);
      HIP_CHECK(hipMemsetAsync(S.tie_used.p, 0, sizeof(unsigned long long), st));
      tie.buf = S.tie_buf.p; tie.used = S.tie_used.p;
    }
  }
  // values of the scalar reducer fields per incidence (gdb_core.hpp: ScalarPre)
  ScalarPre pre;
  memset(&pre, 0, sizeof(pre));
  for (int f = 0; f < GDB_MAX_FIELDS; ++f) pre.slot[f] = -1;
  if (T > 0) {
    std::vector<std::pair<int, int>> fields;   // (plan field, keep_spanning)
    for (int i = 0; i < pl.n_info; ++i) {
      const GdbFieldDesc& fd = pl.field[pl.info_field[i]];
      if ((fd.combine_op == GDB_OP_MEDIAN || fd.combine_op == GDB_OP_SUM || fd.combine_op == GDB_OP_MEAN) && (fd.elem == GDB_ET_FLOAT || fd.elem == GDB_ET_INT) && fd.length != GDB_VL_A)
        fields.push_back(std::make_pair(pl.info_field[i], 0));
    }
    if (pl.qual_combine_op != GDB_OP_UNKNOWN && pl.f_QUAL >= 0) fields.push_back(std::make_pair(pl.f_QUAL, 1));
    if (!fields.empty()) {
      S.scalar_pre.ensure(fields.size() * (size_t)T);
      for (size_t s = 0; s < fields.size(); ++s) {
        hipLaunchKernelGGL(k_scalar_values, dim3(blocks_for(T)), dim3(kBlock), 0, st, fr, pl, cm, rec, (const uint64_t*)S.inc_keys_sorted.p, (const int64_t*)S.inc_vals_sorted.p, T,
                           (int64_t)N, fields[s].first, fields[s].second, S.scalar_pre.p + s * (size_t)T);
        pre.slot[fields[s].first] = (int8_t)s;
      }
      pre.val = S.scalar_pre.p; pre.stride = T; pre.enabled = 1;
    }
  }
  // records with very many variant calls: their call walk is done by one workgroup each, before the per-thread site pass
  HugeSites huge;
  memset(&huge, 0, sizeof(huge));
  SiteCtx sx0{fr, pl, cm, rec, hl, pc, nt, qw, so, med, big, tie, pre, huge};
  // Which records get a workgroup of their own: the ones with more than kHugeRecord variant calls - and, when the interval's records
  // carry many calls on average (tens of thousands of samples: BASELINE configs[4] has ~100 calls on every record and a few
  // thousand records per piece, i.e. a few dozen wavefronts of one-thread-per-record walks of 100 calls each), every record with
  // more than 24.
  const int32_t huge_threshold = (P > 0 && T / P >= 24 && P <= (int64_t)kMaxHugeRecords) ? 24 : kHugeRecord;
  if (T > (int64_t)huge_threshold && !getenv("GDBAMD_NO_HUGE_SITES")) {
    S.huge_index.ensure((size_t)P); S.huge_list.ensure(kMaxHugeRecords); S.huge_out.ensure(kMaxHugeRecords); S.d_sx0.ensure(1);
    HIP_CHECK(hipMemcpyAsync(S.d_sx0.p, &sx0, sizeof(SiteCtx), hipMemcpyHostToDevice, st));
    STAGE("k_site_huge");
    hipLaunchKernelGGL(k_huge_records, dim3(blocks_for(P)), dim3(kBlock), 0, st, (const int64_t*)S.hbase.p, P, huge_threshold, S.huge_index.p, S.huge_list.p, S.counters.p + 3);
    hipLaunchKernelGGL(k_site_huge, dim3(1024), dim3(kHugeBlock), 0, st, (const SiteCtx*)S.d_sx0.p, (const int32_t*)S.huge_list.p, (const int32_t*)(S.counters.p + 3), S.huge_out.p, S.err.p);
    huge.index = S.huge_index.p; huge.out = S.huge_out.p; huge.enabled = 1;
  }
  SiteCtx sx{fr, pl, cm, rec, hl, pc, nt, qw, so, med, big, tie, pre, huge};
  S.d_sx.ensure(1);
  HIP_CHECK(hipMemcpyAsync(S.d_sx.p, &sx, sizeof(SiteCtx), hipMemcpyHostToDevice, st));   // (sx outlives the copy: the function synchronises before it returns)
  STAGE("k_site_size");
  S.site_staging.ensure((size_t)P * kSiteStride + 64);
  S.spill_buf.ensure((size_t)P * kSpillTail); S.spill_chunk.ensure((size_t)P);
  const int32_t* site_order = nullptr;
  {
    static const bool sorted_sites = !(getenv("GDBAMD_SITE_ORDER") && atoi(getenv("GDBAMD_SITE_ORDER")) == 0);
    if (sorted_sites && P >= 4096) {
      S.site_key.ensure((size_t)P); S.site_key_sorted.ensure((size_t)P); S.site_ord_in.ensure((size_t)P); S.site_ord.ensure((size_t)P);
      hipLaunchKernelGGL(k_site_order_keys, dim3(blocks_for(P)), dim3(kBlock), 0, st, (const int64_t*)S.hbase.p, P, S.site_key.p, S.site_ord_in.p);
      S.sort_pairs(S.site_key.p, S.site_key_sorted.p, S.site_ord_in.p, S.site_ord.p, (size_t)P, 8);
      site_order = S.site_ord.p;
    }
  }
  hipLaunchKernelGGL(k_site_size, dim3(blocks_for(P, 64)), dim3(64), 0, st, S.d_sx.p, site_order, S.site_staging.p, SpillPool{S.spill_buf.p}, S.spill_chunk.p, S.err.p);
  S.remap_total.ensure(1);
  HIP_CHECK(hipMemsetAsync(S.remap_total.p, 0, sizeof(unsigned long long), st));
  for (int i = 0; i < pl.n_format; ++i)
    if (pl.format_field[i] == pl.f_PL && pl.f_PL >= 0)
      hipLaunchKernelGGL(k_remap_elements, dim3(blocks_for(P)), dim3(kBlock), 0, st, so, (const int32_t*)(d_fmt + (size_t)i * stride), P, i, S.remap_total.p);
  HIP_CHECK(hipEventRecord(ev[2], st));
  // ---- S8 sample-column sizes + offsets ---------------------------------------------------------------------------
  const int nchunks = (N + kAsmRows - 1) / kAsmRows;
  const size_t nchunk_total = (size_t)P * nchunks;
  S.chunk_size.ensure(nchunk_total + 1); S.chunk_off.ensure(nchunk_total + 2); S.rec_off.ensure(P + 2);
  RowIndex ri{S.row_ptr.p, S.perm.p, S.rm_begin.p};
  EntryCtx ex{fr, pl, cm, hl};
  // every pipeline owns one element of c_ex (several handles per process is the reference's rule: they no longer serialise)
  HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_ex), &ex, sizeof(EntryCtx), (size_t)S.ctx_slot * sizeof(EntryCtx), hipMemcpyHostToDevice, st));
  // ---- S8a entry text table: record types, slots of (plain cell, type) / (record, heavy call) / no-call, text pool ------------
  if (T >= (1ll << 32)) throw GenomicsDBDeviceException("more than 2^32 (record, variant call) incidences in one interval: split the query interval");
  S.type_hkeys.ensure(kTypeHash); S.type_hrep.ensure(kTypeHash); S.type_hid.ensure(kTypeHash); S.type_rep.ensure(kMaxTypes); S.rtype.ensure(P);
  HIP_CHECK(hipMemsetAsync(S.type_hkeys.p, 0xFF, kTypeHash * sizeof(unsigned long long), st));
  HIP_CHECK(hipMemsetAsync(S.type_hrep.p, 0x7F, kTypeHash * sizeof(int32_t), st));
  HIP_CHECK(hipMemsetAsync(S.type_hid.p, 0xFF, kTypeHash, st));
  HIP_CHECK(hipMemsetAsync(S.type_rep.p, 0, kMaxTypes * sizeof(int32_t), st));
  STAGE("k_type_insert");
  hipLaunchKernelGGL(k_type_insert, dim3(blocks_for(P)), dim3(kBlock), 0, st, so, P, S.type_hkeys.p, S.type_hrep.p);
  hipLaunchKernelGGL(k_type_assign, dim3(1), dim3(64), 0, st, S.type_hkeys.p, S.type_hrep.p, max_tabled_types(), S.type_hid.p, S.type_rep.p, S.counters.p + 1);
  S.untabled.ensure(P + 1); S.ubase.ensure(P + 1); S.urec.ensure(P + 1);
  hipLaunchKernelGGL(k_type_lookup, dim3(blocks_for(P)), dim3(kBlock), 0, st, so, P, S.type_hkeys.p, S.type_hid.p, S.rtype.p, S.untabled.p);
  S.excl_scan(S.untabled.p, S.ubase.p, (size_t)P);
  S.tmask.ensure(CW); S.nslots.ensure(CW + 1); S.tbase.ensure(CW + 1);
  STAGE("k_cell_types");
  {
    const size_t nocc = (size_t)kMaxTypes * (size_t)(P + 1);
    S.type_occ.ensure(nocc);
    HIP_CHECK(hipMemsetAsync(S.type_occ.p, 0, nocc * sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_type_onehot, dim3(blocks_for(P)), dim3(kBlock), 0, st, (const uint8_t*)S.rtype.p, P, S.type_occ.p);
    S.excl_scan(S.type_occ.p, S.type_occ.p, nocc);
  }
  hipLaunchKernelGGL(k_cell_types, dim3(blocks_for(CW)), dim3(kBlock), 0, st, S.cflags.p, S.k_lo.p, S.k_hi.p, (const uint32_t*)S.type_occ.p, P,
                     (const int32_t*)(S.counters.p + 1), c_base, CW, S.tmask.p, S.nslots.p);
  S.excl_scan(S.nslots.p, S.tbase.p, (size_t)CW);
  uint32_t ur_a = 0, ur_b = 0, sl_a = 0, sl_b = 0;
  int32_t ntypes = 0;
  S.read_back_many({{&ur_a, S.ubase.p + (P - 1), 4}, {&ur_b, S.untabled.p + (P - 1), 4}, {&sl_a, S.tbase.p + (CW - 1), 4}, {&sl_b, S.nslots.p + (CW - 1), 4},
                    {&ntypes, S.counters.p + 1, 4}});
  const int64_t UR = (int64_t)ur_a + ur_b;
  const uint64_t SL = (uint64_t)sl_a + sl_b;
  if (UR > 0) hipLaunchKernelGGL(k_untabled_records, dim3(blocks_for(P)), dim3(kBlock), 0, st, S.untabled.p, S.ubase.p, P, S.urec.p);
  const uint64_t NS = (uint64_t)kMaxTypes + SL + (uint64_t)T + (uint64_t)UR * (uint64_t)N;
  if (NS >= (1ull << 32)) throw GenomicsDBDeviceException("entry text table exceeds 2^32 slots: split the query interval");
  S.slot_len.ensure(NS + 1); S.slot_units.ensure(NS + 1); S.slot_off.ensure(NS + 1); S.slot_desc.ensure(NS + 1);
  if (NS * (uint64_t)(kSlotStride / 16) >= (uint64_t)kOverflowBit) throw GenomicsDBDeviceException("entry text table exceeds 32 GiB: split the query interval");
  S.pool.ensure((size_t)NS * kSlotStride + 64);
  // Long texts (more than an inline slot): pass 0 places the ones that fit its LDS strip itself, out of an overflow pool that is
  // sized from what earlier intervals of this pipeline needed (grow-only).  The wide strip (256 bytes per lane, half the resident
  // wavefronts) is used once an interval has shown long texts; GDBAMD_SLOT_STRIP = 128 / 256 forces one.
  S.slot_bump.ensure(kBumpShards * 4);
  HIP_CHECK(hipMemsetAsync(S.slot_bump.p, 0, kBumpShards * 4 * sizeof(unsigned int), st));
  S.slot_over.ensure(kBumpShards);
  HIP_CHECK(hipMemsetAsync(S.slot_over.p, 0, kBumpShards * sizeof(unsigned int), st));
  bool wide = S.long_texts_seen;
  if (const char* e = getenv("GDBAMD_SLOT_STRIP")) wide = atoi(e) >= 256;
  // the overflow pool in kBumpShards equal parts; a part holds what its shard needed at most so far, and an eighth more
  auto shard_units_of = [&](size_t cap_bytes) { return (uint32_t)std::min<uint64_t>((uint64_t)(cap_bytes > 64 ? (cap_bytes - 64) / 16 : 0) / kBumpShards, ((uint64_t)kOverflowBit - 1) / kBumpShards); };
  if (wide) S.pool_ovf.ensure(std::max<size_t>((size_t)kBumpShards * ((size_t)S.pool_ovf_need * 16 + ((size_t)S.pool_ovf_need * 16 >> 3)), (size_t)1 << 20) + 64);
  SlotTable stt{S.slot_len.p, S.slot_off.p, S.pool.p, S.pool_ovf.p, (uint32_t)kMaxTypes, (uint32_t)(kMaxTypes + SL), (uint32_t)(kMaxTypes + SL + (uint64_t)T), S.slot_off.p, S.slot_bump.p,
                wide ? shard_units_of(S.pool_ovf.cap) : 0u, (int32_t)S.ctx_slot, pl.bcf_mode, pl.bcf_mode ? (uint32_t)kSlotStride : (uint32_t)kInlineText,
                (const int64_t*)S.hoff.p, (const int32_t*)S.k_lo.p, c_base, S.slot_over.p};
  STAGE("k_slots<0>");
#define GDB_SLOT_KERNELS(PASSN, W) do { \
  hipLaunchKernelGGL((k_slots_nocall<PASSN, W>), dim3(1), dim3(kMaxTypes), 0, st, stt, so, S.type_rep.p, ntypes, S.err.p); \
  if (SL > 0) hipLaunchKernelGGL((k_slots_light<PASSN, W>), dim3(blocks_for((int64_t)SL, light_block<W>())), dim3(light_block<W>()), 0, st, stt, so, S.type_rep.p, S.tmask.p, S.tbase.p, (const uint32_t*)S.slot_cell.p, c_base, (int64_t)SL, slot_regroup() ? 1 : 0, S.err.p); \
  if (T > 0) hipLaunchKernelGGL((k_slots_heavy<PASSN, W>), dim3(blocks_for(T, 64)), dim3(64), 0, st, stt, so, S.inc_keys_sorted.p, S.inc_vals_sorted.p, T, (int64_t)N, S.err.p); \
  if (UR > 0) hipLaunchKernelGGL((k_slots_untabled<PASSN, W>), dim3(blocks_for(UR * N, 64)), dim3(64), 0, st, stt, so, ri, rec, S.urec.p, UR, N, S.err.p); \
} while (0)
  if (SL > 0) {
    S.slot_cell.ensure(SL + 1);
    hipLaunchKernelGGL(k_slot_cells, dim3(blocks_for(CW)), dim3(kBlock), 0, st, (const uint32_t*)S.tbase.p, (const uint32_t*)S.nslots.p, CW, S.slot_cell.p);
  }
  if (wide) GDB_SLOT_KERNELS(0, kStripWordsWide); else GDB_SLOT_KERNELS(0, kStripWords);
  unsigned int bump_sh[kBumpShards * 4], over_sh[kBumpShards];
  S.read_back_many({{bump_sh, S.slot_bump.p, sizeof(bump_sh)}, {over_sh, S.slot_over.p, sizeof(over_sh)}});
  bool any_over255 = false;
  for (int sh = 0; sh < kBumpShards; ++sh) any_over255 |= over_sh[sh] != 0;
  uint64_t left_texts = 0, long_texts = 0, shard_need = 0, total_units = 0;   // (per shard: units taken - also the ones that ran past its part - plus units still wanted)
  for (int sh = 0; sh < kBumpShards; ++sh) {
    left_texts += bump_sh[sh * 4 + 1]; long_texts += bump_sh[sh * 4 + 3];
    shard_need = std::max<uint64_t>(shard_need, (uint64_t)bump_sh[sh * 4] + bump_sh[sh * 4 + 2]);
    total_units += (uint64_t)bump_sh[sh * 4] + bump_sh[sh * 4 + 2];
  }
  uint64_t pool_units = total_units;
  if (shard_need * kBumpShards >= (uint64_t)kOverflowBit) throw GenomicsDBDeviceException("entry text overflow pool exceeds 32 GiB: split the query interval");
  constexpr uint64_t kBumpLimit = 1u << 21;          // more texts than this left for pass 1: their places come from a scan (one atomic each would queue up)
  if (left_texts > 0 && left_texts <= kBumpLimit) {
    // room in every shard's part for the texts pass 0 left, behind what it has placed there (kept when the pool has to grow)
    const uint32_t old_su = stt.bump_cap;
    // (pass 0 with the narrow strip places nothing - old_su == 0 -: the pool of an earlier interval is taken as it is when it is large enough;
    //  a hipFree would synchronise the whole device and stall the other pipeline's overlapped staging once per interval)
    const bool reuse = old_su == 0 && (uint64_t)shard_units_of(S.pool_ovf.cap) >= shard_need;
    if ((uint64_t)old_su < shard_need && !reuse) {
      const uint64_t su = shard_need + (shard_need >> 3) + 4;
      const size_t ncap = (size_t)su * 16 * kBumpShards + 64;
      char* np = nullptr;
      HIP_CHECK(hipMalloc((void**)&np, ncap));
      for (int sh = 0; sh < kBumpShards && old_su; ++sh) {
        const size_t keep = (size_t)std::min<uint64_t>(bump_sh[sh * 4], old_su) * 16;      // (a counter runs past its part when a request did not fit)
        if (keep) HIP_CHECK(hipMemcpyAsync(np + (size_t)sh * su * 16, S.pool_ovf.p + (size_t)sh * old_su * 16, keep, hipMemcpyDeviceToDevice, st));
      }
      if (old_su) {   // the texts pass 0 placed move with their shard: their descriptors follow
        hipLaunchKernelGGL(k_slot_rebase, dim3(blocks_for((int64_t)NS)), dim3(kBlock), 0, st, (const uint32_t*)S.slot_len.p, S.slot_off.p, (int64_t)NS, old_su, (uint32_t)su, stt.inline_max);
      }
      HIP_CHECK(hipStreamSynchronize(st));
      if (S.pool_ovf.p) (void)hipFree(S.pool_ovf.p);
      S.pool_ovf.p = np; S.pool_ovf.cap = ncap;
    }
    stt.pool_ovf = S.pool_ovf.p;
    stt.bump_cap = shard_units_of(S.pool_ovf.cap);
    STAGE("k_slots<1>");
    GDB_SLOT_KERNELS(1, kStripWords);
  } else if (left_texts > 0) {
    hipLaunchKernelGGL(k_slot_units, dim3(blocks_for((int64_t)NS)), dim3(kBlock), 0, st, S.slot_len.p, (int64_t)NS, S.slot_units.p, stt.inline_max);
    S.excl_scan(S.slot_units.p, S.slot_off.p, (size_t)NS);
    pool_units = (uint64_t)S.read_back_sum(S.slot_off.p + (NS - 1), S.slot_units.p + (NS - 1));
    if (pool_units >= (uint64_t)kOverflowBit) throw GenomicsDBDeviceException("entry text overflow pool exceeds 32 GiB: split the query interval");
    S.pool_ovf.ensure((size_t)pool_units * 16 + 64);
    stt.pool_ovf = S.pool_ovf.p;
    stt.bump_cap = 0;
    STAGE("k_slots<1>");
    GDB_SLOT_KERNELS(1, kStripWords);   // every text longer than an inline slot is formatted again, into its scanned place
  }
#undef GDB_SLOT_KERNELS
  S.long_texts_seen = long_texts * 16 > NS;    // the wide strip of pass 0 pays when a sixteenth of the texts is long
  S.pool_ovf_need = std::max<uint64_t>(S.pool_ovf_need, shard_need);   // (units per shard)
  int asm_path = events_enabled() ? 0 : (pl.bcf_mode && (asm_path_wanted() & 1)) ? 2 : asm_path_wanted();
  if (asm_path == 3 && UR > 0) asm_path = 0;      // (records of untabled types have a slot per (record, sample): the piece lists do not carry them)
  const bool piece_path = asm_path == 1 || asm_path == 3;   // matrix-free page assembly
  hipLaunchKernelGGL(k_slot_desc, dim3(blocks_for((int64_t)NS)), dim3(kBlock), 0, st, S.slot_len.p, S.slot_off.p, (int64_t)NS, asm_path == 1 ? (uint2*)nullptr : S.slot_desc.p,
                     pl.bcf_mode ? (char*)nullptr : S.pool.p, stt.inline_max);
  stats.num_record_types = ntypes;
  stats.resolved_entry_bytes = (asm_path == 1 || asm_path == 3 || events_enabled()) ? 0 : ((asm_path == 0 && !any_over255 && res_compact_wanted()) ? 5 : 8);
  stats.num_text_slots = (int64_t)NS;
  stats.text_pool_bytes = (int64_t)(NS * kSlotStride + pool_units * 16);
  STAGE("k_walk_window");
  if (++S.walk_epoch >= (1u << 24)) {   // the stamp wrapped: wipe the old ones
    hipLaunchKernelGGL(k_walk_static, dim3(blocks_for(C + N)), dim3(kBlock), 0, st, S.perm.p, fr.row, S.row_ptr.p, N, S.eff_end.p, S.cflags.p, C, S.walk.p, S.walk_inv.p);
    S.walk_epoch = 1;
  }
  hipLaunchKernelGGL(k_walk_window, dim3(blocks_for(CW)), dim3(kBlock), 0, st, S.walk_inv.p, S.cflags.p, S.hoff.p, S.k_lo.p, S.k_hi.p, S.tbase.p, S.tmask.p,
                     (const uint32_t*)S.slot_len.p, (uint32_t)kMaxTypes, (uint32_t)(kMaxTypes + SL), c_base, CW, S.walk_epoch, S.walk.p);
  AsmCtx ac{S.row_ptr.p, S.rm_begin.p, S.walk.p, S.rstart.p, S.rtype.p, S.prefix_len.p, S.slot_desc.p, S.pool.p, S.ubase.p,
            (uint32_t)kMaxTypes, (uint32_t)(kMaxTypes + SL), (uint32_t)(kMaxTypes + SL + (uint64_t)T), N};
  S.jhint.ensure((size_t)((P + kHintStep - 1) / kHintStep) * (size_t)N + 1);
  PieceCtx pc2{S.row_ptr.p, S.rm_begin.p, S.walk.p, S.rstart.p, S.rtype.p, S.ubase.p, S.type_occ.p, S.slot_len.p, S.prefix_len.p, S.jhint.p,
               (uint32_t)kMaxTypes, (uint32_t)(kMaxTypes + SL), (uint32_t)(kMaxTypes + SL + (uint64_t)T), S.walk_epoch, N, P};
  // ---- S8b sample-column sizes + offsets ---------------------------------------------------------------------------
  const int run = size_run_length();
  const unsigned run_blocks = (unsigned)((P + run - 1) / run);
  STAGE("k_assemble_size");
  if (!piece_path) S.order_by_type(0, P);
  const uint64_t resolved_bytes = (uint64_t)P * nchunks * kAsmRows * sizeof(uint2);
  bool use_events = false;
  EventBuf ebuf{nullptr, nullptr, nullptr, 0};
  const bool resolved_whole = !piece_path && (pl.bcf_mode || (resolved_bytes <= resolved_budget_bytes() && !(events_enabled() && ((1 << order_block_log2()) % event_run_length()) == 0)));
  // compact layout: the default kernels (text and BCF2), no entry longer than a byte can say
  const bool res_compact = asm_path == 0 && !events_enabled() && !any_over255 && res_compact_wanted() && (!pl.bcf_mode || resolved_whole);
  if (resolved_whole) {
    if (res_compact) { S.res_off.ensure((size_t)P * nchunks * kAsmRows); S.res_len8.ensure((size_t)P * nchunks * kAsmRows); }
    else S.resolved.ensure((size_t)P * nchunks * kAsmRows);
  }
  auto res_view = [&S, res_compact]() { return res_compact ? ResMatrix{nullptr, S.res_off.p, S.res_len8.p} : ResMatrix{S.resolved.p, nullptr, nullptr}; };
  S.max_record.ensure(1);
  HIP_CHECK(hipMemsetAsync(S.max_record.p, 0, sizeof(unsigned long long), st));
  BcfLayout lay{nullptr, nullptr, nullptr, nullptr};
  const int bcf_F = std::max(1, pl.n_format);
  const uint64_t size2_units = (uint64_t)((P + kSizeBlock - 1) / kSizeBlock) * (uint64_t)nchunks;
  if (asm_path != 0 && size2_units >= (1ull << 31)) throw GenomicsDBDeviceException("more than 2^31 (record block, sample chunk) units in one interval: split the query interval");
  const int frun = fill_run_length();
  const unsigned fill_units = (unsigned)(((P + frun - 1) / frun) * nchunks);
  if (pl.bcf_mode) {
    // BCF2: resolve the (record, sample) matrix, reduce the entries' summaries to vector length + type per (record, field), size the records
    if (asm_path == 2) {   // (k_size2 for the walkers' starting points; its text sizes mean nothing here)
      hipLaunchKernelGGL(k_size2, dim3((unsigned)size2_units), dim3(kAsmRows), 0, st, pc2, nchunks, S.chunk_size.p);
      hipLaunchKernelGGL(k_fill2, dim3(fill_units), dim3(kAsmRows), 0, st, pc2, (const uint2*)S.slot_desc.p, (const int32_t*)S.order.p, P, nchunks, frun, S.resolved.p, (int64_t)0);
    } else
    launch_assemble_size(S, st, dim3(run_blocks * (unsigned)nchunks), ac, S.order.p, P, N, nchunks, run, (uint64_t*)nullptr, 0, res_view(), (int64_t)0, (int64_t)0);
    S.bcf_part.ensure(nchunk_total * (size_t)bcf_F + 1); S.bcf_fmeta.ensure((size_t)P * bcf_F + 1); S.bcf_foff.ensure((size_t)P * bcf_F + 1); S.bcf_lindiv.ensure((size_t)P + 1);
    S.bcf_rec_size.ensure((size_t)P + 2);
    lay = BcfLayout{S.bcf_fmeta.p, S.bcf_foff.p, S.bcf_lindiv.p, S.bcf_rec_size.p};
    STAGE("k_bcf_field_meta");
    hipLaunchKernelGGL(k_bcf_field_meta, dim3((unsigned)(((P + kBcfRun - 1) / kBcfRun) * nchunks)), dim3(kAsmRows), 0, st, res_view(), (const char*)S.pool.p,
                       (const char*)S.pool_ovf.p, (const uint32_t*)S.fmt_mask.p, (const int32_t*)S.order.p, P, nchunks, bcf_F, S.bcf_part.p);
    hipLaunchKernelGGL(k_bcf_layout, dim3(blocks_for(P)), dim3(kBlock), 0, st, pl, (const uint32_t*)S.bcf_part.p, (const uint32_t*)S.fmt_mask.p, (const uint32_t*)S.prefix_len.p, P,
                       nchunks, bcf_F, lay);
    HIP_CHECK(hipMemsetAsync(S.bcf_rec_size.p + P, 0, sizeof(uint64_t), st));
    S.excl_scan(S.bcf_rec_size.p, S.rec_off.p, (size_t)P + 1);
    hipLaunchKernelGGL(k_bcf_max_record, dim3(blocks_for(P)), dim3(kBlock), 0, st, (const uint64_t*)S.bcf_rec_size.p, P, S.max_record.p);
  } else {
  const int evrun = event_run_length();
  const uint64_t ev_blocks = (uint64_t)((P + evrun - 1) / evrun) * (uint64_t)nchunks;
  use_events = events_enabled() && ev_blocks * (uint64_t)evrun * kAsmRows * sizeof(uint2) <= resolved_budget_bytes() && ((1 << order_block_log2()) % evrun) == 0;
  if (use_events) {
    // sizing pass that leaves the change list of every (run, chunk) instead of the dense matrix
    S.ev_init.ensure((size_t)ev_blocks * kAsmRows); S.ev_buf.ensure((size_t)ev_blocks * evrun * kAsmRows); S.ev_count.ensure((size_t)ev_blocks);
    ebuf = EventBuf{S.ev_init.p, S.ev_buf.p, S.ev_count.p, evrun};
    hipLaunchKernelGGL(k_assemble_size_ev, dim3((unsigned)ev_blocks), dim3(kAsmRows), 0, st, ac, S.order.p, P, N, nchunks, evrun, S.chunk_size.p, ebuf, S.err.p);
    S.order_iv.ensure((size_t)P);
    HIP_CHECK(hipMemcpyAsync(S.order_iv.p, S.order.p, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    S.order_iv_valid = true;
  } else if (asm_path != 0) {
    hipLaunchKernelGGL(k_size2, dim3((unsigned)size2_units), dim3(kAsmRows), 0, st, pc2, nchunks, S.chunk_size.p);
    if (asm_path == 3) {   // piece lists: count, offsets, fill
      const size_t nlists = (size_t)((P + kSizeBlock - 1) / kSizeBlock) * (size_t)ntypes * (size_t)N;
      S.pl_cnt.ensure(nlists + 1); S.pl_ofs.ensure(nlists + 1);
      hipLaunchKernelGGL(k_plist<false>, dim3((unsigned)size2_units), dim3(kAsmRows), 0, st, pc2, (const uint2*)S.slot_desc.p, ntypes, nchunks, S.pl_cnt.p, (PieceEntry*)nullptr);
      S.excl_scan(S.pl_cnt.p, S.pl_ofs.p, nlists);
      const int64_t nent = S.read_back_sum(S.pl_ofs.p + (nlists - 1), S.pl_cnt.p + (nlists - 1));
      if (nent >= (1ll << 32)) throw GenomicsDBDeviceException("more than 2^32 piece-list entries in one interval: split the query interval");
      S.plist.ensure((size_t)nent + 4);
      hipLaunchKernelGGL(k_plist<true>, dim3((unsigned)size2_units), dim3(kAsmRows), 0, st, pc2, (const uint2*)S.slot_desc.p, ntypes, nchunks, S.pl_ofs.p, S.plist.p);
      S.iv.NT = ntypes;
    }
    if (asm_path == 2 && resolved_whole)
      hipLaunchKernelGGL(k_fill2, dim3(fill_units), dim3(kAsmRows), 0, st, pc2, (const uint2*)S.slot_desc.p, (const int32_t*)S.order.p, P, nchunks, frun, S.resolved.p, (int64_t)0);
  } else
  launch_assemble_size(S, st, dim3(run_blocks * (unsigned)nchunks), ac, S.order.p, P, N, nchunks, run, S.chunk_size.p, nchunk_total, resolved_whole ? res_view() : ResMatrix{nullptr, nullptr, nullptr}, (int64_t)0, res_chunk_major() ? P : (int64_t)0);
  HIP_CHECK(hipMemsetAsync(S.chunk_size.p + nchunk_total, 0, sizeof(uint64_t), st));
  S.excl_scan(S.chunk_size.p, S.chunk_off.p, nchunk_total + 1);
  STAGE("k_gather_record_offsets");
  hipLaunchKernelGGL(k_gather_record_offsets, dim3(blocks_for(P + 1)), dim3(kBlock), 0, st, S.chunk_off.p, nchunks, P, S.rec_off.p, S.max_record.p);
  }
  S.iv.rec_off.clear();             // fetched by next_page only when the interval does not fit one page
  uint64_t totals[2] = {0, 0};      // bytes of the interval, bytes of its largest record
  STAGE("before-sync-offsets");
  HIP_CHECK(hipEventRecord(ev[3], st));
  uint32_t eb = 0;
  uint64_t remap_elems = 0;
  S.read_back_many({{&totals[0], S.rec_off.p + P, sizeof(uint64_t)}, {&totals[1], S.max_record.p, sizeof(uint64_t)}, {&eb, S.err.p, sizeof(uint32_t)},
                    {&remap_elems, S.remap_total.p, sizeof(uint64_t)}});
  stats.bytes_out = totals[0];
  stats.num_remap_elements = remap_elems;
  HIP_CHECK(hipEventElapsedTime(&stats.ms_sweep, ev[0], ev[1]));
  HIP_CHECK(hipEventElapsedTime(&stats.ms_site, ev[1], ev[2]));
  HIP_CHECK(hipEventElapsedTime(&stats.ms_size, ev[2], ev[3]));
  if (eb) throw GenomicsDBDeviceException(err_bits_text(eb));
  S.iv.max_record_bytes = totals[1]; S.iv.total_bytes = totals[0];
  S.iv.P = P; S.iv.nchunks = nchunks; S.iv.kp = 0;
  S.iv.sx = sx; S.iv.ex = ex; S.iv.ri = ri; S.iv.so = so; S.iv.rec = rec; S.iv.ac = ac; S.iv.pc2 = pc2; S.iv.piece_path = piece_path; S.iv.asm_path = asm_path; S.iv.resolved_whole = resolved_whole; S.iv.res_compact = res_compact; S.iv.P_rows = P;
  S.iv.bcf = pl.bcf_mode != 0; S.iv.bcf_F = bcf_F; S.iv.lay = lay;
  S.iv.events = use_events; S.iv.evrun = ebuf.run; S.iv.eb = ebuf;
  S.iv.active = true;
}

// Page production in two steps so that a consumer can drain page p (copy engine, its own stream) while page p + 1 is being
// assembled: begin_page() only enqueues the kernels of the next <= arena_bytes of whole records into arena `arena_idx`;
// finish_page() waits for them and raises their error bits.  Before it overwrites an arena the compute stream waits for the
// event the consumer registered with set_arena_release_event() (its last read of that arena).
bool DevicePipeline::begin_page(uint64_t arena_bytes, int arena_idx, PageTicket* ticket) {
  Impl& S = *m_;
  Impl::IntervalState& iv = S.iv;
  if (!iv.active || iv.kp >= iv.P) { iv.active = false; return false; }
  HIP_CHECK(hipSetDevice(S.device));
  hipStream_t st = S.stream;
  const int ai = arena_idx & 1;
  const int32_t N = S.hp.plan.num_query_rows;
  const uint64_t arena_cap = std::max<uint64_t>(arena_bytes, iv.max_record_bytes);
  const int64_t kp = iv.kp, P = iv.P;
  int64_t ke = P;
  uint64_t page_base = 0, page_bytes = iv.stats.bytes_out;
  if (kp > 0 || iv.stats.bytes_out > arena_cap) {   // paging: the record offsets are needed on the host
    std::vector<uint64_t>& rec_off = iv.rec_off;
    if (rec_off.empty()) {
      rec_off.resize((size_t)P + 1);
      HIP_CHECK(hipMemcpyAsync(rec_off.data(), S.rec_off.p, (size_t)(P + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
    }
    int64_t lo = kp + 1, hi = P;  // largest k_end with rec_off[k_end] - rec_off[kp] <= arena_cap
    while (lo < hi) { int64_t mid = (lo + hi + 1) >> 1; if (rec_off[(size_t)mid] - rec_off[(size_t)kp] <= arena_cap) lo = mid; else hi = mid - 1; }
    ke = lo;
    // the change list is laid out over the interval's order blocks: a page that ends on a block boundary can use it as it is
    if (iv.events && !iv.bcf) {
      const int64_t blk = (int64_t)1 << order_block_log2();
      if (ke < P && (kp % blk) == 0 && (ke / blk) * blk > kp) ke = (ke / blk) * blk;
    }
    page_base = rec_off[(size_t)kp];
    page_bytes = rec_off[(size_t)ke] - page_base;
  }
  if (S.arena_release[ai]) { HIP_CHECK(hipStreamWaitEvent(st, S.arena_release[ai], 0)); S.arena_release[ai] = nullptr; }
  if (S.arena[ai].cap < (S.hp.bgzf ? bgzf_bound(page_bytes) : page_bytes) + 64) {   // (re)allocation frees memory a copy may still read: the stream has to be idle
    HIP_CHECK(hipStreamSynchronize(st));
    const uint64_t want = std::min<uint64_t>(arena_cap, iv.stats.bytes_out);
    S.arena[ai].ensure((S.hp.bgzf ? bgzf_bound(want) : want) + 64);      // (a page of incompressible bytes grows by the block framing)
  }
  char* const arena = S.arena[ai].p;
  const int64_t np = ke - kp;
  hipEvent_t* w = S.ev_page[ai];
  HIP_CHECK(hipEventRecord(w[0], st));
  // the stream of the page kernel: the compute stream, or (several windows in flight) a high-priority one forked from it and joined behind the kernel
  const auto fork_page_stream = [&S, st]() -> hipStream_t {
    if (!(page_priority_env() >= 0 ? page_priority_env() == 1 : S.page_priority)) return st;
    if (!S.stream_page) {
      int least = 0, greatest = 0;
      HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
      HIP_CHECK(hipStreamCreateWithPriority(&S.stream_page, hipStreamNonBlocking, greatest));
      HIP_CHECK(hipEventCreateWithFlags(&S.ev_page_fork, hipEventDisableTiming));
      HIP_CHECK(hipEventCreateWithFlags(&S.ev_page_join, hipEventDisableTiming));
    }
    HIP_CHECK(hipEventRecord(S.ev_page_fork, st));
    HIP_CHECK(hipStreamWaitEvent(S.stream_page, S.ev_page_fork, 0));
    return S.stream_page;
  };
  const auto join_page_stream = [&S, st](hipStream_t st_w) { if (st_w != st) { HIP_CHECK(hipEventRecord(S.ev_page_join, st_w)); HIP_CHECK(hipStreamWaitEvent(st, S.ev_page_join, 0)); } };
  if (iv.bcf) {
    STAGE("k_bcf_shared");
    hipLaunchKernelGGL(k_bcf_shared, dim3(blocks_for(np, 64)), dim3(64), 0, st, S.d_sx.p, (const char*)S.site_staging.p, (const char*)S.spill_buf.p, (const int32_t*)S.spill_chunk.p, kp, ke,
                       (const uint64_t*)S.rec_off.p, page_base, iv.lay, iv.bcf_F, arena, S.err.p);
    STAGE("k_bcf_write");
    S.order_by_type(kp, np);
    S.bcf_same.ensure((size_t)np + 1);
    hipLaunchKernelGGL(k_bcf_same_layout, dim3(blocks_for(np)), dim3(kBlock), 0, st, (const int32_t*)S.order.p, (const uint32_t*)S.fmt_mask.p, (const uint32_t*)iv.lay.fmeta, np, iv.bcf_F, S.bcf_same.p);
    const hipStream_t st_w = fork_page_stream();
    HIP_CHECK(hipEventRecord(w[1], st_w));   // [w1, w2]: the values kernel alone
    if (S.hp.plan.n_format > 0 && !S.hp.plan.sites_only_query)
      hipLaunchKernelGGL(k_bcf_write, dim3((unsigned)(((np + kBcfRun - 1) / kBcfRun) * iv.nchunks)), dim3(kAsmRows), 0, st_w, S.hp.plan, iv.res_compact ? ResMatrix{nullptr, S.res_off.p, S.res_len8.p} : ResMatrix{S.resolved.p, nullptr, nullptr}, (const char*)S.pool.p,
                         (const char*)S.pool_ovf.p, (const uint32_t*)S.fmt_mask.p, (const int32_t*)S.order.p, (const uint8_t*)S.bcf_same.p, np, iv.nchunks, iv.bcf_F, N, iv.lay, (const uint64_t*)S.rec_off.p, page_base, arena BCF_PROF_ARG);
#ifdef GDB_BCF_PROF
    { unsigned long long h[12]; HIP_CHECK(hipStreamSynchronize(st_w)); HIP_CHECK(hipMemcpy(h, bcf_prof_buffer(), sizeof h, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemset(bcf_prof_buffer(), 0, sizeof h));
      fprintf(stderr, "k_bcf_write cycles per step (100 MHz ticks x wavefronts / steps): top %.1f layout %.1f lanes-pass %.1f groups %.1f flush %.1f slow %.1f; steps %llu, with group passes %llu\n", (double)h[0] / h[8], (double)h[1] / h[8],
              (double)h[2] / h[8], (double)h[3] / h[8], (double)h[4] / h[8], (double)h[5] / h[8], h[8], h[9]); }
#endif
    HIP_CHECK(hipEventRecord(w[2], st_w));
    join_page_stream(st_w);
    HIP_CHECK(hipMemcpyAsync(&S.hb->page_err[ai], S.err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(w[3], st));
    iv.kp = ke;
    ticket->arena = ai; ticket->dev = arena; ticket->nbytes = page_bytes; ticket->done_event = (void*)w[3]; S.queue_compression(ai, arena, page_bytes);
    return true;
  }
  STAGE("k_site_write");
  hipLaunchKernelGGL(k_site_copy, dim3(blocks_for(np, 4)), dim3(256), 0, st, (const uint32_t*)S.prefix_len.p, (const char*)S.site_staging.p, (const char*)S.spill_buf.p, (const int32_t*)S.spill_chunk.p, kp, ke,
                     (const uint64_t*)S.chunk_off.p, iv.nchunks, page_base, arena);
  hipLaunchKernelGGL(k_site_write, dim3(blocks_for(np, 64)), dim3(64), 0, st, S.d_sx.p, (const char*)S.site_staging.p, (const char*)S.spill_buf.p, (const int32_t*)S.spill_chunk.p, kp, ke, S.chunk_off.p, iv.nchunks, page_base, arena, S.err.p);
  STAGE("k_assemble_write");
  {
    const int64_t blk = (int64_t)1 << order_block_log2();
    if (iv.events && S.order_iv_valid && (kp % blk) == 0 && (ke == P || (ke % blk) == 0)) {
      const int evrun = iv.evrun;
      const unsigned eruns = (unsigned)((np + evrun - 1) / evrun);
      HIP_CHECK(hipEventRecord(w[1], st));
      const unsigned units = eruns * (unsigned)iv.nchunks;
      if (write_waves_per_group() >= 4)
        hipLaunchKernelGGL(k_assemble_write_ev<4>, dim3((units + 3u) / 4u), dim3(kAsmRows * 4), 0, st, (const char*)S.pool.p, (const char*)S.pool_ovf.p, (const uint32_t*)S.prefix_len.p,
                           iv.eb, (int64_t)(kp / evrun) * iv.nchunks, (const int32_t*)(S.order_iv.p + kp), np, iv.nchunks, evrun, (const uint64_t*)S.chunk_off.p, page_base, arena);
      else
        hipLaunchKernelGGL(k_assemble_write_ev<1>, dim3(units), dim3(kAsmRows), 0, st, (const char*)S.pool.p, (const char*)S.pool_ovf.p, (const uint32_t*)S.prefix_len.p,
                           iv.eb, (int64_t)(kp / evrun) * iv.nchunks, (const int32_t*)(S.order_iv.p + kp), np, iv.nchunks, evrun, (const uint64_t*)S.chunk_off.p, page_base, arena);
      HIP_CHECK(hipEventRecord(w[2], st));
      HIP_CHECK(hipMemcpyAsync(&S.hb->page_err[ai], S.err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipEventRecord(w[3], st));
      iv.kp = ke;
      ticket->arena = ai; ticket->dev = arena; ticket->nbytes = page_bytes; ticket->done_event = (void*)w[3]; S.queue_compression(ai, arena, page_bytes);
      return true;
    }
  }
  const int wrun = iv.piece_path ? write2_run_length() : write_run_length();
  S.order_by_type(kp, np);
  const unsigned wruns = (unsigned)((np + wrun - 1) / wrun);
  const dim3 wgrid(wruns * (unsigned)iv.nchunks);
  if (iv.piece_path) {
    HIP_CHECK(hipEventRecord(w[1], st));
#define GDB_LAUNCH_WRITE2(W, L) hipLaunchKernelGGL((k_write2<W, L>), dim3((wgrid.x + (W) - 1u) / (W)), dim3(kAsmRows * (W)), 0, st, iv.pc2, (const char*)S.pool.p, (const char*)S.pool_ovf.p, \
    (const int32_t*)S.order.p, np, iv.nchunks, wrun, (const uint64_t*)S.chunk_off.p, page_base, arena, (xcd_aware_numbering() ? 1 : 0) | (w2dbg << 1))
    const int ww = write_waves_per_group(), wl = write_image_kb(iv.P > 0 && iv.nchunks > 0 ? iv.total_bytes / ((uint64_t)iv.P * (uint64_t)iv.nchunks) : 0);
    const int w2dbg = getenv("GDBAMD_W2_DBG") ? atoi(getenv("GDBAMD_W2_DBG")) : 0;
    S.dbg_counters.ensure(8);
    if (w2dbg & 64) HIP_CHECK(hipMemsetAsync(S.dbg_counters.p, 0, 8 * sizeof(unsigned long long), st));
#define GDB_LAUNCH_WRITE3(W, L) hipLaunchKernelGGL((k_write3<W, L>), dim3((wgrid.x + (W) - 1u) / (W)), dim3(kAsmRows * (W)), 0, st, iv.pc2, (const PieceEntry*)S.plist.p, (const uint32_t*)S.pl_ofs.p, iv.NT, \
    (const char*)S.pool.p, (const char*)S.pool_ovf.p, (const int32_t*)S.order.p, np, iv.nchunks, wrun, (const uint64_t*)S.chunk_off.p, page_base, arena, (xcd_aware_numbering() ? 1 : 0) | (w2dbg << 1), S.dbg_counters.p)
    if (iv.asm_path == 3) { if (wl <= 4) GDB_LAUNCH_WRITE3(1, 4096); else GDB_LAUNCH_WRITE3(1, 8192); }
    else
    if (ww >= 4 && wl <= 4) GDB_LAUNCH_WRITE2(4, 4096);
    else if (ww >= 4) GDB_LAUNCH_WRITE2(4, 8192);
    else if (wl <= 4) GDB_LAUNCH_WRITE2(1, 4096);
    else if (wl <= 6) GDB_LAUNCH_WRITE2(1, 6144);
    else GDB_LAUNCH_WRITE2(1, 8192);
#undef GDB_LAUNCH_WRITE2
#undef GDB_LAUNCH_WRITE3
    if ((w2dbg & 64) && iv.asm_path == 3) {
      unsigned long long c[4];
      HIP_CHECK(hipMemcpyAsync(c, S.dbg_counters.p, sizeof(c), hipMemcpyDeviceToHost, st));
      HIP_CHECK(hipStreamSynchronize(st));
      fprintf(stderr, "[k_write3] wavefront cycles %.3e, waiting for texts %.3e (%.1f %%), waits %llu, steps %llu: %.0f cycles per step, %.0f per wait\n", (double)c[0], (double)c[1],
              100.0 * (double)c[1] / (double)std::max<unsigned long long>(1, c[0]), c[2], c[3], (double)c[0] / (double)std::max<unsigned long long>(1, c[3]), (double)c[1] / (double)std::max<unsigned long long>(1, c[2]));
    }
    HIP_CHECK(hipEventRecord(w[2], st));
    HIP_CHECK(hipMemcpyAsync(&S.hb->page_err[ai], S.err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    HIP_CHECK(hipEventRecord(w[3], st));
    iv.kp = ke;
    ticket->arena = ai; ticket->dev = arena; ticket->nbytes = page_bytes; ticket->done_event = (void*)w[3]; S.queue_compression(ai, arena, page_bytes);
    return true;
  }
  const auto res_view = [&S, &iv]() { return iv.res_compact ? ResMatrix{nullptr, S.res_off.p, S.res_len8.p} : ResMatrix{S.resolved.p, nullptr, nullptr}; };
  if (!iv.resolved_whole) {   // the interval's matrix exceeded the budget: resolve this page's records now
    if (iv.res_compact) { S.res_off.ensure((size_t)np * iv.nchunks * kAsmRows); S.res_len8.ensure((size_t)np * iv.nchunks * kAsmRows); }
    else S.resolved.ensure((size_t)np * iv.nchunks * kAsmRows);
    if (iv.asm_path == 2) {
      const int frun = fill_run_length();
      hipLaunchKernelGGL(k_fill2, dim3((unsigned)(((np + frun - 1) / frun) * iv.nchunks)), dim3(kAsmRows), 0, st, iv.pc2, (const uint2*)S.slot_desc.p, (const int32_t*)S.order.p, np, iv.nchunks, frun,
                         S.resolved.p, kp);
    } else
    launch_assemble_size(S, st, wgrid, iv.ac, S.order.p, np, N, iv.nchunks, wrun, (uint64_t*)nullptr, 0, res_view(), kp, res_chunk_major() && iv.asm_path != 2 ? np : (int64_t)0);
  }
  const hipStream_t st_w = fork_page_stream();
  HIP_CHECK(hipEventRecord(w[1], st_w));   // [w1, w2] brackets the page-assembly kernel alone (its duration feeds the roofline figure)
#define GDB_LAUNCH_WRITE(W, L) hipLaunchKernelGGL((k_assemble_write<W, L>), dim3((wgrid.x + (W) - 1u) / (W)), dim3(kAsmRows * (W)), 0, st_w, (const char*)S.pool.p, (const char*)S.pool_ovf.p, \
    (const uint32_t*)S.prefix_len.p, res_view(), iv.resolved_whole ? (int64_t)0 : kp, (const int32_t*)S.order.p, np, iv.nchunks, wrun, (const uint64_t*)S.chunk_off.p, page_base, arena, xcd_aware_numbering() ? 1 : 0, \
    (res_chunk_major() && iv.asm_path != 2) ? (iv.resolved_whole ? (int64_t)iv.P_rows : np) : (int64_t)0)
  {
    const int ww = write_waves_per_group(), wl = write_image_kb(iv.P > 0 && iv.nchunks > 0 ? iv.total_bytes / ((uint64_t)iv.P * (uint64_t)iv.nchunks) : 0);
    // the largest record's average entry is a long one (copied by the whole wavefront): the variant that keeps 4 words per lane in flight
    const bool long_entries = coop_unroll_wanted() && iv.nchunks > 0 && iv.max_record_bytes / ((uint64_t)iv.nchunks * kAsmRows) > (uint64_t)kCooperativeEntry;
    if (long_entries && ww == 1) {
      hipLaunchKernelGGL((k_assemble_write<1, 8192, 4>), dim3(wgrid.x), dim3(kAsmRows), 0, st_w, (const char*)S.pool.p, (const char*)S.pool_ovf.p,
        (const uint32_t*)S.prefix_len.p, res_view(), iv.resolved_whole ? (int64_t)0 : kp, (const int32_t*)S.order.p, np, iv.nchunks, wrun, (const uint64_t*)S.chunk_off.p, page_base, arena, xcd_aware_numbering() ? 1 : 0,
        (res_chunk_major() && iv.asm_path != 2) ? (iv.resolved_whole ? (int64_t)iv.P_rows : np) : (int64_t)0);
    } else
    if (ww >= 4 && wl <= 4) GDB_LAUNCH_WRITE(4, 4096);
    else if (ww >= 4 && wl <= 6) GDB_LAUNCH_WRITE(4, 6144);
    else if (ww >= 4) GDB_LAUNCH_WRITE(4, 8192);
    else if (ww == 2 && wl <= 4) GDB_LAUNCH_WRITE(2, 4096);
    else if (ww == 2) GDB_LAUNCH_WRITE(2, 8192);
    else if (wl <= 4) GDB_LAUNCH_WRITE(1, 4096);
    else GDB_LAUNCH_WRITE(1, 8192);
  }
#undef GDB_LAUNCH_WRITE
  HIP_CHECK(hipEventRecord(w[2], st_w));
  join_page_stream(st_w);
  HIP_CHECK(hipMemcpyAsync(&S.hb->page_err[ai], S.err.p, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_CHECK(hipEventRecord(w[3], st));
  iv.kp = ke;
  ticket->arena = ai; ticket->dev = arena; ticket->nbytes = page_bytes; ticket->done_event = (void*)w[3]; S.queue_compression(ai, arena, page_bytes);
  return true;
}

void DevicePipeline::finish_page(PageTicket& ticket) {
  Impl& S = *m_;
  Impl::IntervalState& iv = S.iv;
  hipEvent_t* w = S.ev_page[ticket.arena & 1];
  HIP_CHECK(hipEventSynchronize(w[3]));
  float ms_site = 0, ms_entry = 0;
  HIP_CHECK(hipEventElapsedTime(&ms_site, w[0], w[1]));
  HIP_CHECK(hipEventElapsedTime(&ms_entry, w[1], w[2]));
  iv.stats.err_bits = S.hb->page_err[ticket.arena & 1];
  iv.stats.ms_write += ms_site + ms_entry;
  iv.write_kernel_ms += ms_entry;
  iv.stats.write_launches++;
  iv.stats.ms_write_kernel_avg = iv.write_kernel_ms / iv.stats.write_launches;
  iv.stats.pages++;
  iv.stats.ms_total = iv.stats.ms_sweep + iv.stats.ms_site + iv.stats.ms_size + iv.stats.ms_write;
  if (iv.stats.err_bits) throw GenomicsDBDeviceException(err_bits_text(iv.stats.err_bits));
  if (S.hp.bgzf && ticket.nbytes) {     // "z" / "b": only compressed bytes leave the GPU
    float ms = 0;
    ticket.nbytes = S.bgzf->finish(ticket.arena, &ms);       // (queued by begin_page)
    iv.stats.bytes_compressed += ticket.nbytes;
    iv.stats.ms_compress += ms;
    iv.stats.ms_total += ms;
  }
}

void DevicePipeline::set_arena_release_event(int arena_idx, void* hip_event) { m_->arena_release[arena_idx & 1] = (hipEvent_t)hip_event; }

bool DevicePipeline::next_page(uint64_t arena_bytes, const char** dev_ptr, uint64_t* nbytes) {
  PageTicket t;
  if (!begin_page(arena_bytes, 0, &t)) return false;
  finish_page(t);
  *dev_ptr = t.dev;
  *nbytes = t.nbytes;
  return true;
}

const IntervalStats& DevicePipeline::interval_stats() const { return m_->iv.stats; }
FragmentView DevicePipeline::fragment_view() const { return m_->fr; }
uint64_t DevicePipeline::fragment_generation() const { return m_->fragment_generation; }
void DevicePipeline::set_page_priority(bool on) { m_->page_priority = on; }

IntervalStats DevicePipeline::run_interval(int64_t qb, int64_t qe, uint64_t arena_bytes, PageCallback cb, void* user) {
  prepare_interval(qb, qe);
  const char* p = nullptr;
  uint64_t n = 0;
  while (next_page(arena_bytes, &p, &n)) if (cb) cb(user, p, n);
  return m_->iv.stats;
}

}  // namespace genomicsdb_amd
