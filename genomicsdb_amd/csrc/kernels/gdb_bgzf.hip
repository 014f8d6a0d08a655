// gdb_bgzf.hip - BGZF blocks deflated on the device (see gdb_bgzf.h).
//
// k_bgzf_deflate: one wavefront = one block of 4 / 8 / 16 KiB of input (8 by default), held in LDS together with a 512- / 1 024- /
// 2 048-entry hash table of the most recent position of every 4-byte group, a queue of tokens and a ring of output words
// (11.3 KB for 8 KiB blocks: 14 wavefronts per CU).  The wavefront takes 64 consecutive positions per step:
//   * every lane hashes the 4 bytes at its position, reads the candidate the table holds (a position of an earlier step) and
//     leaves its own position there;
//   * every lane measures its match against the candidate over 16 bytes; ALL LDS accesses are aligned dwords that are
//     funnel-shifted in registers (v_alignbyte): an 8-byte LDS read off its natural alignment is replayed at 64 cycles;
//   * the greedy parse of RFC 1951's usual compressors - take the match if it is >= 4 bytes, else a literal, continue behind
//     it - is a chain through the 64 positions that the scalar unit follows with v_readlane (runs of literals in one go; a match
//     whose 16-byte probe ran to its end is extended by the whole wavefront, 64 bytes per ballot, up to 258 bytes);
//   * the tokens are compacted into the queue, and 64 at a time are encoded with the FIXED Huffman code (literal 8-9 bits; length
//     code + extra bits, 5-bit distance code + extra bits, <= 31 bits, all by arithmetic - no tables), placed by a DPP scan of
//     the bit counts and OR-ed into the ring; full halves of the ring leave as coalesced stores.
// A block that does not shrink is written as a stored block.  The CRC-32 of the block's input is taken in the same kernel: every
// lane runs the table-driven CRC (slicing-by-4) over its own 33 dwords (an odd count: the lanes' reads fall into different LDS
// banks), advances it over the bytes behind its piece with a per-lane precomputed GF(2) operator (4 x 256 words per lane) and the
// 64 contributions are XOR-ed (the CRC register is linear in its start value and the message).
// k_bgzf_pack: headers ('BC' extra field with the block size), payloads and trailers (CRC-32, input size) at their final,
// exclusive-scanned offsets.
#include "gdb_bgzf.h"

#include <hip/hip_runtime.h>
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <rocprim/device/device_scan.hpp>

namespace genomicsdb_amd {

const unsigned char kBgzfEofBlock[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, 0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#define BGZF_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw std::runtime_error(std::string("BGZF: ") + #expr + " failed: " + hipGetErrorString(_e)); } while (0)

namespace {

constexpr int kProbe = 16;                         // bytes every position compares against its candidate (a multiple of 8)
constexpr int kRingWords = 128, kFlushWords = 64;  // (64 tokens x 31 bits = 62 words at most join the < 64 words that wait: < 126)
// room for the payload of one block while it is being produced (a block that grows is cut off early, at most one ring flush beyond its input size)
__host__ __device__ constexpr uint32_t slot_bytes(uint32_t block_input) { return block_input + 1024u; }
constexpr uint32_t kNoCand = 0xFFFFu;
constexpr uint32_t kTokQueue = 128;
// hash bits of the byte-level kernels' tables at 8 KiB blocks.  Round 6: 9 instead of 10 - the kernel runs on residency, and 2 KB less LDS per pair
// of wavefronts are 13 instead of 11 workgroups per CU: BCF2 pages 235 -> 261 GB/s at the same ratio (4.09).  8 bits: 285 GB/s on BCF2 pages at the
// same ratio again, but text-like input loses 2-3 % - so 8 only where the producer says the pages are BCF2 records (BgzfDeviceCompressor::set_bcf2);
// profiles/r6_ab_bgzf_hash_bits.txt
#ifndef GDBAMD_BGZF_HASH_BITS_8K
#define GDBAMD_BGZF_HASH_BITS_8K 9
#endif

typedef __attribute__((address_space(3))) uint8_t lds_u8;

__device__ __forceinline__ uint32_t lds_read_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t lds_read_u64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31
  return v;
}
__device__ __forceinline__ uint32_t wave_xor(uint32_t v) {
  for (int off = 32; off; off >>= 1) v ^= (uint32_t)__shfl_xor((int)v, off, 64);
  return v;
}

// fixed Huffman code of a literal / length symbol (RFC 1951 3.2.6), already bit-reversed for the LSB-first stream
__device__ __forceinline__ void fixed_litlen(uint32_t sym, uint32_t& bits, uint32_t& nb) {
  uint32_t code;
  if (sym < 144u) { code = 0x30u + sym; nb = 8; }
  else if (sym < 256u) { code = 0x190u + (sym - 144u); nb = 9; }
  else if (sym < 280u) { code = sym - 256u; nb = 7; }
  else { code = 0xC0u + (sym - 280u); nb = 8; }
  bits = __brev(code) >> (32u - nb);
}

// One wavefront's share of a block: the positions [begin, end) of the `n` input bytes in LDS, deflated as ONE fixed-Huffman block
// whose first three bits (BFINAL + BTYPE) are already in ring[0].  `table` is the wavefront's own hash table: empty, or primed with
// the positions in front of `begin` (its dictionary).  Returns the bit position behind the last token (no end-of-block symbol yet);
// gave_up: the output has outgrown the input, the whole block will be stored.
template <int kBgzfBlockInput, int kHashBits>
__device__ __forceinline__ uint32_t deflate_range(const uint8_t* in, uint32_t n, uint32_t begin, uint32_t end, uint16_t* table, uint32_t* ring, uint32_t* tokq,
                                                  uint32_t* out_words, int lane, bool& gave_up) {
  uint32_t bitpos = 3, flushed = 0;
  uint32_t p = begin;
  int e = 0;
  gave_up = false;
  uint32_t qhead = 0, qtail = 0;                               // token queue (uniform counters)
  // ---- tokens -> bits: the next `ntok` <= 64 tokens of the queue, one per lane ----------------------------------------------------------
  // (the kernel is bound by the instructions it issues - SQ counters, profiles/r3_*: a step of 64 positions yields ~6 tokens, so
  // encoding per step kept 58 lanes busy with nothing; now the Huffman arithmetic, the scan and the ring update run once per 64 tokens)
  auto emit_tokens = [&](uint32_t ntok) {
    const uint32_t tok = tokq[(qhead + (uint32_t)lane) & (kTokQueue - 1)];
    const bool mine = (uint32_t)lane < ntok;
    qhead += ntok;
    const uint32_t L = tok & 511u, d = (tok >> 9) & 0x7FFFu, lit = tok >> 24;
    uint32_t bits, nb;
    {
      // both encodings are computed by every lane, then one is picked (no divergent branches)
      const uint32_t l = L - 3u;                               // (garbage when L == 0: not used then)
      const uint32_t leb = l < 8u ? 0u : (31u - (uint32_t)__clz(l | 8u)) - 2u;
      uint32_t lsym = l < 8u ? 257u + l : 261u + 4u * leb + ((l >> leb) & 3u);
      uint32_t lextra = l & ((1u << leb) - 1u);
      uint32_t lextra_bits = leb;
      if (L == 258u) { lsym = 285u; lextra = 0; lextra_bits = 0; }
      const uint32_t sym = L ? lsym : lit;
      uint32_t sb, sn;
      fixed_litlen(sym, sb, sn);
      const uint32_t deb = d < 4u ? 0u : (31u - (uint32_t)__clz(d | 4u)) - 1u;
      const uint32_t dsym = d < 4u ? d : 2u * deb + 2u + ((d >> deb) & 1u);
      const uint32_t dextra = d & ((1u << deb) - 1u);
      uint32_t mb = sb | (lextra << sn), mn = sn + lextra_bits;
      mb |= (__brev(dsym) >> 27) << mn; mn += 5u;
      mb |= dextra << mn; mn += deb;
      bits = L ? mb : sb;
      nb = mine ? (L ? mn : sn) : 0u;
    }
    const uint32_t incl = wave_incl_scan(nb);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (mine) {
      const uint32_t off = bitpos + incl - nb;
      const uint32_t w = off >> 5, sh = off & 31u;
      atomicOr(&ring[w & (kRingWords - 1)], bits << sh);
      const uint32_t hi = sh ? (bits >> (32u - sh)) : 0u;
      if (hi) atomicOr(&ring[(w + 1u) & (kRingWords - 1)], hi);
    }
    bitpos += total;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if ((bitpos >> 3) > end - begin) { gave_up = true; return; }   // uniform: the range is not shrinking, the block will be stored
    while ((bitpos >> 5) - flushed >= (uint32_t)kFlushWords) {   // uniform: full words leave, their ring slots are zeroed for reuse
      for (uint32_t i = lane; i < (uint32_t)kFlushWords; i += 64) {
        const uint32_t slot = (flushed + i) & (kRingWords - 1);
        out_words[flushed + i] = ring[slot];
        ring[slot] = 0;
      }
      flushed += kFlushWords;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
  };
  while (p < end) {                                           // uniform
    // The body of a step is written without branches on per-lane conditions (every such branch costs exec-mask bookkeeping on
    // the scalar unit, and this kernel is bound by the instructions it issues - SQ counters, profiles/r3_*): positions behind
    // the block's end hash the zero padding into a spare table slot and come out with no candidate.
    const uint32_t q = p + (uint32_t)lane;
    const bool can_hash = q + 4u <= n && q < end;
    // LDS accesses are aligned dwords only: a ds_read_b64 off its natural alignment is replayed at 64 LDS cycles per wave
    // instruction (2 when aligned), and with four of them per step the kernel was bound by the LDS array (SQ_LDS_IDX_ACTIVE at
    // 90 % of the kernel's cycles).  A lane reads the five aligned dwords around its position and funnel-shifts its 16 bytes out
    // of them (v_alignbyte); the same for the candidate.
    const uint32_t* const in32 = reinterpret_cast<const uint32_t*>(in);
    uint32_t w0, w1, w2, w3;                                    // the position's first 16 bytes: hash input, compare words, literal
    {
      const uint32_t wq = q >> 2, sh = q & 3u;
      const uint32_t o0 = in32[wq], o1 = in32[wq + 1], o2 = in32[wq + 2], o3 = in32[wq + 3], o4 = in32[wq + 4];
      w0 = __builtin_amdgcn_alignbyte(o1, o0, sh); w1 = __builtin_amdgcn_alignbyte(o2, o1, sh);
      w2 = __builtin_amdgcn_alignbyte(o3, o2, sh); w3 = __builtin_amdgcn_alignbyte(o4, o3, sh);
    }
    const uint32_t h = can_hash ? (w0 * 2654435761u) >> (32 - kHashBits) : (1u << kHashBits);
    const uint32_t cand_raw = table[h];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");     // every lane has read the table before any lane writes it
    table[h] = (uint16_t)q;
    const bool has_cand = can_hash && cand_raw != kNoCand;
    const uint32_t cand = has_cand ? cand_raw : 0u;
    // Every lane measures its match only up to kProbe = 16 bytes: nearly every position of a text like this one lies INSIDE a
    // long match of an earlier position, so full-length compares in all lanes are wasted on positions the parse never visits
    // (and the slowest lane sets the wavefront's time).  The few tokens the parse takes with a probe that ran to its end are
    // extended by the whole wavefront, 64 bytes per step, in the chain walk.
    uint32_t L;
    {
      static_assert(kProbe == 16, "the probe compares four dwords");
      const uint32_t cw = cand >> 2, csh = cand & 3u;
      const uint32_t c0 = in32[cw], c1 = in32[cw + 1], c2 = in32[cw + 2], c3 = in32[cw + 3], c4 = in32[cw + 4];
      const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, csh) ^ w0, x1 = __builtin_amdgcn_alignbyte(c2, c1, csh) ^ w1;
      const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, csh) ^ w2, x3 = __builtin_amdgcn_alignbyte(c4, c3, csh) ^ w3;
      uint32_t k = (uint32_t)kProbe;
      k = x3 ? 12u + ((uint32_t)__builtin_ctz(x3) >> 3) : k;
      k = x2 ? 8u + ((uint32_t)__builtin_ctz(x2) >> 3) : k;
      k = x1 ? 4u + ((uint32_t)__builtin_ctz(x1) >> 3) : k;
      k = x0 ? ((uint32_t)__builtin_ctz(x0) >> 3) : k;
      const uint32_t room = end > q ? end - q : 0u;           // (a match never runs past the range: behind it lies the other wavefront's share, or the zero padding)
      k = k < room ? k : room;
      L = (has_cand && k >= 4u) ? k : 0u;
    }
    // ---- greedy parse: the chain of token starts through this step's positions (scalar) --------------------------------------------
    // A run of literals up to the next position that has a match is taken in one go (one bit trick on the ballot of the matches).
    const int lim = (end - p) < 64u ? (int)(end - p) : 64;
    const uint64_t has_match = __ballot(L != 0u);
    uint64_t sel = 0;
    int cur = e;
    while (cur < lim) {
      const uint64_t rest = has_match >> cur;
      if (!(rest & 1ull)) {
        int run = rest ? (int)__builtin_ctzll(rest) : 64;
        run = run < lim - cur ? run : lim - cur;
        sel |= (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << cur;
        cur += run;
        continue;
      }
      uint32_t Lc = (uint32_t)__builtin_amdgcn_readlane((int)L, cur);
      if (Lc == (uint32_t)kProbe) {                              // uniform: the probe ran to its end - how far does the match really go?
        const uint32_t cpos = (uint32_t)__builtin_amdgcn_readlane((int)cand, cur), qpos = p + (uint32_t)cur;
        const uint32_t mx = (end - qpos) < 258u ? (end - qpos) : 258u;
        uint32_t k = (uint32_t)kProbe;
        while (k < mx) {
          const uint32_t j = k + (uint32_t)lane;
          const bool differ = j < mx && in[cpos + j] != in[qpos + j];
          const uint64_t m = __ballot(differ);
          if (m) { k += (uint32_t)__builtin_ctzll(m); break; }
          k += 64;
        }
        Lc = k < mx ? k : mx;
        if (lane == cur) L = Lc;
      }
      sel |= 1ull << cur;
      cur += (int)Lc;
    }
    e = cur - 64;
    // ---- the step's tokens join the queue: (length, distance - 1, literal) in one word --------------------------------------------------
    {
      const bool mine = (sel >> lane) & 1ull;
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel, 0u));
      const uint32_t tok = L | (((q - cand - 1u) & 0x7FFFu) << 9) | ((w0 & 0xFFu) << 24);
      if (mine) tokq[(qtail + rank) & (kTokQueue - 1)] = tok;
      qtail += (uint32_t)__builtin_popcountll(sel);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (qtail - qhead >= 64u) {                                // uniform: 64 tokens are encoded at a time (all lanes busy)
      emit_tokens(64u);
      if (gave_up) break;
    }
    p += 64;
    while (e >= 64) { e -= 64; p += 64; }                     // a match that covers whole steps: nothing to do there
  }
  if (!gave_up && qtail != qhead) emit_tokens(qtail - qhead);
  // (what is left in the ring is written by the caller, once the block's end is in it)
  (void)flushed;
  return bitpos;
}
// full ring words behind `bitpos` that emit_tokens has not flushed: everything from the last multiple of kFlushWords below bitpos / 32
__device__ __forceinline__ void flush_ring_tail(const uint32_t* ring, uint32_t* out_words, uint32_t bitpos_before_end, uint32_t bitpos, int lane) {
  const uint32_t flushed = ((bitpos_before_end >> 5) / (uint32_t)kFlushWords) * (uint32_t)kFlushWords;
  const uint32_t nwords = (bitpos + 31u) >> 5;
  for (uint32_t i = flushed + lane; i < nwords; i += 64) out_words[i] = ring[i & (kRingWords - 1)];
}
// CRC-32 of the n <= kBgzfBlockInput input bytes in LDS, by ONE wavefront (every lane its piece + the GF(2) shift over what follows it)
template <int kBgzfBlockInput>
__device__ __forceinline__ uint32_t block_crc(const uint8_t* in, uint32_t n, int lane, const uint32_t* __restrict__ crc_slice, const uint32_t* __restrict__ crc_shift) {
  constexpr int kPieceWords = kBgzfBlockInput / 256 + 1;       // dwords of the block whose CRC a lane takes: an ODD count, so that the lanes' reads fall into different banks
  const uint32_t* T0 = crc_slice, *T1 = crc_slice + 256, *T2 = crc_slice + 512, *T3 = crc_slice + 768;
  if (n == kBgzfBlockInput) {
    // (pieces of 128 bytes put every lane's dword i into the same LDS bank: a 64-way conflict per read; with 33 dwords per lane
    // - the last lanes get the short rest, or nothing - the lanes of a group hit different banks)
    uint32_t r = lane == 0 ? 0xFFFFFFFFu : 0u;
    const uint32_t* const words = reinterpret_cast<const uint32_t*>(in);
    const uint32_t w_begin = (uint32_t)kPieceWords * (uint32_t)lane;
    const uint32_t w_end = w_begin + (uint32_t)kPieceWords < (uint32_t)(kBgzfBlockInput / 4) ? w_begin + (uint32_t)kPieceWords : (uint32_t)(kBgzfBlockInput / 4);
    for (uint32_t i = w_begin; i < w_end; ++i) {
      const uint32_t x = r ^ words[i];
      r = T3[x & 0xFFu] ^ T2[(x >> 8) & 0xFFu] ^ T1[(x >> 16) & 0xFFu] ^ T0[x >> 24];
    }
    const uint32_t* S = crc_shift + (size_t)lane * 1024;      // this lane's contribution after the bytes behind its piece
    const uint32_t c = S[r & 0xFFu] ^ S[256 + ((r >> 8) & 0xFFu)] ^ S[512 + ((r >> 16) & 0xFFu)] ^ S[768 + (r >> 24)];
    return wave_xor(c) ^ 0xFFFFFFFFu;
  }
  uint32_t r = 0xFFFFFFFFu;                                    // the short last block of a page: one lane, serially
  if (lane == 0) {
    uint32_t i = 0;
    for (; i + 4 <= n; i += 4) { const uint32_t x = r ^ lds_read_u32(in + i); r = T3[x & 0xFFu] ^ T2[(x >> 8) & 0xFFu] ^ T1[(x >> 16) & 0xFFu] ^ T0[x >> 24]; }
    for (; i < n; ++i) r = T0[(r ^ in[i]) & 0xFFu] ^ (r >> 8);
  }
  return r ^ 0xFFFFFFFFu;
}
// the block's bytes into LDS (zero behind its end: the probes read up to 39 bytes past a position), by all threads of the workgroup
template <int kBgzfBlockInput>
__device__ __forceinline__ void load_block(uint4* in4, const uint8_t* blk_src, uint32_t n, int tid, int nthreads) {
  for (uint32_t q = tid; q < kBgzfBlockInput / 16 + 3; q += nthreads) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q * 16u + 16u <= n) v = reinterpret_cast<const uint4*>(blk_src)[q];
    else if (q * 16u < n) {
      uint32_t w[4] = {0, 0, 0, 0};
      for (uint32_t b = q * 16u; b < n; ++b) w[(b & 15u) >> 2] |= (uint32_t)blk_src[b] << (8u * (b & 3u));
      v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    in4[q] = v;
  }
}

// csize / coff hold kParts payload sizes / offsets inside the block's slot (the pack kernel lays the parts behind each other)
constexpr int kParts = 4;
template <int kBgzfBlockInput>
__global__ void __launch_bounds__(64) k_bgzf_deflate(const uint8_t* __restrict__ src, uint64_t n_total, uint8_t* __restrict__ slots, uint32_t* __restrict__ csize, uint32_t* __restrict__ coff,
                                                     uint64_t* __restrict__ bsize, uint32_t* __restrict__ crc_out, const uint32_t* __restrict__ crc_slice,
                                                     const uint32_t* __restrict__ crc_shift) {
  const uint64_t blk = blockIdx.x;
  const uint64_t base = blk * (uint64_t)kBgzfBlockInput;
  const uint32_t n = (uint32_t)((n_total - base) < (uint64_t)kBgzfBlockInput ? (n_total - base) : (uint64_t)kBgzfBlockInput);
  const int lane = threadIdx.x;
  __shared__ uint4 in4[kBgzfBlockInput / 16 + 3];             // 48 bytes of zeros behind the block: the probes read up to 39 bytes past a position
  constexpr int kHashBits = kBgzfBlockInput > 8192 ? 11 : kBgzfBlockInput > 4096 ? GDBAMD_BGZF_HASH_BITS_8K : 9;
  constexpr uint32_t kSlotBytes = slot_bytes((uint32_t)kBgzfBlockInput);
  __shared__ uint32_t table32[(1 << kHashBits) / 2 + 2];      // (+ a spare slot for the positions behind the block's end)
  __shared__ uint32_t ring[kRingWords];
  __shared__ uint32_t tokq[kTokQueue];                        // tokens waiting to be encoded (< 64 before a step, < 128 after it)
  uint8_t* const in = reinterpret_cast<uint8_t*>(in4);
  load_block<kBgzfBlockInput>(in4, src + base, n, lane, 64);
  for (uint32_t q = lane; q < (1u << kHashBits) / 2 + 2; q += 64) table32[q] = 0xFFFFFFFFu;
  for (uint32_t q = lane; q < kRingWords; q += 64) ring[q] = (q == 0) ? 3u : 0u;      // BFINAL = 1, BTYPE = 01 (fixed Huffman)
  __syncthreads();
  uint32_t* const out_words = reinterpret_cast<uint32_t*>(slots + blk * (uint64_t)kSlotBytes);
  bool gave_up;
  const uint32_t body_bits = deflate_range<kBgzfBlockInput, kHashBits>(in, n, 0u, n, reinterpret_cast<uint16_t*>(table32), ring, tokq, out_words, lane, gave_up);
  uint32_t payload = 0;
  if (!gave_up) {
    const uint32_t bitpos = body_bits + 7u;                   // end of block: symbol 256 = seven zero bits
    payload = (bitpos + 7u) >> 3;
    if (payload >= n + 5u) gave_up = true;
    else flush_ring_tail(ring, out_words, body_bits, bitpos, lane);
  }
  if (gave_up) {
    // stored block: 0x01 (BFINAL, BTYPE 00, padding), LEN, NLEN, the bytes
    uint8_t* o = slots + blk * (uint64_t)kSlotBytes;
    if (lane == 0) { o[0] = 1; o[1] = (uint8_t)(n & 0xFFu); o[2] = (uint8_t)(n >> 8); o[3] = (uint8_t)(~n & 0xFFu); o[4] = (uint8_t)((~n >> 8) & 0xFFu); }
    for (uint32_t i = lane; i < n; i += 64) o[5 + i] = in[i];
    payload = n + 5u;
  }
  const uint32_t crc = block_crc<kBgzfBlockInput>(in, n, lane, crc_slice, crc_shift);
  if (lane < kParts) { csize[kParts * blk + lane] = lane ? 0u : payload; coff[kParts * blk + lane] = 0u; }
  if (lane == 0) { bsize[blk] = (uint64_t)payload + kBgzfHeaderBytes + kBgzfTrailerBytes; crc_out[blk] = crc; }
}

// TWO wavefronts per block (round 5).  A block's time is a chain of ~128 dependent steps and a wavefront issues an instruction every
// third cycle (LDS round trips, the scalar parse): more wavefronts per CU and shorter chains are what it runs on.  The two halves of a
// block go to two wavefronts, pigz's way: wavefront 1 first enters the positions of the FIRST half into its own hash table (hash and
// store only: ~5 % of a half's work), so that its matches reach back into the first half like a sequential parse's would; each half
// becomes a DEFLATE block of its own, the first one ends with an empty stored block (3 bits + padding + 00 00 FF FF: zlib's sync flush),
// which makes the second one start on a byte boundary; the pack kernel puts the two payloads behind each other.  Both wavefronts share
// the 8 KiB of input in LDS: 14.4 KB per pair = 22 wavefronts per CU instead of 14.  Ratio: a match cannot cross the middle, + 5 bytes.
template <int kBgzfBlockInput, int kHashBitsAt8K = GDBAMD_BGZF_HASH_BITS_8K>
__global__ void __launch_bounds__(128) k_bgzf_deflate2(const uint8_t* __restrict__ src, uint64_t n_total, uint8_t* __restrict__ slots, uint32_t* __restrict__ csize, uint32_t* __restrict__ coff,
                                                       uint64_t* __restrict__ bsize, uint32_t* __restrict__ crc_out, const uint32_t* __restrict__ crc_slice,
                                                       const uint32_t* __restrict__ crc_shift) {
  const uint64_t blk = blockIdx.x;
  const uint64_t base = blk * (uint64_t)kBgzfBlockInput;
  const uint32_t n = (uint32_t)((n_total - base) < (uint64_t)kBgzfBlockInput ? (n_total - base) : (uint64_t)kBgzfBlockInput);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);    // (wave-uniform)
  constexpr uint32_t kHalf = kBgzfBlockInput / 2;
  __shared__ uint4 in4[kBgzfBlockInput / 16 + 3];
  constexpr int kHashBits = kBgzfBlockInput > 8192 ? 11 : kBgzfBlockInput > 4096 ? kHashBitsAt8K : 9;
  constexpr uint32_t kSlotBytes = slot_bytes((uint32_t)kBgzfBlockInput);
  constexpr uint32_t kTableWords = (1 << kHashBits) / 2 + 2;
  __shared__ uint32_t table32[2][kTableWords];
  __shared__ uint32_t ring2[2][kRingWords];
  __shared__ uint32_t tokq2[2][kTokQueue];
  __shared__ uint32_t s_payload[2], s_gave_up[2];
  uint8_t* const in = reinterpret_cast<uint8_t*>(in4);
  load_block<kBgzfBlockInput>(in4, src + base, n, tid, 128);
  const bool two = n > kHalf;                                 // (a short last block: the first wavefront alone)
  uint32_t* const ring = ring2[wv];
  for (uint32_t q = lane; q < kTableWords; q += 64) table32[wv][q] = 0xFFFFFFFFu;
  for (uint32_t q = lane; q < kRingWords; q += 64) ring[q] = (q == 0) ? ((wv == 1 || !two) ? 3u : 2u) : 0u;   // BTYPE = 01; BFINAL only on the block's last part
  if (tid < 2) { s_payload[tid] = 0u; s_gave_up[tid] = 0u; }
  __syncthreads();
  uint16_t* const table = reinterpret_cast<uint16_t*>(table32[wv]);
  const uint32_t begin = wv == 0 ? 0u : kHalf, end = wv == 0 ? (two ? kHalf : n) : n;
  uint32_t* const out_words = reinterpret_cast<uint32_t*>(slots + blk * (uint64_t)kSlotBytes + (uint64_t)wv * (kSlotBytes / 2));
  if (wv == 0 || two) {
    if (wv == 1) {
      // the dictionary: every position of the first half enters the table (the most recent one of a hash value stays, as in a sequential parse)
      const uint32_t* const in32 = reinterpret_cast<const uint32_t*>(in);
      for (uint32_t pp = 0; pp < kHalf; pp += 64) {
        const uint32_t q = pp + (uint32_t)lane, wq = q >> 2, sh = q & 3u;
        const uint32_t w0 = __builtin_amdgcn_alignbyte(in32[wq + 1], in32[wq], sh);
        table[(w0 * 2654435761u) >> (32 - kHashBits)] = (uint16_t)q;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    bool gave_up;
    const uint32_t body_bits = deflate_range<kBgzfBlockInput, kHashBits>(in, n, begin, end, table, ring, tokq2[wv], out_words, lane, gave_up);
    uint32_t payload = 0;
    if (!gave_up) {
      uint32_t bitpos = body_bits + 7u;                       // end of block: seven zero bits
      if (wv == 0 && two) {
        // an empty stored block behind it: BFINAL = 0, BTYPE = 00 (three zero bits), padding to the byte, LEN = 0000, NLEN = FFFF
        bitpos = (bitpos + 3u + 7u) & ~7u;
        if (lane == 0) {                                      // the two FF bytes: bits [bitpos + 16, bitpos + 32)
          const uint32_t off = bitpos + 16u, w = off >> 5, sh = off & 31u;   // (sh is 0, 8, 16 or 24)
          atomicOr(&ring[w & (kRingWords - 1)], 0xFFFFu << sh);
          if (sh > 16u) atomicOr(&ring[(w + 1u) & (kRingWords - 1)], 0xFFFFu >> (32u - sh));
        }
        bitpos += 32u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
      payload = (bitpos + 7u) >> 3;
      if (payload >= (end - begin) + 16u) gave_up = true;
      else flush_ring_tail(ring, out_words, body_bits, bitpos, lane);
    }
    if (lane == 0) { s_payload[wv] = payload; s_gave_up[wv] = gave_up ? 1u : 0u; }
  }
  __syncthreads();
  const uint32_t total = s_payload[0] + s_payload[1];
  const bool store = s_gave_up[0] || s_gave_up[1] || total >= n + 5u;
  if (store) {
    uint8_t* o = slots + blk * (uint64_t)kSlotBytes;          // (the stored block spans both wavefronts' parts of the slot)
    if (tid == 0) { o[0] = 1; o[1] = (uint8_t)(n & 0xFFu); o[2] = (uint8_t)(n >> 8); o[3] = (uint8_t)(~n & 0xFFu); o[4] = (uint8_t)((~n >> 8) & 0xFFu); }
    for (uint32_t i = tid; i < n; i += 128) o[5 + i] = in[i];
  }
  if (wv == 0) {
    const uint32_t crc = block_crc<kBgzfBlockInput>(in, n, lane, crc_slice, crc_shift);
    if (lane == 0) {
      const uint32_t p0 = store ? n + 5u : s_payload[0], p1 = store ? 0u : s_payload[1];
      csize[kParts * blk] = p0; csize[kParts * blk + 1] = p1; csize[kParts * blk + 2] = 0u; csize[kParts * blk + 3] = 0u;
      coff[kParts * blk] = 0u; coff[kParts * blk + 1] = kSlotBytes / 2; coff[kParts * blk + 2] = 0u; coff[kParts * blk + 3] = 0u;
      bsize[blk] = (uint64_t)(p0 + p1) + kBgzfHeaderBytes + kBgzfTrailerBytes; crc_out[blk] = crc;
    }
  }
}


// ---- VCF TEXT: matches only where an entry begins ("anchored" LZ77) + a Huffman code made for VCF text (round 6) -----------------------
// The byte-level kernels above hash and probe all 64 positions of a step although the greedy parse visits ~6 of them (every position
// inside a match is wasted work): 13.3 K vector instructions per 8 KiB block, and the kernel is bound by the instructions it issues.
// VCF text says where matches begin: a sample column ("\t./.:99:.:.:0,297,4455,...") repeats the column of some earlier sample with
// the same leading fields.  So a lane here is not a byte position but an ANCHOR - a tab or newline (where 64 bytes go by without one:
// the first ':' or ',' of a 32-byte chunk, else the chunk's first byte; at most one anchor per 16 bytes, so a block has <= 512 and a
// segment - the bytes from an anchor to the next - is at most 110 bytes long):
//   * hash of the anchor's first 8 bytes -> the most recent earlier anchor with that hash (a table per wavefront, entered sixteen lanes at a
//     time so that an anchor finds candidates among the lanes in front of it; the three lanes right in front are compared directly);
//   * ONE match per anchor, measured to its end but never beyond the next anchor, then the rest of the segment as literals: no lane
//     depends on another lane's match, there is no parse chain.
// FOUR wavefronts share a block (its 8 KiB of input sit in LDS once): the anchors are found by all 256 threads, each wavefront takes a
// quarter of them and primes its table with the anchors in front of its quarter.  PASS 1 finds the matches and counts every anchor's
// bits (kept in LDS: length, distance, bit count); after a barrier every lane knows where its bits go in the block's ONE bit string;
// PASS 2 ORs them - the match, then the literals four at a time - into an LDS image of the payload, which leaves as coalesced stores.
// The block is ONE DEFLATE block of BTYPE = 10 whose Huffman code is the same in every block (gdb_bgzf_text_code.inc, generated by
// tests/tools/bgzf_text_code.py from VCF text: digits 4 bits, ':' 3, ',' 5, tab 8, every other byte 8 - 12; the 67-byte header that
// describes the code is a constant bit string): a literal of this text costs 4.2 bits instead of the fixed code's 8.
// The CRC-32 is taken by all four wavefronts.  A block that does not shrink is stored.  ANY set of anchors gives a valid stream - they
// only decide how much is found - so binary pages (BCF2) keep the byte-level kernel and the producer of the page says which one runs.
#include "gdb_bgzf_text_code.inc"
constexpr int kTW = 4;                          // wavefronts per block
constexpr int kTThreads = 64 * kTW;
constexpr int kTMaxAnch = 512;                  // (one per 16 bytes of an 8 KiB block)
constexpr int kTHashBits = 8;
constexpr uint32_t kTNoCand = 0xFFFFu;
#ifndef GDBAMD_BGZF_MERGE
#define GDBAMD_BGZF_MERGE 1          // 0: every anchor's match is a token of its own (variant builds: what merging costs and brings)
#endif
#ifndef GDBAMD_BGZF_CONTINUE_FROM
#define GDBAMD_BGZF_CONTINUE_FROM 12
#endif
constexpr int kTContinueFrom = GDBAMD_BGZF_CONTINUE_FROM;   // secondary anchors a step of 64 must have before its lanes try to continue the lane in front
constexpr int kTCodeWords = 192;                // the code as the kernels read it: u16 (bits << 12 | reversed code) x 286 literal / length + 30 distance symbols, header words behind
constexpr int kTHeaderAt = 160;                 // word index of the header in that table: [nbits][words ...]

// 0x80 in every byte of v that equals the byte repeated in pat (exact, no carries between bytes)
__device__ __forceinline__ uint32_t eq_bytes(uint32_t v, uint32_t pat) {
  const uint32_t x = v ^ pat;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}
__device__ __forceinline__ uint32_t marks_to_nibble(uint32_t z) { return ((z >> 7) & 1u) | ((z >> 14) & 2u) | ((z >> 21) & 4u) | ((z >> 28) & 8u); }
// the four bytes at byte position q of the block (aligned LDS dwords, funnel-shifted)
__device__ __forceinline__ uint32_t ld4u(const uint32_t* in32, uint32_t q) {
  const uint32_t wq = q >> 2;
  return __builtin_amdgcn_alignbyte(in32[wq + 1], in32[wq], q & 3u);
}
__device__ __forceinline__ void image_or(uint32_t* image, uint32_t bitoff, uint32_t bits, uint32_t nb) {   // nb <= 32, bits < 2^nb
  const uint32_t w = bitoff >> 5, sh = bitoff & 31u;
  atomicOr(&image[w], bits << sh);
  if (sh + nb > 32u) atomicOr(&image[w + 1u], bits >> (32u - sh));
}
// length / distance symbols of a match of L bytes at distance d + 1 (RFC 1951 3.2.5): symbol, extra bits, their count
__device__ __forceinline__ void match_symbols(uint32_t L, uint32_t d, uint32_t& lsym, uint32_t& lextra, uint32_t& lnb, uint32_t& dsym, uint32_t& dextra, uint32_t& dnb) {
  const uint32_t l = L - 3u;
  lnb = l < 8u ? 0u : (31u - (uint32_t)__clz(l | 8u)) - 2u;
  lsym = l < 8u ? 257u + l : 261u + 4u * lnb + ((l >> lnb) & 3u);
  lextra = l & ((1u << lnb) - 1u);
  if (L == 258u) { lsym = 285u; lextra = 0; lnb = 0; }
  dnb = d < 4u ? 0u : (31u - (uint32_t)__clz(d | 4u)) - 1u;
  dsym = d < 4u ? d : 2u * dnb + 2u + ((d >> dnb) & 1u);
  dextra = d & ((1u << dnb) - 1u);
}

// Matches of consecutive anchors that continue each other (the lane in front matched its whole segment, and this lane's match lies at the
// same distance) are ONE match: `cont` marks the lanes that continue the lane in front.  Returns, for the first lane of every run, the
// run's length (<= 258: runs are cut where the sum would pass it); `cont` comes back with the cuts made.
__device__ __forceinline__ uint32_t wave_incl_max(uint32_t v) {        // (values >= 0: the 0 the DPP moves fill in is the identity)
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));   // row_shr:1
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));   // row_shr:2
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));   // row_shr:4
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));   // row_shr:8
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));  // row_bcast:15
  v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return v;
}
__device__ __forceinline__ uint32_t merged_run_length(uint32_t L, bool& cont, int lane) {
  const uint32_t P = wave_incl_scan(L);                         // sums of L up to and including the lane
  const uint32_t E = P - L;
  uint32_t S = 0, head = 0, cut_guard = 0;
  for (;;) {                                                     // uniform: as many rounds as a run has to be cut (nearly always one)
    head = wave_incl_max(cont ? 0u : (uint32_t)lane + 1u) - 1u;  // the run's first lane (lane 0 never continues anything)
    S = P - (uint32_t)__shfl((int)E, (int)head, 64);             // the run's length up to and including the lane
    const bool over = cont && S > 258u;
    if (!__any((int)over)) break;
    if (++cut_guard > 8u) { cont = false; continue; }            // (never seen: give up merging in this step - the next round then finds nothing over)
    const int prev_over = __shfl_up((int)over, 1, 64);
    if (over && !(lane > 0 && prev_over)) cont = false;         // the first lane of a run that passes 258 starts a run of its own
  }
  // the last lane of a run hands the sum to the run's first lane (a push through the LDS crossbar, no memory: every other lane pushes to the lane
  // behind it, which continues it and has no use for what it receives - so no two lanes push to the same place)
  const int next_cont = __shfl_down((int)cont, 1, 64);
  const bool last = lane == 63 || !next_cont;
  const uint32_t total = (uint32_t)__builtin_amdgcn_ds_permute((int)((last ? head : (uint32_t)lane + 1u) << 2), (int)S);
  return cont ? 0u : total;
}

template <int kBgzfBlockInput>
__global__ void __launch_bounds__(kTThreads) __attribute__((amdgpu_waves_per_eu(7))) k_bgzf_deflate_text(const uint8_t* __restrict__ src, uint64_t n_total, uint8_t* __restrict__ slots, uint32_t* __restrict__ csize,
                                                                 uint32_t* __restrict__ coff, uint64_t* __restrict__ bsize, uint32_t* __restrict__ crc_out,
                                                                 const uint32_t* __restrict__ crc_slice, const uint32_t* __restrict__ crc_shift256, const uint32_t* __restrict__ text_code) {
  static_assert(kBgzfBlockInput == 8192, "256 threads x 32 bytes");
  const uint64_t blk = blockIdx.x;
  const uint64_t base = blk * (uint64_t)kBgzfBlockInput;
  const uint32_t n = (uint32_t)((n_total - base) < (uint64_t)kBgzfBlockInput ? (n_total - base) : (uint64_t)kBgzfBlockInput);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr uint32_t kSlotBytes = slot_bytes((uint32_t)kBgzfBlockInput);
  constexpr int kImageWords = kBgzfBlockInput / 4 + 32;            // the payload while it is put together: never more than the input (else the block is stored)
  __shared__ uint4 in4[kBgzfBlockInput / 16 + 3];
  __shared__ uint32_t image[kImageWords];
  __shared__ uint32_t minfo[kTMaxAnch];                            // per anchor: match length inside its segment (7 bits: <= 110) | continues the anchor in front << 7 | (distance - 1) << 8 (13 bits) | bits of its tokens << 21 (<= 31 + 110 x 12)
  __shared__ uint16_t anch[kTMaxAnch + 2];
  __shared__ uint16_t table_all[kTW][(1 << kTHashBits) + 2];
  __shared__ uint32_t code_w[kTCodeWords];
  __shared__ uint32_t s_cnt[kTW], s_crc[kTW], s_step_bits[kTW * 4];
  __shared__ uint8_t s_has_tab[kTThreads];
  uint8_t* const in = reinterpret_cast<uint8_t*>(in4);
  const uint32_t* const in32 = reinterpret_cast<const uint32_t*>(in4);
  const uint16_t* const code = reinterpret_cast<const uint16_t*>(code_w);       // [0, 286): literals / lengths, [288, 318): distances
  load_block<kBgzfBlockInput>(in4, src + base, n, tid, kTThreads);
  uint16_t* const table = table_all[wv];
  for (uint32_t q = lane; q < (1u << kTHashBits) + 2u; q += 64) table[q] = (uint16_t)kTNoCand;
  for (uint32_t q = tid; q < (uint32_t)kImageWords; q += kTThreads) image[q] = 0u;
  if (tid < kTCodeWords) code_w[tid] = text_code[tid];
  if (tid < kTW * 4) s_step_bits[tid] = 0u;
  __syncthreads();
  // ---- anchors: thread t looks at the 32 bytes [32 t, 32 t + 32) ----------------------------------------------------------------------
  const uint32_t chunk = 32u * (uint32_t)tid;
  uint32_t mt = 0, ms = 0;                                     // bit b: byte chunk + b is a tab / newline, a ':' / ','
  {
    const uint4 q0 = in4[2 * tid], q1 = in4[2 * tid + 1];
    const uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mt |= marks_to_nibble(eq_bytes(w[k], 0x09090909u) | eq_bytes(w[k], 0x0A0A0A0Au)) << (4 * k);
      ms |= marks_to_nibble(eq_bytes(w[k], 0x3A3A3A3Au) | eq_bytes(w[k], 0x2C2C2C2Cu)) << (4 * k);
    }
  }
  const uint32_t valid = chunk >= n ? 0u : (n - chunk >= 32u ? ~0u : ((1u << (n - chunk)) - 1u));
  mt &= valid; ms &= valid;
  s_has_tab[tid] = mt ? 1 : 0;
  __syncthreads();
  uint32_t m = mt;
  if (!mt && valid && tid > 0 && !s_has_tab[tid - 1]) m = ms ? (ms & (0u - ms)) : 1u;   // 32 .. 63 bytes without a tab in front: a secondary anchor
  if (tid == 0 && n) m |= 1u;                                                             // the block's first byte
  {                                                                                       // at most one anchor per 16 bytes: the first of each half of the chunk
    const uint32_t lo16 = m & 0xFFFFu, hi16 = m >> 16;
    m = (lo16 & (0u - lo16)) | ((hi16 & (0u - hi16)) << 16);
  }
  const uint32_t cnt = (uint32_t)__popc(m);
  const uint32_t incl_c = wave_incl_scan(cnt);
  if (lane == 63) s_cnt[wv] = incl_c;
  __syncthreads();
  uint32_t A = 0, before = 0;
#pragma unroll
  for (int k = 0; k < kTW; ++k) { const uint32_t c = s_cnt[k]; if (k < wv) before += c; A += c; }
  {
    uint32_t pos = before + incl_c - cnt;
    while (m) { const uint32_t b = (uint32_t)__builtin_ctz(m); anch[pos++] = (uint16_t)(chunk + b); m &= m - 1u; }
  }
  if (tid == 0) anch[A] = (uint16_t)n;                          // (A <= 512; n <= 8 192 fits)
  __syncthreads();
  // ---- PASS 1: this wavefront's quarter of the anchors: matches and bit counts --------------------------------------------------------------
  const uint32_t per = A ? (A + kTW - 1) / kTW : 1u;           // (<= 128: at most two steps of 64 per wavefront)
  const uint32_t lo = (uint32_t)wv * per < A ? (uint32_t)wv * per : A, hi = lo + per < A ? lo + per : A;
  // the dictionary: the anchors in front of the quarter enter the table (which of two lanes with one hash stays is not defined: either is a candidate)
  for (uint32_t j0 = 0; j0 < lo; j0 += 64) {
    const uint32_t j = j0 + (uint32_t)lane;
    if (j < lo) {
      const uint32_t a = anch[j];
      const uint32_t h = ((ld4u(in32, a) * 2654435761u) ^ (ld4u(in32, a + 4u) * 2246822519u)) >> (32 - kTHashBits);
      table[h] = (uint16_t)a;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  {
    int step = 0;
    for (uint32_t s0 = lo; s0 < hi; s0 += 64, ++step) {         // uniform
      const uint32_t j = s0 + (uint32_t)lane;
      const bool act = j < hi;
      const uint32_t a = act ? anch[j] : 0u, e = act ? anch[j + 1] : 0u;
      const uint32_t w0 = ld4u(in32, a), w1 = ld4u(in32, a + 4u);
      const uint32_t h = act ? ((w0 * 2654435761u) ^ (w1 * 2246822519u)) >> (32 - kTHashBits) : (1u << kTHashBits);
      int32_t cand = -1;
#pragma unroll
      for (int q = 0; q < 4; ++q) {                              // sixteen lanes at a time: read the table, then enter it
        if ((lane >> 4) == q) { const uint32_t c = table[h]; cand = c == kTNoCand ? -1 : (int32_t)c; }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if ((lane >> 4) == q) table[h] = (uint16_t)a;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
      if (!act) cand = -1;
#pragma unroll
      for (int k = 3; k >= 1; --k) {                             // the three anchors right in front (more recent than anything in the table)
        const uint32_t nw0 = (uint32_t)__shfl_up((int)w0, k, 64), nw1 = (uint32_t)__shfl_up((int)w1, k, 64);
        const int32_t na = __shfl_up((int)a, k, 64);
        if (act && lane >= k && nw0 == w0 && nw1 == w1 && na > cand) cand = na;
      }
      // the match: cand's bytes against the anchor's, to the first difference, the segment's end or 258
      const uint32_t seg = e - a;
      const uint32_t maxL = seg < 258u ? seg : 258u;
      uint32_t L = 0;
      {
        bool go = cand >= 0;
        const uint32_t cpos = go ? (uint32_t)cand : 0u;
        while (__any((int)go)) {
          const uint32_t x = ld4u(in32, a + L) ^ ld4u(in32, cpos + L);
          if (go) {
            L += x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
            if (x || L >= maxL) go = false;
          }
        }
        L = L < maxL ? L : maxL;
        if (L < 4u) L = 0;
      }
      // A column longer than two chunks has secondary anchors inside it (wide cohorts: PL vectors of 20 and more values).  Such an anchor
      // first tries to CONTINUE the anchor in front: if that one matched its whole segment at some candidate, the bytes behind the candidate
      // are the obvious place to look - and the two matches then merge into one token (below).  A few rounds: a continuation adopted by
      // lane j is what lane j + 1 continues.  Steps with few secondary anchors (c2's columns of ~40 bytes: the occasional variant call) skip
      // the rounds: there they cost a sixth of the kernel's speed for 4 % of ratio (profiles/r6_ab_bgzf_text_kernel.txt).
      const bool sec = act && a != 0u && in[a] != (uint8_t)'\t' && in[a] != (uint8_t)'\n';
      if (__popcll(__ballot(sec)) >= kTContinueFrom) {           // uniform
        for (int round = 0; round < 4; ++round) {
          const uint32_t pL = (uint32_t)__shfl_up((int)L, 1, 64), pa = (uint32_t)__shfl_up((int)a, 1, 64);
          const int32_t pcand = __shfl_up(cand, 1, 64);
          const int32_t cc = pcand + (int32_t)(a - pa);
          const bool tryit = sec && lane > 0 && pL > 0u && pL == a - pa && pcand >= 0 && cc != cand;
          if (!__any((int)tryit)) break;
          uint32_t L2 = 0;
          bool go = tryit;
          const uint32_t cpos = go ? (uint32_t)cc : 0u;
          while (__any((int)go)) {
            const uint32_t x = ld4u(in32, a + L2) ^ ld4u(in32, cpos + L2);
            if (go) {
              L2 += x ? ((uint32_t)__builtin_ctz(x) >> 3) : 4u;
              if (x || L2 >= maxL) go = false;
            }
          }
          L2 = L2 < maxL ? L2 : maxL;
          if (tryit && L2 >= 4u && L2 >= L) { cand = cc; L = L2; }
        }
      }
      const uint32_t d = L ? (a - (uint32_t)cand) - 1u : 0u;      // distance - 1
      bool cont = false;
      uint32_t T = L;                                            // length of the match token this lane emits (0: none, or merged into the lane in front)
      {
        const uint32_t pL = (uint32_t)__shfl_up((int)L, 1, 64), pa = (uint32_t)__shfl_up((int)a, 1, 64), pd = (uint32_t)__shfl_up((int)d, 1, 64);
        cont = GDBAMD_BGZF_MERGE && act && lane > 0 && L > 0u && pL > 0u && pL == a - pa && pd == d;
        if (__any((int)cont)) T = merged_run_length(L, cont, lane);      // uniform
      }
      uint32_t B = 0;
      if (T) {
        uint32_t lsym, lextra, lnb, dsym, dextra, dnb;
        match_symbols(T, d, lsym, lextra, lnb, dsym, dextra, dnb);
        B = (uint32_t)(code[lsym] >> 12) + lnb + (uint32_t)(code[288u + dsym] >> 12) + dnb;
      }
      // the literals behind it: their codes' lengths
      const uint32_t ls = a + L, nl = act ? seg - L : 0u;
      for (uint32_t t = 0; __any((int)(t < nl)); t += 4) {
        if (t < nl) {
          const uint32_t v = ld4u(in32, ls + t);
          const uint32_t k = nl - t < 4u ? nl - t : 4u;
#pragma unroll
          for (uint32_t q = 0; q < 4u; ++q) if (q < k) B += (uint32_t)(code[(v >> (8u * q)) & 0xFFu] >> 12);
        }
      }
      if (act) minfo[j] = L | ((cont ? 1u : 0u) << 7) | (d << 8) | (B << 21);
      const uint32_t incl = wave_incl_scan(B);
      if (lane == 63) s_step_bits[wv * 4 + step] = incl;
    }
  }
  // ---- CRC-32 of the block: every thread its nine dwords + the GF(2) shift over what follows them ---------------------------------------
  {
    const uint32_t* T0 = crc_slice, *T1 = crc_slice + 256, *T2 = crc_slice + 512, *T3 = crc_slice + 768;
    uint32_t c = 0;
    if (n == (uint32_t)kBgzfBlockInput) {
      constexpr uint32_t kPiece = kBgzfBlockInput / 4 / kTThreads + 1;      // 9: an odd count (the threads' reads fall into different banks)
      uint32_t r = tid == 0 ? 0xFFFFFFFFu : 0u;
      const uint32_t w_begin = kPiece * (uint32_t)tid;
      const uint32_t w_end = w_begin + kPiece < (uint32_t)(kBgzfBlockInput / 4) ? w_begin + kPiece : (uint32_t)(kBgzfBlockInput / 4);
      for (uint32_t i = w_begin; i < w_end; ++i) {
        const uint32_t x = r ^ in32[i];
        r = T3[x & 0xFFu] ^ T2[(x >> 8) & 0xFFu] ^ T1[(x >> 16) & 0xFFu] ^ T0[x >> 24];
      }
      const uint32_t* S = crc_shift256 + (size_t)tid * 1024;
      c = S[r & 0xFFu] ^ S[256 + ((r >> 8) & 0xFFu)] ^ S[512 + ((r >> 16) & 0xFFu)] ^ S[768 + (r >> 24)];
      c = wave_xor(c);
    } else if (tid == 0) {                                        // the short last block of a page: one thread, serially
      uint32_t r = 0xFFFFFFFFu, i = 0;
      for (; i + 4 <= n; i += 4) { const uint32_t x = r ^ lds_read_u32(in + i); r = T3[x & 0xFFu] ^ T2[(x >> 8) & 0xFFu] ^ T1[(x >> 16) & 0xFFu] ^ T0[x >> 24]; }
      for (; i < n; ++i) r = T0[(r ^ in[i]) & 0xFFu] ^ (r >> 8);
      c = r;
    }
    if (lane == 0) s_crc[wv] = c;
  }
  __syncthreads();
  // ---- where everybody's bits go: [3 bits BFINAL / BTYPE][header][steps in anchor order][end of block] ------------------------------------
  const uint32_t hdr_bits = code_w[kTHeaderAt];
  uint32_t total_bits = 3u + hdr_bits, my_base = 0;
#pragma unroll
  for (int k = 0; k < kTW * 4; ++k) { if (k == wv * 4) my_base = total_bits; total_bits += s_step_bits[k]; }
  const uint32_t eob_bits = (uint32_t)(code[256] >> 12);
  const uint32_t payload = (total_bits + eob_bits + 7u) >> 3;
  const bool store = payload >= n + 5u || n == 0u;              // uniform over the block
  uint8_t* const o = slots + blk * (uint64_t)kSlotBytes;
  if (!store) {
    // ---- PASS 2: the tokens into the image ---------------------------------------------------------------------------------------------
    if (wv == 0) {
      if (lane == 0) { atomicOr(&image[0], 5u); image_or(image, total_bits, (uint32_t)code[256] & 0xFFFu, eob_bits); }      // BFINAL = 1, BTYPE = 10; end of block
      for (uint32_t w = lane; w * 32u < hdr_bits; w += 64) {
        const uint32_t nb = hdr_bits - w * 32u < 32u ? hdr_bits - w * 32u : 32u;
        image_or(image, 3u + w * 32u, code_w[kTHeaderAt + 1 + w], nb);
      }
    }
    uint32_t step_base = my_base;
    int step = 0;
    for (uint32_t s0 = lo; s0 < hi; s0 += 64, ++step) {         // uniform
      const uint32_t j = s0 + (uint32_t)lane;
      const bool act = j < hi;
      const uint32_t a = act ? anch[j] : 0u, e = act ? anch[j + 1] : 0u;
      const uint32_t info = act ? minfo[j] : 0u;
      const uint32_t L = info & 127u, d = (info >> 8) & 8191u, B = info >> 21;
      bool cont = ((info >> 7) & 1u) != 0u;
      uint32_t T = L;
      if (__any((int)cont)) T = merged_run_length(L, cont, lane);      // (the same runs as in pass 1: nothing is cut a second time)
      const uint32_t incl = wave_incl_scan(B);
      uint32_t ob = step_base + incl - B;
      if (T) {
        uint32_t lsym, lextra, lnb, dsym, dextra, dnb;
        match_symbols(T, d, lsym, lextra, lnb, dsym, dextra, dnb);
        const uint32_t lc = code[lsym], dc = code[288u + dsym];
        const uint32_t ln = lc >> 12, dn = dc >> 12;
        image_or(image, ob, (lc & 0xFFFu) | (lextra << ln), ln + lnb);                // <= 12 + 5 bits
        image_or(image, ob + ln + lnb, (dc & 0xFFFu) | (dextra << dn), dn + dnb);     // <= 12 + 13 bits
        ob += ln + lnb + dn + dnb;
      }
      const uint32_t ls = a + L, nl = act ? (e - a) - L : 0u;
      for (uint32_t t = 0; __any((int)(t < nl)); t += 4) {
        if (t < nl) {
          const uint32_t v = ld4u(in32, ls + t);
          const uint32_t k = nl - t < 4u ? nl - t : 4u;
          uint32_t b01 = 0, n01 = 0, b23 = 0, n23 = 0;           // two literals per word: <= 24 bits
          { const uint32_t c0 = code[v & 0xFFu]; b01 = c0 & 0xFFFu; n01 = c0 >> 12; }
          if (k > 1u) { const uint32_t c1 = code[(v >> 8) & 0xFFu]; b01 |= (c1 & 0xFFFu) << n01; n01 += c1 >> 12; }
          if (k > 2u) { const uint32_t c2 = code[(v >> 16) & 0xFFu]; b23 = c2 & 0xFFFu; n23 = c2 >> 12; }
          if (k > 3u) { const uint32_t c3 = code[v >> 24]; b23 |= (c3 & 0xFFFu) << n23; n23 += c3 >> 12; }
          image_or(image, ob, b01, n01);
          if (n23) image_or(image, ob + n01, b23, n23);
          ob += n01 + n23;
        }
      }
      step_base += s_step_bits[wv * 4 + step];
    }
  }
  __syncthreads();
  if (store) {
    if (tid == 0) { o[0] = 1; o[1] = (uint8_t)(n & 0xFFu); o[2] = (uint8_t)(n >> 8); o[3] = (uint8_t)(~n & 0xFFu); o[4] = (uint8_t)((~n >> 8) & 0xFFu); }
    for (uint32_t i = tid; i < n; i += kTThreads) o[5 + i] = in[i];
  } else {
    uint32_t* const ow = reinterpret_cast<uint32_t*>(o);
    for (uint32_t i = tid; i < (payload + 3u) >> 2; i += kTThreads) ow[i] = image[i];
  }
  if (tid == 0) {
    const uint32_t crc = (s_crc[0] ^ s_crc[1] ^ s_crc[2] ^ s_crc[3]) ^ 0xFFFFFFFFu;
    const uint32_t p0 = store ? n + 5u : payload;
    for (int k = 0; k < kParts; ++k) { csize[kParts * blk + k] = k ? 0u : p0; coff[kParts * blk + k] = 0u; }
    bsize[blk] = (uint64_t)p0 + kBgzfHeaderBytes + kBgzfTrailerBytes; crc_out[blk] = crc;
  }
}

// `len` bytes from the 4-byte aligned `s` to the arbitrary `d`, by one wavefront: aligned words of the destination assembled from two aligned source words
__device__ __forceinline__ void pack_copy(const uint8_t* s, uint8_t* d, uint32_t c, int lane) {
  if (c == 0u) return;
  const uint32_t head = (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) < c ? (uint32_t)((4u - ((uintptr_t)d & 3u)) & 3u) : c;
  if ((uint32_t)lane < head) d[lane] = s[lane];
  const uint32_t nw = (c - head) >> 2;
  const uint32_t* sw = reinterpret_cast<const uint32_t*>(s);
  uint32_t* dw = reinterpret_cast<uint32_t*>(d + head);
  for (uint32_t i = lane; i < nw; i += 64) {
    const uint32_t lo = sw[i], hi = sw[i + 1];                // (the slot has room behind the payload)
    dw[i] = head ? __builtin_amdgcn_alignbyte(hi, lo, head) : lo;
  }
  const uint32_t tail_at = head + (nw << 2);
  if ((uint32_t)lane < c - tail_at) d[tail_at + lane] = s[tail_at + lane];
}
__global__ void __launch_bounds__(64) k_bgzf_pack(const uint8_t* __restrict__ slots, const uint32_t* __restrict__ csize, const uint32_t* __restrict__ coff, const uint64_t* __restrict__ boff,
                                                  const uint32_t* __restrict__ crc, uint64_t n_total, uint8_t* __restrict__ dst, uint32_t kBgzfBlockInput) {
  const uint64_t blk = blockIdx.x;
  const int lane = threadIdx.x;
  uint32_t c = 0;                                                               // the block's payload: up to kParts pieces of the slot
  for (int k = 0; k < kParts; ++k) c += csize[kParts * blk + k];
  uint8_t* o = dst + boff[blk];
  const uint64_t base = blk * (uint64_t)kBgzfBlockInput;
  const uint32_t n = (uint32_t)((n_total - base) < (uint64_t)kBgzfBlockInput ? (n_total - base) : (uint64_t)kBgzfBlockInput);
  const uint32_t bs = c + kBgzfHeaderBytes + kBgzfTrailerBytes - 1u;      // BSIZE = total block size - 1
  if (lane < 18) {
    const uint8_t hdr[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, (uint8_t)(bs & 0xFFu), (uint8_t)(bs >> 8)};
    o[lane] = hdr[lane];
  } else if (lane < 26) {
    const uint32_t v = lane < 22 ? crc[blk] : n;
    o[kBgzfHeaderBytes + c + (uint32_t)(lane - 18)] = (uint8_t)((v >> (8 * ((lane - 18) & 3))) & 0xFFu);
  }
  const uint8_t* s = slots + blk * (uint64_t)slot_bytes(kBgzfBlockInput);
  uint32_t at = 0;
  for (int k = 0; k < kParts; ++k) {                                            // (uniform)
    const uint32_t ck = csize[kParts * blk + k];
    pack_copy(s + coff[kParts * blk + k], o + kBgzfHeaderBytes + at, ck, lane);   // (part offsets are multiples of 4)
    at += ck;
  }
}

// ---- CRC-32 tables (reflected polynomial 0xEDB88320, the gzip CRC) -------------------------------------------------------------
// the text kernel's code as it reads it: u16 (bits << 12 | code, bit-reversed for the LSB-first stream) for the 286 literal / length symbols at
// [0, 286) and the 30 distance symbols at [288, 318), then at word kTHeaderAt the header: its bit count and its bits.  The canonical codes follow
// from the lengths (RFC 1951 3.2.2); a set of lengths that is not a complete prefix code (Kraft sum != 1) would make every stream unreadable: refused.
std::vector<uint32_t> build_text_code() {
  std::vector<uint32_t> words(kTCodeWords, 0u);
  uint16_t* c16 = reinterpret_cast<uint16_t*>(words.data());
  auto canonical = [&](const uint8_t* bits, int nsym, int at) {
    uint32_t count[16] = {0}, next[16] = {0};
    uint64_t kraft = 0;
    for (int i = 0; i < nsym; ++i) { if (bits[i] > 12) throw std::runtime_error("BGZF: text code longer than 12 bits"); if (bits[i]) { ++count[bits[i]]; kraft += 1ull << (12 - bits[i]); } }
    if (kraft != (1ull << 12)) throw std::runtime_error("BGZF: the text code's lengths are not a complete prefix code");
    uint32_t code = 0;
    for (int b = 1; b <= 12; ++b) { code = (code + count[b - 1]) << 1; next[b] = code; }
    for (int i = 0; i < nsym; ++i) {
      const uint32_t len = bits[i];
      if (!len) { c16[at + i] = 0; continue; }
      const uint32_t cw = next[len]++;
      uint32_t rev = 0;
      for (uint32_t k = 0; k < len; ++k) rev |= ((cw >> k) & 1u) << (len - 1 - k);
      c16[at + i] = (uint16_t)((len << 12) | rev);
    }
  };
  canonical(kTextLitLenBits, 286, 0);
  canonical(kTextDistBits, 30, 288);
  if ((kTextHeaderNBits + 31) / 32 + kTHeaderAt + 1 > (uint32_t)kTCodeWords) throw std::runtime_error("BGZF: text code header too long");
  words[kTHeaderAt] = kTextHeaderNBits;
  for (uint32_t i = 0; i < sizeof(kTextHeader); ++i) words[kTHeaderAt + 1 + i / 4] |= (uint32_t)kTextHeader[i] << (8 * (i & 3));
  return words;
}

void build_crc_tables(std::vector<uint32_t>& slice, std::vector<uint32_t>& shift, int block_bytes, int lanes = 64) {
  slice.assign(4 * 256, 0);
  for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; slice[i] = c; }
  for (uint32_t i = 0; i < 256; ++i) for (int t = 1; t < 4; ++t) { const uint32_t prev = slice[(t - 1) * 256 + i]; slice[t * 256 + i] = (prev >> 8) ^ slice[prev & 0xFFu]; }
  // lane i takes the dwords [i * pw, min((i + 1) * pw, nw)) of a full block (pw = nw / 64 + 1, as in the kernel) and is followed by
  // behind(i) bytes: its register has to be advanced over that many zero bytes.  cols = the images of the 32 unit vectors under
  // "advance over k zero bytes", stepped from k = 0 upwards; a lane's table is written when k reaches its count.
  auto zero_byte = [&](uint32_t r) { return slice[r & 0xFFu] ^ (r >> 8); };
  const int nw = block_bytes / 4, pw = nw / lanes + 1;
  std::vector<int> behind(lanes);
  for (int lane = 0; lane < lanes; ++lane) { const int end = std::min((lane + 1) * pw, nw), begin = std::min(lane * pw, nw); behind[lane] = begin < end ? 4 * (nw - end) : -1; }
  shift.assign((size_t)lanes * 1024, 0);   // (a lane without dwords contributes nothing: its register stays 0, any table will do)
  uint32_t cols[32];
  for (int j = 0; j < 32; ++j) cols[j] = 1u << j;
  auto apply = [&](uint32_t v) { uint32_t o = 0; for (int j = 0; j < 32; ++j) if ((v >> j) & 1u) o ^= cols[j]; return o; };
  for (int k = 0; k <= block_bytes; ++k) {
    for (int lane = 0; lane < lanes; ++lane) if (behind[lane] == k)
      for (int t = 0; t < 4; ++t) for (uint32_t b = 0; b < 256; ++b) shift[(size_t)lane * 1024 + t * 256 + b] = apply(b << (8 * t));
    for (int j = 0; j < 32; ++j) cols[j] = zero_byte(cols[j]);
  }
}

}  // namespace

uint32_t bgzf_block_input() {
  static const uint32_t v = []() { const char* e = getenv("GDBAMD_BGZF_BLOCK"); const int v = e ? atoi(e) : 0; return v == 16384 ? 16384u : v == 4096 ? 4096u : v == 6144 ? 6144u : 8192u; }();
  return v;
}

// wavefronts per 8 KiB block: 2 (default: k_bgzf_deflate2) or 1 (GDBAMD_BGZF_WAVES=1: the kernel of rounds 3-4, for A/B runs)
// pages of VCF text through the anchored kernel (k_bgzf_deflate_text, 8 KiB blocks only); GDBAMD_BGZF_TEXT=0: the byte-level kernel for everything (A/B runs)
static bool bgzf_text_kernel() { static const bool v = []() { const char* e = getenv("GDBAMD_BGZF_TEXT"); return !(e && *e == '0'); }(); return v; }
static int bgzf_waves_per_block() { static const int v = []() { const char* e = getenv("GDBAMD_BGZF_WAVES"); return e && *e == '1' ? 1 : 2; }(); return v; }

std::string bgzf_compress_host(const std::string& bytes) {
  std::string out;
  const size_t kHostBlock = 0xff00;                            // htslib's BGZF_BLOCK_SIZE
  for (size_t at = 0; at < bytes.size(); at += kHostBlock) {
    const size_t n = std::min<size_t>(kHostBlock, bytes.size() - at);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("BGZF: deflateInit2 failed");
    std::vector<unsigned char> buf(deflateBound(&zs, (uLong)n) + 16);
    zs.next_in = (Bytef*)bytes.data() + at; zs.avail_in = (uInt)n;
    zs.next_out = buf.data(); zs.avail_out = (uInt)buf.size();
    const int rc = deflate(&zs, Z_FINISH);
    const size_t c = buf.size() - zs.avail_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) throw std::runtime_error("BGZF: deflate failed");
    const uint32_t bs = (uint32_t)(c + kBgzfHeaderBytes + kBgzfTrailerBytes - 1);
    if (bs > 0xFFFFu) throw std::runtime_error("BGZF: block too large");
    const unsigned char hdr[18] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0, (unsigned char)(bs & 0xFFu), (unsigned char)(bs >> 8)};
    out.append((const char*)hdr, 18);
    out.append((const char*)buf.data(), c);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)bytes.data() + at, (uInt)n), isize = (uint32_t)n;
    out.append((const char*)&crc, 4);
    out.append((const char*)&isize, 4);
  }
  return out;
}

struct BgzfDeviceCompressor::Impl {
  uint32_t* d_slice = nullptr; uint32_t* d_shift = nullptr; uint32_t* d_shift256 = nullptr; uint32_t block = 0;
  uint32_t* coff = nullptr; uint32_t* d_text_code = nullptr;
  bool text = false, bcf2 = false;
  uint8_t* slots = nullptr; size_t slots_cap = 0;
  uint32_t* csize = nullptr; uint32_t* crc = nullptr; uint64_t* bsize = nullptr; uint64_t* boff = nullptr; size_t blocks_cap = 0;
  void* temp = nullptr; size_t temp_cap = 0;
  // two jobs may be queued behind each other on one stream (the stream's two page arenas); they share the scratch buffers -
  // stream order keeps them apart - and have their own events and their own pinned word for the size of the result
  struct Job { hipEvent_t ev0 = nullptr, ev1 = nullptr, done = nullptr; bool pending = false, used = false; } job[2];
  uint64_t* h_total = nullptr;          // pinned, [2]
  void release() {
    for (void* p : {(void*)d_slice, (void*)d_shift, (void*)d_shift256, (void*)d_text_code, (void*)slots, (void*)csize, (void*)coff, (void*)crc, (void*)bsize, (void*)boff, temp}) if (p) (void)hipFree(p);
    for (Job& j : job) for (hipEvent_t e : {j.ev0, j.ev1, j.done}) if (e) (void)hipEventDestroy(e);
    if (h_total) (void)hipHostFree(h_total);
  }
};

BgzfDeviceCompressor::BgzfDeviceCompressor() : m_(new Impl) {}
BgzfDeviceCompressor::~BgzfDeviceCompressor() { m_->release(); delete m_; }

void BgzfDeviceCompressor::enqueue(int slot, const char* dev_src, uint64_t n, char* dev_dst, void* hip_stream) {
  slot &= 1;
  Impl& S = *m_;
  // (a job still pending in this slot was abandoned by its owner: the new one simply queues behind it)
  if (n == 0) throw std::runtime_error("BGZF: empty input");
  if ((uintptr_t)dev_src & 15u) throw std::runtime_error("BGZF: the page is not 16-byte aligned");
  hipStream_t st = (hipStream_t)hip_stream;
  if (!S.d_slice) {
    std::vector<uint32_t> slice, shift;
    S.block = bgzf_block_input();
    build_crc_tables(slice, shift, (int)S.block);
    BGZF_HIP(hipMalloc((void**)&S.d_slice, slice.size() * 4));
    BGZF_HIP(hipMalloc((void**)&S.d_shift, shift.size() * 4));
    BGZF_HIP(hipMemcpy(S.d_slice, slice.data(), slice.size() * 4, hipMemcpyHostToDevice));
    BGZF_HIP(hipMemcpy(S.d_shift, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
    if (S.block == 8192u) {                                     // the text kernel's CRC: 256 threads x 9 dwords
      build_crc_tables(slice, shift, 8192, kTThreads);
      BGZF_HIP(hipMalloc((void**)&S.d_shift256, shift.size() * 4));
      BGZF_HIP(hipMemcpy(S.d_shift256, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
      const std::vector<uint32_t> tc = build_text_code();
      BGZF_HIP(hipMalloc((void**)&S.d_text_code, tc.size() * 4));
      BGZF_HIP(hipMemcpy(S.d_text_code, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
    }
    for (Impl::Job& j : S.job) {
      BGZF_HIP(hipEventCreate(&j.ev0));
      BGZF_HIP(hipEventCreate(&j.ev1));
      BGZF_HIP(hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
    }
    BGZF_HIP(hipHostMalloc((void**)&S.h_total, 2 * sizeof(uint64_t), hipHostMallocDefault));
  }
  const uint32_t kBgzfBlockInput = S.block;
  const uint64_t nblocks = (n + kBgzfBlockInput - 1) / kBgzfBlockInput;
  if (nblocks >= (1ull << 31)) throw std::runtime_error("BGZF: page too large");
  // (growing a scratch buffer frees memory the other queued job may still use: the stream has to be idle for that)
  if (S.blocks_cap < nblocks + 1) {
    BGZF_HIP(hipStreamSynchronize(st));
    for (void* p : {(void*)S.csize, (void*)S.coff, (void*)S.crc, (void*)S.bsize, (void*)S.boff}) if (p) (void)hipFree(p);
    const size_t cap = (size_t)nblocks + (size_t)(nblocks >> 3) + 64;
    BGZF_HIP(hipMalloc((void**)&S.csize, cap * 4 * kParts)); BGZF_HIP(hipMalloc((void**)&S.coff, cap * 4 * kParts)); BGZF_HIP(hipMalloc((void**)&S.crc, cap * 4));   // (kParts payload parts per block)
    BGZF_HIP(hipMalloc((void**)&S.bsize, cap * 8)); BGZF_HIP(hipMalloc((void**)&S.boff, cap * 8));
    S.blocks_cap = cap;
  }
  const size_t need_slots = (size_t)nblocks * slot_bytes(kBgzfBlockInput) + 64;
  if (S.slots_cap < need_slots) {
    BGZF_HIP(hipStreamSynchronize(st));
    if (S.slots) (void)hipFree(S.slots);
    S.slots = nullptr; S.slots_cap = 0;
    BGZF_HIP(hipMalloc((void**)&S.slots, need_slots + (need_slots >> 4)));
    S.slots_cap = need_slots + (need_slots >> 4);
  }
  size_t bytes = 0;
  BGZF_HIP(rocprim::exclusive_scan(nullptr, bytes, S.bsize, S.boff, (uint64_t)0, (size_t)nblocks + 1, rocprim::plus<uint64_t>(), st));
  if (S.temp_cap < bytes) {
    BGZF_HIP(hipStreamSynchronize(st));
    if (S.temp) (void)hipFree(S.temp);
    BGZF_HIP(hipMalloc(&S.temp, bytes + 256));
    S.temp_cap = bytes + 256;
  }
  Impl::Job& J = S.job[slot];
  BGZF_HIP(hipEventRecord(J.ev0, st));
  if (kBgzfBlockInput == 16384u)
    hipLaunchKernelGGL(k_bgzf_deflate<16384>, dim3((unsigned)nblocks), dim3(64), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift);
  else if (kBgzfBlockInput == 6144u)
    hipLaunchKernelGGL(k_bgzf_deflate<6144>, dim3((unsigned)nblocks), dim3(64), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift);
  else if (kBgzfBlockInput == 4096u)
    hipLaunchKernelGGL(k_bgzf_deflate<4096>, dim3((unsigned)nblocks), dim3(64), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift);
  else if (S.text && bgzf_text_kernel())
    hipLaunchKernelGGL(k_bgzf_deflate_text<8192>, dim3((unsigned)nblocks), dim3(kTThreads), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift256, (const uint32_t*)S.d_text_code);
  else if (bgzf_waves_per_block() >= 2 && S.bcf2)
    hipLaunchKernelGGL((k_bgzf_deflate2<8192, 8>), dim3((unsigned)nblocks), dim3(128), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift);
  else if (bgzf_waves_per_block() >= 2)
    hipLaunchKernelGGL(k_bgzf_deflate2<8192>, dim3((unsigned)nblocks), dim3(128), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift);
  else
    hipLaunchKernelGGL(k_bgzf_deflate<8192>, dim3((unsigned)nblocks), dim3(64), 0, st, (const uint8_t*)dev_src, n, S.slots, S.csize, S.coff, S.bsize, S.crc, (const uint32_t*)S.d_slice,
                       (const uint32_t*)S.d_shift);
  BGZF_HIP(hipMemsetAsync(S.bsize + nblocks, 0, sizeof(uint64_t), st));
  BGZF_HIP(rocprim::exclusive_scan(S.temp, bytes, S.bsize, S.boff, (uint64_t)0, (size_t)nblocks + 1, rocprim::plus<uint64_t>(), st));
  hipLaunchKernelGGL(k_bgzf_pack, dim3((unsigned)nblocks), dim3(64), 0, st, (const uint8_t*)S.slots, (const uint32_t*)S.csize, (const uint32_t*)S.coff, (const uint64_t*)S.boff, (const uint32_t*)S.crc, n,
                     (uint8_t*)dev_dst, kBgzfBlockInput);
  BGZF_HIP(hipEventRecord(J.ev1, st));
  BGZF_HIP(hipMemcpyAsync(S.h_total + slot, S.boff + nblocks, sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  BGZF_HIP(hipEventRecord(J.done, st));
  J.pending = true; J.used = true;
}

void BgzfDeviceCompressor::set_text(bool pages_are_vcf_text) { m_->text = pages_are_vcf_text; }
void BgzfDeviceCompressor::set_bcf2(bool pages_are_bcf2_records) { m_->bcf2 = pages_are_bcf2_records; }

uint64_t BgzfDeviceCompressor::finish(int slot, float* ms_kernels) {
  slot &= 1;
  Impl& S = *m_;
  Impl::Job& J = S.job[slot];
  if (!J.pending) throw std::runtime_error("BGZF: no job queued in this slot");
  J.pending = false;
  BGZF_HIP(hipEventSynchronize(J.done));
  if (ms_kernels) BGZF_HIP(hipEventElapsedTime(ms_kernels, J.ev0, J.ev1));
  return S.h_total[slot];
}

void BgzfDeviceCompressor::cancel(int slot) { m_->job[slot & 1].pending = false; }
void* BgzfDeviceCompressor::done_event(int slot) const { return m_->job[slot & 1].used ? (void*)m_->job[slot & 1].done : nullptr; }

uint64_t BgzfDeviceCompressor::compress(const char* dev_src, uint64_t n, char* dev_dst, void* hip_stream, float* ms_kernels) {
  if (ms_kernels) *ms_kernels = 0;
  if (n == 0) return 0;
  enqueue(0, dev_src, n, dev_dst, hip_stream);
  return finish(0, ms_kernels);
}

}  // namespace genomicsdb_amd
