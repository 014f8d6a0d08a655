#include "vcf_index.h"

#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <vector>

#include "../kernels/gdb_bgzf.h"

namespace genomicsdb_amd {
namespace {

constexpr int kMinShift = 14, kDepth = 5;

// bin of [beg, end) in the binning scheme of min_shift / depth (SAM specification 5.3; CSIv1 generalises the constants)
int reg2bin(int64_t beg, int64_t end) {
  int l, s = kMinShift, t = ((1 << (kDepth * 3)) - 1) / 7;
  for (--end, l = kDepth; l > 0; --l, s += 3, t -= 1 << (l * 3))
    if (beg >> s == end >> s) return t + (int)(beg >> s);
  return 0;
}
int bin_first(int l) { return ((1 << (3 * l)) - 1) / 7; }
// level of a bin and the first position it covers -> used for CSI's per-bin loffset
struct Chunk { uint64_t beg, end; };
struct RefIndex {
  std::map<uint32_t, std::vector<Chunk>> bins;
  std::map<uint32_t, uint64_t> loffset;        // CSI: virtual offset of the first record that overlaps the bin's first 16 kb window (fill_linear)
  std::vector<uint64_t> linear;                // 16 kb windows -> virtual offset of the first record overlapping the window
  uint64_t off_beg = ~0ull, off_end = 0, n_mapped = 0;
  uint32_t last_bin = ~0u;
};

struct BgzfReader {
  FILE* f = nullptr;
  uint64_t coff = 0;              // file offset of the block in `buf`
  uint64_t next_coff = 0;
  std::vector<uint8_t> buf;       // inflated bytes of the current block
  size_t at = 0;
  bool eof = false;
  explicit BgzfReader(const std::string& path) {
    f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("index: cannot open " + path);
  }
  ~BgzfReader() { if (f) fclose(f); }
  bool next_block() {
    for (;;) {
      uint8_t h[18];
      coff = next_coff;
      const size_t n = fread(h, 1, 18, f);
      if (n == 0) { eof = true; buf.clear(); at = 0; return false; }
      if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4) || h[12] != 'B' || h[13] != 'C') throw std::runtime_error("index: not a BGZF block");
      const uint32_t bsize = (uint32_t)h[16] | ((uint32_t)h[17] << 8);
      std::vector<uint8_t> comp(bsize + 1 - 18);
      if (fread(comp.data(), 1, comp.size(), f) != comp.size()) throw std::runtime_error("index: truncated BGZF block");
      next_coff = coff + bsize + 1;
      const uint32_t isize = (uint32_t)comp[comp.size() - 4] | ((uint32_t)comp[comp.size() - 3] << 8) | ((uint32_t)comp[comp.size() - 2] << 16) | ((uint32_t)comp[comp.size() - 1] << 24);
      buf.resize(isize);
      at = 0;
      if (isize == 0) continue;   // (the EOF block, or an empty one)
      z_stream z;
      memset(&z, 0, sizeof(z));
      if (inflateInit2(&z, -15) != Z_OK) throw std::runtime_error("index: inflateInit2");
      z.next_in = comp.data(); z.avail_in = (uInt)(comp.size() - 8);
      z.next_out = buf.data(); z.avail_out = isize;
      const int rc = inflate(&z, Z_FINISH);
      inflateEnd(&z);
      if (rc != Z_STREAM_END || z.avail_out != 0) throw std::runtime_error("index: a BGZF block does not inflate to its ISIZE");
      return true;
    }
  }
  // virtual offset of the next byte to be read
  uint64_t tell() { if (at >= buf.size() && !eof) next_block(); return eof ? (next_coff << 16) : ((coff << 16) | (uint64_t)at); }
  int getc() { if (at >= buf.size()) { if (eof || !next_block()) return -1; } return buf[at++]; }
  bool read(void* dst, size_t n) { uint8_t* d = (uint8_t*)dst; while (n) { if (at >= buf.size()) { if (eof || !next_block()) return false; } const size_t k = std::min(n, buf.size() - at); memcpy(d, buf.data() + at, k); d += k; at += k; n -= k; } return true; }
};

void note_record(std::vector<RefIndex>& refs, int tid, int64_t beg, int64_t end, uint64_t vbeg, uint64_t vend) {
  if (tid < 0) return;
  if ((size_t)tid >= refs.size()) refs.resize((size_t)tid + 1);
  RefIndex& r = refs[(size_t)tid];
  if (end <= beg) end = beg + 1;
  const uint32_t bin = (uint32_t)reg2bin(beg, end);
  std::vector<Chunk>& ch = r.bins[bin];
  if (r.last_bin == bin && !ch.empty()) ch.back().end = vend;       // records of one bin in a row: one chunk
  else ch.push_back(Chunk{vbeg, vend});
  r.last_bin = bin;
  const size_t w0 = (size_t)(beg >> kMinShift), w1 = (size_t)((end - 1) >> kMinShift);
  if (r.linear.size() <= w1) r.linear.resize(w1 + 1, 0);
  for (size_t w = w0; w <= w1; ++w) if (r.linear[w] == 0) r.linear[w] = vbeg;
  r.off_beg = std::min(r.off_beg, vbeg);
  r.off_end = std::max(r.off_end, vend);
  ++r.n_mapped;
}
// htslib's update_loff (hts.c) AS IT WAS IN THE 1.x RELEASES OF THE REFERENCE'S TIME (its dependencies/htslib is a 2016-17 fork whose
// pinned commit cannot be read here: SURVEY 8(c)): windows in front of a contig's first record take the offset of that record, every
// other window without a record the offset of the window BEFORE it.  htslib 1.10 and later fill an empty window from the window BEHIND
// it ("the last entry is always valid").  Both are lower bounds of the first overlapping record's offset, so every reader finds the same
// records with either; only the stored values of empty windows (and of bins that begin in one) differ - byte identity with a given
// htslib build's .tbi / .csi is therefore NOT claimed for files with empty 16 kb windows, query equivalence is (tests/test_vcf_index.py).
// Then the loffset of a bin = the linear offset of the bin's FIRST 16 kb
// window, i.e. the offset of the first record that OVERLAPS that window - not of the first record that begins in the bin (a
// reference block that begins in the window before and reaches into it comes first in the file, and hts_itr_query drops every
// chunk that ends at or below the loffset)
int bin_level(uint32_t bin) { int l = 0; for (uint32_t b = bin; b; b = (b - 1) >> 3) ++l; return l; }
void fill_linear(std::vector<RefIndex>& refs) {
  for (RefIndex& r : refs) {
    size_t i = 0;
    for (; i < r.linear.size() && r.linear[i] == 0; ++i) r.linear[i] = r.n_mapped ? r.off_beg : 0;
    for (; i < r.linear.size(); ++i) if (r.linear[i] == 0) r.linear[i] = r.linear[i - 1];
    r.loffset.clear();
    for (const auto& b : r.bins) {
      const int l = bin_level(b.first);
      const uint64_t bot = (uint64_t)(b.first - (uint32_t)bin_first(l)) << ((kDepth - l) * 3);
      r.loffset[b.first] = bot < r.linear.size() ? r.linear[bot] : 0;
    }
  }
}
template <class T> void put(std::string& o, T v) { o.append((const char*)&v, sizeof(T)); }
constexpr uint32_t kMetaBin = ((1u << ((kDepth + 1) * 3)) - 1u) / 7u + 1u;   // 37450: htslib's pseudo-bin with a contig's offsets and record count

void write_bgzf(const std::string& path, const std::string& bytes) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("index: cannot write " + path);
  const std::string blk = bgzf_compress_host(bytes);          // (blocks of 0xff00 input bytes, like htslib's)
  if (fwrite(blk.data(), 1, blk.size(), f) != blk.size()) { fclose(f); throw std::runtime_error("index: short write"); }
  (void)fwrite(kBgzfEofBlock, 1, sizeof(kBgzfEofBlock), f);
  fclose(f);
}

}  // namespace

void build_tbi_index(const std::string& path) {
  BgzfReader in(path);
  std::vector<RefIndex> refs;
  std::vector<std::string> names;
  std::map<std::string, int> name_id;
  std::string line;
  for (;;) {
    const uint64_t vbeg = in.tell();
    if (in.eof) break;
    line.clear();
    int c;
    while ((c = in.getc()) >= 0 && c != '\n') line.push_back((char)c);
    if (c < 0 && line.empty()) break;
    const uint64_t vend = in.tell();
    if (line.empty() || line[0] == '#') continue;
    // CHROM \t POS \t ID \t REF \t ALT \t QUAL \t FILTER \t INFO
    size_t t[8], k = 0;
    for (size_t i = 0; i < line.size() && k < 8; ++i) if (line[i] == '\t') t[k++] = i;
    if (k < 4) continue;
    const std::string chrom = line.substr(0, t[0]);
    const int64_t pos = atoll(line.c_str() + t[0] + 1);
    int64_t end = pos - 1 + (int64_t)(t[3] - t[2] - 1);              // beg + length of REF
    if (k >= 7) {                                                     // INFO END=<n> overrides (tabix's VCF preset)
      const size_t ib = t[6] + 1, ie = k >= 8 ? t[7] : line.size();
      size_t p = ib;
      while (p < ie) {
        size_t q = line.find(';', p);
        if (q == std::string::npos || q > ie) q = ie;
        if (q - p > 4 && line.compare(p, 4, "END=") == 0) { end = atoll(line.c_str() + p + 4); break; }
        p = q + 1;
      }
    }
    auto it = name_id.find(chrom);
    int tid;
    if (it == name_id.end()) { tid = (int)names.size(); name_id[chrom] = tid; names.push_back(chrom); } else tid = it->second;
    note_record(refs, tid, pos - 1, end, vbeg, vend);
  }
  fill_linear(refs);
  refs.resize(names.size());
  std::string o;
  o.append("TBI\1", 4);
  put<int32_t>(o, (int32_t)names.size());
  put<int32_t>(o, 2);                                                 // format: VCF
  put<int32_t>(o, 1); put<int32_t>(o, 2); put<int32_t>(o, 0);        // col_seq, col_beg, col_end
  put<int32_t>(o, '#'); put<int32_t>(o, 0);                           // meta, skip
  std::string nm;
  for (const std::string& n : names) { nm += n; nm.push_back('\0'); }
  put<int32_t>(o, (int32_t)nm.size());
  o += nm;
  for (const RefIndex& r : refs) {
    put<int32_t>(o, (int32_t)r.bins.size() + (r.n_mapped ? 1 : 0));
    for (const auto& b : r.bins) {
      put<uint32_t>(o, b.first); put<int32_t>(o, (int32_t)b.second.size());
      for (const Chunk& c : b.second) { put<uint64_t>(o, c.beg); put<uint64_t>(o, c.end); }
    }
    if (r.n_mapped) { put<uint32_t>(o, kMetaBin); put<int32_t>(o, 2); put<uint64_t>(o, r.off_beg); put<uint64_t>(o, r.off_end); put<uint64_t>(o, r.n_mapped); put<uint64_t>(o, 0); }
    put<int32_t>(o, (int32_t)r.linear.size());
    for (uint64_t v : r.linear) put<uint64_t>(o, v);
  }
  put<uint64_t>(o, 0);                                                // n_no_coor
  write_bgzf(path + ".tbi", o);
}

void build_csi_index(const std::string& path) {
  BgzfReader in(path);
  char magic[5];
  uint32_t l_text = 0;
  if (!in.read(magic, 5) || memcmp(magic, "BCF\2\2", 5) != 0 || !in.read(&l_text, 4)) throw std::runtime_error("index: " + path + " is not a BCF2 file");
  std::string text(l_text, '\0');
  if (!in.read(&text[0], l_text)) throw std::runtime_error("index: truncated BCF header");
  int n_ref = 0;
  for (size_t p = 0; (p = text.find("##contig=", p)) != std::string::npos; ++p) ++n_ref;
  std::vector<RefIndex> refs;
  for (;;) {
    const uint64_t vbeg = in.tell();
    if (in.eof) break;
    uint32_t l_shared, l_indiv;
    if (!in.read(&l_shared, 4)) break;
    if (!in.read(&l_indiv, 4)) throw std::runtime_error("index: truncated BCF record");
    std::vector<uint8_t> rec((size_t)l_shared + l_indiv);
    if (!in.read(rec.data(), rec.size()) || l_shared < 12) throw std::runtime_error("index: truncated BCF record");
    int32_t chrom, pos, rlen;
    memcpy(&chrom, rec.data(), 4); memcpy(&pos, rec.data() + 4, 4); memcpy(&rlen, rec.data() + 8, 4);
    const uint64_t vend = in.tell();
    note_record(refs, chrom, pos, (int64_t)pos + std::max(1, rlen), vbeg, vend);
  }
  if ((int)refs.size() < n_ref) refs.resize((size_t)n_ref);
  fill_linear(refs);
  std::string o;
  o.append("CSI\1", 4);
  put<int32_t>(o, kMinShift); put<int32_t>(o, kDepth); put<int32_t>(o, 0);   // l_aux = 0
  put<int32_t>(o, (int32_t)refs.size());
  for (const RefIndex& r : refs) {
    put<int32_t>(o, (int32_t)r.bins.size() + (r.n_mapped ? 1 : 0));
    for (const auto& b : r.bins) {
      put<uint32_t>(o, b.first);
      auto lo = r.loffset.find(b.first);
      put<uint64_t>(o, lo == r.loffset.end() ? 0 : lo->second);
      put<int32_t>(o, (int32_t)b.second.size());
      for (const Chunk& c : b.second) { put<uint64_t>(o, c.beg); put<uint64_t>(o, c.end); }
    }
    if (r.n_mapped) { put<uint32_t>(o, kMetaBin); put<uint64_t>(o, 0); put<int32_t>(o, 2); put<uint64_t>(o, r.off_beg); put<uint64_t>(o, r.off_end); put<uint64_t>(o, r.n_mapped); put<uint64_t>(o, 0); }
  }
  put<uint64_t>(o, 0);
  write_bgzf(path + ".csi", o);
}

}  // namespace genomicsdb_amd
