// gdb_bgzf.h - BGZF (blocked gzip) output of the combined-gVCF stream, compressed ON THE DEVICE.
//
// The reference's VCFAdapter opens its output through htslib with mode "w" + vcf_output_format: "z" = BGZF-compressed VCF text,
// "b" = BGZF-compressed BCF2 (reference src/main/cpp/src/vcf/vcf_adapter.cc:340-372, src/config/genomicsdb_config_base.cc:34,
// 156-165).  Here the finished page (text or BCF2 records, in HBM) is cut into 8 / 16 KiB pieces, every piece becomes one BGZF
// block - gzip member with the 'BC' extra field, RFC 1952 + the SAM specification section 4.1 - deflated by ONE wavefront
// (LZ77 over the piece held in LDS + fixed-Huffman codes, RFC 1951), and only the compressed bytes leave the GPU: the stream is
// PCIe-bound, so this is the lever on what a caller sees (the same page assembly, ~7x fewer bytes over the link).
// Compressed bytes are not comparable with htslib's (any valid DEFLATE of the same bytes is a valid file); parity is defined
// on the inflated stream, which equals the "" / "bu" stream byte for byte, and on the BGZF framing (EOF block included).
#pragma once
#include <cstdint>
#include <string>

namespace genomicsdb_amd {

// uncompressed bytes per block: 64 lanes x 64, 128 or 256 bytes (GDBAMD_BGZF_BLOCK = 4096 / 6144 / 8192 / 16384; smaller blocks leave room for more
// resident wavefronts, larger ones find a little more to match)
uint32_t bgzf_block_input();
constexpr uint32_t kBgzfHeaderBytes = 18, kBgzfTrailerBytes = 8;
extern const unsigned char kBgzfEofBlock[28];               // the empty block that ends every BGZF file

// worst case of the compressed stream of n bytes (every block stored: 5 bytes of DEFLATE framing + header + trailer)
inline uint64_t bgzf_bound(uint64_t n) {
  const uint64_t nblocks = (n + 4096 - 1) / 4096;      // (the smallest block size there is)
  return n + nblocks * (kBgzfHeaderBytes + kBgzfTrailerBytes + 5) + 64;
}

// BGZF blocks of a host buffer (the VCF / BCF header; zlib on the host - a few KB once per stream)
std::string bgzf_compress_host(const std::string& bytes);

class BgzfDeviceCompressor {
 public:
  BgzfDeviceCompressor();
  ~BgzfDeviceCompressor();
  BgzfDeviceCompressor(const BgzfDeviceCompressor&) = delete;
  BgzfDeviceCompressor& operator=(const BgzfDeviceCompressor&) = delete;
  // Enqueues on `hip_stream` (a hipStream_t) the compression of the n bytes at dev_src (16-byte aligned, device memory) into
  // consecutive BGZF blocks at dev_dst (device memory of at least bgzf_bound(n) bytes; MAY BE dev_src itself: the packed stream
  // is written only after every piece has been read), then waits for the stream and returns the number of bytes at dev_dst.
  // ms_kernels (optional): device time of the two kernels.
  uint64_t compress(const char* dev_src, uint64_t n, char* dev_dst, void* hip_stream, float* ms_kernels = nullptr);
  // The same in two halves, so that the host need not wait between the kernels that produce a page and the ones that compress it:
  // enqueue() only queues work on the stream (n > 0), finish() waits for it and returns the size.  Two jobs (slot 0 / 1) may be
  // queued behind each other on ONE stream; cancel() forgets a queued job whose result nobody will ask for.
  void enqueue(int slot, const char* dev_src, uint64_t n, char* dev_dst, void* hip_stream);
  // The pages are VCF TEXT (output format "z"): matches are looked for where a column begins (tabs, newlines) instead of at every byte -
  // k_bgzf_deflate_text, about a third of the instructions per block at ~5 % of the ratio.  Any bytes still give a valid stream (the anchors
  // only decide how much is found), but binary pages (BCF2, "b") compress far better with the byte-level kernel, which stays the default.
  void set_text(bool pages_are_vcf_text);
  // The pages are BCF2 records ("b"): the byte-level kernel with smaller hash tables (more resident workgroups: 285 against 261 GB/s at the same
  // ratio on BCF2; text-like bytes would lose 2-3 % of ratio, which is why it is not the default for arbitrary input)
  void set_bcf2(bool pages_are_bcf2_records);
  uint64_t finish(int slot, float* ms_kernels = nullptr);
  void cancel(int slot);
  // the hipEvent_t recorded behind the slot's last job (nullptr: none yet): for a producer that reuses the job's buffers from ANOTHER stream
  void* done_event(int slot) const;
  struct Impl;
 private:
  Impl* m_;
};

}  // namespace genomicsdb_amd
