// store_bw.hip - what the chip sustains for STORE-dominated streaming (the page assembly is 97 % stores): plain fill, fill with
// non-temporal stores, read + write copy, and a fill in the page assembly's shape (every wavefront writes ~2.9 KB runs at an
// arbitrary byte alignment, head / tail bytes as byte stores).  Build: hipcc --offload-arch=gfx950 -O3 store_bw.hip -o store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_fill(uint4* __restrict__ p, size_t n16) {
  const uint4 v = make_uint4(1, 2, 3, 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_fill_nt(uint4* __restrict__ p, size_t n16) {
  const u32x4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p) + i);
}
__global__ void k_copy(const uint4* __restrict__ s, uint4* __restrict__ d, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void k_read(const uint4* __restrict__ s, size_t n16, uint32_t* out) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = s[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *out = acc;
}
// one wavefront per workgroup; wavefront w writes runs [w * run_bytes + skew, ...) of run_bytes (not a multiple of 16) back to back,
// `runs` of them at a stride of stride_bytes - the page assembly's pattern (a 64-sample chunk of a record, then the next record)
__global__ void __launch_bounds__(64) k_chunks(char* __restrict__ base, size_t run_bytes, size_t stride_bytes, int runs, int chunks_per_record) {
  const int lane = threadIdx.x;
  const size_t rec0 = (size_t)(blockIdx.x / chunks_per_record) * runs;
  const int ch = blockIdx.x % chunks_per_record;
  for (int r = 0; r < runs; ++r) {
    char* g = base + (rec0 + r) * stride_bytes + (size_t)ch * run_bytes;
    const uint32_t al = (uint32_t)((uintptr_t)g & 15u);
    uint32_t head = (16u - al) & 15u;
    const uint32_t nwords = (uint32_t)((run_bytes - head) >> 4);
    const uint32_t tail_at = head + (nwords << 4);
    if ((uint32_t)lane < head) g[lane] = (char)lane;
    uint4* gw = reinterpret_cast<uint4*>(g + head);
    const uint4 v = make_uint4(lane, r, 3, 4);
    for (uint32_t wq = lane; wq < nwords; wq += 64) gw[wq] = v;
    if ((uint32_t)lane < run_bytes - tail_at) g[tail_at + lane] = (char)lane;
  }
}

// variants of the same: `mode` 0 = as the page assembly maps blocks (chunk fast), 1 = XCD-aware (workgroups are dealt round-robin to
// the 8 XCDs: renumber so that the chunks of one record run share an XCD and its L2), 2 = one wavefront writes all chunks of its
// records itself (contiguous 44 KB per record); wpb = wavefronts per workgroup (each takes the next chunk)
__global__ void k_chunks2(char* __restrict__ base, size_t run_bytes, size_t stride_bytes, int runs, int chunks_per_record, int mode, unsigned nlogical) {
  const int lane = threadIdx.x & 63;
  const unsigned wpb = blockDim.x >> 6;
  unsigned bid = blockIdx.x;
  if (mode == 1) { const unsigned per = gridDim.x / 8u; bid = (blockIdx.x % 8u) * per + blockIdx.x / 8u; if (blockIdx.x >= per * 8u) bid = blockIdx.x; }
  const unsigned lid = bid * wpb + (threadIdx.x >> 6);
  if (lid >= nlogical) return;
  if (mode == 2) {
    const size_t rec0 = (size_t)lid * runs;
    for (int r = 0; r < runs; ++r)
      for (int ch = 0; ch < chunks_per_record; ++ch) {
        char* g = base + (rec0 + r) * stride_bytes + (size_t)ch * run_bytes;
        const uint32_t al = (uint32_t)((uintptr_t)g & 15u);
        uint32_t head = (16u - al) & 15u;
        const uint32_t nwords = (uint32_t)((run_bytes - head) >> 4);
        const uint32_t tail_at = head + (nwords << 4);
        if ((uint32_t)lane < head) g[lane] = (char)lane;
        uint4* gw = reinterpret_cast<uint4*>(g + head);
        const uint4 v = make_uint4(lane, r, 3, 4);
        for (uint32_t wq = lane; wq < nwords; wq += 64) gw[wq] = v;
        if ((uint32_t)lane < run_bytes - tail_at) g[tail_at + lane] = (char)lane;
      }
    return;
  }
  const size_t rec0 = (size_t)(lid / chunks_per_record) * runs;
  const int ch = lid % chunks_per_record;
  for (int r = 0; r < runs; ++r) {
    char* g = base + (rec0 + r) * stride_bytes + (size_t)ch * run_bytes;
    const uint32_t al = (uint32_t)((uintptr_t)g & 15u);
    uint32_t head = (16u - al) & 15u;
    const uint32_t nwords = (uint32_t)((run_bytes - head) >> 4);
    const uint32_t tail_at = head + (nwords << 4);
    if ((uint32_t)lane < head) g[lane] = (char)lane;
    uint4* gw = reinterpret_cast<uint4*>(g + head);
    const uint4 v = make_uint4(lane, r, 3, 4);
    for (uint32_t wq = lane; wq < nwords; wq += 64) gw[wq] = v;
    if ((uint32_t)lane < run_bytes - tail_at) g[tail_at + lane] = (char)lane;
  }
}

// Which XCD does workgroup b run on?  (XCC_ID hardware register, gfx940+.)  The XCD-aware numbering assumes blockIdx % 8.
__global__ void k_xcc(uint32_t* out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)); }
// mode 3: every workgroup asks the hardware which XCD it is on and takes the next unit of THAT XCD's eighth of the units
// (stealing from the others when its own is exhausted): robust against any dispatch order
__global__ void __launch_bounds__(64) k_chunks_dyn(char* __restrict__ base, size_t run_bytes, size_t stride_bytes, int runs, int chunks_per_record, unsigned nlogical, unsigned* ctr) {
  const int lane = threadIdx.x;
  __shared__ unsigned s_unit;
  if (lane == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;
    const unsigned per = (nlogical + 7u) / 8u;
    unsigned unit = 0xFFFFFFFFu;
    for (unsigned t = 0; t < 8u && unit == 0xFFFFFFFFu; ++t) {
      const unsigned x = (xcc + t) & 7u;
      const unsigned lo = x * per, hi = min(nlogical, lo + per);
      if (lo >= hi) continue;
      const unsigned u = atomicAdd(&ctr[x], 1u);
      if (lo + u < hi) unit = lo + u;
    }
    s_unit = unit;
  }
  __syncthreads();
  const unsigned lid = s_unit;
  if (lid == 0xFFFFFFFFu) return;
  const size_t rec0 = (size_t)(lid / chunks_per_record) * runs;
  const int ch = lid % chunks_per_record;
  for (int r = 0; r < runs; ++r) {
    char* g = base + (rec0 + r) * stride_bytes + (size_t)ch * run_bytes;
    const uint32_t al = (uint32_t)((uintptr_t)g & 15u);
    uint32_t head = (16u - al) & 15u;
    const uint32_t nwords = (uint32_t)((run_bytes - head) >> 4);
    const uint32_t tail_at = head + (nwords << 4);
    if ((uint32_t)lane < head) g[lane] = (char)lane;
    uint4* gw = reinterpret_cast<uint4*>(g + head);
    const uint4 v = make_uint4(lane, r, 3, 4);
    for (uint32_t wq = lane; wq < nwords; wq += 64) gw[wq] = v;
    if ((uint32_t)lane < run_bytes - tail_at) g[tail_at + lane] = (char)lane;
  }
}

int main(int argc, char** argv) {
  const size_t gb = argc > 1 ? (size_t)atoll(argv[1]) : 32;
  const size_t bytes = gb << 30, n16 = bytes / 16;
  char *a, *b;
  CK(hipMalloc(&a, bytes + 65536)); CK(hipMalloc(&b, bytes + 65536));
  uint32_t* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, double moved, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int i = 0; i < 3; ++i) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, moved / best / 1e6);
  };
  for (int blocks : {2048, 8192, 32768}) {
    char nm[96];
    snprintf(nm, 96, "fill 16B stores, %d x 256", blocks);
    time(nm, (double)bytes, [&]() { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, (uint4*)a, n16); });
  }
  time("fill non-temporal, 8192 x 256", (double)bytes, [&]() { hipLaunchKernelGGL(k_fill_nt, dim3(8192), dim3(256), 0, 0, (uint4*)a, n16); });
  time("read only, 8192 x 256", (double)bytes, [&]() { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const uint4*)a, n16, out); });
  time("copy (read + write bytes), 8192 x 256", 2.0 * bytes, [&]() { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, n16); });
  {
    // c2 shape: 16 chunks of 2 786 bytes per 44 577-byte record, 32 records per wavefront
    const size_t run_bytes = 2786, stride = 44577; const int cpr = 16, runs = 32;
    const size_t records = bytes / stride / runs * runs;
    const unsigned nblk = (unsigned)(records / runs * cpr);
    time("page-assembly shaped fill (2.8 KB runs, unaligned)", (double)records * cpr * run_bytes, [&]() { hipLaunchKernelGGL(k_chunks, dim3(nblk), dim3(64), 0, 0, a, run_bytes, stride, runs, cpr); });
    const size_t run2 = 2784, stride2 = 44544;   // the same with 16-byte aligned runs
    const size_t records2 = bytes / stride2 / runs * runs;
    time("the same, 16-byte aligned runs", (double)records2 * cpr * run2, [&]() { hipLaunchKernelGGL(k_chunks, dim3((unsigned)(records2 / runs * cpr)), dim3(64), 0, 0, a, run2, stride2, runs, cpr); });
  }
  {
    const size_t run_bytes = 2786, stride = 44577; const int cpr = 16;
    for (int runs : {32, 128}) for (int wpb : {1, 4, 16}) for (int mode : {0, 1}) {
      const size_t records = bytes / stride / runs * runs;
      const unsigned nlogical = (unsigned)(records / runs * cpr);
      char nm[128];
      snprintf(nm, 128, "shaped: %3d rec/wave, %2d waves/WG, %s", runs, wpb, mode ? "XCD-aware" : "chunk-fast");
      time(nm, (double)records * cpr * run_bytes, [&]() { hipLaunchKernelGGL(k_chunks2, dim3((nlogical + wpb - 1) / wpb), dim3(64 * wpb), 0, 0, a, run_bytes, stride, runs, cpr, mode, nlogical); });
    }
    {
      const int runs = 32;
      const size_t records = bytes / stride / runs * runs;
      const unsigned nlogical = (unsigned)(records / runs * cpr);
      unsigned* ctr; CK(hipMalloc(&ctr, 64));
      time("shaped:  32 rec/wave, units taken per XCC_ID", (double)records * cpr * run_bytes, [&]() { CK(hipMemsetAsync(ctr, 0, 64, 0)); hipLaunchKernelGGL(k_chunks_dyn, dim3(nlogical), dim3(64), 0, 0, a, run_bytes, stride, runs, cpr, nlogical, ctr); });
      // does blockIdx % 8 name the XCD?
      const unsigned nb = 1u << 16;
      uint32_t* dx; CK(hipMalloc(&dx, nb * 4));
      hipLaunchKernelGGL(k_xcc, dim3(nb), dim3(64), 0, 0, dx);
      std::vector<uint32_t> hx(nb);
      CK(hipMemcpy(hx.data(), dx, nb * 4, hipMemcpyDeviceToHost));
      unsigned best = 0; int best_c = 0; unsigned hist[16] = {0};
      for (unsigned b2 = 0; b2 < nb; ++b2) hist[hx[b2] & 15u]++;
      for (int c = 0; c < 8; ++c) { unsigned ok = 0; for (unsigned b2 = 0; b2 < nb; ++b2) ok += ((hx[b2] & 7u) == ((b2 + c) & 7u)); if (ok > best) { best = ok; best_c = c; } }
      printf("XCC_ID of 65536 workgroups: blockIdx %% 8 names the XCD (rotation %d) for %.1f %% of them; per-XCD counts", best_c, 100.0 * best / nb);
      for (int x = 0; x < 8; ++x) printf(" %u", hist[x]);
      printf("\n");
    }
    for (int runs : {2, 8}) {
      const size_t records = bytes / stride / runs * runs;
      const unsigned nlogical = (unsigned)(records / runs);
      char nm[128];
      snprintf(nm, 128, "whole records per wave (%d rec/wave)", runs);
      time(nm, (double)records * cpr * run_bytes, [&]() { hipLaunchKernelGGL(k_chunks2, dim3(nlogical), dim3(64), 0, 0, a, run_bytes, stride, runs, cpr, 2, nlogical); });
    }
  }
  return 0;
}
