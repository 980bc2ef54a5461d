// lds_fill_probe.hip — how fast can ONE CU pull L2-resident bytes into LDS on MI355X, and does the answer depend on the
// way the bytes travel?  The small-map layers of the step (encoder conv_6 / conv_8, renderer conv_1..4, VGG conv5: 128 x 64
// tiles that stream 300-600 KB of filter rows per workgroup) run at the ~30 GB/s per CU the LDS-DMA ring of conv_hdeep
// delivers (profiles/r06_v1_hdeep_lds_dma_roof.txt: 5.8-8.7 TB/s over the chip).  Is that the part, or the issue pattern?
//
// One workgroup per CU (256 workgroups), W waves each; every workgroup streams `total` bytes:
//   region SHARED  : every workgroup walks the same 1.25 MB region (a filter image: all CUs of an XCD hit the same L2 lines)
//   region PRIVATE : each workgroup walks its own 512 KB region (halo-like; 128 MB in all: L2-missing, Infinity-Cache-resident)
// through one of
//   dma   : buffer_load_dwordx4 ... offen lds (1 KB per wave instruction) into a per-wave ring of D slots, s_waitcnt vmcnt(D-1)
//   regs  : global_load_dwordx4 into D registers in flight, ds_write_b128 into the same ring
//   sink  : global_load_dwordx4 into D registers, xor-folded (no LDS): the L2 -> register rate alone
// Prints GB/s per CU and TB/s for the chip.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_fill_probe.hip -o /tmp/lfp && /tmp/lfp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ void dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" :: "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// MODE 0 dma | 1 regs -> ds_write | 2 sink
template <int MODE, int W, int D>
__global__ __launch_bounds__(W * 64) void fill_kernel(const uint4* __restrict__ src, uint32_t region_bytes, uint32_t wg_stride_bytes,
                                                      int iters, unsigned long long* t_out, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];      // W * D slots of 1 KB
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t base = (uint64_t)src + (uint64_t)blockIdx.x * wg_stride_bytes;
  const u32x4_t rs = {(uint32_t)base, (uint32_t)(base >> 32) & 0xffffu, region_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem + (uint32_t)wid * D * 1024u;
  const uint4* gp = (const uint4*)base;
  const uint32_t n_chunks = region_bytes >> 10;                    // 1 KB chunks in the region
  uint32_t chunk = (uint32_t)wid % n_chunks;
  uint4 r[D];
  uint32_t fold = 0;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // slot d of this wave was requested D instructions ago: at most D - 1 younger ones may still be in flight
        if (it > 0) wait_vm<D - 1>();
        dma16(rs, lds_base + d * 1024u, chunk * 1024u + lane * 16u);
        chunk += W; if (chunk >= n_chunks) chunk -= n_chunks;
      }
    }
    wait_vm<0>();
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) { r[d] = gp[chunk * 64u + lane]; chunk += W; if (chunk >= n_chunks) chunk -= n_chunks; }
    for (int it = 1; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        wait_vm<D - 1>();
        asm volatile("" : "+v"(r[d].x), "+v"(r[d].y), "+v"(r[d].z), "+v"(r[d].w));
        if (MODE == 1) smem[(wid * D + d) * 64 + lane] = r[d];
        else fold ^= r[d].x ^ r[d].y ^ r[d].z ^ r[d].w;
        r[d] = gp[chunk * 64u + lane];
        chunk += W; if (chunk >= n_chunks) chunk -= n_chunks;
      }
    }
    wait_vm<0>();
#pragma unroll
    for (int d = 0; d < D; ++d) fold ^= r[d].x;
  }
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
  if (MODE != 0 && fold == 0x12345678u) sink[0] = fold;
  if (MODE != 2 && smem[threadIdx.x].x == 0x12345678u) sink[1] = 1;
}


// ---- the question behind the 30 GB/s: a kernel starts with a cold L2 and its workgroups walk a shared filter image in lockstep, so
// EVERY workgroup waits out the L2 miss of every line (hit-on-miss: the traffic is shared, the latency is not).  One pass over a
// slice of a filter image (n_slices slices; workgroup b on XCD b & 7 takes slice (b >> 3) % n_slices), DMA ring of D KB per wave,
// with an optional touch-ahead: PF = 0 none | 1 every workgroup touches its whole slice at kernel start (one dword per 128-B line)
// | 2 the workgroups that share a slice on an XCD touch a share each.
template <int W, int D, int PF>
__global__ __launch_bounds__(W * 64) void cold_kernel(const uint4* __restrict__ src, uint32_t slice_bytes, int n_slices,
                                                      unsigned long long* t_out, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int in_xcd = blockIdx.x >> 3;
  const int slice = in_xcd % n_slices, sharer = in_xcd / n_slices, n_sharers = (gridDim.x >> 3) / n_slices;
  const uint64_t base = (uint64_t)src + (uint64_t)slice * slice_bytes;
  const u32x4_t rs = {(uint32_t)base, (uint32_t)(base >> 32) & 0xffffu, slice_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem + (uint32_t)wid * D * 1024u;
  const uint32_t n_chunks = slice_bytes >> 10;
  const unsigned long long t0 = wall_clock64();
  uint32_t dummy = 0;
  if (PF) {
    // 64 lanes x one dword per 128-byte line = 8 KB per wave instruction
    const uint32_t n_touch = slice_bytes >> 13;                   // 8 KB pieces in the slice
    const uint32_t lo = PF == 2 ? (n_touch * sharer) / n_sharers : 0, hi = PF == 2 ? (n_touch * (sharer + 1)) / n_sharers : n_touch;
    for (uint32_t p = lo + wid; p < hi; p += W) {
      const uint32_t* q = (const uint32_t*)(base + (uint64_t)p * 8192u + lane * 128u);
      uint32_t v;
      asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(q) : "memory");
      dummy ^= v;      // (forces a wait at the end of the loop body: the touches of one wave are serial; fine for a probe of the effect)
    }
  }
  uint32_t chunk = wid;
  int issued = 0;
  for (; chunk < n_chunks; chunk += W) {
    const int d = issued % D;
    if (issued >= D) wait_vm<D - 1>();
    dma16(rs, lds_base + d * 1024u, chunk * 1024u + lane * 16u);
    ++issued;
  }
  wait_vm<0>();
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
  if (dummy == 0x12345678u || smem[threadIdx.x].x == 0x12345678u) sink[1] = 1;
}

template <int W, int D, int PF>
static void run_cold(const uint4* src, uint32_t slice_bytes, int n_slices, int n_wg, unsigned long long* t_dev, uint32_t* sink, uint4* flush, size_t flush_bytes) {
  const size_t lds = (size_t)W * D * 1024;
  CHECK(hipFuncSetAttribute((const void*)cold_kernel<W, D, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<double> med;
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL((cold_kernel<W, D, PF>), dim3(n_wg), dim3(W * 64), lds, 0, src, slice_bytes, n_slices, t_dev, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> t(n_wg);
    CHECK(hipMemcpy(t.data(), t_dev, n_wg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(t.begin(), t.end());
    if (rep) med.push_back(t[n_wg / 2] / 100.0);
  }
  std::sort(med.begin(), med.end());
  const double us = med[med.size() / 2];
  printf("cold  %4u KB slice x %d slices  waves %d  ring %2d KB per wave  touch-ahead %-12s  %6.2f us per workgroup = %6.1f GB/s per CU\n",
         slice_bytes >> 10, n_slices, W, D, PF == 0 ? "none" : PF == 1 ? "whole slice" : "shared", us, slice_bytes / us * 1e-3);
}

template <int MODE, int W, int D>
static void run(const char* mode, const char* region, const uint4* src, uint32_t region_bytes, uint32_t wg_stride, int n_wg,
                unsigned long long* t_dev, uint32_t* sink) {
  const size_t total = 8u << 20;                                   // bytes per workgroup
  const int iters = (int)(total / ((size_t)W * D * 1024));
  const size_t lds = (size_t)W * D * 1024;
  CHECK(hipFuncSetAttribute((const void*)fill_kernel<MODE, W, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((fill_kernel<MODE, W, D>), dim3(n_wg), dim3(W * 64), lds, 0, src, region_bytes, wg_stride, iters, t_dev, sink);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = std::min(best, ms);
  }
  std::vector<unsigned long long> t(n_wg);
  CHECK(hipMemcpy(t.data(), t_dev, n_wg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(t.begin(), t.end());
  const double bytes = (double)iters * W * D * 1024.0;
  const double us_med = t[n_wg / 2] / 100.0;                       // wall_clock64: 100 MHz
  printf("%-5s %-8s waves %2d  in flight per wave %2d (%3d KB per CU)   %7.1f GB/s per CU (median workgroup)   %6.2f TB/s chip (launch %.1f us)\n",
         mode, region, W, D, W * D, bytes / us_med * 1e-3, bytes * n_wg / (best * 1e-3) * 1e-12, best * 1e3);
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int n_wg = p.multiProcessorCount;
  printf("# %s, %d CUs; one workgroup per CU, 8 MB per workgroup, 16 B per lane (1 KB per wave instruction)\n", p.gcnArchName, n_wg);
  const uint32_t shared_bytes = 1280u << 10, priv_bytes = 512u << 10;
  uint4* src; CHECK(hipMalloc(&src, (size_t)priv_bytes * n_wg));
  CHECK(hipMemset(src, 1, (size_t)priv_bytes * n_wg));
  unsigned long long* t_dev; CHECK(hipMalloc(&t_dev, n_wg * sizeof(unsigned long long)));
  uint32_t* sink; CHECK(hipMalloc(&sink, 64));
#define BOTH(MODE, NAME, W, D) run<MODE, W, D>(NAME, "shared", src, shared_bytes, 0, n_wg, t_dev, sink); \
                               run<MODE, W, D>(NAME, "private", src, priv_bytes, priv_bytes, n_wg, t_dev, sink);
  BOTH(0, "dma", 1, 8)  BOTH(0, "dma", 1, 16) BOTH(0, "dma", 1, 32)
  BOTH(0, "dma", 4, 4)  BOTH(0, "dma", 4, 8)  BOTH(0, "dma", 4, 16)
  BOTH(0, "dma", 8, 4)  BOTH(0, "dma", 8, 8)  BOTH(0, "dma", 8, 16)
  BOTH(0, "dma", 16, 4) BOTH(0, "dma", 16, 8)
  BOTH(1, "regs", 4, 4) BOTH(1, "regs", 4, 8) BOTH(1, "regs", 4, 16)
  BOTH(1, "regs", 8, 4) BOTH(1, "regs", 8, 8) BOTH(1, "regs", 8, 16)
  BOTH(1, "regs", 16, 4) BOTH(1, "regs", 16, 8)
  BOTH(2, "sink", 4, 8) BOTH(2, "sink", 8, 8) BOTH(2, "sink", 16, 8)

  printf("# cold-L2 single pass (a kernel's first walk over a filter image shared by the workgroups of an XCD)\n");
#define COLD(W, D, SL, NS) run_cold<W, D, 0>(src, SL, NS, n_wg, t_dev, sink, 0, 0); run_cold<W, D, 1>(src, SL, NS, n_wg, t_dev, sink, 0, 0); \
                           run_cold<W, D, 2>(src, SL, NS, n_wg, t_dev, sink, 0, 0);
  COLD(4, 2, 288u << 10, 4) COLD(4, 4, 288u << 10, 4) COLD(4, 8, 288u << 10, 4) COLD(4, 12, 288u << 10, 4)
  COLD(8, 2, 1152u << 10, 4) COLD(8, 4, 1152u << 10, 4)
  COLD(4, 4, 576u << 10, 8)
  return 0;
}
