// lds_pattern_probe.hip — ds_read_b128 cost of the A-fragment access patterns of conv_halo2_kernel (16x16x32 tiles: 16 consecutive
// halo pixels x one 16-byte chunk per 16-lane quarter) and conv_halo2x_kernel (32x32x16 tiles: 2 halo rows x 16 pixels, chunk pair
// per wave half) on 128-byte halo pixels, with their chunk swizzles and without.  256 threads (one wave per SIMD), no MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_pattern_probe.hip -o /tmp/lpp && /tmp/lpp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// mode 0: halo2 (16x16): lane (frow = lane & 15, q = lane >> 4): pixel frow + kx, chunk (q ^ swz) [k-step: ^ 4]
// mode 1: halo2x (32x32): lane (px = lane & 15, half = (lane >> 4) & 1, h5 = lane >> 5): pixel (2 half rows down) px + kx, chunk (2 ks + h5) ^ swz
template <int MODE, bool SWZ>
__global__ __launch_bounds__(256) void probe(uint4* out, int iters) {
  extern __shared__ uint4 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  for (int i = tid; i < 192 * 8 * 4; i += 256) smem[i] = make_uint4(i, i * 3, i * 5, i * 7);
  __syncthreads();
  int off[3][4];
  for (int kx = 0; kx < 3; ++kx)
    for (int ks = 0; ks < 4; ++ks) {
      if (MODE == 0) {
        const int frow = lane & 15, q = lane >> 4, hx = frow + kx;
        const int sw = SWZ ? (((hx >> 1) & 3) << 1) : 0;
        off[kx][ks] = (((wid >> 1) * 4 * 18 + hx) * 8 + ((q ^ sw) ^ ((ks & 1) * 4))) + (ks >> 1) * 18 * 8;   // ks >= 2: next row (6 rows x 2 k-steps in the kernel)
      } else {
        const int px = lane & 15, half = (lane >> 4) & 1, h5 = lane >> 5, hx = px + kx;
        const int sw = SWZ ? ((hx >> 1) & 7) : 0;
        off[kx][ks] = (((wid >> 1) * 4 + 2 * half) * 18 + hx) * 8 + ((2 * ks + h5) ^ sw);
      }
    }
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    const int stage = (it & 3) * 192 * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const uint4 v = smem[stage + off[kx][ks] + r * 18 * 8];
          acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
  }
  if (acc.x == 0x12345678u) out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE, bool SWZ>
static void run(const char* name, uint4* out, int n_cu) {
  const int iters = 400;
  const size_t lds = 192 * 8 * 4 * 16;
  CHECK(hipFuncSetAttribute((const void*)probe<MODE, SWZ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, SWZ>), dim3(n_cu), dim3(256), lds, 0, out, iters);
  CHECK(hipEventRecord(e0));
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((probe<MODE, SWZ>), dim3(n_cu), dim3(256), lds, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double reads_per_wave = 10.0 * iters * 48;
  printf("%-44s %7.1f us per launch, %6.2f ns per ds_read_b128 of a wave (4 waves per CU) = %5.1f B/ns per CU\n", name, ms * 100, ms * 1e6 / reads_per_wave,
         4.0 * 1024.0 / (ms * 1e6 / reads_per_wave));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  uint4* out; CHECK(hipMalloc(&out, (size_t)p.multiProcessorCount * 256 * 16));
  run<0, true>("16x16 pattern (conv_halo2), swizzled", out, p.multiProcessorCount);
  run<0, false>("16x16 pattern, no swizzle", out, p.multiProcessorCount);
  run<1, true>("32x32 pattern (conv_halo2x), swizzled", out, p.multiProcessorCount);
  run<1, false>("32x32 pattern, no swizzle", out, p.multiProcessorCount);
  return 0;
}
