// mfma_rate_probe.hip — what matrix rate does the part sustain, per MFMA shape, in the regimes the step's convolutions run in?
// (every MFMA kernel of this repo issues v_mfma_f32_16x16x32_bf16; MI355X_MICROARCH.md lists it at ~17
// cycles per SIMD back to back against 32 for the twice-as-large v_mfma_f32_32x32x16_bf16 — 6 % — and the VGG classes of the step
// track the clock of the box.)  Pure register-resident MFMA streams, 16 independent accumulator tiles (16x16) / 4 (32x32) per wave:
//   * shape 16x16x32 vs 32x32x16,
//   * one or two waves per SIMD (256- / 512-thread workgroups, one workgroup per CU),
//   * a short launch (~30 us, the length of a VGG layer) after an idle gap vs a sustained stream of back-to-back launches (~20 ms).
// Operands are random bf16 (data-dependent power: zeros would flatter the clock).
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o /tmp/mrp && /tmp/mrp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// 16 accumulator tiles of 16x16 (64 VGPRs), 4 A x 4 B fragments: the wave tile of conv_hdeep6.hip
template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfma16_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  uint4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123456.789f) sink[t] = s;
}

// 4 accumulator tiles of 32x32 (64 VGPRs), 2 A x 2 B fragments: the same 64x64 wave tile, the same operand registers per FLOP
template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfma32_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  uint4 a[2][2], b[2][2];                               // [tile][k half of a 32-deep step]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) { a[i][h] = src[(t * 8 + i * 2 + h) & 0xffff]; b[i][h] = src[(t * 8 + 4 + i * 2 + h) & 0xffff]; }
  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {                  // one iteration = the FLOPs of one iteration of mfma16_kernel (k = 32)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i][h]), __builtin_bit_cast(bf16x8_t, b[j][h]), acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 123456.789f) sink[t] = s;
}


// dependent-accumulator distance: NACC independent 32x32 accumulators in rotation (1 = every MFMA waits for the one before)
template <int THREADS, int NACC>
__global__ __launch_bounds__(THREADS) void mfma32_dep_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  uint4 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + 4 + i) & 0xffff]; }
  f32x16_t acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {                  // 8 MFMAs per iteration = the FLOPs of one iteration of the kernels above
#pragma unroll
    for (int m = 0; m < 8; ++m)
      acc[m % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[m & 3]), __builtin_bit_cast(bf16x8_t, b[(m >> 1) & 3]), acc[m % NACC], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123456.789f) sink[t] = s;
}

// the operand pattern of conv_halo2x_kernel: 36 filter fragments (144 registers: the compiler keeps most of them in AGPRs) as the first
// operand, a 4-deep ring of pixel fragments as the second, two accumulators; 72 MFMAs per iteration (= 4.5 iterations of the kernels above)
template <int THREADS, int NB>
__global__ __launch_bounds__(THREADS) void mfma32_taps_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters) {
  const int t = blockIdx.x * THREADS + threadIdx.x;
  uint4 bw[NB], fa[4];
#pragma unroll
  for (int i = 0; i < NB; ++i) bw[i] = src[(t * 40 + i) & 0xffff];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = src[(t * 8 + 36 + i) & 0xffff];
  f32x16_t acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 72; ++m)
      acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, bw[(m >> 1) % NB]), __builtin_bit_cast(bf16x8_t, fa[(m >> 1) & 3]), acc[m & 1], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123456.789f) sink[t] = s;
}

template <typename F>
static void run(const char* name, F launch, int n_cu, int threads, int iters) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double flop = (double)n_cu * (threads / 64) * iters * 16.0 * 16384.0;   // per launch
  // (a) short launches after an idle gap
  float best = 1e30f, sum = 0.f;
  for (int r = 0; r < 12; ++r) {
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); launch(iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) { sum += ms; if (ms < best) best = ms; }
  }
  const float short_ms = sum / 10.f;
  // (b) sustained: ~600 launches back to back
  const int reps = 600;
  CHECK(hipDeviceSynchronize());
  for (int r = 0; r < 50; ++r) launch(iters);
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch(iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-40s  one launch %6.1f us (best %6.1f) = %6.0f TFLOP/s | sustained %6.1f us per launch = %6.0f TFLOP/s  (%d launches, %.1f ms)\n", name,
         short_ms * 1e3, best * 1e3, flop / (short_ms * 1e-3) / 1e12, ms / reps * 1e3, flop / (ms / reps * 1e-3) / 1e12, reps, ms);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int n_cu = p.multiProcessorCount;
  printf("# %s, %d CUs; bf16 MFMA streams, 64 accumulator registers per wave, one workgroup per CU; FLOPs = 2 x MACs\n", p.gcnArchName, n_cu);
  std::vector<uint16_t> h(65536 * 8);
  uint32_t x = 12345u;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (uint16_t)(0x3c00u + ((x >> 12) & 0x3ffu)) | (uint16_t)((x >> 31) << 15); }   // +-[0.008, 0.03]
  uint4* src; float* sink;
  CHECK(hipMalloc(&src, h.size() * 2)); CHECK(hipMalloc(&sink, (size_t)n_cu * 512 * 4));
  CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  const int it1 = 220, it2 = 110;   // ~30 us launches at one / two waves per SIMD
  run("16x16x32, one wave per SIMD", [&](int it) { hipLaunchKernelGGL(mfma16_kernel<256>, dim3(n_cu), dim3(256), 0, 0, src, sink, it); }, n_cu, 256, it1);
  run("32x32x16, one wave per SIMD", [&](int it) { hipLaunchKernelGGL(mfma32_kernel<256>, dim3(n_cu), dim3(256), 0, 0, src, sink, it); }, n_cu, 256, it1);
  run("16x16x32, two waves per SIMD", [&](int it) { hipLaunchKernelGGL(mfma16_kernel<512>, dim3(n_cu), dim3(512), 0, 0, src, sink, it); }, n_cu, 512, it2);
  run("32x32x16, two waves per SIMD", [&](int it) { hipLaunchKernelGGL(mfma32_kernel<512>, dim3(n_cu), dim3(512), 0, 0, src, sink, it); }, n_cu, 512, it2);
  run("32x32x16, one wave, 1 accumulator (dependent)", [&](int it) { hipLaunchKernelGGL((mfma32_dep_kernel<256, 1>), dim3(n_cu), dim3(256), 0, 0, src, sink, it); }, n_cu, 256, it1);
  run("32x32x16, one wave, 2 accumulators", [&](int it) { hipLaunchKernelGGL((mfma32_dep_kernel<256, 2>), dim3(n_cu), dim3(256), 0, 0, src, sink, it); }, n_cu, 256, it1);
  run("32x32x16, one wave, 4 accumulators", [&](int it) { hipLaunchKernelGGL((mfma32_dep_kernel<256, 4>), dim3(n_cu), dim3(256), 0, 0, src, sink, it); }, n_cu, 256, it1);
  // 72 MFMAs of 32x32x16 per iteration = 9 x the FLOPs of an iteration above: iters / 9 (225 -> 25) for the same FLOPs per launch
  run("32x32x16, one wave, 36 tap fragments (AGPRs)", [&](int it) { hipLaunchKernelGGL((mfma32_taps_kernel<256, 36>), dim3(n_cu), dim3(256), 0, 0, src, sink, it / 9); }, n_cu, 256, 225);
  run("32x32x16, one wave, 4 tap fragments", [&](int it) { hipLaunchKernelGGL((mfma32_taps_kernel<256, 4>), dim3(n_cu), dim3(256), 0, 0, src, sink, it / 9); }, n_cu, 256, 225);
  CHECK(hipFree(src)); CHECK(hipFree(sink));
  return 0;
}
