// lds_mfma_ratio_probe.hip — the matrix rate a CU sustains when every v_mfma_f32_16x16x32_bf16 needs r operand fetches
// (ds_read_b128) from LDS: the measurement behind the Winograd F(2x2,3x3) no-go (VERDICT r5 item 2 -> DESIGN.md item 60).
//
// A wave holds NA x NB accumulator tiles (x NPOS independent "positions", Winograd's 16 transform points), and per k-step
// reads NA A fragments + NB B fragments and issues NA*NB MFMAs per position:
//   direct conv_hdeep6 tile : NA = NB = 4, 1 position, 2 waves per SIMD        -> r = 0.5 fetches per MFMA, 64 accumulator VGPRs
//   Winograd, what fits     : NA = NB = 2, 16 positions, 1 wave per SIMD       -> r = 1.0, 256 accumulator VGPRs
//   Winograd, quarter tile  : NA = 1, NB = 4 / NA = NB = 1, 16 positions        -> r = 1.25 / 2.0
// Fragments are double-buffered by hand (next k-step's reads issued before this k-step's MFMAs), addresses move with the k-step
// (nothing can be hoisted), LDS is filled once; no global traffic in the loop.  Prints TFLOP/s for the whole chip (one workgroup
// per CU, 256 CUs) and nanoseconds per MFMA per SIMD (8.3 ns = 16 cycles at 1.9 GHz).
//
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_mfma_ratio_probe.hip -o /tmp/lmr && /tmp/lmr
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int NA, int NB, int NPOS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void probe_kernel(float* out, int ksteps, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];     // 64 KB
  constexpr int N_U4 = 4096;
  for (int i = threadIdx.x; i < N_U4; i += WAVES * 64) smem[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4_t acc[NPOS][NA][NB];
#pragma unroll
  for (int p = 0; p < NPOS; ++p)
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[p][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  uint4 a[2][NA], b[2][NB];
  auto fetch = [&](int buf, int step) {
    // conflict-free: 64 lanes read 64 consecutive 16-byte words; the base moves with (step, fragment, wave)
#pragma unroll
    for (int i = 0; i < NA; ++i) a[buf][i] = smem[(lane + 64 * ((step * 7 + i * 3 + wv) & 31)) & (N_U4 - 1)];
#pragma unroll
    for (int j = 0; j < NB; ++j) b[buf][j] = smem[(lane + 64 * ((step * 5 + j * 11 + wv + 32) & 63)) & (N_U4 - 1)];
  };
  const unsigned long long t0 = clock64();
  fetch(0, 0);
  for (int k = 0; k < ksteps; k += 2) {                 // two k-steps per trip: the fragment buffer index is a compile-time constant
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int p = 0; p < NPOS; ++p) {
        const int step = (k + u) * NPOS + p;
        const int cur = (u * NPOS + p) & 1;
        fetch(cur ^ 1, step + 1);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[cur][i]), __builtin_bit_cast(bf16x8_t, b[cur][j]),
                                                                   acc[p][i][j], 0, 0, 0);
      }
  }
  const unsigned long long t1 = clock64();
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < NPOS; ++p)
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) s += acc[p][i][j][0] + acc[p][i][j][1] + acc[p][i][j][2] + acc[p][i][j][3];
  out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The Winograd F(2x2,3x3) inner loop with its INPUT TRANSFORM done in registers on the A operands (the form that adds no LDS
// traffic): per 32-channel slice a wave (2 M blocks of 16 tiles x 2 N blocks of 16 output channels, 16 positions = 256 accumulator
// registers, one wave per SIMD) reads the 4x4 input patch of its lane's tile (16 x ds_read_b128 per M block: 8 channels each),
// computes V = B^T d B in f32 (unpack, 32 adds per channel, pack to bf16: the real arithmetic), then per position 2 B fragments and
// 4 MFMAs.  No output transform, no filter DMA: an upper bound of what the loop of such a kernel sustains.
__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t bf_pack(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2_t;
  const f2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2_t));
}

__global__ __launch_bounds__(256) void wino_loop_kernel(float* out, int slices) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  constexpr int N_U4 = 4096;
  for (int i = threadIdx.x; i < N_U4; i += 256) smem[i] = make_uint4(0x3c003c00u + i, 0x3c003f80u, 0x3c003c00u, 0x3f803c00u);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4_t acc[16][2][2];
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[p][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < slices; ++sl) {
    uint4 a[2][16];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      uint4 d[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) d[q] = smem[(lane + 64 * ((sl * 3 + mb * 16 + q + wv) & 63)) & (N_U4 - 1)];
      // V = B^T d B on the 4x4 patch, channel pair by channel pair (4 x uint32 = 8 channels per lane)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float lo[4][4], hi[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint32_t u = w == 0 ? d[r * 4 + c].x : w == 1 ? d[r * 4 + c].y : w == 2 ? d[r * 4 + c].z : d[r * 4 + c].w;
            lo[r][c] = bf_lo(u); hi[r][c] = bf_hi(u);
          }
        auto bt = [](float (&m)[4][4]) {
          float t[4][4];
#pragma unroll
          for (int c = 0; c < 4; ++c) { t[0][c] = m[0][c] - m[2][c]; t[1][c] = m[1][c] + m[2][c]; t[2][c] = m[2][c] - m[1][c]; t[3][c] = m[1][c] - m[3][c]; }
#pragma unroll
          for (int r = 0; r < 4; ++r) { m[r][0] = t[r][0] - t[r][2]; m[r][1] = t[r][1] + t[r][2]; m[r][2] = t[r][2] - t[r][1]; m[r][3] = t[r][1] - t[r][3]; }
        };
        bt(lo); bt(hi);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const uint32_t u = bf_pack(lo[r][c], hi[r][c]);
            if (w == 0) a[mb][r * 4 + c].x = u; else if (w == 1) a[mb][r * 4 + c].y = u; else if (w == 2) a[mb][r * 4 + c].z = u; else a[mb][r * 4 + c].w = u;
          }
      }
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      uint4 b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = smem[(lane + 64 * ((sl * 5 + p * 2 + j + wv + 32) & 63)) & (N_U4 - 1)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[p][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[i][p]), __builtin_bit_cast(bf16x8_t, b[j]), acc[p][i][j], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int p = 0; p < 16; ++p)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) s += acc[p][i][j][0] + acc[p][i][j][1] + acc[p][i][j][2] + acc[p][i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static void run_wino(float* out) {
  const int cus = 256, slices = 2048;                 // 64 MFMAs per slice and wave
  CHECK(hipFuncSetAttribute((const void*)wino_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(wino_loop_kernel, dim3(cus), dim3(256), 65536, 0, out, slices);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  const double mfma = (double)cus * 4 * (double)slices * 64;
  const double tflops = mfma * 16384.0 / (best * 1e-3) / 1e12;
  printf("%-64s r = 1.00  waves/SIMD 1  acc VGPRs 256   %7.1f TFLOP/s   %5.2f ns / MFMA / SIMD   (%.3f ms)   = %.1f TFLOP/s of DIRECT-convolution FLOPs (x 2.25)\n",
         "Winograd loop incl. the register input transform (B^T d B)", tflops, best * 1e6 / ((double)slices * 64), best, tflops * 2.25);
}

template <int NA, int NB, int NPOS, int WAVES>
static void run(const char* what, float* out, unsigned long long* cyc, int wg_per_cu) {
  const int cus = 256, ksteps = 131072 / (NPOS * NA * NB);      // 131072 MFMAs per wave in every variant (~1 ms)
  auto k = probe_kernel<NA, NB, NPOS, WAVES>;
  CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k, dim3(cus * wg_per_cu), dim3(WAVES * 64), 65536, 0, out, ksteps, cyc);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  unsigned long long c0 = 0;
  CHECK(hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost));
  const double mfma = (double)cus * wg_per_cu * WAVES * (double)ksteps * NPOS * NA * NB;
  const double tflops = mfma * 16.0 * 16.0 * 32.0 * 2.0 / (best * 1e-3) / 1e12;
  const double mfma_per_simd = (double)WAVES * wg_per_cu / 4.0 * ksteps * NPOS * NA * NB;
  (void)c0;
  printf("%-64s r = %4.2f  waves/SIMD %d  acc VGPRs %3d   %7.1f TFLOP/s   %5.2f ns / MFMA / SIMD   (%.3f ms)\n", what,
         (double)(NA + NB) / (NA * NB), WAVES * wg_per_cu / 4, NPOS * NA * NB * 4, tflops, best * 1e6 / mfma_per_simd, best);
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 2 * 512 * 4));
  CHECK(hipMalloc(&cyc, 256 * 2 * 8));
  printf("# v_mfma_f32_16x16x32_bf16 fed from LDS by ds_read_b128, whole chip (256 CUs), no global traffic; r = operand fetches per MFMA\n");
  run<4, 4, 1, 8>("direct tile (conv_hdeep6): 4 x 4 blocks, 2 waves/SIMD", out, cyc, 1);
  run<4, 4, 1, 4>("direct tile, 1 wave/SIMD", out, cyc, 1);
  run<2, 2, 16, 4>("Winograd F(2x2,3x3): 2 x 2 blocks x 16 positions, 1 wave/SIMD", out, cyc, 1);
  run<2, 2, 4, 8>("(hypothetical) 2 x 2 blocks x 4 positions, 2 waves/SIMD", out, cyc, 1);
  run<1, 4, 16, 4>("Winograd quarter tile: 1 x 4 blocks x 16 positions, 1 wave/SIMD", out, cyc, 1);
  run<1, 1, 16, 4>("Winograd 1 x 1 blocks x 16 positions, 1 wave/SIMD", out, cyc, 1);
  run_wino(out);
  run<4, 4, 1, 8>("direct tile again (clock check)", out, cyc, 1);
  return 0;
}
