// common.h — shared device helpers for libimm_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <type_traits>

#include "../../include/imm_hip.h"

#define IMM_WAVE 64

// debug-only ablation bits in imm_conv_desc.flags (tools/bench_conv.py); never set by the product path
#define IMM_DBG_NO_GLOAD 0x100
#define IMM_DBG_NO_LDS_STORE 0x200
#define IMM_DBG_NO_MFMA 0x400
#define IMM_DBG_NO_EPILOGUE 0x800

// ---------------------------------------------------------------------------------------------
// error plumbing (thread-local message; no exceptions cross the C boundary)
// ---------------------------------------------------------------------------------------------
extern thread_local char imm_err_buf[512];
int imm_fail(int code, const char* fmt, ...);

#define IMM_REQUIRE(cond, ...)                              \
  do {                                                      \
    if (!(cond)) return imm_fail(IMM_E_INVALID, __VA_ARGS__); \
  } while (0)

#define IMM_CHECK_LAUNCH(name)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) return imm_fail(IMM_E_HIP, "%s: %s", name, hipGetErrorString(e__)); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// 16-bit element traits: storage is uint16_t in memory; arithmetic in f32; MFMA operand vectors
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

struct BF16 {
  static constexpr int kEnum = IMM_BF16;
  using T = uint16_t;     // storage element
  using V8 = uint4;       // eight stored elements (one 16-byte access)
  __device__ static __forceinline__ V8 zero8() { return make_uint4(0, 0, 0, 0); }
  __device__ static __forceinline__ float to_f32(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
  // round to nearest even in hardware (v_cvt_pk_bf16_f32 on gfx950)
  __device__ static __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  }
  // c + a.lo*b.lo + a.hi*b.hi on packed pairs (v_dot2c_f32_bf16)
  __device__ static __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
  }
  __device__ static __forceinline__ f32x4_t mfma(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                   c, 0, 0, 0);
  }
};

struct F16 {
  static constexpr int kEnum = IMM_F16;
  using T = uint16_t;
  using V8 = uint4;
  __device__ static __forceinline__ V8 zero8() { return make_uint4(0, 0, 0, 0); }
  __device__ static __forceinline__ float to_f32(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  __device__ static __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
  }
  __device__ static __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
  }
  __device__ static __forceinline__ f32x4_t mfma(uint4 a, uint4 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b),
                                                  c, 0, 0, 0);
  }
};

// f32 STORAGE (IMM_F32): the exact-arithmetic witness of the wiring (round 6; the reference computes in fp32, imm_model.py:97).
// Only the HBM-bound passes (elementwise.hip, loss_optim.hip, bottleneck.hip's soft-argmax) are instantiated on it — eight
// elements are two 16-byte accesses; the convolutions of the f32 engine are the plain kernels of conv_f32.hip (no MFMA
// instantiation exists for it: mfma / dot2 / pack2 are deliberately absent).
struct F32 {
  static constexpr int kEnum = IMM_F32;
  using T = float;
  struct V8 { float4 a, b; };
  __device__ static __forceinline__ V8 zero8() { V8 v; v.a = make_float4(0.f, 0.f, 0.f, 0.f); v.b = v.a; return v; }
  __device__ static __forceinline__ float to_f32(float u) { return u; }
  __device__ static __forceinline__ float from_f32(float f) { return f; }
};

// eight stored elements <-> registers
template <typename ET>
__device__ __forceinline__ typename ET::V8 ld8(const typename ET::T* p) { return *(const uint4*)p; }
template <>
__device__ __forceinline__ F32::V8 ld8<F32>(const float* p) { F32::V8 v; v.a = *(const float4*)p; v.b = *(const float4*)(p + 4); return v; }
template <typename ET>
__device__ __forceinline__ void st8(typename ET::T* p, const typename ET::V8& v) { *(uint4*)p = v; }
template <>
__device__ __forceinline__ void st8<F32>(float* p, const F32::V8& v) { *(float4*)p = v.a; *(float4*)(p + 4) = v.b; }

__device__ __forceinline__ void unpack8_f32(const F32::V8& v, float (&f)[8]) {
  f[0] = v.a.x; f[1] = v.a.y; f[2] = v.a.z; f[3] = v.a.w; f[4] = v.b.x; f[5] = v.b.y; f[6] = v.b.z; f[7] = v.b.w;
}

// unpack / pack 8 16-bit values held in a uint4
template <typename ET>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = ET::to_f32((uint16_t)(w[i] & 0xffffu));
    f[2 * i + 1] = ET::to_f32((uint16_t)(w[i] >> 16));
  }
}
template <typename ET>
__device__ __forceinline__ void unpack8(const F32::V8& v, float (&f)[8]) { unpack8_f32(v, f); }
template <typename ET, typename std::enable_if<!std::is_same<ET, F32>::value, int>::type = 0>
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = ET::pack2(f[2 * i], f[2 * i + 1]);
  return make_uint4(w[0], w[1], w[2], w[3]);
}
template <typename ET, typename std::enable_if<std::is_same<ET, F32>::value, int>::type = 0>
__device__ __forceinline__ F32::V8 pack8(const float (&f)[8]) {
  F32::V8 v;
  v.a = make_float4(f[0], f[1], f[2], f[3]); v.b = make_float4(f[4], f[5], f[6], f[7]);
  return v;
}

// dispatch on the runtime dtype enum (16-bit storage: every kernel; IMM_DISPATCH_DTYPE_F32 below adds f32 storage for the
// HBM-bound passes that are instantiated on it)
#define IMM_DISPATCH_DTYPE(dtype, ...)                                          \
  do {                                                                          \
    if ((dtype) == IMM_BF16) { using ET = BF16; __VA_ARGS__; }                  \
    else if ((dtype) == IMM_F16) { using ET = F16; __VA_ARGS__; }               \
    else return imm_fail(IMM_E_INVALID, "unknown dtype %d", (int)(dtype));      \
  } while (0)

#define IMM_DISPATCH_DTYPE_F32(dtype, ...)                                      \
  do {                                                                          \
    if ((dtype) == IMM_BF16) { using ET = BF16; __VA_ARGS__; }                  \
    else if ((dtype) == IMM_F16) { using ET = F16; __VA_ARGS__; }               \
    else if ((dtype) == IMM_F32) { using ET = F32; __VA_ARGS__; }               \
    else return imm_fail(IMM_E_INVALID, "unknown dtype %d", (int)(dtype));      \
  } while (0)

// ---------------------------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 256; result valid on every thread. `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

static inline int imm_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// IMM_CONV_DISABLE=<comma list>: take a specialised convolution kernel out of the dispatch so that the layer falls back to
// the next more general one (A/B timing and the fallback paths' tests): halo, halo2, hdeep, deepk, group, group32, wgrad_tr,
// wgrad_halo, s2d / s2d_halo (one-launch stride-2 data gradient, deep-K / whole-filter form -> the four class launches), nol (imm_conv2d_nol_supported -> 0),
// hdeep6 (six-k-steps-per-barrier 16x16x128 tile -> conv_hdeep's tap-at-a-time form), s2f (stride-2 forward LDS-halo kernel -> im2col).  The ONLY dispatch switch the kernels read; every tuning constant is compiled in.
bool imm_conv_disabled(const char* name);
// The CU count a persistent / chip-sized launch of THIS THREAD sizes its grid for: the device's, or the caller's smaller
// imm_set_cu_limit() value (runtime.hip) — a launch confined to a share of the chip leaves the other CUs to the kernels of a
// concurrent stream (the frozen VGG's ground-truth half beside the encoder chains, imm_amd/engine.py).
int imm_limit_cus(int device_cus);
