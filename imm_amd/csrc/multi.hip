// multi.hip — table-driven launches: one grid serves many small per-tensor jobs (weight re-packing,
// split-K slab reduction) so a training step does not pay ~70 launch gaps for microsecond kernels.
// A job table is a device array of int64[MULTI_FIELDS]; `blk_first[j]` is the first workgroup of job j
// (prefix sums, blk_first[n_jobs] = grid size); each workgroup binary-searches its job.
#include "common.h"

#define MULTI_FIELDS 12

__device__ __forceinline__ int find_job(const int32_t* __restrict__ blk_first, int n_jobs, int blk) {
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (blk_first[mid] <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// job: {w, wt, mode, kh, kw, ci_real, co_real, c_pad, rows, kpad, 0, 0}; 256 threads x 8 elements per workgroup
template <typename ET>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const int64_t* __restrict__ jobs,
                                                                const int32_t* __restrict__ blk_first, int n_jobs) {
  const int j = find_job(blk_first, n_jobs, blockIdx.x);
  const int64_t* jb = jobs + (int64_t)j * MULTI_FIELDS;
  const float* __restrict__ w = (const float*)jb[0];
  uint16_t* __restrict__ wt = (uint16_t*)jb[1];
  const int mode = (int)jb[2], kh = (int)jb[3], kw = (int)jb[4], ci_real = (int)jb[5], co_real = (int)jb[6];
  const int c_pad = (int)jb[7], rows = (int)jb[8], kpad = (int)jb[9];
  const int64_t total = (int64_t)rows * kpad;
  const int64_t base = ((int64_t)(blockIdx.x - blk_first[j]) * 256 + threadIdx.x) * 8;
  if (base >= total) return;
  // kpad % 8 == 0: the 8 elements share a row; c_pad % 8 == 0: they share a tap
  const int n = (int)(base / kpad), k0 = (int)(base - (int64_t)n * kpad);
  const int tap = k0 / c_pad, c0 = k0 - tap * c_pad;
  float f[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    float v = 0.f;
    if (tap < kh * kw) {
      if (mode == 0) {
        if (n < co_real && c < ci_real) v = w[((int64_t)tap * ci_real + c) * co_real + n];
      } else {
        const int kyf = kh - 1 - tap / kw, kxf = kw - 1 - tap % kw;
        if (n < ci_real && c < co_real) v = w[(((int64_t)kyf * kw + kxf) * ci_real + n) * co_real + c];
      }
    }
    f[e] = v;
  }
  *(uint4*)(wt + base) = pack8<ET>(f);
}

extern "C" int imm_pack_weights_multi(const int64_t* jobs, const int32_t* blk_first, int n_jobs, int n_blocks, int dtype,
                                      void* stream) {
  IMM_REQUIRE(jobs && blk_first && n_jobs > 0 && n_blocks > 0, "pack_weights_multi: args");
  IMM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_weights_multi_kernel<ET>), dim3(n_blocks), dim3(256), 0,
                                               (hipStream_t)stream, jobs, blk_first, n_jobs));
  IMM_CHECK_LAUNCH("imm_pack_weights_multi");
  return 0;
}

// job: {slab, dw, nsplit, ntaps, ci_pad, ci_real, co, kpad, 0...}; 1024 outputs per workgroup (4 per lane)
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const int64_t* __restrict__ jobs,
                                                                const int32_t* __restrict__ blk_first, int n_jobs) {
  const int j = find_job(blk_first, n_jobs, blockIdx.x);
  const int64_t* jb = jobs + (int64_t)j * MULTI_FIELDS;
  const float* __restrict__ slab = (const float*)jb[0];
  float* __restrict__ dw = (float*)jb[1];
  const int nsplit = (int)jb[2], ntaps = (int)jb[3], ci_pad = (int)jb[4], ci_real = (int)jb[5], co = (int)jb[6];
  const int kpad = (int)jb[7];
  const int64_t total = (int64_t)ntaps * ci_real * co;
  const int64_t sstride = (int64_t)kpad * co;
  if ((co & 3) == 0) {
    // 4 consecutive output channels per lane: 16-byte slab loads, 4 independent splits in flight
    const int64_t idx = ((int64_t)(blockIdx.x - blk_first[j]) * 256 + threadIdx.x) * 4;
    if (idx >= total) return;
    const int n = (int)(idx % co);
    const int64_t tc = idx / co;
    const int c = (int)(tc % ci_real), tap = (int)(tc / ci_real);
    const float* sp = slab + ((int64_t)tap * ci_pad + c) * co + n;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
    int s = 0;
    for (; s + 3 < nsplit; s += 4) {
      const float4 v0 = *(const float4*)(sp + (int64_t)s * sstride), v1 = *(const float4*)(sp + (int64_t)(s + 1) * sstride);
      const float4 v2 = *(const float4*)(sp + (int64_t)(s + 2) * sstride), v3 = *(const float4*)(sp + (int64_t)(s + 3) * sstride);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
      a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
    }
    for (; s < nsplit; ++s) {
      const float4 v0 = *(const float4*)(sp + (int64_t)s * sstride);
      a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    }
    dw[idx] = (a0.x + a1.x) + (a2.x + a3.x);
    dw[idx + 1] = (a0.y + a1.y) + (a2.y + a3.y);
    dw[idx + 2] = (a0.z + a1.z) + (a2.z + a3.z);
    dw[idx + 3] = (a0.w + a1.w) + (a2.w + a3.w);
    return;
  }
  for (int e = 0; e < 4; ++e) {
    const int64_t idx = ((int64_t)(blockIdx.x - blk_first[j]) * 256 + threadIdx.x) * 4 + e;
    if (idx >= total) return;
    const int n = (int)(idx % co);
    const int64_t tc = idx / co;
    const int c = (int)(tc % ci_real), tap = (int)(tc / ci_real);
    const float* sp = slab + ((int64_t)tap * ci_pad + c) * co + n;
    float acc = 0.f;
    for (int s = 0; s < nsplit; ++s) acc += sp[s * sstride];
    dw[idx] = acc;
  }
}

extern "C" int imm_wgrad_reduce_multi(const int64_t* jobs, const int32_t* blk_first, int n_jobs, int n_blocks, void* stream) {
  IMM_REQUIRE(jobs && blk_first && n_jobs > 0 && n_blocks > 0, "wgrad_reduce_multi: args");
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, jobs, blk_first, n_jobs);
  IMM_CHECK_LAUNCH("imm_wgrad_reduce_multi");
  return 0;
}
