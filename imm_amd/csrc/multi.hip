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
  typename ET::T* __restrict__ wt = (typename ET::T*)jb[1];
  const int mode = (int)jb[2], kh = (int)jb[3], kw = (int)jb[4], ci_real = (int)jb[5], co_real = (int)jb[6];
  const int c_pad = (int)jb[7], rows = (int)jb[8], kpad = (int)jb[9];
  if (mode == 0) {
    // forward layout Wt[n][tap*c_pad + c] <- w[tap][c][n]: the source is contiguous along n, the destination along k.  A
    // workgroup transposes one 32 (n) x 64 (k) tile through LDS: 128-byte runs on both sides (the element-wise form below read
    // one float per 64-byte sector: 16x the traffic, 31 us per step for 16 MB of filters).  Tiles of a job: ceil(rows / 32) x
    // ceil(kpad / 64), k fastest (imm_pack_weights_multi_blocks).
    __shared__ float tile[64][33];
    const int t = blockIdx.x - blk_first[j], tiles_k = (kpad + 63) >> 6;
    const int n0 = (t / tiles_k) * 32, k0 = (t % tiles_k) * 64;
    const int nn = threadIdx.x & 31;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = (threadIdx.x >> 5) + 8 * i, k = k0 + kk;
      const int tap = k / c_pad, c = k - tap * c_pad;
      float v = 0.f;
      if (tap < kh * kw && c < ci_real && n0 + nn < co_real) v = w[((int64_t)tap * ci_real + c) * co_real + n0 + nn];
      tile[kk][nn] = v;
    }
    __syncthreads();
    const int rn = threadIdx.x >> 3, kq = (threadIdx.x & 7) * 8;
    if (n0 + rn < rows && k0 + kq < kpad) {          // kpad % 8 == 0
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = tile[kq + e][rn];
      st8<ET>(wt + (int64_t)(n0 + rn) * kpad + k0 + kq, pack8<ET>(f));
    }
    return;
  }
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
      } else if (mode == 1) {
        const int kyf = kh - 1 - tap / kw, kxf = kw - 1 - tap % kw;
        if (n < ci_real && c < co_real) v = w[(((int64_t)kyf * kw + kxf) * ci_real + n) * co_real + c];
      } else {      // parity class (py,px) of the stride-2 data gradient
        const int py = ((mode - 4) >> 1) & 1, px = (mode - 4) & 1;
        const int ny = (kh - py + 1) / 2, nx = (kw - px + 1) / 2;
        if (tap < ny * nx) {
          const int jy = tap / nx, jx = tap - jy * nx;
          const int kyf = py + 2 * (ny - 1 - jy), kxf = px + 2 * (nx - 1 - jx);
          if (n < ci_real && c < co_real) v = w[(((int64_t)kyf * kw + kxf) * ci_real + n) * co_real + c];
        }
      }
    }
    f[e] = v;
  }
  st8<ET>(wt + base, pack8<ET>(f));
}

// workgroups of one job of imm_pack_weights_multi (the caller's blk_first prefix sums are built from these)
extern "C" int imm_pack_weights_multi_blocks(int mode, int rows, int kpad) {
  if (rows <= 0 || kpad <= 0 || kpad % 8) return IMM_E_INVALID;
  if (mode == 0) return ((rows + 31) / 32) * ((kpad + 63) / 64);
  return (int)(((int64_t)rows * kpad + 2047) / 2048);
}

extern "C" int imm_pack_weights_multi(const int64_t* jobs, const int32_t* blk_first, int n_jobs, int n_blocks, int dtype,
                                      void* stream) {
  IMM_REQUIRE(jobs && blk_first && n_jobs > 0 && n_blocks > 0, "pack_weights_multi: args");
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((pack_weights_multi_kernel<ET>), dim3(n_blocks), dim3(256), 0,
                                               (hipStream_t)stream, jobs, blk_first, n_jobs));
  IMM_CHECK_LAUNCH("imm_pack_weights_multi");
  return 0;
}

// job: {slab, dw, nsplit, ntaps, ci_pad, ci_real, co, kpad, lanes, 0...}.  A workgroup = 256/lanes quads of 4 consecutive
// outputs x `lanes` split lanes (lane sl sums the slabs sl, sl + lanes, ...; 16-byte loads), then a fixed-order LDS reduction
// over the split lanes: the per-lane dependent chain is nsplit/lanes loads (the kernel is latency-, not bandwidth-bound).
// lanes in {1, 2, 4, 8, 16} (0 = 16) is the caller's choice per job, about the split count: since the multi-problem filter-
// gradient launches most layers have 1-8 slabs, and 16 split lanes left 15/16 of every workgroup idle (64 k workgroups of 16
// live threads each for a step's gradients: 74 us).  Workgroups per job = ceil(outputs / (1024 / lanes)).
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(const int64_t* __restrict__ jobs,
                                                                const int32_t* __restrict__ blk_first, int n_jobs) {
  __shared__ float4 red[256 + 16];
  const int j = find_job(blk_first, n_jobs, blockIdx.x);
  const int64_t* jb = jobs + (int64_t)j * MULTI_FIELDS;
  const float* __restrict__ slab = (const float*)jb[0];
  float* __restrict__ dw = (float*)jb[1];
  const int nsplit = (int)jb[2], ntaps = (int)jb[3], ci_pad = (int)jb[4], ci_real = (int)jb[5], co = (int)jb[6];
  const int kpad = (int)jb[7];
  const int SL = (int)jb[8] > 0 ? (int)jb[8] : 16, QL = 256 / SL;
  const int64_t total = (int64_t)ntaps * ci_real * co;
  const int64_t sstride = (int64_t)kpad * co;
  const int ql = threadIdx.x % QL, sl = threadIdx.x / QL;
  const int64_t idx = ((int64_t)(blockIdx.x - blk_first[j]) * QL + ql) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool vec = (co & 3) == 0;
  if (idx < total) {
    if (vec) {
      const int n = (int)(idx % co);
      const int64_t tc = idx / co;
      const int c = (int)(tc % ci_real), tap = (int)(tc / ci_real);
      const float* sp = slab + ((int64_t)tap * ci_pad + c) * co + n;
      float4 a1 = make_float4(0.f, 0.f, 0.f, 0.f);
      int s = sl;
      for (; s + SL < nsplit; s += 2 * SL) {
        const float4 v0 = *(const float4*)(sp + (int64_t)s * sstride), v1 = *(const float4*)(sp + (int64_t)(s + SL) * sstride);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
      }
      if (s < nsplit) {
        const float4 v0 = *(const float4*)(sp + (int64_t)s * sstride);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      }
      acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
    } else {
      float* ap = &acc.x;
      for (int e = 0; e < 4; ++e) {
        const int64_t id = idx + e;
        if (id >= total) break;
        const int n = (int)(id % co);
        const int64_t tc = id / co;
        const int c = (int)(tc % ci_real), tap = (int)(tc / ci_real);
        const float* sp = slab + ((int64_t)tap * ci_pad + c) * co + n;
        float t = 0.f;
        for (int s = sl; s < nsplit; s += SL) t += sp[s * sstride];
        ap[e] = t;
      }
    }
  }
  if (SL > 1) {
    red[sl * (QL + 1) + ql] = acc;
    __syncthreads();
    if (sl == 0 && idx < total) {
      for (int s = 1; s < SL; ++s) { const float4 v = red[s * (QL + 1) + ql]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
  }
  if (sl == 0 && idx < total) {
    if (idx + 3 < total && ((uintptr_t)(dw + idx) & 15) == 0) *(float4*)(dw + idx) = acc;
    else {
      dw[idx] = acc.x;
      if (idx + 1 < total) dw[idx + 1] = acc.y;
      if (idx + 2 < total) dw[idx + 2] = acc.z;
      if (idx + 3 < total) dw[idx + 3] = acc.w;
    }
  }
}

extern "C" int imm_wgrad_reduce_multi(const int64_t* jobs, const int32_t* blk_first, int n_jobs, int n_blocks, void* stream) {
  IMM_REQUIRE(jobs && blk_first && n_jobs > 0 && n_blocks > 0, "wgrad_reduce_multi: args");
  hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, jobs, blk_first, n_jobs);
  IMM_CHECK_LAUNCH("imm_wgrad_reduce_multi");
  return 0;
}
