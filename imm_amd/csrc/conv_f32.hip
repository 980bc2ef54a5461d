// conv_f32.hip — the convolutions of the f32-storage WITNESS engine (IMM_F32; round 6).
//
// The reference computes in fp32 (imm/models/imm_model.py:97 `dtype=tf.float32`; tf.nn.conv2d at imm/tf_utils/nn_utils.py:100,
// imm/models/selfsup/vgg16.py:182 and their tf.gradients).  The product engine stores activations in 16 bits and contracts on
// the matrix cores; its distance to the fp32 oracle at initialisation is storage noise that no 16-bit engine can tell from a
// wiring error.  This file gives the SAME launch program an exact-arithmetic form: plain f32 FMA implicit-GEMM
// kernels behind the same entry points (imm_conv2d, imm_conv2d_wgrad with dtype IMM_F32), same descriptors, same packed filter
// layouts (f32 elements), same epilogue semantics as conv_common.h.  A test instrument: LDS-tiled SGEMM form, no MFMA (gfx950's
// f32 MFMA rounds like these FMAs but would need a third copy of the tile code), no tuning, ~100x slower than the bf16 kernels.
//
//   y[m][n] = epilogue( sum_k gather(x)[m][k] * wt[n][k] ),  k = (ky * kw + kx) * ci + c
//   slab[s][k][n] = sum_{m in split s} gather(x)[m][k] * dy[m][n]
#include "common.h"

namespace {
constexpr int F_BM = 64, F_BN = 64, F_BK = 16, F_THREADS = 256;

struct ConvF32Args {
  const float* x; const float* wt; const float* bias; float* y; const float* mask;
  int M, hi, wi, ci, ldx, ho, wo, co, ldy, kh, kw, stride, pad_t, pad_l, updiv, kpad, ktot, flags, ldmask;
  int oscale, ooff_y, ooff_x;
};

// gather(x)[m][k]: pixel m of THIS launch (dense index over batch x ho x wo), k = tap * ci + c
__device__ __forceinline__ float gather_x(const ConvF32Args& a, int m, int k) {
  if (m >= a.M || k >= a.ktot) return 0.f;
  const int hw = a.ho * a.wo;
  const int img = m / hw, rem = m - img * hw;
  const int oy = rem / a.wo, ox = rem - oy * a.wo;
  const int tap = k / a.ci, c = k - tap * a.ci;
  const int ky = tap / a.kw, kx = tap - ky * a.kw;
  int iy = oy * a.stride - a.pad_t + ky, ix = ox * a.stride - a.pad_l + kx;
  if (a.updiv == 2) {                       // transposed gather of a stride-2 data gradient: only even positions hit a dy pixel
    if ((iy | ix) & 1) return 0.f;
    iy >>= 1; ix >>= 1;
  }
  if ((unsigned)iy >= (unsigned)a.hi || (unsigned)ix >= (unsigned)a.wi) return 0.f;
  return a.x[(((int64_t)img * a.hi + iy) * a.wi + ix) * a.ldx + c];
}

__global__ __launch_bounds__(F_THREADS) void conv_f32_kernel(const ConvF32Args a) {
  __shared__ float As[F_BK][F_BM + 4];
  __shared__ float Bs[F_BK][F_BN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.x * F_BM, n0 = blockIdx.y * F_BN;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < a.ktot; k0 += F_BK) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * F_THREADS;                 // 1024 = 16 x 64
      const int kk = idx & 15, mm = idx >> 4;
      As[kk][mm] = gather_x(a, m0 + mm, k0 + kk);
      const int n = n0 + mm, k = k0 + kk;
      Bs[kk][mm] = (n < a.co && k < a.ktot) ? a.wt[(int64_t)n * a.kpad + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < F_BK; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[kk][ty * 4 + i]; bv[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  // epilogue (conv_common.h order): + bias, ReLU, mask (v = 0 where mask_ref <= 0), store
  const bool f_bias = a.flags & IMM_CONV_BIAS, f_relu = a.flags & IMM_CONV_RELU, f_mask = a.flags & IMM_CONV_MASK;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mi = m0 + ty * 4 + i;
    if (mi >= a.M) continue;
    int64_t m = mi;
    if (a.oscale != 1) {
      const int hw = a.ho * a.wo;
      const int img = mi / hw, rem = mi - img * hw;
      const int oy = rem / a.wo, ox = rem - oy * a.wo;
      m = ((int64_t)img * a.ho * a.oscale + oy * a.oscale + a.ooff_y) * (a.wo * a.oscale) + ox * a.oscale + a.ooff_x;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= a.co) continue;
      float v = acc[i][j] + (f_bias ? a.bias[n] : 0.f);
      if (f_relu) v = fmaxf(v, 0.f);
      if (f_mask && !(a.mask[m * a.ldmask + n] > 0.f)) v = 0.f;
      a.y[m * a.ldy + n] = v;
    }
  }
}

// Batch-norm partial rows of a forward convolution (IMM_CONV_STATS): with f32 storage the stored y IS the accumulator value,
// so the rows (sum v, sum v^2) are taken from y in a pass of their own: row r = the pixels [r * P, (r + 1) * P), P = ceil(M / R),
// R = imm_conv_stats_blocks(desc) — the row count the engine allocated (any partition of the pixels gives the same totals).
__global__ __launch_bounds__(256) void conv_f32_stats_kernel(const float* __restrict__ y, int M, int co, int ldy, int P,
                                                             float* __restrict__ partial) {
  const int r = blockIdx.x;
  const int p0 = r * P, p1 = min(M, p0 + P);
  for (int n = threadIdx.x; n < co; n += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int p = p0; p < p1; ++p) { const float v = y[(int64_t)p * ldy + n]; s1 += v; s2 = fmaf(v, v, s2); }
    partial[((int64_t)r * 2 + 0) * co + n] = s1;
    partial[((int64_t)r * 2 + 1) * co + n] = s2;
  }
}

struct WgradF32Args {
  const float* x; const float* dy; float* slab;
  int M, hi, wi, ci, ldx, ho, wo, co, lddy, kh, kw, stride, pad_t, pad_l, kpad, ktot, per_split;
};

// slab[s][k][n] over the pixels of split s: tile = 64 k x 64 n, contraction over 16 pixels per trip
__global__ __launch_bounds__(F_THREADS) void wgrad_f32_kernel(const WgradF32Args w) {
  __shared__ float As[F_BK][F_BM + 4];       // [pixel][k]
  __shared__ float Bs[F_BK][F_BN + 4];       // [pixel][n]
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int k0 = blockIdx.x * F_BM, n0 = blockIdx.y * F_BN, s = blockIdx.z;
  const int p0 = s * w.per_split, p1 = min(w.M, p0 + w.per_split);
  ConvF32Args a;
  a.x = w.x; a.M = w.M; a.hi = w.hi; a.wi = w.wi; a.ci = w.ci; a.ldx = w.ldx; a.ho = w.ho; a.wo = w.wo; a.kh = w.kh; a.kw = w.kw;
  a.stride = w.stride; a.pad_t = w.pad_t; a.pad_l = w.pad_l; a.updiv = 1; a.ktot = w.ktot;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int pb = p0; pb < p1; pb += F_BK) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + u * F_THREADS;
      const int col = idx & 63, pp = idx >> 6;             // 16 pixels x 64 columns
      const int p = pb + pp;
      As[pp][col] = p < p1 ? gather_x(a, p, k0 + col) : 0.f;
      const int n = n0 + col;
      Bs[pp][col] = (p < p1 && n < w.co) ? w.dy[(int64_t)p * w.lddy + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int pp = 0; pp < F_BK; ++pp) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[pp][ty * 4 + i]; bv[i] = Bs[pp][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* out = w.slab + (int64_t)s * w.kpad * w.co;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + ty * 4 + i;
    if (k >= w.kpad) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n < w.co) out[(int64_t)k * w.co + n] = acc[i][j];       // (rows k >= ktot hold zeros: gather_x returns 0 there)
    }
  }
}
}  // namespace

int imm_conv_f32(const imm_conv_desc* d, const void* x, const void* wt, const float* bias, void* y, float* stats_partial,
                 const void* mask_ref, hipStream_t s) {
  ConvF32Args a;
  a.x = (const float*)x; a.wt = (const float*)wt; a.bias = bias; a.y = (float*)y; a.mask = (const float*)mask_ref;
  a.M = d->batch * d->ho * d->wo; a.hi = d->hi; a.wi = d->wi; a.ci = d->ci; a.ldx = d->ldx; a.ho = d->ho; a.wo = d->wo; a.co = d->co;
  a.ldy = d->ldy; a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.updiv = d->updiv;
  a.kpad = d->kpad; a.ktot = d->kh * d->kw * d->ci; a.flags = d->flags; a.ldmask = d->ldmask;
  a.oscale = d->out_scale > 1 ? d->out_scale : 1; a.ooff_y = d->out_off_y; a.ooff_x = d->out_off_x;
  if ((d->flags & IMM_CONV_STATS) && ((d->flags & IMM_CONV_MASK) || a.oscale != 1))
    return imm_fail(IMM_E_UNSUPPORTED, "conv(f32): batch-norm partial sums are served for dense forward convolutions only");
  const dim3 grid((a.M + F_BM - 1) / F_BM, (a.co + F_BN - 1) / F_BN);
  hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(F_THREADS), 0, s, a);
  if (d->flags & IMM_CONV_STATS) {
    const int R = imm_conv_stats_blocks(d);
    if (R <= 0) return IMM_E_INVALID;
    const int P = (a.M + R - 1) / R;
    hipLaunchKernelGGL(conv_f32_stats_kernel, dim3(R), dim3(256), 0, s, (const float*)y, a.M, a.co, a.ldy, P, stats_partial);
  }
  IMM_CHECK_LAUNCH("imm_conv2d(f32)");
  return 0;
}

int imm_conv_f32_wgrad(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit, hipStream_t s) {
  WgradF32Args w;
  w.x = (const float*)x; w.dy = (const float*)dy; w.slab = slab;
  w.M = d->batch * d->ho * d->wo; w.hi = d->hi; w.wi = d->wi; w.ci = d->ci; w.ldx = d->ldx; w.ho = d->ho; w.wo = d->wo; w.co = d->co;
  w.lddy = lddy; w.kh = d->kh; w.kw = d->kw; w.stride = d->stride; w.pad_t = d->pad_t; w.pad_l = d->pad_l; w.kpad = d->kpad;
  w.ktot = d->kh * d->kw * d->ci;
  w.per_split = (w.M + nsplit - 1) / nsplit;
  const dim3 grid((w.kpad + F_BM - 1) / F_BM, (w.co + F_BN - 1) / F_BN, nsplit);
  hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(F_THREADS), 0, s, w);
  IMM_CHECK_LAUNCH("imm_conv2d_wgrad(f32)");
  return 0;
}
