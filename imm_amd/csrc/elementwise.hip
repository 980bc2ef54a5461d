// elementwise.hip — HBM-bound passes of the IMM step on gfx950: batch-norm finalize/apply/backward,
// bilinear resampling, 2x2 max pooling, input packing, bias-gradient column sums.
// All 16-bit tensors are NHWC with C % 8 == 0 and pixel stride % 8 == 0, so every lane moves
// 16-byte vectors (8 channels); reductions write per-block partials that a finalize kernel sums in
// a fixed order (deterministic, no atomics, nothing to zero).
//
// Reference call sites: tf.layers.batch_normalization(fused=True)+relu imm/tf_utils/nn_utils.py:201-209;
// tf.image.resize_images imm/models/imm_model.py:175; resize_bilinear(align_corners=True) :334;
// tf.nn.max_pool imm/models/selfsup/ops.py:16-26; tf.nn.bias_add gradient nn_utils.py:108.
#include "common.h"
#include <stdlib.h>

#define EW_THREADS 256

static inline int ew_blocks(int64_t work_items, int cap = 16384) {
  int64_t b = (work_items + EW_THREADS - 1) / EW_THREADS;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

#define EW_REQUIRE_VEC(c, ld, name) \
  IMM_REQUIRE((c) > 0 && (c) % 8 == 0 && (ld) % 8 == 0 && (ld) >= (c), name ": C=%d ld=%d must be multiples of 8", (int)(c), (int)(ld))

// ---------------------------------------------------------------------------------------------
// input packing
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ void pack_image_kernel(const float* __restrict__ src, uint4* __restrict__ dst, int64_t npix) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (int64_t)gridDim.x * blockDim.x) {
    const float f[8] = {src[p * 3], src[p * 3 + 1], src[p * 3 + 2], 0.f, 0.f, 0.f, 0.f, 0.f};
    dst[p] = pack8<ET>(f);
  }
}

extern "C" int imm_pack_image(const float* src, void* dst, int dtype, int64_t npix, void* stream) {
  IMM_REQUIRE(src && dst && npix > 0, "pack_image: args");
  IMM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_image_kernel<ET>), dim3(ew_blocks(npix)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, src, (uint4*)dst, npix));
  IMM_CHECK_LAUNCH("imm_pack_image");
  return 0;
}

// First encoder convolution (7x7x3, imm/models/imm_model.py:190): a 3-channel NHWC image gives the
// matrix cores a K-chunk of 3.  Unroll the kw horizontal taps into the channel axis instead:
// dst[b,y,x, kx*3+ch] = src[b,y,x+kx-pad_l,ch] (zero outside, channels kw*3..ld-1 zero), so the
// layer becomes a (kh x 1) convolution over ld=32 channels whose K tile is one vertical tap.
template <typename ET>
__global__ void pack_image_taps_kernel(const float* __restrict__ src, typename ET::T* __restrict__ dst, int batch, int h, int w,
                                       int kw, int pad_l, int ld) {
  const int c8n = ld / 8;
  const int64_t total = (int64_t)batch * h * w * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    const int64_t p = idx / c8n;
    const int x = (int)(p % w);
    const int64_t row = p - x;          // pixel index of (b, y, 0)
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
      const int kx = c / 3, ch = c - kx * 3;
      const int xx = x + kx - pad_l;
      f[e] = (kx < kw && xx >= 0 && xx < w) ? src[(row + xx) * 3 + ch] : 0.f;
    }
    st8<ET>(dst + p * ld + cg * 8, pack8<ET>(f));
  }
}

// The same through LDS, for ld == 32 and kw <= 10 (the 7x7 first layer: 21 of 32 channels): a workgroup converts RPB whole
// image rows to 16-bit once ([row][pad_l + w + kw - 1 - pad_l][3], zero borders) and every thread assembles the 16 bytes
// of one (pixel, 8-channel chunk) from 8 LDS values — the pass is bound by its 64-byte-per-pixel output (33.5 MB at batch 32:
// 19-25 us with eight scalar global loads + divisions per thread, the store-bound form ~8 us).
#define PIT_ROWS 2
template <typename ET>
__global__ __launch_bounds__(256) void pack_image_taps_rows_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                                                   int nrows, int w, int kw, int pad_l) {
  extern __shared__ uint16_t srow[];                    // [PIT_ROWS][(w + kw - 1) * 3] 16-bit, zero where x is outside
  const int wp = w + kw - 1, rowlen = wp * 3;
  const int r0 = blockIdx.x * PIT_ROWS;
  for (int i = threadIdx.x; i < PIT_ROWS * rowlen; i += 256) {
    const int r = i / rowlen, j = i - r * rowlen;
    const int xx = j / 3 - pad_l;
    float v = 0.f;
    if (r0 + r < nrows && xx >= 0 && xx < w) v = src[((int64_t)(r0 + r) * w + xx) * 3 + (j - (j / 3) * 3)];
    srow[i] = ET::from_f32(v);
  }
  __syncthreads();
  // dst[row][x][kx*3+ch] = srow[row][(x + kx)*3 + ch] = srow[row][x*3 + c], c = kx*3+ch < 3*kw: a contiguous run of the row
  for (int i = threadIdx.x; i < PIT_ROWS * w * 4; i += 256) {
    const int cg = i & 3, px = i >> 2;
    const int r = px / w, x = px - r * w;
    if (r0 + r >= nrows) break;
    const uint16_t* sp = srow + r * rowlen + x * 3 + cg * 8;
    uint16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (cg * 8 + e < 3 * kw) ? sp[e] : (uint16_t)0;
    *(uint4*)(dst + ((int64_t)(r0 + r) * w + x) * 32 + cg * 8) =
        make_uint4(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16), o[4] | ((uint32_t)o[5] << 16), o[6] | ((uint32_t)o[7] << 16));
  }
}

extern "C" int imm_pack_image_taps(const float* src, void* dst, int dtype, int batch, int h, int w, int kw, int pad_l,
                                   int ld, void* stream) {
  IMM_REQUIRE(src && dst && batch > 0 && h > 0 && w > 0 && kw > 0 && ld % 8 == 0 && ld >= 3 * kw, "pack_image_taps: args");
  if (dtype != IMM_F32 && ld == 32 && pad_l >= 0 && pad_l < kw && (w + kw - 1) * 3 * PIT_ROWS * 2 <= 48 * 1024) {
    const int nrows = batch * h;
    const size_t lds = (size_t)PIT_ROWS * (w + kw - 1) * 3 * 2 + 64;      // + slack: the last chunk of a row reads <= 11 values past it
    IMM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((pack_image_taps_rows_kernel<ET>), dim3((nrows + PIT_ROWS - 1) / PIT_ROWS), dim3(256),
                                                 lds, (hipStream_t)stream, src, (uint16_t*)dst, nrows, w, kw, pad_l));
    IMM_CHECK_LAUNCH("imm_pack_image_taps");
    return 0;
  }
  const int64_t total = (int64_t)batch * h * w * (ld / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((pack_image_taps_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, src, (typename ET::T*)dst, batch, h, w, kw, pad_l, ld));
  IMM_CHECK_LAUNCH("imm_pack_image_taps");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// batch norm forward
// ---------------------------------------------------------------------------------------------
// Sums the per-block partials of one 32-channel group with a 1024-thread block (f64 accumulation, fixed order =>
// deterministic); threads (x, y = 0) receive the totals of channel ch = 32*blockIdx.x + x.  The pass is latency-bound
// (up to 4096 partial rows, one block per 32 channels): 16-byte loads, 4 independent row chains per thread, then a
// fixed-order LDS reduction over the row lanes.  c % 4 == 0 takes the vector path.
template <int NS, int CPB = 32>
__device__ __forceinline__ void reduce_partials_32x32(const float* __restrict__ partial, int nblk, int c, int ch,
                                                      double (&out)[NS], int ldp = 0) {
  if (ldp == 0) ldp = c;                       // row layout [NS][ldp]: ldp > c when the producer wrote wider rows
  constexpr int QPB = CPB / 4;                 // float4 columns per sum
  constexpr int COLS = QPB * NS;               // float4 columns of this block's CPB channels x NS sums
  constexpr int RL = 1024 / COLS;              // row lanes
  __shared__ double red[RL][NS * CPB + 1];
  const int tid = threadIdx.y * 32 + threadIdx.x;
  const int ch0 = blockIdx.x * CPB;
  if ((c & 3) == 0) {
    const int col = tid % COLS, rl = tid / COLS;
    const int sidx = col / QPB, q4 = (col % QPB) * 4;
    const bool live = ch0 + q4 < c;            // c % 4 == 0: a float4 is entirely inside or outside
    double acc[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[u][e] = 0.0;
    if (live) {
      const float* base = partial + (int64_t)sidx * ldp + ch0 + q4;
      int b = rl;
      for (; b + 7 * RL < nblk; b += 8 * RL) {        // 8 loads in flight: the pass is a chain of L2 round trips
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(base + (int64_t)(b + u * RL) * NS * ldp);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc[u & 3][0] += (double)v[u].x; acc[u & 3][1] += (double)v[u].y; acc[u & 3][2] += (double)v[u].z; acc[u & 3][3] += (double)v[u].w;
        }
      }
      for (; b + 3 * RL < nblk; b += 4 * RL) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(base + (int64_t)(b + u * RL) * NS * ldp);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u][0] += (double)v[u].x; acc[u][1] += (double)v[u].y; acc[u][2] += (double)v[u].z; acc[u][3] += (double)v[u].w;
        }
      }
      for (; b < nblk; b += RL) {
        const float4 v = *(const float4*)(base + (int64_t)b * NS * ldp);
        acc[0][0] += (double)v.x; acc[0][1] += (double)v.y; acc[0][2] += (double)v.z; acc[0][3] += (double)v.w;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rl][sidx * CPB + q4 + e] = (acc[0][e] + acc[1][e]) + (acc[2][e] + acc[3][e]);
  } else {
    // scalar fallback (CPB = 32 only): thread (x, y) walks rows y, y+32, ... of channel ch; row lanes >= 32 hold zeros
    double acc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s] = 0.0;
    if (ch < c && threadIdx.x < CPB) {
      for (int b = threadIdx.y; b < nblk; b += 32) {
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s] += (double)partial[((int64_t)b * NS + s) * ldp + ch];
      }
    }
    if (threadIdx.x < CPB) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        red[threadIdx.y][s * CPB + threadIdx.x] = acc[s];
        for (int r = threadIdx.y + 32; r < RL; r += 32) red[r][s * CPB + threadIdx.x] = 0.0;
      }
    }
  }
  __syncthreads();
  if (threadIdx.y == 0 && threadIdx.x < CPB) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
      for (int r = 0; r < RL; r += 4) {
        t0 += red[r][s * CPB + threadIdx.x]; t1 += red[r + 1][s * CPB + threadIdx.x];
        t2 += red[r + 2][s * CPB + threadIdx.x]; t3 += red[r + 3][s * CPB + threadIdx.x];
      }
      out[s] = (t0 + t1) + (t2 + t3);
    }
  }
}

// CPB channels per block: 32, or 8 when there are many partial rows (4x the blocks pulling them in)
template <int CPB>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int nblk, int c, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                   float momentum, int training, float* moving_mean, float* moving_var,
                                   float* scale, float* shift, float* mean_out, float* rstd_out) {
  const int ch = blockIdx.x * CPB + threadIdx.x;
  double s[2] = {0.0, 0.0};
  if (training) reduce_partials_32x32<2, CPB>(partial, nblk, c, ch, s);
  if (threadIdx.y != 0 || threadIdx.x >= CPB || ch >= c) return;
  float mean, var;
  if (training) {
    const double m = s[0] / count;
    double v = s[1] / count - m * m;
    if (v < 0.0) v = 0.0;
    mean = (float)m; var = (float)v;
    const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
    moving_mean[ch] = moving_mean[ch] * momentum + mean * (1.f - momentum);
    moving_var[ch] = moving_var[ch] * momentum + (float)unbiased * (1.f - momentum);
  } else {
    mean = moving_mean[ch]; var = moving_var[ch];
  }
  const float rstd = rsqrtf(var + eps);
  const float sc = gamma[ch] * rstd;
  scale[ch] = sc;
  shift[ch] = beta[ch] - mean * sc;
  mean_out[ch] = mean;
  rstd_out[ch] = rstd;
}

extern "C" int imm_bn_finalize(const float* partial, int nblk, int c, int64_t count, const float* gamma,
                               const float* beta, float eps, float momentum, int training, float* moving_mean,
                               float* moving_var, float* scale, float* shift, float* mean, float* rstd, void* stream) {
  IMM_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && mean && rstd, "bn_finalize: null");
  IMM_REQUIRE(!training || (partial && nblk > 0), "bn_finalize: training needs partial sums");
  IMM_REQUIRE(c > 0 && count > 0, "bn_finalize: dims");
  if (training && nblk >= 1024 && c % 8 == 0)
    hipLaunchKernelGGL(bn_finalize_kernel<8>, dim3(c / 8), dim3(32, 32), 0, (hipStream_t)stream, partial, nblk, c,
                       (double)count, gamma, beta, eps, momentum, training, moving_mean, moving_var, scale, shift, mean, rstd);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3((c + 31) / 32), dim3(32, 32), 0, (hipStream_t)stream, partial, nblk, c,
                       (double)count, gamma, beta, eps, momentum, training, moving_mean, moving_var, scale, shift, mean, rstd);
  IMM_CHECK_LAUNCH("imm_bn_finalize");
  return 0;
}

template <typename ET>
__global__ void bn_apply_kernel(const typename ET::T* __restrict__ y, int64_t npix, int c8n, int ldy,
                                const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                typename ET::T* __restrict__ x, int ldx) {
  const int64_t total = npix * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / c8n;
    const int cg = (int)(idx - p * c8n);
    float f[8];
    unpack8<ET>(ld8<ET>(y + p * ldy + cg * 8), f);
    const float4 sa = *(const float4*)(scale + cg * 8), sb = *(const float4*)(scale + cg * 8 + 4);
    const float4 ha = *(const float4*)(shift + cg * 8), hb = *(const float4*)(shift + cg * 8 + 4);
    const float sc[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
    const float sh[8] = {ha.x, ha.y, ha.z, ha.w, hb.x, hb.y, hb.z, hb.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f[i] = f[i] * sc[i] + sh[i];
      if (relu) f[i] = fmaxf(f[i], 0.f);
    }
    st8<ET>(x + p * ldx + cg * 8, pack8<ET>(f));
  }
}

extern "C" int imm_bn_apply_relu(const void* y, int dtype, int64_t npix, int c, int ldy, const float* scale,
                                 const float* shift, int relu, void* x_out, int ldx, void* stream) {
  IMM_REQUIRE(y && scale && shift && x_out && npix > 0, "bn_apply: null");
  EW_REQUIRE_VEC(c, ldy, "bn_apply(y)");
  EW_REQUIRE_VEC(c, ldx, "bn_apply(x)");
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((bn_apply_kernel<ET>), dim3(ew_blocks(npix * (c / 8))), dim3(EW_THREADS),
                                               0, (hipStream_t)stream, (const typename ET::T*)y, npix, c / 8, ldy, scale,
                                               shift, relu, (typename ET::T*)x_out, ldx));
  IMM_CHECK_LAUNCH("imm_bn_apply_relu");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// finalize + apply in ONE launch, for layers with few partial rows (the 16x16 .. 64x64 maps: rows <= 256)
// ---------------------------------------------------------------------------------------------
// Every workgroup owns a 32-channel slice x a pixel range and FIRST redoes the (tiny) finalize for its slice: nblk rows of
// 2 x 32 floats = nblk * 256 B from L2 (same rows, same fixed summation order in every workgroup => bit-identical
// scale / shift everywhere), then streams its pixels.  One launch and one kernel boundary less per layer than
// imm_bn_finalize + imm_bn_apply_relu, whose finalize is a 6-9 us latency chain for a microsecond of arithmetic.
// Workgroup (0, slice) also publishes scale / shift / mean / rstd for the backward pass and updates the moving statistics.
// Shared with the backward twin below: reduce NS sums of this workgroup's 32 channels over the partial rows.
template <int NS>
__device__ __forceinline__ void slice32_reduce(const float* __restrict__ partial, int nblk, int ldp, int ch0, double* out /*[NS*32] LDS*/) {
  constexpr int COLS = 8 * NS, RL = EW_THREADS / COLS;       // float4 columns x row lanes
  __shared__ double red[RL][NS * 32 + 1];
  const int tid = threadIdx.x, col = tid % COLS, rl = tid / COLS;
  const int sidx = col / 8, q4 = (col % 8) * 4;
  double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
  const float* base = partial + (int64_t)sidx * ldp + ch0 + q4;
  int b = rl;
  // eight rows per trip, all eight 16-byte loads issued before the first add (the two-row loop below walked a 256-row table as
  // eight L2 round trips in a row: ~4 of the 6-8 us of a small layer's launch); even rows -> a0, odd -> a1: the sums and
  // their order are those of the two-row loop
  for (; b + 7 * RL < nblk; b += 8 * RL) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *(const float4*)(base + (int64_t)(b + u * RL) * NS * ldp);
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      a0[0] += (double)v[u].x; a0[1] += (double)v[u].y; a0[2] += (double)v[u].z; a0[3] += (double)v[u].w;
      a1[0] += (double)v[u + 1].x; a1[1] += (double)v[u + 1].y; a1[2] += (double)v[u + 1].z; a1[3] += (double)v[u + 1].w;
    }
  }
  for (; b + RL < nblk; b += 2 * RL) {
    const float4 v0 = *(const float4*)(base + (int64_t)b * NS * ldp), v1 = *(const float4*)(base + (int64_t)(b + RL) * NS * ldp);
    a0[0] += (double)v0.x; a0[1] += (double)v0.y; a0[2] += (double)v0.z; a0[3] += (double)v0.w;
    a1[0] += (double)v1.x; a1[1] += (double)v1.y; a1[2] += (double)v1.z; a1[3] += (double)v1.w;
  }
  if (b < nblk) {
    const float4 v0 = *(const float4*)(base + (int64_t)b * NS * ldp);
    a0[0] += (double)v0.x; a0[1] += (double)v0.y; a0[2] += (double)v0.z; a0[3] += (double)v0.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[rl][sidx * 32 + q4 + e] = a0[e] + a1[e];
  __syncthreads();
  if (tid < NS * 32) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int r = 0; r < RL; r += 2) { t0 += red[r][tid]; t1 += red[r + 1][tid]; }
    out[tid] = t0 + t1;
  }
  __syncthreads();
}

template <typename ET>
__global__ __launch_bounds__(EW_THREADS) void bn_apply_fused_kernel(
    const float* __restrict__ partial, int nblk, int c, double count, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, int training, float* moving_mean, float* moving_var,
    float* scale, float* shift, float* mean_out, float* rstd_out, const typename ET::T* __restrict__ y, int64_t npix, int ldy,
    int relu, typename ET::T* __restrict__ x, int ldx, int px_per_blk, typename ET::T* __restrict__ up, int ldu, int h, int w) {
  __shared__ double sums[64];
  __shared__ float ssc[32], ssh[32];
  const int tid = threadIdx.x, ch0 = blockIdx.y * 32;
  const int q = tid & 3;                         // 16-byte chunk of the 64-byte slice of a pixel
  const int64_t p0 = (int64_t)blockIdx.x * px_per_blk;
  const int64_t p1 = p0 + px_per_blk < npix ? p0 + px_per_blk : npix;
  // this thread's first two pixels are requested before the finalize prologue (its L2 round trips hide their latency)
  const int64_t pf = p0 + (tid >> 2);
  constexpr int PSTEP = EW_THREADS / 4;
  typename ET::V8 raw0 = ET::zero8(), raw1 = raw0;
  if (up == nullptr && pf < p1) raw0 = ld8<ET>(y + pf * ldy + ch0 + q * 8);
  if (up == nullptr && pf + PSTEP < p1) raw1 = ld8<ET>(y + (pf + PSTEP) * ldy + ch0 + q * 8);
  if (training) slice32_reduce<2>(partial, nblk, c, ch0, sums);
  if (tid < 32) {
    const int ch = ch0 + tid;
    float mean, var;
    if (training) {
      const double m = sums[tid] / count;
      double v = sums[32 + tid] / count - m * m;
      if (v < 0.0) v = 0.0;
      mean = (float)m; var = (float)v;
      if (blockIdx.x == 0) {
        const double unbiased = count > 1.0 ? v * count / (count - 1.0) : v;
        moving_mean[ch] = moving_mean[ch] * momentum + mean * (1.f - momentum);
        moving_var[ch] = moving_var[ch] * momentum + (float)unbiased * (1.f - momentum);
      }
    } else {
      mean = moving_mean[ch]; var = moving_var[ch];
    }
    const float rstd = rsqrtf(var + eps);
    const float sc = gamma[ch] * rstd, sh = beta[ch] - mean * sc;
    ssc[tid] = sc; ssh[tid] = sh;
    if (blockIdx.x == 0) { scale[ch] = sc; shift[ch] = sh; mean_out[ch] = mean; rstd_out[ch] = rstd; }
  }
  __syncthreads();
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sc[i] = ssc[q * 8 + i]; sh[i] = ssh[q * 8 + i]; }
  auto normq = [&](const typename ET::V8& rawq, float (&f)[8]) {
    unpack8<ET>(rawq, f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f[i] = f[i] * sc[i] + sh[i];
      if (relu) f[i] = fmaxf(f[i], 0.f);
    }
  };
  auto norm = [&](int64_t pp, float (&f)[8]) { normq(ld8<ET>(y + pp * ldy + ch0 + q * 8), f); };
  if (up == nullptr) {
    // plain pass: two pixels per trip, the next trip's loads issued before this trip's stores
    for (int64_t p = pf; p < p1; p += 2 * PSTEP) {
      const bool two = p + PSTEP < p1;
      typename ET::V8 n0 = ET::zero8(), n1 = n0;
      if (p + 2 * PSTEP < p1) n0 = ld8<ET>(y + (p + 2 * PSTEP) * ldy + ch0 + q * 8);
      if (p + 3 * PSTEP < p1) n1 = ld8<ET>(y + (p + 3 * PSTEP) * ldy + ch0 + q * 8);
      float f[8];
      normq(raw0, f);
      st8<ET>(x + p * ldx + ch0 + q * 8, pack8<ET>(f));
      if (two) {
        normq(raw1, f);
        st8<ET>(x + (p + PSTEP) * ldx + ch0 + q * 8, pack8<ET>(f));
      }
      raw0 = n0; raw1 = n1;
    }
    return;
  }
  for (int64_t p = pf; p < p1; p += PSTEP) {
    float f[8];
    norm(p, f);
    if (x != nullptr) st8<ET>(x + p * ldx + ch0 + q * 8, pack8<ET>(f));      // (nobody reads the un-sampled tensor of an up-sampled block)
    {
      // x2 bilinear up-sampling of the normalised activation in the same pass (tf.image.resize_images, legacy
      // align_corners=False, imm_model.py:175: out[2i] = in[i], out[2i+1] = (in[i] + in[min(i+1, n-1)]) / 2), from the
      // 16-bit values the separate kernel would read back (same arithmetic, same results)
      const int j = (int)(p % w);
      const int64_t t = p / w;
      const int i = (int)(t % h);
      const int64_t b = t / h;
      float c00[8], c01[8], c10[8], c11[8], g[8];
      { const typename ET::V8 u = pack8<ET>(f); unpack8<ET>(u, c00); }
      const int64_t pr = p + (j + 1 < w ? 1 : 0), pd = p + (i + 1 < h ? w : 0), pdr = pd + (j + 1 < w ? 1 : 0);
      norm(pr, g); { const typename ET::V8 u = pack8<ET>(g); unpack8<ET>(u, c01); }
      norm(pd, g); { const typename ET::V8 u = pack8<ET>(g); unpack8<ET>(u, c10); }
      norm(pdr, g); { const typename ET::V8 u = pack8<ET>(g); unpack8<ET>(u, c11); }
      typename ET::T* o = up + (((b * 2 * h + 2 * i) * (int64_t)(2 * w)) + 2 * j) * ldu + ch0 + q * 8;
      float r01[8], r10[8], r11[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float top = c00[e] + (c01[e] - c00[e]) * 0.5f, bot = c10[e] + (c11[e] - c10[e]) * 0.5f;
        r01[e] = top; r10[e] = c00[e] + (c10[e] - c00[e]) * 0.5f; r11[e] = top + (bot - top) * 0.5f;
      }
      st8<ET>(o, pack8<ET>(c00));
      st8<ET>(o + ldu, pack8<ET>(r01));
      st8<ET>(o + (int64_t)2 * w * ldu, pack8<ET>(r10));
      st8<ET>(o + (int64_t)2 * w * ldu + ldu, pack8<ET>(r11));
    }
  }
}

static int fused_px_per_blk(int64_t npix, int c) {
  // ~512 workgroups, at least 256 pixels each (the per-workgroup finalize must stay small next to the streamed bytes)
  int64_t chunks = 512 / (c / 32);
  if (chunks < 1) chunks = 1;
  int64_t ppb = (npix + chunks - 1) / chunks;
  if (ppb < 256) ppb = 256;
  return (int)((ppb + 63) / 64 * 64);
}

extern "C" int imm_bn_apply_fused(const float* partial, int nblk, int c, int64_t count, const float* gamma, const float* beta,
                                  float eps, float momentum, int training, float* moving_mean, float* moving_var, float* scale,
                                  float* shift, float* mean, float* rstd, const void* y, int dtype, int ldy, int relu,
                                  void* x_out, int ldx, void* up2x_out, int ldu, int h, int w, void* stream) {
  IMM_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && mean && rstd && y && (x_out || up2x_out), "bn_apply_fused: null");
  IMM_REQUIRE(!training || (partial && nblk > 0), "bn_apply_fused: training needs partial sums");
  IMM_REQUIRE(c > 0 && c % 32 == 0 && count > 0, "bn_apply_fused: C=%d must be a multiple of 32", c);
  EW_REQUIRE_VEC(c, ldy, "bn_apply_fused(y)");
  if (x_out) EW_REQUIRE_VEC(c, ldx, "bn_apply_fused(x)");
  if (up2x_out) {
    EW_REQUIRE_VEC(c, ldu, "bn_apply_fused(up2x)");
    IMM_REQUIRE(h > 0 && w > 0 && count % ((int64_t)h * w) == 0, "bn_apply_fused: up-sampling needs the map size (h=%d w=%d)", h, w);
  }
  const int ppb = fused_px_per_blk(count, c);
  const dim3 grid((unsigned)((count + ppb - 1) / ppb), (unsigned)(c / 32));
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((bn_apply_fused_kernel<ET>), grid, dim3(EW_THREADS), 0, (hipStream_t)stream, partial,
                                               nblk, c, (double)count, gamma, beta, eps, momentum, training, moving_mean, moving_var,
                                               scale, shift, mean, rstd, (const typename ET::T*)y, count, ldy, relu, (typename ET::T*)x_out, ldx, ppb,
                                               (typename ET::T*)up2x_out, ldu, h, w));
  IMM_CHECK_LAUNCH("imm_bn_apply_fused");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Row pre-reduction: out[g][w] = sum of rows [g*group, (g+1)*group) of in[rows][width], f64 accumulation in row order.
// The layers with more than 256 partial rows (128x128 / 64x64 maps) used a finalize launch whose single workgroup per 8..32
// channels walks all rows (a 9-14 us latency chain per layer, 18 launches per step); 32-row groups reduced in parallel by
// rows/32 workgroups leave <= 32 rows, which the fused apply passes (imm_bn_apply_fused / imm_bn_bwd_apply_fused) finish.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void rows_reduce_kernel(const float* __restrict__ in, int rows, int width, int group,
                                                                 float* __restrict__ out) {
  const int w = blockIdx.y * EW_THREADS + threadIdx.x;
  if (w >= width) return;
  const int r0 = blockIdx.x * group, r1 = r0 + group < rows ? r0 + group : rows;
  const float* col = in + w;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int r = r0;
  for (; r + 7 < r1; r += 8) {                       // 8 loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(r + u) * width];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u & 3] += (double)v[u];
  }
  for (; r < r1; ++r) acc[0] += (double)col[(int64_t)r * width];
  out[(int64_t)blockIdx.x * width + w] = (float)((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

extern "C" int imm_rows_reduce(const float* in, int rows, int width, int group, float* out, void* stream) {
  IMM_REQUIRE(in && out && rows > 0 && width > 0 && group > 0, "rows_reduce: args");
  const dim3 grid((rows + group - 1) / group, (width + EW_THREADS - 1) / EW_THREADS);
  hipLaunchKernelGGL(rows_reduce_kernel, grid, dim3(EW_THREADS), 0, (hipStream_t)stream, in, rows, width, group, out);
  IMM_CHECK_LAUNCH("imm_rows_reduce");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// batch norm backward (+ fused ReLU backward):  dz = dout * [scale*y+shift > 0]
//   s1 = sum dz, s2 = sum dz*xhat;  dgamma = s2, dbeta = s1
//   dy = gamma*rstd * (dz - s1/N - xhat*s2/N)
// ---------------------------------------------------------------------------------------------
static int col_reduce_blocks(int64_t npix, int c) {
  const int tpp = c / 8;
  const int rows = EW_THREADS / tpp;
  // 4 pixels per lane for small tensors (the pass is latency-bound: more workgroups, shorter dependent chains),
  // capped at 1024 workgroups (= partial rows the finalize kernel has to sum)
  int64_t b = (npix + (int64_t)rows * 4 - 1) / ((int64_t)rows * 4);
  if (b < 1) b = 1;
  // <= 256 rows for the small tensors (<= 4M elements): imm_bn_bwd_apply_fused re-reduces the rows in every workgroup.  The
  // large ones keep up to 1024 workgroups: MEASURED 256 (one per CU, 8 loads in flight per thread) 3.594 ->
  // 3.631 ms per step, 512 3.587 — the pre-reduction of the rows (imm_rows_reduce) is cheaper than a thinner grid
  constexpr int64_t cap_env = 1024;
  const int64_t cap = (npix * c <= (1LL << 22)) ? 256 : cap_env;
  if (b > cap) b = cap;
  return (int)b;
}

extern "C" int imm_bn_bwd_blocks(int64_t npix, int c) {
  if (c <= 0 || c % 8 || c / 8 > EW_THREADS || EW_THREADS % (c / 8)) return IMM_E_UNSUPPORTED;
  return col_reduce_blocks(npix, c);
}
extern "C" int imm_colsum_blocks(int64_t npix, int c) { return imm_bn_bwd_blocks(npix, c); }

// shared tail: thread-private 8-channel accumulators (NS sums each) -> per-block partial[blk][NS][c]
// COHERENT: the row is stored with agent-scope (sc1, write-through) stores: visible to every XCD once acknowledged, without
// a release fence (which on this part is a write-back of the whole L2)
template <int NS, bool COHERENT = false>
__device__ __forceinline__ void col_reduce_tail(float (&acc)[NS][8], int c, int tpp, int rows, float* out_blk) {
  __shared__ float red[EW_THREADS * NS * 8];
  const int tid = threadIdx.x;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) red[(tid * NS + s) * 8 + i] = acc[s][i];
  __syncthreads();
  for (int t = tid; t < NS * c; t += EW_THREADS) {
    const int s = t / c, ch = t - s * c;
    const int cg = ch >> 3, e = ch & 7;
    float v = 0.f;
    for (int r = 0; r < rows; ++r) v += red[((r * tpp + cg) * NS + s) * 8 + e];
    if (COHERENT) __hip_atomic_store(out_blk + (int64_t)s * c + ch, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else out_blk[(int64_t)s * c + ch] = v;
  }
}

// UP: `dout` is the gradient of the x2 up-sampled tensor ([batch, 2h, 2w], imm_model.py:175); the adjoint of the up-sampling is
// gathered here (the arithmetic of upsample2x_bwd_kernel, same order: bitwise equal), stored 16-bit to dprev (the apply pass
// reads it) and taken as this pass's d_out — the standalone adjoint launch and the re-read of its result are gone.
template <typename ET, bool UP = false>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_reduce_kernel(
    const typename ET::T* __restrict__ dout, int lddo, const typename ET::T* __restrict__ y, int ldy, int64_t npix, int c,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int relu, float* __restrict__ partial, typename ET::T* __restrict__ dprev = nullptr,
    int lddp = 0, int h = 0, int w = 0) {
  const int tpp = c / 8, rows = EW_THREADS / tpp;
  const int r = threadIdx.x / tpp, cg = threadIdx.x - r * tpp;
  const int64_t per_blk = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per_blk;
  const int64_t p1 = p0 + per_blk < npix ? p0 + per_blk : npix;
  float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = scale[cg * 8 + i]; sh[i] = shift[cg * 8 + i]; mu[i] = mean[cg * 8 + i]; rs[i] = rstd[cg * 8 + i];
  }
  float acc[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  auto take = [&](const typename ET::V8& dq, const typename ET::V8& yq) {
    float d[8], v[8];
    unpack8<ET>(dq, d);
    unpack8<ET>(yq, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float dz = d[i];
      if (relu && !(v[i] * sc[i] + sh[i] > 0.f)) dz = 0.f;
      acc[0][i] += dz;
      acc[1][i] += dz * ((v[i] - mu[i]) * rs[i]);
    }
  };
  if constexpr (UP) {
    const int H = 2 * h, W = 2 * w;
    for (int64_t p = p0 + r; p < p1; p += rows) {
      const int j = (int)(p % w);
      const int64_t t = p / w;
      const int i = (int)(t % h);
      const int64_t b = t / h;
      int Ys[3], Xs[3]; float wy[3], wx[3];
      Ys[0] = 2 * i - 1; wy[0] = (i >= 1) ? 0.5f : 0.f;
      Ys[1] = 2 * i;     wy[1] = 1.f;
      Ys[2] = 2 * i + 1; wy[2] = (i == h - 1) ? 1.f : 0.5f;
      Xs[0] = 2 * j - 1; wx[0] = (j >= 1) ? 0.5f : 0.f;
      Xs[1] = 2 * j;     wx[1] = 1.f;
      Xs[2] = 2 * j + 1; wx[2] = (j == w - 1) ? 1.f : 0.5f;
      const typename ET::V8 yq = ld8<ET>(y + p * ldy + cg * 8);
      typename ET::V8 q[9];
      const typename ET::T* base = dout + b * H * W * lddo + cg * 8;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb)
          q[a * 3 + bb] = (wy[a] != 0.f && wx[bb] != 0.f) ? ld8<ET>(base + ((int64_t)Ys[a] * W + Xs[bb]) * lddo) : ET::zero8();
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (wy[a] == 0.f) continue;
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) {
          if (wx[bb] == 0.f) continue;
          float d[8];
          unpack8<ET>(q[a * 3 + bb], d);
          const float wgt = wy[a] * wx[bb];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] += wgt * d[e];
        }
      }
      const typename ET::V8 dq = pack8<ET>(o);
      st8<ET>(dprev + p * lddp + cg * 8, dq);
      take(dq, yq);
    }
    col_reduce_tail<2>(acc, c, tpp, rows, partial + (int64_t)blockIdx.x * 2 * c);
    return;
  }
  // 4 pixels (8 x 16-byte loads) in flight per thread
  int64_t p = p0 + r;
  for (; p + 3 * (int64_t)rows < p1; p += 4 * (int64_t)rows) {
    typename ET::V8 dq[4], yq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dq[u] = ld8<ET>(dout + (p + u * (int64_t)rows) * lddo + cg * 8);
      yq[u] = ld8<ET>(y + (p + u * (int64_t)rows) * ldy + cg * 8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) take(dq[u], yq[u]);
  }
  for (; p < p1; p += rows) take(ld8<ET>(dout + p * lddo + cg * 8), ld8<ET>(y + p * ldy + cg * 8));
  col_reduce_tail<2>(acc, c, tpp, rows, partial + (int64_t)blockIdx.x * 2 * c);
}

extern "C" int imm_bn_bwd_reduce(const void* dout, int lddo, const void* y, int ldy, int dtype, int64_t npix, int c,
                                 const float* scale, const float* shift, const float* mean, const float* rstd,
                                 int relu, float* partial, void* stream) {
  IMM_REQUIRE(dout && y && scale && shift && mean && rstd && partial && npix > 0, "bn_bwd_reduce: null");
  EW_REQUIRE_VEC(c, lddo, "bn_bwd_reduce(dout)");
  EW_REQUIRE_VEC(c, ldy, "bn_bwd_reduce(y)");
  const int nblk = imm_bn_bwd_blocks(npix, c);
  if (nblk < 0) return imm_fail(IMM_E_UNSUPPORTED, "bn_bwd_reduce: C=%d unsupported (C/8 must divide 256)", c);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<ET>), dim3(nblk), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)dout, lddo, (const typename ET::T*)y,
                                               ldy, npix, c, scale, shift, mean, rstd, relu, partial));
  IMM_CHECK_LAUNCH("imm_bn_bwd_reduce");
  return 0;
}

extern "C" int imm_bn_bwd_reduce_up(const void* dy_up, int lddy, void* dprev, int lddp, const void* y, int ldy, int dtype, int batch,
                                    int h, int w, int c, const float* scale, const float* shift, const float* mean,
                                    const float* rstd, int relu, float* partial, void* stream) {
  IMM_REQUIRE(dy_up && dprev && y && scale && shift && mean && rstd && partial && batch > 0 && h > 0 && w > 0, "bn_bwd_reduce_up: null");
  EW_REQUIRE_VEC(c, lddy, "bn_bwd_reduce_up(dy)");
  EW_REQUIRE_VEC(c, lddp, "bn_bwd_reduce_up(dprev)");
  EW_REQUIRE_VEC(c, ldy, "bn_bwd_reduce_up(y)");
  const int64_t npix = (int64_t)batch * h * w;
  const int nblk = imm_bn_bwd_blocks(npix, c);
  if (nblk < 0) return imm_fail(IMM_E_UNSUPPORTED, "bn_bwd_reduce_up: C=%d unsupported (C/8 must divide 256)", c);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<ET, true>), dim3(nblk), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)dy_up, lddy, (const typename ET::T*)y,
                                               ldy, npix, c, scale, shift, mean, rstd, relu, partial, (typename ET::T*)dprev, lddp, h, w));
  IMM_CHECK_LAUNCH("imm_bn_bwd_reduce_up");
  return 0;
}

// from_out != 0: the partial sums come from the epilogue of the data gradient that produced dz (conv STATS|MASK, or
// imm_upsample2x_bwd_bn): row = (sum dz, sum dz*out) with out = relu(gamma*xhat + beta) the layer's stored activation.
// dz is non-zero only where out > 0, and there xhat = (out - beta)/gamma, so  sum dz*xhat = (sum dz*out - beta*sum dz)/gamma.
template <int CPB>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int c, int ldp, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ rstd, int from_out,
                                       float* dgamma, float* dbeta, float* coef) {
  const int ch = blockIdx.x * CPB + threadIdx.x;
  double s[2] = {0.0, 0.0};
  reduce_partials_32x32<2, CPB>(partial, nblk, c, ch, s, ldp);
  if (threadIdx.y != 0 || threadIdx.x >= CPB || ch >= c) return;
  if (from_out) {
    const double g = (double)gamma[ch];
    s[1] = (g != 0.0) ? (s[1] - (double)beta[ch] * s[0]) / g : 0.0;
  }
  dbeta[ch] = (float)s[0];
  dgamma[ch] = (float)s[1];
  coef[ch] = gamma[ch] * rstd[ch];
  coef[c + ch] = (float)(s[0] / count);
  coef[2 * c + ch] = (float)(s[1] / count);
}

extern "C" int imm_bn_bwd_finalize(const float* partial, int nblk, int c, int ldp, int64_t count, const float* gamma,
                                   const float* beta, const float* rstd, int from_out, float* dgamma, float* dbeta,
                                   float* coef, void* stream) {
  IMM_REQUIRE(partial && gamma && beta && rstd && dgamma && dbeta && coef && nblk > 0 && c > 0 && count > 0, "bn_bwd_finalize: args");
  IMM_REQUIRE(ldp >= c && (ldp % 4 == 0 || ldp == c), "bn_bwd_finalize: ldp=%d (c=%d)", ldp, c);
  if (nblk >= 1024 && c % 8 == 0)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<8>, dim3(c / 8), dim3(32, 32), 0, (hipStream_t)stream, partial, nblk, c, ldp,
                       (double)count, gamma, beta, rstd, from_out, dgamma, dbeta, coef);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<32>, dim3((c + 31) / 32), dim3(32, 32), 0, (hipStream_t)stream, partial, nblk, c, ldp,
                       (double)count, gamma, beta, rstd, from_out, dgamma, dbeta, coef);
  IMM_CHECK_LAUNCH("imm_bn_bwd_finalize");
  return 0;
}

// Channel-stationary mapping (same as the reduce pass): a thread owns ONE 8-channel group, keeps its 7x8
// per-channel constants in registers and streams pixels -> 2 loads + 1 store + ~50 VALU per 16 bytes.
template <typename ET>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_kernel(
    const typename ET::T* __restrict__ dout, int lddo, const typename ET::T* __restrict__ y, int ldy, int64_t npix, int c8n, int c,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int relu, const float* __restrict__ coef, typename ET::T* __restrict__ dy, int lddy) {
  const int tpp = c8n, rows = EW_THREADS / tpp;
  const int r = threadIdx.x / tpp, cg = threadIdx.x - r * tpp;
  float sc[8], sh[8], mu[8], rs[8], k0[8], k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = cg * 8 + i;
    sc[i] = scale[ch]; sh[i] = shift[ch]; mu[i] = mean[ch]; rs[i] = rstd[ch];
    k0[i] = coef[ch]; k1[i] = coef[c + ch]; k2[i] = coef[2 * c + ch];
  }
  for (int64_t p = (int64_t)blockIdx.x * rows + r; p < npix; p += (int64_t)gridDim.x * rows) {
    float d[8], v[8], o[8];
    unpack8<ET>(ld8<ET>(dout + p * lddo + cg * 8), d);
    unpack8<ET>(ld8<ET>(y + p * ldy + cg * 8), v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float dz = d[i];
      if (relu && !(v[i] * sc[i] + sh[i] > 0.f)) dz = 0.f;
      const float xhat = (v[i] - mu[i]) * rs[i];
      o[i] = k0[i] * (dz - k1[i] - xhat * k2[i]);
    }
    st8<ET>(dy + p * lddy + cg * 8, pack8<ET>(o));
  }
}

extern "C" int imm_bn_bwd_apply(const void* dout, int lddo, const void* y, int ldy, int dtype, int64_t npix, int c,
                                const float* scale, const float* shift, const float* mean, const float* rstd, int relu,
                                const float* coef, void* dy_out, int lddy, void* stream) {
  IMM_REQUIRE(dout && y && scale && shift && mean && rstd && coef && dy_out && npix > 0, "bn_bwd_apply: null");
  EW_REQUIRE_VEC(c, lddo, "bn_bwd_apply(dout)");
  EW_REQUIRE_VEC(c, ldy, "bn_bwd_apply(y)");
  EW_REQUIRE_VEC(c, lddy, "bn_bwd_apply(dy)");
  if (imm_bn_bwd_blocks(npix, c) < 0) return imm_fail(IMM_E_UNSUPPORTED, "bn_bwd_apply: C=%d unsupported (C/8 must divide 256)", c);
  const int rows_per_blk = EW_THREADS / (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<ET>), dim3(ew_blocks((npix + rows_per_blk - 1) / rows_per_blk * EW_THREADS / 4, 4096)),
                                               dim3(EW_THREADS), 0, (hipStream_t)stream, (const typename ET::T*)dout, lddo,
                                               (const typename ET::T*)y, ldy, npix, c / 8, c, scale, shift, mean, rstd, relu,
                                               coef, (typename ET::T*)dy_out, lddy));
  IMM_CHECK_LAUNCH("imm_bn_bwd_apply");
  return 0;
}

// finalize + apply of the backward pass in one launch (see bn_apply_fused_kernel): every workgroup re-reduces the
// (sum dz, sum dz*xhat) rows of its 32-channel slice, workgroup (0, slice) writes dgamma / dbeta.
template <typename ET>
__global__ __launch_bounds__(EW_THREADS) void bn_bwd_apply_fused_kernel(
    const float* __restrict__ partial, int nblk, int c, double count, const float* __restrict__ gamma,
    const typename ET::T* __restrict__ dout, int lddo, const typename ET::T* __restrict__ y, int ldy, int64_t npix,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ mean,
    const float* __restrict__ rstd, int relu, float* dgamma, float* dbeta, typename ET::T* __restrict__ dy, int lddy, int px_per_blk) {
  __shared__ double sums[64];
  const int tid = threadIdx.x, ch0 = blockIdx.y * 32;
  const int q = tid & 3;
  const int64_t p0 = (int64_t)blockIdx.x * px_per_blk;
  const int64_t p1 = p0 + px_per_blk < npix ? p0 + px_per_blk : npix;
  // this thread's first two pixels and its per-channel constants are requested before the finalize prologue (whose L2 round
  // trips hide their latency)
  const int64_t pf = p0 + (tid >> 2);
  constexpr int PSTEP = EW_THREADS / 4;
  typename ET::V8 d0 = ET::zero8(), y0 = d0, d1 = d0, y1 = d0;
  if (pf < p1) { d0 = ld8<ET>(dout + pf * lddo + ch0 + q * 8); y0 = ld8<ET>(y + pf * ldy + ch0 + q * 8); }
  if (pf + PSTEP < p1) { d1 = ld8<ET>(dout + (pf + PSTEP) * lddo + ch0 + q * 8); y1 = ld8<ET>(y + (pf + PSTEP) * ldy + ch0 + q * 8); }
  float sc[8], sh[8], mu[8], rs[8], k0[8], k1[8], k2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int ch = ch0 + q * 8 + i;
    sc[i] = scale[ch]; sh[i] = shift[ch]; mu[i] = mean[ch]; rs[i] = rstd[ch];
    k0[i] = gamma[ch] * rs[i];
  }
  slice32_reduce<2>(partial, nblk, c, ch0, sums);
  if (blockIdx.x == 0 && tid < 32) { dbeta[ch0 + tid] = (float)sums[tid]; dgamma[ch0 + tid] = (float)sums[32 + tid]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int l = q * 8 + i;
    k1[i] = (float)(sums[l] / count); k2[i] = (float)(sums[32 + l] / count);
  }
  auto one = [&](const typename ET::V8& dq, const typename ET::V8& yq, int64_t p) {
    float d[8], v[8], o[8];
    unpack8<ET>(dq, d);
    unpack8<ET>(yq, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float dz = d[i];
      if (relu && !(v[i] * sc[i] + sh[i] > 0.f)) dz = 0.f;
      const float xhat = (v[i] - mu[i]) * rs[i];
      o[i] = k0[i] * (dz - k1[i] - xhat * k2[i]);
    }
    st8<ET>(dy + p * lddy + ch0 + q * 8, pack8<ET>(o));
  };
  // two pixels per trip, the next trip's loads issued before this trip's stores
  for (int64_t p = pf; p < p1; p += 2 * PSTEP) {
    const bool two = p + PSTEP < p1;
    typename ET::V8 nd0 = ET::zero8(), ny0 = nd0, nd1 = nd0, ny1 = nd0;
    if (p + 2 * PSTEP < p1) { nd0 = ld8<ET>(dout + (p + 2 * PSTEP) * lddo + ch0 + q * 8); ny0 = ld8<ET>(y + (p + 2 * PSTEP) * ldy + ch0 + q * 8); }
    if (p + 3 * PSTEP < p1) { nd1 = ld8<ET>(dout + (p + 3 * PSTEP) * lddo + ch0 + q * 8); ny1 = ld8<ET>(y + (p + 3 * PSTEP) * ldy + ch0 + q * 8); }
    one(d0, y0, p);
    if (two) one(d1, y1, p + PSTEP);
    d0 = nd0; y0 = ny0; d1 = nd1; y1 = ny1;
  }
}

extern "C" int imm_bn_bwd_apply_fused(const float* partial, int nblk, int c, int64_t count, const float* gamma, const void* dout,
                                      int lddo, const void* y, int ldy, int dtype, const float* scale, const float* shift,
                                      const float* mean, const float* rstd, int relu, float* dgamma, float* dbeta, void* dy_out,
                                      int lddy, void* stream) {
  IMM_REQUIRE(partial && gamma && dout && y && scale && shift && mean && rstd && dgamma && dbeta && dy_out && nblk > 0 && count > 0,
              "bn_bwd_apply_fused: args");
  IMM_REQUIRE(c > 0 && c % 32 == 0, "bn_bwd_apply_fused: C=%d must be a multiple of 32", c);
  EW_REQUIRE_VEC(c, lddo, "bn_bwd_apply_fused(dout)");
  EW_REQUIRE_VEC(c, ldy, "bn_bwd_apply_fused(y)");
  EW_REQUIRE_VEC(c, lddy, "bn_bwd_apply_fused(dy)");
  const int ppb = fused_px_per_blk(count, c);
  const dim3 grid((unsigned)((count + ppb - 1) / ppb), (unsigned)(c / 32));
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((bn_bwd_apply_fused_kernel<ET>), grid, dim3(EW_THREADS), 0, (hipStream_t)stream, partial,
                                               nblk, c, (double)count, gamma, (const typename ET::T*)dout, lddo, (const typename ET::T*)y, ldy,
                                               count, scale, shift, mean, rstd, relu, dgamma, dbeta, (typename ET::T*)dy_out, lddy, ppb));
  IMM_CHECK_LAUNCH("imm_bn_bwd_apply_fused");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// bias gradient: out[n] = sum_p dy[p][n]   (convs without batch norm)
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ __launch_bounds__(EW_THREADS) void colsum_kernel(const typename ET::T* __restrict__ dy, int ld, int64_t npix,
                                                            int c, float* __restrict__ partial) {
  const int tpp = c / 8, rows = EW_THREADS / tpp;
  const int r = threadIdx.x / tpp, cg = threadIdx.x - r * tpp;
  const int64_t per_blk = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per_blk;
  const int64_t p1 = p0 + per_blk < npix ? p0 + per_blk : npix;
  float acc[1][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[0][i] = 0.f;
  for (int64_t p = p0 + r; p < p1; p += rows) {
    float d[8];
    unpack8<ET>(ld8<ET>(dy + p * ld + cg * 8), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[0][i] += d[i];
  }
  col_reduce_tail<1>(acc, c, tpp, rows, partial + (int64_t)blockIdx.x * c);
}

__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int c, int c_real, float* out) {
  const int ch = blockIdx.x * 32 + threadIdx.x;
  double s[1] = {0.0};
  reduce_partials_32x32<1>(partial, nblk, c, ch, s);
  if (threadIdx.y != 0 || ch >= c_real) return;
  out[ch] = (float)s[0];
}

// c = padded channel count of the buffer rows that are summed (multiple of 8); the first c_out sums are written.
extern "C" int imm_colsum(const void* dy, int dtype, int64_t npix, int c, int c_out, int ld, float* partial, float* out,
                          void* stream) {
  IMM_REQUIRE(dy && partial && out && npix > 0 && c_out > 0 && c_out <= c, "colsum: args");
  EW_REQUIRE_VEC(c, ld, "colsum");
  const int nblk = imm_colsum_blocks(npix, c);
  if (nblk < 0) return imm_fail(IMM_E_UNSUPPORTED, "colsum: C=%d unsupported", c);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((colsum_kernel<ET>), dim3(nblk), dim3(EW_THREADS), 0, (hipStream_t)stream,
                                               (const typename ET::T*)dy, ld, npix, c, partial));
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((c + 31) / 32), dim3(32, 32), 0, (hipStream_t)stream, partial, nblk, c, c_out, out);
  IMM_CHECK_LAUNCH("imm_colsum");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// x2 bilinear upsample, TF legacy mapping (src = dst/2): out[2i] = in[i], out[2i+1] = (in[i]+in[min(i+1,n-1)])/2
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ void upsample2x_fwd_kernel(const typename ET::T* __restrict__ x, typename ET::T* __restrict__ y, int batch, int h, int w,
                                      int c8n, int ldx, int ldy) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)batch * H * W * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int X = (int)(t % W); t /= W;
    const int Y = (int)(t % H);
    const int b = (int)(t / H);
    const int y0 = Y >> 1, x0 = X >> 1;
    const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
    const float ly = (Y & 1) ? 0.5f : 0.f, lx = (X & 1) ? 0.5f : 0.f;
    const typename ET::T* base = x + (int64_t)b * h * w * ldx + cg * 8;
    float tl[8], tr[8], bl[8], br[8], o[8];
    unpack8<ET>(ld8<ET>(base + ((int64_t)y0 * w + x0) * ldx), tl);
    unpack8<ET>(ld8<ET>(base + ((int64_t)y0 * w + x1) * ldx), tr);
    unpack8<ET>(ld8<ET>(base + ((int64_t)y1 * w + x0) * ldx), bl);
    unpack8<ET>(ld8<ET>(base + ((int64_t)y1 * w + x1) * ldx), br);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float top = tl[i] + (tr[i] - tl[i]) * lx;
      const float bot = bl[i] + (br[i] - bl[i]) * lx;
      o[i] = top + (bot - top) * ly;
    }
    st8<ET>(y + (((int64_t)b * H + Y) * W + X) * ldy + cg * 8, pack8<ET>(o));
  }
}

// adjoint as a gather: dx[i][j] = sum_{Y,X} wy(i,Y)*wx(j,X)*dy[Y][X]
template <typename ET>
__global__ void upsample2x_bwd_kernel(const typename ET::T* __restrict__ dy, typename ET::T* __restrict__ dx, int batch, int h,
                                      int w, int c8n, int lddy, int lddx) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)batch * h * w * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int j = (int)(t % w); t /= w;
    const int i = (int)(t % h);
    const int b = (int)(t / h);
    // contributing output rows/cols and weights
    int Ys[3], Xs[3]; float wy[3], wx[3];
    Ys[0] = 2 * i - 1; wy[0] = (i >= 1) ? 0.5f : 0.f;
    Ys[1] = 2 * i;     wy[1] = 1.f;
    Ys[2] = 2 * i + 1; wy[2] = (i == h - 1) ? 1.f : 0.5f;
    Xs[0] = 2 * j - 1; wx[0] = (j >= 1) ? 0.5f : 0.f;
    Xs[1] = 2 * j;     wx[1] = 1.f;
    Xs[2] = 2 * j + 1; wx[2] = (j == w - 1) ? 1.f : 0.5f;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const typename ET::T* base = dy + (int64_t)b * H * W * lddy + cg * 8;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        if (wx[bb] == 0.f) continue;
        float d[8];
        unpack8<ET>(ld8<ET>(base + ((int64_t)Ys[a] * W + Xs[bb]) * lddy), d);
        const float wgt = wy[a] * wx[bb];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += wgt * d[e];
      }
    }
    st8<ET>(dx + (((int64_t)b * h + i) * w + j) * lddx + cg * 8, pack8<ET>(o));
  }
}

extern "C" int imm_upsample2x_fwd(const void* x, void* y, int dtype, int batch, int h, int w, int c, int ldx, int ldy,
                                  void* stream) {
  IMM_REQUIRE(x && y && batch > 0 && h > 0 && w > 0, "upsample2x_fwd: args");
  EW_REQUIRE_VEC(c, ldx, "upsample2x_fwd(x)");
  EW_REQUIRE_VEC(c, ldy, "upsample2x_fwd(y)");
  const int64_t total = (int64_t)batch * 4 * h * w * (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((upsample2x_fwd_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)x, (typename ET::T*)y, batch, h, w, c / 8,
                                               ldx, ldy));
  IMM_CHECK_LAUNCH("imm_upsample2x_fwd");
  return 0;
}

extern "C" int imm_upsample2x_bwd(const void* dy, void* dx, int dtype, int batch, int h, int w, int c, int lddy,
                                  int lddx, void* stream) {
  IMM_REQUIRE(dy && dx && batch > 0 && h > 0 && w > 0, "upsample2x_bwd: args");
  EW_REQUIRE_VEC(c, lddy, "upsample2x_bwd(dy)");
  EW_REQUIRE_VEC(c, lddx, "upsample2x_bwd(dx)");
  const int64_t total = (int64_t)batch * h * w * (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((upsample2x_bwd_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)dy, (typename ET::T*)dx, batch, h, w,
                                               c / 8, lddy, lddx));
  IMM_CHECK_LAUNCH("imm_upsample2x_bwd");
  return 0;
}

// The same adjoint fused with what follows it when dx is the output gradient of a conv+BN+ReLU block (renderer conv_2/4/6,
// imm_model.py:166-175): dz = dx * [out > 0] is what gets stored, and every workgroup writes the partial sums
// (sum dz, sum dz*out) per channel — the pass imm_bn_bwd_reduce would otherwise make over dx and the conv output.
template <typename ET>
__global__ __launch_bounds__(EW_THREADS) void upsample2x_bwd_bn_kernel(
    const typename ET::T* __restrict__ dy, typename ET::T* __restrict__ dx, int batch, int h, int w, int c8n, int lddy, int lddx,
    const typename ET::T* __restrict__ out, int ldo, float* __restrict__ partial) {
  const int H = 2 * h, W = 2 * w;
  const int64_t total = (int64_t)batch * h * w * c8n;
  float acc[2][8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { acc[0][e] = 0.f; acc[1][e] = 0.f; }
  // gridDim.x * blockDim.x is a multiple of c8n (c8n | 256): a thread keeps its channel group over the grid-stride loop
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int j = (int)(t % w); t /= w;
    const int i = (int)(t % h);
    const int b = (int)(t / h);
    int Ys[3], Xs[3]; float wy[3], wx[3];
    Ys[0] = 2 * i - 1; wy[0] = (i >= 1) ? 0.5f : 0.f;
    Ys[1] = 2 * i;     wy[1] = 1.f;
    Ys[2] = 2 * i + 1; wy[2] = (i == h - 1) ? 1.f : 0.5f;
    Xs[0] = 2 * j - 1; wx[0] = (j >= 1) ? 0.5f : 0.f;
    Xs[1] = 2 * j;     wx[1] = 1.f;
    Xs[2] = 2 * j + 1; wx[2] = (j == w - 1) ? 1.f : 0.5f;
    float o[8], m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const typename ET::T* base = dy + (int64_t)b * H * W * lddy + cg * 8;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (wy[a] == 0.f) continue;
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        if (wx[bb] == 0.f) continue;
        float d[8];
        unpack8<ET>(ld8<ET>(base + ((int64_t)Ys[a] * W + Xs[bb]) * lddy), d);
        const float wgt = wy[a] * wx[bb];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += wgt * d[e];
      }
    }
    const int64_t p = ((int64_t)b * h + i) * w + j;
    unpack8<ET>(ld8<ET>(out + p * ldo + cg * 8), m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (!(m[e] > 0.f)) o[e] = 0.f;
      acc[0][e] += o[e];
      acc[1][e] += o[e] * m[e];
    }
    st8<ET>(dx + p * lddx + cg * 8, pack8<ET>(o));
  }
  col_reduce_tail<2>(acc, c8n * 8, c8n, EW_THREADS / c8n, partial + (int64_t)blockIdx.x * 2 * c8n * 8);
}

extern "C" int imm_upsample2x_bwd_bn_blocks(int batch, int h, int w, int c) {
  if (c <= 0 || c % 8 || c / 8 > EW_THREADS || EW_THREADS % (c / 8) || batch <= 0 || h <= 0 || w <= 0) return IMM_E_UNSUPPORTED;
  const int64_t total = (int64_t)batch * h * w * (c / 8);
  int64_t b = (total + EW_THREADS * 4 - 1) / (EW_THREADS * 4);      // >= 4 pixels-groups per thread
  if (b < 1) b = 1;
  if (b > 512) b = 512;
  return (int)b;
}

extern "C" int imm_upsample2x_bwd_bn(const void* dy, void* dx, int dtype, int batch, int h, int w, int c, int lddy, int lddx,
                                     const void* out, int ldo, float* partial, void* stream) {
  IMM_REQUIRE(dy && dx && out && partial && batch > 0 && h > 0 && w > 0, "upsample2x_bwd_bn: args");
  EW_REQUIRE_VEC(c, lddy, "upsample2x_bwd_bn(dy)");
  EW_REQUIRE_VEC(c, lddx, "upsample2x_bwd_bn(dx)");
  EW_REQUIRE_VEC(c, ldo, "upsample2x_bwd_bn(out)");
  const int nblk = imm_upsample2x_bwd_bn_blocks(batch, h, w, c);
  if (nblk < 0) return imm_fail(IMM_E_UNSUPPORTED, "upsample2x_bwd_bn: C=%d unsupported (C/8 must divide 256)", c);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((upsample2x_bwd_bn_kernel<ET>), dim3(nblk), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)dy, (typename ET::T*)dx, batch, h, w,
                                               c / 8, lddy, lddx, (const typename ET::T*)out, ldo, partial));
  IMM_CHECK_LAUNCH("imm_upsample2x_bwd_bn");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// resize_bilinear(align_corners=True): src = dst*(in-1)/(out-1)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ac_coord(int o, int n_in, int n_out, int& lo, int& hi, float& lerp) {
  const float scale = (n_out > 1) ? (float)(n_in - 1) / (float)(n_out - 1) : 0.f;
  const float src = o * scale;
  lo = (int)floorf(src);
  hi = min(lo + 1, n_in - 1);
  lerp = src - (float)lo;
}

template <typename ET>
__global__ void resize_ac_fwd_kernel(const typename ET::T* __restrict__ x, typename ET::T* __restrict__ y, int batch, int hi, int wi,
                                     int ho, int wo, int c8n, int ldx, int ldy) {
  const int64_t total = (int64_t)batch * ho * wo * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int X = (int)(t % wo); t /= wo;
    const int Y = (int)(t % ho);
    const int b = (int)(t / ho);
    int y0, y1, x0, x1; float ly, lx;
    ac_coord(Y, hi, ho, y0, y1, ly);
    ac_coord(X, wi, wo, x0, x1, lx);
    const typename ET::T* base = x + (int64_t)b * hi * wi * ldx + cg * 8;
    float tl[8], tr[8], bl[8], br[8], o[8];
    unpack8<ET>(ld8<ET>(base + ((int64_t)y0 * wi + x0) * ldx), tl);
    unpack8<ET>(ld8<ET>(base + ((int64_t)y0 * wi + x1) * ldx), tr);
    unpack8<ET>(ld8<ET>(base + ((int64_t)y1 * wi + x0) * ldx), bl);
    unpack8<ET>(ld8<ET>(base + ((int64_t)y1 * wi + x1) * ldx), br);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float top = tl[i] + (tr[i] - tl[i]) * lx;
      const float bot = bl[i] + (br[i] - bl[i]) * lx;
      o[i] = top + (bot - top) * ly;
    }
    st8<ET>(y + (((int64_t)b * ho + Y) * wo + X) * ldy + cg * 8, pack8<ET>(o));
  }
}

// adjoint by gather: each input pixel scans the (few) output rows/cols whose stencil touches it
template <typename ET>
__global__ void resize_ac_bwd_kernel(const typename ET::T* __restrict__ dy, typename ET::T* __restrict__ dx, int batch, int hi,
                                     int wi, int ho, int wo, int c8n, int lddy, int lddx) {
  const int64_t total = (int64_t)batch * hi * wi * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int j = (int)(t % wi); t /= wi;
    const int i = (int)(t % hi);
    const int b = (int)(t / hi);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    const typename ET::T* base = dy + (int64_t)b * ho * wo * lddy + cg * 8;
    for (int Y = 0; Y < ho; ++Y) {
      int y0, y1; float ly;
      ac_coord(Y, hi, ho, y0, y1, ly);
      float wy = 0.f;
      if (y0 == i) wy += 1.f - ly;
      if (y1 == i) wy += ly;
      if (wy == 0.f) continue;
      for (int X = 0; X < wo; ++X) {
        int x0, x1; float lx;
        ac_coord(X, wi, wo, x0, x1, lx);
        float wx = 0.f;
        if (x0 == j) wx += 1.f - lx;
        if (x1 == j) wx += lx;
        if (wx == 0.f) continue;
        float d[8];
        unpack8<ET>(ld8<ET>(base + ((int64_t)Y * wo + X) * lddy), d);
        const float wgt = wy * wx;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += wgt * d[e];
      }
    }
    st8<ET>(dx + (((int64_t)b * hi + i) * wi + j) * lddx + cg * 8, pack8<ET>(o));
  }
}

extern "C" int imm_resize_ac_fwd(const void* x, void* y, int dtype, int batch, int hi, int wi, int ho, int wo, int c,
                                 int ldx, int ldy, void* stream) {
  IMM_REQUIRE(x && y && batch > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, "resize_ac_fwd: args");
  EW_REQUIRE_VEC(c, ldx, "resize_ac_fwd(x)");
  EW_REQUIRE_VEC(c, ldy, "resize_ac_fwd(y)");
  const int64_t total = (int64_t)batch * ho * wo * (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((resize_ac_fwd_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)x, (typename ET::T*)y, batch, hi, wi, ho,
                                               wo, c / 8, ldx, ldy));
  IMM_CHECK_LAUNCH("imm_resize_ac_fwd");
  return 0;
}

extern "C" int imm_resize_ac_bwd(const void* dy, void* dx, int dtype, int batch, int hi, int wi, int ho, int wo, int c,
                                 int lddy, int lddx, void* stream) {
  IMM_REQUIRE(dy && dx && batch > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, "resize_ac_bwd: args");
  EW_REQUIRE_VEC(c, lddy, "resize_ac_bwd(dy)");
  EW_REQUIRE_VEC(c, lddx, "resize_ac_bwd(dx)");
  const int64_t total = (int64_t)batch * hi * wi * (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((resize_ac_bwd_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)dy, (typename ET::T*)dx, batch, hi, wi, ho,
                                               wo, c / 8, lddy, lddx));
  IMM_CHECK_LAUNCH("imm_resize_ac_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// 2x2/2 max pool (dense NHWC, ld == c)
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ void maxpool2_fwd_kernel(const typename ET::T* __restrict__ x, typename ET::T* __restrict__ y, int batch, int h, int w,
                                    int c8n) {
  const int ho = h / 2, wo = w / 2, c = c8n * 8;
  const int64_t total = (int64_t)batch * ho * wo * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int X = (int)(t % wo); t /= wo;
    const int Y = (int)(t % ho);
    const int b = (int)(t / ho);
    const typename ET::T* base = x + (((int64_t)b * h + 2 * Y) * w + 2 * X) * c + cg * 8;
    float a0[8], a1[8], a2[8], a3[8], o[8];
    unpack8<ET>(ld8<ET>(base), a0);
    unpack8<ET>(ld8<ET>(base + c), a1);
    unpack8<ET>(ld8<ET>(base + (int64_t)w * c), a2);
    unpack8<ET>(ld8<ET>(base + (int64_t)w * c + c), a3);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = fmaxf(fmaxf(a0[i], a1[i]), fmaxf(a2[i], a3[i]));
    st8<ET>(y + (((int64_t)b * ho + Y) * wo + X) * c + cg * 8, pack8<ET>(o));
  }
}

template <typename ET>
__global__ void maxpool2_bwd_kernel(const typename ET::T* __restrict__ x, const typename ET::T* __restrict__ dy,
                                    typename ET::T* __restrict__ dx, int batch, int h, int w, int c8n, int relu_mask) {
  const int ho = h / 2, wo = w / 2, c = c8n * 8;
  const int64_t total = (int64_t)batch * ho * wo * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int X = (int)(t % wo); t /= wo;
    const int Y = (int)(t % ho);
    const int b = (int)(t / ho);
    const int64_t off = (((int64_t)b * h + 2 * Y) * w + 2 * X) * c + cg * 8;
    float a[4][8], g[8], o[4][8];
    unpack8<ET>(ld8<ET>(x + off), a[0]);
    unpack8<ET>(ld8<ET>(x + off + c), a[1]);
    unpack8<ET>(ld8<ET>(x + off + (int64_t)w * c), a[2]);
    unpack8<ET>(ld8<ET>(x + off + (int64_t)w * c + c), a[3]);
    unpack8<ET>(ld8<ET>(dy + (((int64_t)b * ho + Y) * wo + X) * c + cg * 8), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int am = 0; float mv = a[0][i];
#pragma unroll
      for (int q = 1; q < 4; ++q) if (a[q][i] > mv) { mv = a[q][i]; am = q; }
      const float gv = (relu_mask && !(mv > 0.f)) ? 0.f : g[i];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q][i] = (q == am) ? gv : 0.f;
    }
    st8<ET>(dx + off, pack8<ET>(o[0]));
    st8<ET>(dx + off + c, pack8<ET>(o[1]));
    st8<ET>(dx + off + (int64_t)w * c, pack8<ET>(o[2]));
    st8<ET>(dx + off + (int64_t)w * c + c, pack8<ET>(o[3]));
  }
}

extern "C" int imm_maxpool2_fwd(const void* x, void* y, int dtype, int batch, int h, int w, int c, void* stream) {
  IMM_REQUIRE(x && y && batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "maxpool2_fwd: even sides required");
  EW_REQUIRE_VEC(c, c, "maxpool2_fwd");
  const int64_t total = (int64_t)batch * (h / 2) * (w / 2) * (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((maxpool2_fwd_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)x, (typename ET::T*)y, batch, h, w, c / 8));
  IMM_CHECK_LAUNCH("imm_maxpool2_fwd");
  return 0;
}

extern "C" int imm_maxpool2_bwd(const void* x, const void* dy, void* dx, int dtype, int batch, int h, int w, int c,
                                int relu_mask, void* stream) {
  IMM_REQUIRE(x && dy && dx && batch > 0 && h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "maxpool2_bwd: even sides required");
  EW_REQUIRE_VEC(c, c, "maxpool2_bwd");
  const int64_t total = (int64_t)batch * (h / 2) * (w / 2) * (c / 8);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((maxpool2_bwd_kernel<ET>), dim3(ew_blocks(total)), dim3(EW_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)x, (const typename ET::T*)dy,
                                               (typename ET::T*)dx, batch, h, w, c / 8, relu_mask));
  IMM_CHECK_LAUNCH("imm_maxpool2_bwd");
  return 0;
}
