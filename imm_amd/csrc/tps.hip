// tps.hip — thin-plate-spline image warp on gfx950: the reference's data augmentation (imm/utils/tps_sampler.py:76-99
// TPSRandomSampler.forward, :142-157 TPSGridGen.forward; used with pad=False by imm/datasets/tps_dataset.py:70-96 on
// the batched mask||image tensor).  The reference evaluates it on the CPU inside a tf.py_func with
// num_parallel_calls=1; here the sampling grid (basis @ parameters) and the bilinear gather are one kernel, and the
// results can be written straight into the training step's input buffers.
//
//   grid[b][p] = sum_j basis[p][j] * W[b][j]         (j over M control-point kernels, then 1, x, y;  (x, y) output)
//   dst[b][p]  = bilinear(src[b], grid[b][p])        (align_corners=True mapping, zero padding: F.grid_sample of the
//                                                     torch 0.4.1 the reference pins)
// One thread = one output pixel for TB = 8 samples: the basis value of (pixel, j) is loaded once (basis stored
// transposed, [j][pixel], coalesced) and feeds 16 accumulators; W is wave-uniform (scalar loads).  HBM-bound:
// algorithmic bytes = B*h*w*c*4 in + out, plus the 4*(M+3)*h*w-byte basis per 8 samples (L2-resident).
//
// pad=True (tps_sampler.py:24-29,89-92) is the same kernel with a geometry: the source is addressed as if replicate-padded
// by (pad_y, pad_x) (coordinates clamped into the image instead of a padded copy), the sampling grid has its own size
// (gh, gw) and only the window (crop_y.., crop_x..) of size (oh, ow) of the warped result is produced.
#include "common.h"

#define TPS_TB 8

struct TpsGeom {
  int pad_y, pad_x;      // replicate padding of the source (rows, columns)
  int gh, gw;            // sampling grid = size of the un-cropped warp result = what basis_t was built for
  int crop_y, crop_x;    // first grid row / column that is kept
  int oh, ow;            // output size
};

__global__ __launch_bounds__(256) void tps_warp_kernel(const float* __restrict__ src, int ld_src, int batch, int h, int w, int c, TpsGeom g,
                                                       const float* __restrict__ basis_t, int m3,
                                                       const float* __restrict__ w_tps, float* __restrict__ dst, int ld_dst,
                                                       float* __restrict__ dst_c0, float* __restrict__ dst_rest, int ld_rest) {
  extern __shared__ float2 wsh[];                    // [m3][TPS_TB]: the parameters of this block's samples (zeros beyond nb)
  const int npix = g.gh * g.gw, nout = g.oh * g.ow, nsrc = h * w;
  const int po = blockIdx.x * 256 + threadIdx.x;       // output pixel
  const int oy = po / g.ow;
  const int p = (oy + g.crop_y) * g.gw + (po - oy * g.ow) + g.crop_x;   // its grid point
  const int b0 = blockIdx.y * TPS_TB;
  const int nb = batch - b0 < TPS_TB ? batch - b0 : TPS_TB;
  for (int t = threadIdx.x; t < m3 * TPS_TB; t += 256) {
    const int j = t / TPS_TB, i = t - j * TPS_TB;
    wsh[t] = i < nb ? *(const float2*)(w_tps + ((int64_t)(b0 + i) * m3 + j) * 2) : make_float2(0.f, 0.f);
  }
  __syncthreads();
  if (po >= nout) return;
  float gx[TPS_TB], gy[TPS_TB];
#pragma unroll
  for (int i = 0; i < TPS_TB; ++i) { gx[i] = 0.f; gy[i] = 0.f; }
  // 8 basis loads in flight per thread (the loop is a chain of L2 latencies otherwise); summation order stays j = 0, 1, ...
  int j = 0;
  for (; j + 8 <= m3; j += 8) {
    float l[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l[u] = basis_t[(int64_t)(j + u) * npix + p];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < TPS_TB; ++i) {
        const float2 wv = wsh[(j + u) * TPS_TB + i];   // broadcast LDS read
        gx[i] = fmaf(l[u], wv.x, gx[i]);
        gy[i] = fmaf(l[u], wv.y, gy[i]);
      }
  }
  for (; j < m3; ++j) {
    const float l = basis_t[(int64_t)j * npix + p];
#pragma unroll
    for (int i = 0; i < TPS_TB; ++i) {
      const float2 wv = wsh[j * TPS_TB + i];
      gx[i] = fmaf(l, wv.x, gx[i]);
      gy[i] = fmaf(l, wv.y, gy[i]);
    }
  }
  const int pw = w + 2 * g.pad_x, ph = h + 2 * g.pad_y;    // the padded source the normalised coordinates refer to
  const float sx = 0.5f * (float)(pw - 1), sy = 0.5f * (float)(ph - 1);
#pragma unroll
  for (int i = 0; i < TPS_TB; ++i) {
    if (i >= nb) break;
    const int b = b0 + i;
    const float fx = (gx[i] + 1.f) * sx, fy = (gy[i] + 1.f) * sy;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float ax = fx - x0f, ay = fy - y0f;
    // far-away coordinates (|f| > 2^30) would overflow the int conversion: they are outside the image anyway
    const bool sane = fabsf(fx) < 1.0e9f && fabsf(fy) < 1.0e9f;
    const int x0 = sane ? (int)x0f : -4, y0 = sane ? (int)y0f : -4;
    float acc[8];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) acc[ch] = 0.f;
    const float* sb = src + (int64_t)b * nsrc * ld_src;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int xi = x0 + dx, yi = y0 + dy;
        const float wgt = (dy ? ay : 1.f - ay) * (dx ? ax : 1.f - ax);
        if ((unsigned)xi < (unsigned)pw && (unsigned)yi < (unsigned)ph) {
          const int xs = min(max(xi - g.pad_x, 0), w - 1), ys = min(max(yi - g.pad_y, 0), h - 1);
          const float* sp = sb + ((int64_t)ys * w + xs) * ld_src;
          if (c == 4 && (ld_src & 3) == 0) {
            const float4 v = *(const float4*)sp;
            acc[0] += wgt * v.x; acc[1] += wgt * v.y; acc[2] += wgt * v.z; acc[3] += wgt * v.w;
          } else {
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) if (ch < c) acc[ch] += wgt * sp[ch];
          }
        }
      }
    const int64_t o = (int64_t)b * nout + po;
    if (dst) {
      float* dp = dst + o * ld_dst;
      if (c == 4 && (ld_dst & 3) == 0) *(float4*)dp = make_float4(acc[0], acc[1], acc[2], acc[3]);
      else {
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) if (ch < c) dp[ch] = acc[ch];
      }
    }
    if (dst_c0) dst_c0[o] = acc[0];
    if (dst_rest) {
      float* rp = dst_rest + o * ld_rest;
#pragma unroll
      for (int ch = 1; ch < 8; ++ch) if (ch < c) rp[ch - 1] = acc[ch];
    }
  }
}

static int tps_launch(const float* src, int ld_src, int batch, int h, int w, int c, TpsGeom g, const float* basis_t, int m3,
                      const float* w_tps, float* dst, int ld_dst, float* dst_c0, float* dst_rest, int ld_rest, void* stream) {
  IMM_REQUIRE(src && basis_t && w_tps, "tps_warp: null input");
  IMM_REQUIRE(dst || dst_c0 || dst_rest, "tps_warp: no output");
  IMM_REQUIRE(batch > 0 && h > 1 && w > 1 && c >= 1 && c <= 8 && m3 >= 3, "tps_warp: dims (1 <= c <= 8, h, w >= 2)");
  IMM_REQUIRE(ld_src >= c && (!dst || ld_dst >= c) && (!dst_rest || ld_rest >= c - 1), "tps_warp: leading dimensions");
  IMM_REQUIRE(g.pad_y >= 0 && g.pad_x >= 0 && g.gh > 1 && g.gw > 1 && g.crop_y >= 0 && g.crop_x >= 0 && g.oh > 0 && g.ow > 0 &&
              g.crop_y + g.oh <= g.gh && g.crop_x + g.ow <= g.gw, "tps_warp: geometry (crop window %d+%d x %d+%d of a %d x %d grid)",
              g.crop_y, g.oh, g.crop_x, g.ow, g.gh, g.gw);
  const int64_t big = (int64_t)h * w > (int64_t)g.oh * g.ow ? (int64_t)h * w : (int64_t)g.oh * g.ow;
  IMM_REQUIRE((int64_t)batch * big * (int64_t)(ld_src > ld_dst ? ld_src : ld_dst) < (1LL << 40) && (int64_t)g.gh * g.gw < (1LL << 30), "tps_warp: size");
  const dim3 grid((g.oh * g.ow + 255) / 256, (batch + TPS_TB - 1) / TPS_TB);
  IMM_REQUIRE((size_t)m3 * TPS_TB * sizeof(float2) <= 60 * 1024, "tps_warp: too many control points (%d)", m3 - 3);
  hipLaunchKernelGGL(tps_warp_kernel, grid, dim3(256), (size_t)m3 * TPS_TB * sizeof(float2), (hipStream_t)stream, src, ld_src, batch, h, w, c, g,
                     basis_t, m3, w_tps, dst, ld_dst, dst_c0, dst_rest, ld_rest);
  IMM_CHECK_LAUNCH("imm_tps_warp");
  return 0;
}

extern "C" int imm_tps_warp(const float* src, int ld_src, int batch, int h, int w, int c, const float* basis_t, int m3,
                            const float* w_tps, float* dst, int ld_dst, float* dst_c0, float* dst_rest, int ld_rest,
                            void* stream) {
  const TpsGeom g{0, 0, h, w, 0, 0, h, w};
  return tps_launch(src, ld_src, batch, h, w, c, g, basis_t, m3, w_tps, dst, ld_dst, dst_c0, dst_rest, ld_rest, stream);
}

extern "C" int imm_tps_warp_pad(const float* src, int ld_src, int batch, int h, int w, int c, int pad_y, int pad_x, int grid_h,
                                int grid_w, int crop_y, int crop_x, int out_h, int out_w, const float* basis_t, int m3,
                                const float* w_tps, float* dst, int ld_dst, void* stream) {
  const TpsGeom g{pad_y, pad_x, grid_h, grid_w, crop_y, crop_x, out_h, out_w};
  return tps_launch(src, ld_src, batch, h, w, c, g, basis_t, m3, w_tps, dst, ld_dst, nullptr, nullptr, 0, stream);
}
