// image_io.hip — ingest of decoded images on gfx950: the resize / central-crop / to-float stage of the reference's
// input pipelines (imm/datasets/celeba_dataset.py:136-174: tf.to_float -> tf.image.resize_images(BILINEAR,
// align_corners=True) to round(size/0.8) -> central crop; imm/datasets/aflw_dataset.py:81-114: resize straight to size).
// The reference runs this per image on CPU threads of tf.data; here the JPEG decoder's u8 HWC outputs (different sizes
// per image) are packed back to back into one staging buffer, copied to HBM once, and ONE launch produces the f32 NHWC
// batch — only the pixels that survive the crop are ever computed, and they can be written at a channel offset of a wider
// buffer (the mask||image stack the thin-plate-spline warp reads, imm/datasets/tps_dataset.py:79).
//
//   scale = (in - 1) / (resize - 1)      (TF1 align_corners; in / resize when resize == 1)
//   s = (crop0 + o) * scale;  lo = floor(s);  hi = min(ceil(s), in - 1);  t = s - lo
//   out = top + (bottom - top) * ty,  top = tl + (tr - tl) * tx,  bottom = bl + (br - bl) * tx     (float32, unfused)
// HBM-bound and tiny: B * (in_h*in_w*c bytes read at most) + B*oh*ow*c*4 bytes written.
#include "common.h"

__global__ __launch_bounds__(256) void resize_crop_u8_kernel(const uint8_t* __restrict__ src, const int64_t* __restrict__ offs,
                                                             const int32_t* __restrict__ hw, int c, int rh, int rw, int y0,
                                                             int x0, int oh, int ow, float* __restrict__ dst, int ld_dst) {
#pragma clang fp contract(off)   // a fused a*s - floor(a*s) would differ from the separately rounded host evaluation
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= oh * ow) return;
  const int b = blockIdx.y;
  const int ih = hw[2 * b], iw = hw[2 * b + 1];
  const uint8_t* s = src + offs[b];
  const int oy = p / ow, ox = p - oy * ow;
  // every operation rounded separately (no fma contraction): bit-identical to a float32 host evaluation in the same order
  // (oracle/image_oracle.py)
  // the scale is the correctly rounded float quotient of two small integers: formed in double and rounded once (the
  // operands are < 2^24, so the double quotient can never sit within a double ulp of a float rounding boundary); the
  // device's float '/' may be the approximate-reciprocal form
  const float sy = rh > 1 ? (float)((double)(ih - 1) / (double)(rh - 1)) : (float)((double)ih / (double)rh);
  const float sx = rw > 1 ? (float)((double)(iw - 1) / (double)(rw - 1)) : (float)((double)iw / (double)rw);
  const float fy = (float)(y0 + oy) * sy, fx = (float)(x0 + ox) * sx;
  const int yl = (int)floorf(fy), xl = (int)floorf(fx);
  const int yh = min((int)ceilf(fy), ih - 1), xh = min((int)ceilf(fx), iw - 1);
  const float ty = fy - (float)yl, tx = fx - (float)xl;
  const uint8_t* r0 = s + (int64_t)yl * iw * c;
  const uint8_t* r1 = s + (int64_t)yh * iw * c;
  float* d = dst + ((int64_t)b * oh * ow + p) * ld_dst;
  for (int ch = 0; ch < c; ++ch) {
    const float tl = (float)r0[xl * c + ch], tr = (float)r0[xh * c + ch];
    const float bl = (float)r1[xl * c + ch], br = (float)r1[xh * c + ch];
    const float top = tl + (tr - tl) * tx;
    const float bot = bl + (br - bl) * tx;
    d[ch] = top + (bot - top) * ty;
  }
}

extern "C" int imm_resize_crop_u8(const uint8_t* src, const int64_t* offsets, const int32_t* hw, int batch, int c, int resize_h,
                                  int resize_w, int crop_y0, int crop_x0, int out_h, int out_w, float* dst, int ld_dst,
                                  void* stream) {
  IMM_REQUIRE(src && offsets && hw && dst, "resize_crop_u8: null pointer");
  IMM_REQUIRE(batch > 0 && c >= 1 && c <= 4 && ld_dst >= c, "resize_crop_u8: batch > 0, 1 <= c <= 4, ld_dst >= c");
  IMM_REQUIRE(resize_h > 0 && resize_w > 0 && out_h > 0 && out_w > 0, "resize_crop_u8: sizes");
  IMM_REQUIRE(crop_y0 >= 0 && crop_x0 >= 0 && crop_y0 + out_h <= resize_h && crop_x0 + out_w <= resize_w,
              "resize_crop_u8: crop window [%d+%d, %d+%d] outside the resized image %dx%d", crop_y0, out_h, crop_x0, out_w,
              resize_h, resize_w);
  IMM_REQUIRE(batch <= 65535, "resize_crop_u8: batch %d > 65535", batch);
  const dim3 grid((out_h * out_w + 255) / 256, batch);
  hipLaunchKernelGGL(resize_crop_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, offsets, hw, c, resize_h, resize_w,
                     crop_y0, crop_x0, out_h, out_w, dst, ld_dst);
  IMM_CHECK_LAUNCH("imm_resize_crop_u8");
  return 0;
}
