// vgg_first.hip — the C_in = 1 head of the frozen perceptual VGG16 on gfx950, forward and backward.
//
// Reference: imm/models/selfsup/build_vgg16.py:22-26 (mean over RGB, /255, -114.451/255) feeding
// imm/models/selfsup/vgg16.py:345 conv1_1 (3x3 SAME, 1->64, +bias, ReLU); input batch is
// concat([gt, pred], 0) (imm/models/imm_model.py:126).  K = 9 is far too short for the matrix
// cores (0.07 % of the step's FLOPs): this is a direct VALU kernel bound by its 128 B/pixel write.
// Backward returns d loss / d pred (3 channels) = conv1_1^T(dz)/(3*255) plus the 'input' feature term
// of the perceptual loss (imm_model.py:142-147, feature name 'input'), written as the 16-bit
// gradient of the renderer's last convolution (pixel stride lddp, channels >= 3 zero: S10).
#include "common.h"

#define VF_TILE 16

template <typename ET>
__global__ __launch_bounds__(256) void vgg_conv1_1_fwd_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                              int ldp, int batch, int s, const float* __restrict__ w,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ out) {
  __shared__ float gray[(VF_TILE + 2) * (VF_TILE + 2)];
  __shared__ float sw[9 * 64 + 64];
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  const int ty0 = blockIdx.y * VF_TILE, tx0 = blockIdx.x * VF_TILE;
  const float* src = img < batch ? gt + (int64_t)img * s * s * 3 : pred + (int64_t)(img - batch) * s * s * ldp;
  const int ld = img < batch ? 3 : ldp;
  for (int i = tid; i < 9 * 64 + 64; i += 256) sw[i] = i < 576 ? w[i] : bias[i - 576];
  for (int i = tid; i < (VF_TILE + 2) * (VF_TILE + 2); i += 256) {
    const int yy = ty0 + i / (VF_TILE + 2) - 1, xx = tx0 + i % (VF_TILE + 2) - 1;
    float g = 0.f;   // SAME zero padding applies to the NORMALISED gray image
    if (yy >= 0 && yy < s && xx >= 0 && xx < s) {
      const float* p = src + ((int64_t)yy * s + xx) * ld;
      g = (p[0] + p[1] + p[2]) / 3.0f / 255.0f - 114.451f / 255.0f;
    }
    gray[i] = g;
  }
  __syncthreads();
  const int ly = tid / VF_TILE, lx = tid % VF_TILE;
  const int yy = ty0 + ly, xx = tx0 + lx;
  if (yy >= s || xx >= s) return;
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = gray[(ly + t / 3) * (VF_TILE + 2) + lx + t % 3];
  uint16_t* op = out + (((int64_t)img * s + yy) * s + xx) * 64;
#pragma unroll
  for (int cg = 0; cg < 8; ++cg) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc += g[t] * sw[t * 64 + c];
      o[e] = fmaxf(acc + sw[576 + c], 0.f);
    }
    *(uint4*)(op + cg * 8) = pack8<ET>(o);
  }
}

template <typename ET>
__global__ __launch_bounds__(256) void vgg_conv1_1_bwd_kernel(const uint16_t* __restrict__ dz, int batch, int s,
                                                              const float* __restrict__ w, const float* __restrict__ gt,
                                                              const float* __restrict__ pred, int ldp,
                                                              const float* __restrict__ mask, const float* __restrict__ coef,
                                                              uint16_t* __restrict__ dpred, int lddp) {
  __shared__ float sw[9 * 64];
  const int tid = threadIdx.x;
  for (int i = tid; i < 576; i += 256) sw[i] = w[i];
  __syncthreads();
  const int img = blockIdx.z;
  const int yy = blockIdx.y * VF_TILE + tid / VF_TILE, xx = blockIdx.x * VF_TILE + tid % VF_TILE;
  if (yy >= s || xx >= s) return;
  // z[q][c] = sum_t gray[q + (ky-1,kx-1)] w[t][c]  =>  dgray[p] = sum_t sum_c dz[p - (ky-1,kx-1)][c] w[t][c]
  float dgray = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int qy = yy - (t / 3 - 1), qx = xx - (t % 3 - 1);
    if (qy < 0 || qy >= s || qx < 0 || qx >= s) continue;
    const uint16_t* q = dz + (((int64_t)img * s + qy) * s + qx) * 64;
#pragma unroll
    for (int cg = 0; cg < 8; ++cg) {
      float d[8];
      unpack8<ET>(*(const uint4*)(q + cg * 8), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) dgray += d[e] * sw[t * 64 + cg * 8 + e];
    }
  }
  const int64_t p = ((int64_t)img * s + yy) * s + xx;
  const float c0 = coef[0] * (mask ? mask[p] : 1.f);
  const float dg = dgray / (3.0f * 255.0f);
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) o[ch] = dg + c0 * (pred[p * ldp + ch] - gt[p * 3 + ch]);
  uint16_t* op = dpred + p * lddp;
  *(uint4*)op = pack8<ET>(o);
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (int cg = 1; cg < lddp / 8; ++cg) *(uint4*)(op + cg * 8) = z;
}

extern "C" int imm_vgg_conv1_1_fwd(const float* gt, const float* pred, int ldp, int batch, int s, const float* w9x64,
                                   const float* b64, void* out, int dtype, void* stream) {
  IMM_REQUIRE(gt && pred && w9x64 && b64 && out, "vgg_conv1_1_fwd: null");
  IMM_REQUIRE(batch > 0 && s > 0 && ldp >= 3, "vgg_conv1_1_fwd: dims");
  const dim3 grid((s + VF_TILE - 1) / VF_TILE, (s + VF_TILE - 1) / VF_TILE, 2 * batch);
  IMM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vgg_conv1_1_fwd_kernel<ET>), grid, dim3(256), 0, (hipStream_t)stream, gt,
                                               pred, ldp, batch, s, w9x64, b64, (uint16_t*)out));
  IMM_CHECK_LAUNCH("imm_vgg_conv1_1_fwd");
  return 0;
}

extern "C" int imm_vgg_conv1_1_bwd(const void* dz, int dtype, int batch, int s, const float* w9x64, const float* gt,
                                   const float* pred, int ldp, const float* mask, const float* coef, void* dpred,
                                   int lddp, void* stream) {
  IMM_REQUIRE(dz && w9x64 && gt && pred && coef && dpred, "vgg_conv1_1_bwd: null");
  IMM_REQUIRE(batch > 0 && s > 0 && ldp >= 3 && lddp >= 8 && lddp % 8 == 0, "vgg_conv1_1_bwd: dims");
  const dim3 grid((s + VF_TILE - 1) / VF_TILE, (s + VF_TILE - 1) / VF_TILE, batch);
  IMM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vgg_conv1_1_bwd_kernel<ET>), grid, dim3(256), 0, (hipStream_t)stream,
                                               (const uint16_t*)dz, batch, s, w9x64, gt, pred, ldp, mask, coef,
                                               (uint16_t*)dpred, lddp));
  IMM_CHECK_LAUNCH("imm_vgg_conv1_1_bwd");
  return 0;
}
