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

// Thread mapping (both kernels): a workgroup owns a 16x16 pixel tile; 8 lanes share a pixel, each owning 8 of the 64
// channels, so a wave touches 8 neighbouring pixels x 128 contiguous bytes (fully coalesced 16-byte stores/loads) and
// every lane keeps its 9x8 filter taps in registers.
template <typename ET>
__global__ __launch_bounds__(256) void vgg_conv1_1_fwd_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                              int ldp, int batch, int s, const float* __restrict__ w,
                                                              const float* __restrict__ bias, typename ET::T* __restrict__ out,
                                                              int img0) {
  __shared__ float gray[(VF_TILE + 2) * (VF_TILE + 2)];
  const int tid = threadIdx.x;
  const int img = blockIdx.z + img0;
  const int ty0 = blockIdx.y * VF_TILE, tx0 = blockIdx.x * VF_TILE;
  const float* src = img < batch ? gt + (int64_t)img * s * s * 3 : pred + (int64_t)(img - batch) * s * s * ldp;
  const int ld = img < batch ? 3 : ldp;
  for (int i = tid; i < (VF_TILE + 2) * (VF_TILE + 2); i += 256) {
    const int yy = ty0 + i / (VF_TILE + 2) - 1, xx = tx0 + i % (VF_TILE + 2) - 1;
    float g = 0.f;   // SAME zero padding applies to the NORMALISED gray image
    if (yy >= 0 && yy < s && xx >= 0 && xx < s) {
      const float* p = src + ((int64_t)yy * s + xx) * ld;
      g = (p[0] + p[1] + p[2]) / 3.0f / 255.0f - 114.451f / 255.0f;
    }
    gray[i] = g;
  }
  const int cg = tid & 7, pl = tid >> 3;      // channel group, pixel slot (32 pixels per pass)
  // The kernel is VALU-bound, not store-bound (1.2 GFLOP of f32 FMAs = 15 us at the part's 79 TFLOP/s, plus everything around
  // them): channel PAIRS as packed-f32 FMAs (v_pk_fma_f32: 36 per pixel instead of hipcc's own 32 v_pk_mul + 75 v_add + 46 moves
  // for the scalar form), accumulators starting at the bias.
  f32x2_t wr[9][4], br[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    br[e] = f32x2_t{bias[cg * 8 + 2 * e], bias[cg * 8 + 2 * e + 1]};
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[t][e] = f32x2_t{w[t * 64 + cg * 8 + 2 * e], w[t * 64 + cg * 8 + 2 * e + 1]};
  }
  __syncthreads();
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int pix = pass * 32 + pl;
    const int ly = pix / VF_TILE, lx = pix % VF_TILE;
    const int yy = ty0 + ly, xx = tx0 + lx;
    if (yy >= s || xx >= s) continue;
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) g[t] = gray[(ly + t / 3) * (VF_TILE + 2) + lx + t % 3];
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x2_t acc = br[e];
#pragma unroll
      for (int t = 0; t < 9; ++t) acc = __builtin_elementwise_fma(f32x2_t{g[t], g[t]}, wr[t][e], acc);
      o[2 * e] = fmaxf(acc[0], 0.f);
      o[2 * e + 1] = fmaxf(acc[1], 0.f);
    }
    st8<ET>(out + (((int64_t)img * s + yy) * s + xx) * 64 + cg * 8, pack8<ET>(o));
  }
}

template <typename ET>
__global__ __launch_bounds__(256) void vgg_conv1_1_bwd_kernel(const uint16_t* __restrict__ dz, int batch, int s,
                                                              const float* __restrict__ w, const float* __restrict__ gt,
                                                              const float* __restrict__ pred, int ldp,
                                                              const float* __restrict__ mask, const float* __restrict__ coef,
                                                              int input_idx, int l1, uint16_t* __restrict__ dpred, int lddp) {
  constexpr int HT = VF_TILE + 2;
  __shared__ uint4 sdz[HT * HT * 8];          // dz tile + halo, 64 channels = 8 x 16 B per pixel (41 KB)
  __shared__ float sdir[VF_TILE * VF_TILE][3]; // the 'input' feature's direct term c0 * mask * (pred - gt) of the tile's pixels
  const int tid = threadIdx.x;
  const int img = blockIdx.z;
  const int ty0 = blockIdx.y * VF_TILE, tx0 = blockIdx.x * VF_TILE;
  const float c0 = input_idx >= 0 ? coef[input_idx] : 0.f;      // 'input' feature not in perceptual.comp: no direct term
  {
    // thread t = pixel t of the tile: its pred / gt / mask values are requested HERE, together with the dz tile (they were 7
    // dependent 4-byte loads per pixel at the end of every pass of the loop below, by one lane in eight: a chain of HBM
    // latencies; 44 -> 35 us.  A GEMM + gather formulation on the matrix cores was no faster: with the 64-byte gradient
    // pixels of the renderer head and the strided pred reads the kernel moves ~150 MB, it is HBM-bound)
    const int yy = ty0 + tid / VF_TILE, xx = tx0 + tid % VF_TILE;
    float dir[3] = {0.f, 0.f, 0.f};
    if (yy < s && xx < s) {
      const int64_t p = ((int64_t)img * s + yy) * s + xx;
      const float cm = c0 * (mask ? mask[p] : 1.f);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float d = pred[p * ldp + ch] - gt[p * 3 + ch];
        if (l1) d = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);       // perceptual.l2: False => |.|, gradient sign(d)
        dir[ch] = cm * d;
      }
    }
    // all 11 loads of a thread in flight before the first LDS store (the tile load is a chain of HBM latencies otherwise)
    constexpr int NL = (HT * HT * 8 + 255) / 256;
    uint4 v[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + k * 256;
      const int hp = i >> 3, c = i & 7;
      const int yy = ty0 + hp / HT - 1, xx = tx0 + hp % HT - 1;
      v[k] = make_uint4(0, 0, 0, 0);
      if (i < HT * HT * 8 && yy >= 0 && yy < s && xx >= 0 && xx < s)
        v[k] = *(const uint4*)(dz + (((int64_t)img * s + yy) * s + xx) * 64 + c * 8);
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + k * 256;
      if (i < HT * HT * 8) sdz[i] = v[k];
    }
    sdir[tid][0] = dir[0]; sdir[tid][1] = dir[1]; sdir[tid][2] = dir[2];
  }
  const int cg = tid & 7, pl = tid >> 3;
  // filter taps of this lane's 8 channels as packed 16-bit pairs: the contraction runs on v_dot2c (2 MACs per issue;
  // weights at the activation precision, like every other VGG data gradient)
  uint32_t wr[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wr[t][e] = ET::pack2(w[t * 64 + cg * 8 + 2 * e], w[t * 64 + cg * 8 + 2 * e + 1]);
  __syncthreads();
#pragma unroll 2
  for (int pass = 0; pass < 8; ++pass) {
    const int pix = pass * 32 + pl;
    const int ly = pix / VF_TILE, lx = pix % VF_TILE;
    const int yy = ty0 + ly, xx = tx0 + lx;
    // z[q][c] = sum_t gray[q + (ky-1,kx-1)] w[t][c]  =>  dgray[p] = sum_t sum_c dz[p - (ky-1,kx-1)][c] w[t][c]
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int hy = ly + 1 - (t / 3 - 1), hx = lx + 1 - (t % 3 - 1);     // halo coordinates of q
      const uint4 d = sdz[(hy * HT + hx) * 8 + cg];
      part = ET::dot2(d.x, wr[t][0], part);
      part = ET::dot2(d.y, wr[t][1], part);
      part = ET::dot2(d.z, wr[t][2], part);
      part = ET::dot2(d.w, wr[t][3], part);
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (yy >= s || xx >= s) continue;
    const int nvec = lddp / 8;
    if (cg < nvec) {
      float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int64_t p = ((int64_t)img * s + yy) * s + xx;
      if (cg == 0) {
        const float dg = part / (3.0f * 255.0f);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) o[ch] = dg + sdir[pix][ch];
      }
      *(uint4*)(dpred + p * lddp + cg * 8) = pack8<ET>(o);
    }
  }
}

// f32 storage (the witness engine, IMM_F32): the same gradient in plain f32 arithmetic, one thread per pixel — dgray[p] =
// sum_t sum_c dz[p - (ky-1, kx-1)][c] w[t][c] (zero outside the image), dpred[p][ch < 3] = dgray / (3 * 255) + c0 * mask * (pred -
// gt) (sign with l1), channels 3 .. lddp - 1 zero.
__global__ __launch_bounds__(256) void vgg_conv1_1_bwd_f32_kernel(const float* __restrict__ dz, int batch, int s,
                                                                  const float* __restrict__ w, const float* __restrict__ gt,
                                                                  const float* __restrict__ pred, int ldp,
                                                                  const float* __restrict__ mask, const float* __restrict__ coef,
                                                                  int input_idx, int l1, float* __restrict__ dpred, int lddp) {
  __shared__ float sw[9 * 64];
  for (int i = threadIdx.x; i < 9 * 64; i += 256) sw[i] = w[i];
  __syncthreads();
  const int64_t npix = (int64_t)batch * s * s;
  const float c0 = input_idx >= 0 ? coef[input_idx] : 0.f;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (int64_t)gridDim.x * 256) {
    const int xx = (int)(p % s);
    const int64_t t0 = p / s;
    const int yy = (int)(t0 % s);
    const int64_t img = t0 / s;
    float part = 0.f;
    for (int t = 0; t < 9; ++t) {
      const int qy = yy - (t / 3 - 1), qx = xx - (t % 3 - 1);
      if (qy < 0 || qy >= s || qx < 0 || qx >= s) continue;
      const float* d = dz + ((img * s + qy) * s + qx) * 64;
      for (int c = 0; c < 64; ++c) part = fmaf(d[c], sw[t * 64 + c], part);
    }
    const float dg = part / (3.0f * 255.0f);
    const float cm = c0 * (mask ? mask[p] : 1.f);
    for (int ch = 0; ch < lddp; ++ch) {
      float o = 0.f;
      if (ch < 3) {
        float d = pred[p * ldp + ch] - gt[p * 3 + ch];
        if (l1) d = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        o = dg + cm * d;
      }
      dpred[p * lddp + ch] = o;
    }
  }
}

extern "C" int imm_vgg_conv1_1_fwd(const float* gt, const float* pred, int ldp, int batch, int s, const float* w9x64,
                                   const float* b64, void* out, int dtype, int halves, void* stream) {
  IMM_REQUIRE(gt && pred && w9x64 && b64 && out, "vgg_conv1_1_fwd: null");
  IMM_REQUIRE(batch > 0 && s > 0 && ldp >= 3, "vgg_conv1_1_fwd: dims");
  IMM_REQUIRE(halves >= 1 && halves <= 3, "vgg_conv1_1_fwd: halves must be 1 (gt), 2 (pred) or 3 (both)");
  const dim3 grid((s + VF_TILE - 1) / VF_TILE, (s + VF_TILE - 1) / VF_TILE, halves == 3 ? 2 * batch : batch);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((vgg_conv1_1_fwd_kernel<ET>), grid, dim3(256), 0, (hipStream_t)stream, gt,
                                                   pred, ldp, batch, s, w9x64, b64, (typename ET::T*)out, halves == 2 ? batch : 0));
  IMM_CHECK_LAUNCH("imm_vgg_conv1_1_fwd");
  return 0;
}

extern "C" int imm_vgg_conv1_1_bwd(const void* dz, int dtype, int batch, int s, const float* w9x64, const float* gt,
                                   const float* pred, int ldp, const float* mask, const float* coef, int input_idx, int l1,
                                   void* dpred, int lddp, void* stream) {
  IMM_REQUIRE(dz && w9x64 && gt && pred && coef && dpred, "vgg_conv1_1_bwd: null");
  IMM_REQUIRE(batch > 0 && s > 0 && ldp >= 3 && lddp >= 8 && lddp % 8 == 0 && lddp <= 64, "vgg_conv1_1_bwd: dims");
  if (dtype == IMM_F32) {
    const int64_t npix = (int64_t)batch * s * s;
    hipLaunchKernelGGL(vgg_conv1_1_bwd_f32_kernel, dim3((unsigned)((npix + 255) / 256 > 8192 ? 8192 : (npix + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float*)dz, batch, s, w9x64, gt, pred, ldp, mask, coef, input_idx, l1, (float*)dpred, lddp);
    IMM_CHECK_LAUNCH("imm_vgg_conv1_1_bwd(f32)");
    return 0;
  }
  const dim3 grid((s + VF_TILE - 1) / VF_TILE, (s + VF_TILE - 1) / VF_TILE, batch);
  IMM_DISPATCH_DTYPE(dtype, hipLaunchKernelGGL((vgg_conv1_1_bwd_kernel<ET>), grid, dim3(256), 0, (hipStream_t)stream,
                                               (const uint16_t*)dz, batch, s, w9x64, gt, pred, ldp, mask, coef, input_idx, l1,
                                               (uint16_t*)dpred, lddp));
  IMM_CHECK_LAUNCH("imm_vgg_conv1_1_bwd");
  return 0;
}
