// loss_optim.hip — perceptual-loss reductions / gradient injection and the fused clip+Adam optimizer.
//
// Reference: imm/models/imm_model.py:111-151 (_colorization_reconstruction_loss: masked squared
// feature differences, normalised by a DIFFERENTIABLE running average, base_model.py:39-50),
// :408-410 (_loss_mask = legacy resize of the mask == strided pick), base_model.py:33-37 (_decay),
// imm/train/cnn_train_multi.py:86-98,231-243 (tower mean -> tf.clip_by_norm per tensor) and
// scripts/train.py:92-98 (staircase lr, tf.train.AdamOptimizer).
// Everything the host would otherwise need to read back between launches (loss normalisers,
// gradient coefficients, step count, learning rate) lives in device memory, so the whole step is a
// fixed launch sequence that a HIP graph can replay.
#include "common.h"

#define LO_THREADS 256
#define IMM_NORM2_SANE 1e30f   // a chunk's sum of squared gradients beyond this counts as an overflow (grad_prepare_kernel)

// ---------------------------------------------------------------------------------------------
// masked sum of squared differences (per feature; IMM_SSE_BLOCKS deterministic partials)
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ __launch_bounds__(LO_THREADS) void masked_sse_kernel(const typename ET::T* __restrict__ a, const typename ET::T* __restrict__ b,
                                                                int batch, int s, int c8n, const float* __restrict__ mask,
                                                                int S, int l1, float* __restrict__ partial) {
  __shared__ float red[4];
  const int r = S / s;
  const int64_t total = (int64_t)batch * s * s * c8n;
  float acc = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / c8n;
    float mk = 1.f;
    if (mask) {
      const int xx = (int)(p % s);
      const int64_t t = p / s;
      const int yy = (int)(t % s);
      const int64_t bi = t / s;
      mk = mask[(bi * S + (int64_t)yy * r) * S + (int64_t)xx * r];
    }
    float fa[8], fb[8];
    unpack8<ET>(ld8<ET>(a + idx * 8), fa);
    unpack8<ET>(ld8<ET>(b + idx * 8), fb);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = fa[i] - fb[i]; sq += l1 ? fabsf(d) : d * d; }   // perceptual.l2 (imm_model.py:132)
    acc += mk * sq;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// Several features in ONE launch (blockIdx.y = feature): the error sums of the deep tapped layers (conv3_2, conv4_2, conv5_2)
// sit back to back on the one-lane critical path in front of the loss; each is a 5-11 us launch.
struct SseMultiArgs {
  const void* a[8]; const void* b[8];      // 16-bit or f32 features (the kernel's element type)
  int s[8], c8n[8];
  float* partial[8];
};
template <typename ET>
__global__ __launch_bounds__(LO_THREADS) void masked_sse_multi_kernel(const SseMultiArgs g, int batch, const float* __restrict__ mask,
                                                                      int S, int l1) {
  __shared__ float red[4];
  const int f = blockIdx.y;
  const typename ET::T* __restrict__ a = (const typename ET::T*)g.a[f];
  const typename ET::T* __restrict__ b = (const typename ET::T*)g.b[f];
  const int s = g.s[f], c8n = g.c8n[f];
  const int r = S / s;
  const int64_t total = (int64_t)batch * s * s * c8n;
  float acc = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / c8n;
    float mk = 1.f;
    if (mask) {
      const int xx = (int)(p % s);
      const int64_t t = p / s;
      const int yy = (int)(t % s);
      const int64_t bi = t / s;
      mk = mask[(bi * S + (int64_t)yy * r) * S + (int64_t)xx * r];
    }
    float fa[8], fb[8];
    unpack8<ET>(ld8<ET>(a + idx * 8), fa);
    unpack8<ET>(ld8<ET>(b + idx * 8), fb);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = fa[i] - fb[i]; sq += l1 ? fabsf(d) : d * d; }
    acc += mk * sq;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) g.partial[f][blockIdx.x] = acc;
}

// The same sum fused with the 2x2 max-pool that follows the tapped layer (conv1_2, conv2_2: imm/models/selfsup/vgg16.py
// pool1 / pool2 right after the feature the loss reads): one pass over the two feature halves instead of two.
// a = ground-truth half, b = prediction half [batch,s,s,c]; pool_a / pool_b [batch,s/2,s/2,c].
template <typename ET>
__global__ __launch_bounds__(LO_THREADS) void masked_sse_pool_kernel(const typename ET::T* __restrict__ a, const typename ET::T* __restrict__ b,
                                                                     int batch, int s, int c8n, const float* __restrict__ mask,
                                                                     int S, float* __restrict__ partial,
                                                                     typename ET::T* __restrict__ pool_a, typename ET::T* __restrict__ pool_b) {
  __shared__ float red[4];
  const int r = S / s, so = s / 2, c = c8n * 8;
  const int64_t total = (int64_t)batch * so * so * c8n;
  float acc = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int X = (int)(t % so); t /= so;
    const int Y = (int)(t % so);
    const int64_t bi = t / so;
    float ma[8], mb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { ma[i] = -__builtin_huge_valf(); mb[i] = -__builtin_huge_valf(); }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int yy = 2 * Y + dy, xx = 2 * X + dx;
        const int64_t off = ((bi * s + yy) * s + xx) * c + cg * 8;
        float fa[8], fb[8];
        unpack8<ET>(ld8<ET>(a + off), fa);
        unpack8<ET>(ld8<ET>(b + off), fb);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float d = fa[i] - fb[i];
          sq += d * d;
          ma[i] = fmaxf(ma[i], fa[i]); mb[i] = fmaxf(mb[i], fb[i]);
        }
        const float mk = mask ? mask[(bi * S + (int64_t)yy * r) * S + (int64_t)xx * r] : 1.f;
        acc += mk * sq;
      }
    const int64_t po = ((bi * so + Y) * so + X) * c + cg * 8;
    if (pool_a) st8<ET>(pool_a + po, pack8<ET>(ma));
    st8<ET>(pool_b + po, pack8<ET>(mb));
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// this thread's share of sum_p mask[p] * sum_ch (a - b)^2 over f32 pixels (grid-stride over gridDim.x blocks)
__device__ __forceinline__ float sse_f32_thread_sum(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
                                                    int64_t npix, int c, const float* __restrict__ mask, int l1) {
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 3) {
    // RGB (the 'input' term of the perceptual loss): four pixels' loads in flight per thread, added in the order of the plain
    // loop below (this launch sits on the one-lane path in front of the loss: 12.4 us as four dependent trips of scalar loads)
    const bool a4 = lda == 4 && ((uintptr_t)a & 15) == 0, b4 = ldb == 4 && ((uintptr_t)b & 15) == 0;
    for (; p + 3 * stride < npix; p += 4 * stride) {
      float va[4][3], vb[4][3], mk[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t q = p + u * stride;
        if (a4) { const float4 t = *(const float4*)(a + q * 4); va[u][0] = t.x; va[u][1] = t.y; va[u][2] = t.z; }
        else { va[u][0] = a[q * lda]; va[u][1] = a[q * lda + 1]; va[u][2] = a[q * lda + 2]; }
        if (b4) { const float4 t = *(const float4*)(b + q * 4); vb[u][0] = t.x; vb[u][1] = t.y; vb[u][2] = t.z; }
        else { vb[u][0] = b[q * ldb]; vb[u][1] = b[q * ldb + 1]; vb[u][2] = b[q * ldb + 2]; }
        mk[u] = mask ? mask[q] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float sq = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { const float d = va[u][ch] - vb[u][ch]; sq += l1 ? fabsf(d) : d * d; }
        acc += mk[u] * sq;
      }
    }
  }
  for (; p < npix; p += stride) {
    float sq = 0.f;
    for (int ch = 0; ch < c; ++ch) { const float d = a[p * lda + ch] - b[p * ldb + ch]; sq += l1 ? fabsf(d) : d * d; }
    acc += (mask ? mask[p] : 1.f) * sq;
  }
  return acc;
}

__global__ __launch_bounds__(LO_THREADS) void masked_sse_f32_kernel(const float* __restrict__ a, int lda,
                                                                    const float* __restrict__ b, int ldb, int64_t npix,
                                                                    int c, const float* __restrict__ mask, int l1,
                                                                    float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = sse_f32_thread_sum(a, lda, b, ldb, npix, c, mask, l1);
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// The error sums in front of the loss as ONE launch (round 6): blockIdx.y < n16 = the 16-bit features of masked_sse_multi_kernel,
// blockIdx.y == n16 = the f32 image pair of masked_sse_f32_kernel (the 'input' feature of perceptual.comp, imm_model.py:126-131).
// Block and thread mapping, and so every partial sum, are those of the two launches it replaces.
struct SseRgbArgs { const float* a; const float* b; int lda, ldb, c; int64_t npix; float* partial; };
template <typename ET>
__global__ __launch_bounds__(LO_THREADS) void masked_sse_all_kernel(const SseMultiArgs g, int n16, const SseRgbArgs rgb, int batch,
                                                                    const float* __restrict__ mask, int S, int l1) {
  __shared__ float red[4];
  const int f = blockIdx.y;
  float acc = 0.f;
  if (f == n16) {
    acc = sse_f32_thread_sum(rgb.a, rgb.lda, rgb.b, rgb.ldb, rgb.npix, rgb.c, mask, l1);
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) rgb.partial[blockIdx.x] = acc;
    return;
  }
  const typename ET::T* __restrict__ a = (const typename ET::T*)g.a[f];
  const typename ET::T* __restrict__ b = (const typename ET::T*)g.b[f];
  const int s = g.s[f], c8n = g.c8n[f];
  const int r = S / s;
  const int64_t total = (int64_t)batch * s * s * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / c8n;
    float mk = 1.f;
    if (mask) {
      const int xx = (int)(p % s);
      const int64_t t = p / s;
      const int yy = (int)(t % s);
      const int64_t bi = t / s;
      mk = mask[(bi * S + (int64_t)yy * r) * S + (int64_t)xx * r];
    }
    float fa[8], fb[8];
    unpack8<ET>(ld8<ET>(a + idx * 8), fa);
    unpack8<ET>(ld8<ET>(b + idx * 8), fb);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = fa[i] - fb[i]; sq += l1 ? fabsf(d) : d * d; }
    acc += mk * sq;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) g.partial[f][blockIdx.x] = acc;
}

extern "C" int imm_masked_sse(const void* a, const void* b, int dtype, int batch, int s, int c, const float* mask, int S,
                              int l1, float* partial, void* stream) {
  IMM_REQUIRE(a && b && partial && batch > 0 && s > 0 && c > 0 && c % 8 == 0, "masked_sse: args");
  IMM_REQUIRE(mask == nullptr || (S >= s && S % s == 0), "masked_sse: mask side %d not a multiple of feature side %d", S, s);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((masked_sse_kernel<ET>), dim3(IMM_SSE_BLOCKS), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)a, (const typename ET::T*)b, batch, s, c / 8,
                                               mask, S, l1, partial));
  IMM_CHECK_LAUNCH("imm_masked_sse");
  return 0;
}

extern "C" int imm_masked_sse_multi(int n, const void* const* a, const void* const* b, const int32_t* s_host, const int32_t* c_host,
                                    float* const* partial, int dtype, int batch, const float* mask, int S, int l1, void* stream) {
  IMM_REQUIRE(n >= 1 && n <= 8 && a && b && s_host && c_host && partial && batch > 0, "masked_sse_multi: args");
  SseMultiArgs g;
  for (int i = 0; i < 8; ++i) {
    const int k = i < n ? i : 0;
    IMM_REQUIRE(a[k] && b[k] && partial[k] && s_host[k] > 0 && c_host[k] > 0 && c_host[k] % 8 == 0, "masked_sse_multi: feature %d", k);
    IMM_REQUIRE(mask == nullptr || (S >= s_host[k] && S % s_host[k] == 0), "masked_sse_multi: mask side %d vs feature side %d", S, s_host[k]);
    g.a[i] = a[k]; g.b[i] = b[k]; g.s[i] = s_host[k]; g.c8n[i] = c_host[k] / 8; g.partial[i] = partial[k];
  }
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((masked_sse_multi_kernel<ET>), dim3(IMM_SSE_BLOCKS, n), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, g, batch, mask, S, l1));
  IMM_CHECK_LAUNCH("imm_masked_sse_multi");
  return 0;
}

extern "C" int imm_masked_sse_all(int n, const void* const* a, const void* const* b, const int32_t* s_host, const int32_t* c_host,
                                  float* const* partial, int dtype, int batch, const float* mask, int S, int l1,
                                  const float* img_a, int lda, const float* img_b, int ldb, int img_c, float* img_partial, void* stream) {
  IMM_REQUIRE(n >= 1 && n <= 8 && a && b && s_host && c_host && partial && batch > 0, "masked_sse_all: args");
  IMM_REQUIRE(img_a && img_b && img_partial && img_c > 0 && lda >= img_c && ldb >= img_c && S > 0, "masked_sse_all: image pair");
  SseMultiArgs g;
  for (int i = 0; i < 8; ++i) {
    const int k = i < n ? i : 0;
    IMM_REQUIRE(a[k] && b[k] && partial[k] && s_host[k] > 0 && c_host[k] > 0 && c_host[k] % 8 == 0, "masked_sse_all: feature %d", k);
    IMM_REQUIRE(S >= s_host[k] && S % s_host[k] == 0, "masked_sse_all: image side %d vs feature side %d", S, s_host[k]);
    g.a[i] = a[k]; g.b[i] = b[k]; g.s[i] = s_host[k]; g.c8n[i] = c_host[k] / 8; g.partial[i] = partial[k];
  }
  SseRgbArgs rgb;
  rgb.a = img_a; rgb.b = img_b; rgb.lda = lda; rgb.ldb = ldb; rgb.c = img_c; rgb.npix = (int64_t)batch * S * S; rgb.partial = img_partial;
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((masked_sse_all_kernel<ET>), dim3(IMM_SSE_BLOCKS, n + 1), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, g, n, rgb, batch, mask, S, l1));
  IMM_CHECK_LAUNCH("imm_masked_sse_all");
  return 0;
}

extern "C" int imm_masked_sse_pool(const void* a, const void* b, int dtype, int batch, int s, int c, const float* mask, int S,
                                   float* partial, void* pool_a, void* pool_b, void* stream) {
  IMM_REQUIRE(a && b && partial && pool_b && batch > 0 && s > 0 && s % 2 == 0 && c > 0 && c % 8 == 0, "masked_sse_pool: args");
  IMM_REQUIRE(mask == nullptr || (S >= s && S % s == 0), "masked_sse_pool: mask side %d not a multiple of feature side %d", S, s);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((masked_sse_pool_kernel<ET>), dim3(IMM_SSE_BLOCKS), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, (const typename ET::T*)a, (const typename ET::T*)b, batch, s, c / 8,
                                               mask, S, partial, (typename ET::T*)pool_a, (typename ET::T*)pool_b));
  IMM_CHECK_LAUNCH("imm_masked_sse_pool");
  return 0;
}

extern "C" int imm_masked_sse_f32(const float* a, int lda, const float* b, int ldb, int batch, int s, int c,
                                  const float* mask, int l1, float* partial, void* stream) {
  IMM_REQUIRE(a && b && partial && batch > 0 && s > 0 && c > 0 && lda >= c && ldb >= c, "masked_sse_f32: args");
  hipLaunchKernelGGL(masked_sse_f32_kernel, dim3(IMM_SSE_BLOCKS), dim3(LO_THREADS), 0, (hipStream_t)stream, a, lda, b, ldb,
                     (int64_t)batch * s * s, c, mask, l1, partial);
  IMM_CHECK_LAUNCH("imm_masked_sse_f32");
  return 0;
}

// m_k = SSE_k / nel_k ; wl_k = a_k + 0.01 (m_k - a_k) ; term_k = m_k / wl_k
// d(1000*sum term)/d a_pred = c_k * mask * (a_pred - a_gt),  c_k = 1000 * (1/wl - 0.01 m/wl^2) * 2/nel
// l1 (perceptual.l2: False, imm_model.py:132): the sums are of |d|, c_k = 1000 * (1/wl - 0.01 m/wl^2) / nel multiplies sign(d).
// mode IMM_LOSS_L2 (reconstruction_loss: 'l2', imm_model.py:385-387,399): one feature (the image), no normaliser:
//   reconstruction = 1000 * m, total = reconstruction / 255 + weight decay, c_0 = (1000/255) * 2/nel.
#define PF_THREADS 1024
__global__ __launch_bounds__(PF_THREADS) void perceptual_finalize_kernel(const float* __restrict__ partial, int nfeat,
                                                                         const float* __restrict__ nel, float* agg,
                                                                         int training, const float* __restrict__ wd_loss,
                                                                         int l1, int mode, const float* __restrict__ loss_scale,
                                                                         float* __restrict__ out) {
  __shared__ double sse[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the scalars of the tail, requested first (thread f owns feature f)
  const float nel_f = tid < nfeat ? nel[tid] : 1.f;
  const float agg_f = tid < nfeat ? agg[tid] : 1.f;
  const float ls = (tid < nfeat && loss_scale) ? loss_scale[0] : 1.f;
  const float wd = (tid == 0 && wd_loss) ? wd_loss[0] : 0.f;
  // One WAVE per feature (16 waves, nfeat <= 16): a lane's eight partials in flight at once, added in index order in f64, one
  // butterfly, one barrier.  Round 6: the form before this one kept every feature's sums in every thread (16x unrolled loads +
  // f64 butterflies: 28 KB of straight-line code that ONE workgroup walked through a cold instruction cache — 17 us on the
  // critical path between the last error sum and the first gradient for a few hundred bytes of input); this one is < 2 KB.
  if (wave < nfeat) {
    const float* pf = partial + wave * IMM_SSE_BLOCKS + lane;
    float x[IMM_SSE_BLOCKS / 64];
#pragma unroll
    for (int i = 0; i < IMM_SSE_BLOCKS / 64; ++i) x[i] = pf[64 * i];
    double v = 0.0;
#pragma unroll
    for (int i = 0; i < IMM_SSE_BLOCKS / 64; ++i) v += (double)x[i];
#pragma unroll 1
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) sse[wave] = v;
  }
  __syncthreads();
  // loss scaling (16-bit gradient storage with f16's range): every seed of the backward pass carries the factor S; the
  // backward chain is linear in its seeds, imm_clip_adam_step divides the flat gradients by S again
  if (mode == IMM_LOSS_L2) {
    if (tid == 0) {
      const float m = (float)(sse[0] / (double)nel_f);
      out[0] = m; out[nfeat] = m;
      out[2 * nfeat] = ls * ((1000.f / 255.f) * 2.f / nel_f);
      out[3 * nfeat] = 1000.f * m;
      out[3 * nfeat + 1] = wd;
      out[3 * nfeat + 2] = 1000.f * m / 255.f + wd;
    }
    return;
  }
  // one thread per feature (their nel / agg / scale loads are one round trip instead of nfeat dependent ones on thread 0:
  // 18 -> 7 us between the last error sum and the first gradient launch); the terms are then added in feature order
  __shared__ float terms[16];
  if (tid < nfeat) {
    const int f = tid;
    const float m = (float)(sse[f] / (double)nel_f);
    const float a = agg_f;
    const float wl = a + 0.01f * (m - a);
    const float term = m / wl;
    out[f] = term;
    out[nfeat + f] = m;
    out[2 * nfeat + f] = ls * (1000.f * (1.f / wl - 0.01f * m / (wl * wl)) * (l1 ? 1.f : 2.f) / nel_f);
    if (training) agg[f] = wl;
    terms[f] = term;
  }
  __syncthreads();
  if (tid == 0) {
    float rec = 0.f;
    for (int f = 0; f < nfeat; ++f) rec += terms[f];
    rec *= 1000.f;
    out[3 * nfeat] = rec;
    out[3 * nfeat + 1] = wd;
    out[3 * nfeat + 2] = rec + wd;
  }
}

extern "C" int imm_perceptual_finalize(const float* partial, int nfeat, const float* nel, float* agg, int training,
                                       const float* wd_loss, int l1, int mode, const float* loss_scale, float* out,
                                       void* stream) {
  IMM_REQUIRE(partial && nel && agg && out && nfeat > 0 && nfeat <= 16, "perceptual_finalize: args");
  IMM_REQUIRE(mode == IMM_LOSS_PERCEPTUAL || (mode == IMM_LOSS_L2 && nfeat == 1 && !l1), "perceptual_finalize: mode");
  hipLaunchKernelGGL(perceptual_finalize_kernel, dim3(1), dim3(PF_THREADS), 0, (hipStream_t)stream, partial, nfeat, nel,
                     agg, training, wd_loss, l1, mode, loss_scale, out);
  IMM_CHECK_LAUNCH("imm_perceptual_finalize");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// gradient injection at a perceptual tap (+ fused ReLU backward of the tapped activation)
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ void tap_grad_kernel(typename ET::T* __restrict__ da, int has_in, const typename ET::T* __restrict__ ap,
                                const typename ET::T* __restrict__ ag, int batch, int s, int c8n, const float* __restrict__ mask,
                                int S, const float* __restrict__ coef, int idx_coef, int relu, int l1) {
  const int r = S / s;
  const float ck = coef[idx_coef];
  const int64_t total = (int64_t)batch * s * s * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / c8n;
    float mk = 1.f;
    if (mask) {
      const int xx = (int)(p % s);
      const int64_t t = p / s;
      const int yy = (int)(t % s);
      const int64_t bi = t / s;
      mk = mask[(bi * S + (int64_t)yy * r) * S + (int64_t)xx * r];
    }
    float fp[8], fg[8], d[8];
    unpack8<ET>(ld8<ET>(ap + idx * 8), fp);
    unpack8<ET>(ld8<ET>(ag + idx * 8), fg);
    if (has_in) unpack8<ET>(ld8<ET>(da + idx * 8), d);
    const float cm = ck * mk;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float e = fp[i] - fg[i];
      if (l1) e = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
      float v = (has_in ? d[i] : 0.f) + cm * e;
      if (relu && !(fp[i] > 0.f)) v = 0.f;
      d[i] = v;
    }
    st8<ET>(da + idx * 8, pack8<ET>(d));
  }
}

// tap_grad fused with the max-pool backward in front of it (conv1_2, conv2_2: the tapped layer is also the pooled one):
// da[p] = relu'(ap[p]) * ( [p is the argmax of its 2x2 window] * dpool[window] + c_k * mask[p] * (ap[p] - ag[p]) ).
// One pass instead of maxpool2_bwd (write da) + tap_grad (read da, ap again); bitwise equal to that sequence.
template <typename ET>
__global__ void unpool_tap_grad_kernel(typename ET::T* __restrict__ da, const typename ET::T* __restrict__ dpool,
                                       const typename ET::T* __restrict__ ap, const typename ET::T* __restrict__ ag, int batch, int s,
                                       int c8n, const float* __restrict__ mask, int S, const float* __restrict__ coef,
                                       int idx_coef) {
  const int r = S / s, so = s / 2, c = c8n * 8;
  const float ck = coef[idx_coef];
  const int64_t total = (int64_t)batch * so * so * c8n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % c8n);
    int64_t t = idx / c8n;
    const int X = (int)(t % so); t /= so;
    const int Y = (int)(t % so);
    const int64_t bi = t / so;
    const int64_t off = ((bi * s + 2 * Y) * s + 2 * X) * c + cg * 8;
    const int64_t qoff[4] = {off, off + c, off + (int64_t)s * c, off + (int64_t)s * c + c};
    float fp[4][8], fg[4][8], g[8], o[4][8], cm[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unpack8<ET>(ld8<ET>(ap + qoff[q]), fp[q]);
      unpack8<ET>(ld8<ET>(ag + qoff[q]), fg[q]);
      const int yy = 2 * Y + (q >> 1), xx = 2 * X + (q & 1);
      cm[q] = ck * (mask ? mask[(bi * S + (int64_t)yy * r) * S + (int64_t)xx * r] : 1.f);
    }
    unpack8<ET>(ld8<ET>(dpool + ((bi * so + Y) * so + X) * c + cg * 8), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int am = 0; float mv = fp[0][i];
#pragma unroll
      for (int q = 1; q < 4; ++q) if (fp[q][i] > mv) { mv = fp[q][i]; am = q; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = ((q == am) ? g[i] : 0.f) + cm[q] * (fp[q][i] - fg[q][i]);
        if (!(fp[q][i] > 0.f)) v = 0.f;
        o[q][i] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) st8<ET>(da + qoff[q], pack8<ET>(o[q]));
  }
}

extern "C" int imm_unpool_tap_grad(void* da, const void* dpool, const void* a_pred, const void* a_gt, int dtype, int batch, int s,
                                   int c, const float* mask, int S, const float* coef, int idx, void* stream) {
  IMM_REQUIRE(da && dpool && a_pred && a_gt && coef && batch > 0 && s > 0 && s % 2 == 0 && c > 0 && c % 8 == 0 && idx >= 0,
              "unpool_tap_grad: args");
  IMM_REQUIRE(mask == nullptr || (S >= s && S % s == 0), "unpool_tap_grad: mask side");
  const int64_t total = (int64_t)batch * (s / 2) * (s / 2) * (c / 8);
  int64_t blocks = (total + LO_THREADS - 1) / LO_THREADS;
  if (blocks > 16384) blocks = 16384;
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((unpool_tap_grad_kernel<ET>), dim3((int)blocks), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, (typename ET::T*)da, (const typename ET::T*)dpool,
                                               (const typename ET::T*)a_pred, (const typename ET::T*)a_gt, batch, s, c / 8, mask, S, coef, idx));
  IMM_CHECK_LAUNCH("imm_unpool_tap_grad");
  return 0;
}

extern "C" int imm_tap_grad(void* da, int has_in, const void* a_pred, const void* a_gt, int dtype, int batch, int s, int c,
                            const float* mask, int S, const float* coef, int idx, int relu, int l1, void* stream) {
  IMM_REQUIRE(da && a_pred && a_gt && coef && batch > 0 && s > 0 && c > 0 && c % 8 == 0 && idx >= 0, "tap_grad: args");
  IMM_REQUIRE(mask == nullptr || (S >= s && S % s == 0), "tap_grad: mask side");
  const int64_t total = (int64_t)batch * s * s * (c / 8);
  int64_t blocks = (total + LO_THREADS - 1) / LO_THREADS;
  if (blocks > 16384) blocks = 16384;
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((tap_grad_kernel<ET>), dim3((int)blocks), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, (typename ET::T*)da, has_in, (const typename ET::T*)a_pred,
                                               (const typename ET::T*)a_gt, batch, s, c / 8, mask, S, coef, idx, relu, l1));
  IMM_CHECK_LAUNCH("imm_tap_grad");
  return 0;
}

// Gradient of an image-space loss term w.r.t. the prediction, when no VGG feature is tapped (reconstruction_loss: 'l2',
// imm_model.py:385-387, or perceptual.comp == ['input']): dpred[p][ch] = coef[idx] * mask[p] * (pred - gt) (sign(.) with
// l1) for ch < 3, zero for the other lddp - 3 channels of the 16-bit gradient image.
template <typename ET>
__global__ void image_loss_grad_kernel(const float* __restrict__ gt, const float* __restrict__ pred, int ldp, int64_t npix,
                                       const float* __restrict__ mask, const float* __restrict__ coef, int idx, int l1,
                                       typename ET::T* __restrict__ dpred, int lddp) {
  const int nvec = lddp / 8;
  const float ck = coef[idx];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / nvec;
    const int cg = (int)(i - p * nvec);
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cg == 0) {
      const float cm = ck * (mask ? mask[p] : 1.f);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float d = pred[p * ldp + ch] - gt[p * 3 + ch];
        if (l1) d = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        o[ch] = cm * d;
      }
    }
    st8<ET>(dpred + p * lddp + cg * 8, pack8<ET>(o));
  }
}

extern "C" int imm_image_loss_grad(const float* gt, const float* pred, int ldp, int batch, int s, const float* mask,
                                   const float* coef, int idx, int l1, void* dpred, int lddp, int dtype, void* stream) {
  IMM_REQUIRE(gt && pred && coef && dpred && batch > 0 && s > 0 && ldp >= 3 && idx >= 0, "image_loss_grad: args");
  IMM_REQUIRE(lddp >= 8 && lddp % 8 == 0, "image_loss_grad: lddp=%d must be a multiple of 8", lddp);
  const int64_t npix = (int64_t)batch * s * s;
  int64_t blocks = (npix * (lddp / 8) + LO_THREADS - 1) / LO_THREADS;
  if (blocks > 8192) blocks = 8192;
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((image_loss_grad_kernel<ET>), dim3((int)blocks), dim3(LO_THREADS), 0,
                                               (hipStream_t)stream, gt, pred, ldp, npix, mask, coef, idx, l1,
                                               (typename ET::T*)dpred, lddp));
  IMM_CHECK_LAUNCH("imm_image_loss_grad");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// optimizer: flat f32 buffers, block table (one block = one chunk of one tensor)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LO_THREADS) void wd_partial_kernel(const float* __restrict__ params,
                                                                const int32_t* __restrict__ blk_seg,
                                                                const int32_t* __restrict__ blk_begin,
                                                                const int32_t* __restrict__ blk_end,
                                                                const float* __restrict__ seg_wd,
                                                                float* __restrict__ blk_partial) {
  __shared__ float red[4];
  const int blk = blockIdx.x;
  const float wd = seg_wd[blk_seg[blk]];
  float acc = 0.f;
  if (wd != 0.f)
    for (int i = blk_begin[blk] + threadIdx.x; i < blk_end[blk]; i += LO_THREADS) { const float w = params[i]; acc += w * w; }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) blk_partial[blk] = 0.5f * wd * acc;
}

__global__ __launch_bounds__(LO_THREADS) void sum_partials_kernel(const float* __restrict__ partial, int n, float* out) {
  __shared__ double dred[LO_THREADS];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += LO_THREADS) v += (double)partial[i];
  dred[threadIdx.x] = v;
  __syncthreads();
  for (int o = LO_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) dred[threadIdx.x] += dred[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)dred[0];
}

extern "C" int imm_weight_decay_loss(const float* params, const int32_t* blk_seg, const int32_t* blk_begin,
                                     const int32_t* blk_end, int nblk, const float* seg_wd, float* blk_partial, float* out,
                                     void* stream) {
  IMM_REQUIRE(params && blk_seg && blk_begin && blk_end && seg_wd && blk_partial && out && nblk > 0, "weight_decay_loss: args");
  hipLaunchKernelGGL(wd_partial_kernel, dim3(nblk), dim3(LO_THREADS), 0, (hipStream_t)stream, params, blk_seg, blk_begin,
                     blk_end, seg_wd, blk_partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(LO_THREADS), 0, (hipStream_t)stream, blk_partial, nblk, out);
  IMM_CHECK_LAUNCH("imm_weight_decay_loss");
  return 0;
}

// ---------------------------------------------------------------------------------------------
// summaries (round 5): the reference's cost moving averages and per-VGG-layer activation scale
// ---------------------------------------------------------------------------------------------
// BaseModel._add_cost_summary (base_model.py:52-60): tf.train.ExponentialMovingAverage(0.99).apply([cost]) for reconstruction_loss,
// weights_loss and loss_total, run with every training step (avg_ops, cnn_train_multi.py:61-62).  For a TENSOR the shadow value
// starts at 0: shadow <- decay * shadow + (1 - decay) * cost, and — tensorflow 1.10, zero_debias=False — the value summarised is
// the shadow itself.  state = {shadow[3], local_step} (the step count is bookkeeping only).
__global__ void cost_ema_kernel(const float* __restrict__ cost3, float* __restrict__ state, float decay) {
  if (threadIdx.x < 3) state[threadIdx.x] = decay * state[threadIdx.x] + (1.f - decay) * cost3[threadIdx.x];
  if (threadIdx.x == 3) state[3] += 1.f;
}

extern "C" int imm_cost_ema(const float* cost3, float* state4, float decay, void* stream) {
  IMM_REQUIRE(cost3 && state4 && decay >= 0.f && decay < 1.f, "cost_ema: args");
  hipLaunchKernelGGL(cost_ema_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, cost3, state4, decay);
  IMM_CHECK_LAUNCH("imm_cost_ema");
  return 0;
}

// selfsup/vgg16.py:232-234: tf.summary.scalar('activation/<layer>', sqrt(reduce_mean(z^2))) of every VGG layer's output.
template <typename ET>
__global__ __launch_bounds__(LO_THREADS) void sumsq16_kernel(const typename ET::T* __restrict__ x, int64_t n8, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * LO_THREADS + threadIdx.x; i < n8; i += (int64_t)gridDim.x * LO_THREADS) {
    float f[8];
    unpack8<ET>(ld8<ET>(x + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(f[e], f[e], acc);
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

__global__ __launch_bounds__(LO_THREADS) void rms_finish_kernel(const float* __restrict__ partial, int n, double count, float* out) {
  __shared__ double dred[LO_THREADS];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += LO_THREADS) v += (double)partial[i];
  dred[threadIdx.x] = v;
  __syncthreads();
  for (int o = LO_THREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) dred[threadIdx.x] += dred[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sqrt(dred[0] / count);
}

extern "C" int imm_rms16(const void* x, int64_t n, int dtype, float* partial, int nblk, float* out, void* stream) {
  IMM_REQUIRE(x && partial && out && n > 0 && n % 8 == 0 && nblk > 0 && nblk <= 4096, "rms16: args (n must be a multiple of 8)");
  IMM_REQUIRE(((uintptr_t)x % 16) == 0, "rms16: 16-byte alignment");
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((sumsq16_kernel<ET>), dim3(nblk), dim3(LO_THREADS), 0, (hipStream_t)stream,
                                              (const typename ET::T*)x, n / 8, partial));
  hipLaunchKernelGGL(rms_finish_kernel, dim3(1), dim3(LO_THREADS), 0, (hipStream_t)stream, partial, nblk, (double)n, out);
  IMM_CHECK_LAUNCH("imm_rms16");
  return 0;
}

// The optimizer step is TWO launches (round 3; three before, with a one-workgroup "tick" between them whose threads walked their
// tensor's chunk sums serially: 12 us of latency chain on the strictly serial tail of the step):
//   grad_prepare: g <- g * grad_scale / S + wd * w and the chunk sums of g^2; one extra workgroup computes the step's learning
//                 rate (three f64 pow: microseconds on one thread — once, not per workgroup of the update: measured +45 us);
//   clip_adam:    EVERY workgroup rebuilds what it needs from the chunk sums (a few KB from L2): its tensor's squared norm (f64,
//                 fixed order) and the skip decision (any chunk sum NaN / inf / absurd), then updates its chunk; one extra
//                 workgroup advances the state nobody in this launch reads (counters, loss scale).
// No atomics: a ticket per workgroup (tried first) serialises on its one address, ~30 ns each — 2000 workgroups, +60 us; agent-scope
// loads of a shared flag bypass the L2 and queue up at one memory channel the same way.
// learning rate of this step: staircase decay on TF's global_step, Adam's bias correction on its own update count
__device__ __forceinline__ void opt_learning_rate(const imm_opt_hparams& hp, int gs, int tt, float* lr_t, float* lr) {
  const double l = (double)hp.lr_multiple * (double)hp.lr_start * pow((double)hp.lr_decay, (double)(gs / hp.lr_step));
  *lr_t = (float)(l * sqrt(1.0 - pow((double)hp.beta2, (double)tt)) / (1.0 - pow((double)hp.beta1, (double)tt)));
  *lr = (float)l;
}

__global__ __launch_bounds__(LO_THREADS) void grad_prepare_kernel(const float* __restrict__ params, float* __restrict__ grads,
                                                                  const int32_t* __restrict__ blk_seg,
                                                                  const int32_t* __restrict__ blk_begin,
                                                                  const int32_t* __restrict__ blk_end,
                                                                  const float* __restrict__ seg_wd, float grad_scale,
                                                                  const float* __restrict__ loss_scale,
                                                                  float* __restrict__ blk_partial, int nblk,
                                                                  const int32_t* __restrict__ step_count,
                                                                  const int32_t* __restrict__ adam_t, float* __restrict__ lr_state,
                                                                  imm_opt_hparams hp) {
  __shared__ float red[4];
  const int blk = blockIdx.x;
  if (blk == nblk) {           // the extra workgroup: this step's learning rate (the counters move in the update launch)
    if (threadIdx.x == 0) opt_learning_rate(hp, step_count[0], adam_t[0] + 1, lr_state, lr_state + 1);
    return;
  }
  if (loss_scale) grad_scale *= 1.f / loss_scale[0];     // S is a power of two: exact
  const float wd = seg_wd[blk_seg[blk]];
  float acc = 0.f;
  // 16-byte accesses over the aligned body of the chunk, scalars on the (<= 3 element) head and tail
  const int b0 = blk_begin[blk], b1 = blk_end[blk];
  const int a0 = min((b0 + 3) & ~3, b1), a1 = max(a0, b1 & ~3);
  for (int i = b0 + threadIdx.x; i < a0; i += LO_THREADS) {
    const float g = grads[i] * grad_scale + wd * params[i];
    grads[i] = g; acc += g * g;
  }
  for (int i = a0 + 4 * threadIdx.x; i < a1; i += 4 * LO_THREADS) {
    float4 g = *(const float4*)(grads + i);
    const float4 w = *(const float4*)(params + i);
    g.x = g.x * grad_scale + wd * w.x; g.y = g.y * grad_scale + wd * w.y;
    g.z = g.z * grad_scale + wd * w.z; g.w = g.w * grad_scale + wd * w.w;
    *(float4*)(grads + i) = g;
    acc += (g.x * g.x + g.y * g.y) + (g.z * g.z + g.w * g.w);
  }
  for (int i = a1 + threadIdx.x; i < b1; i += LO_THREADS) {
    const float g = grads[i] * grad_scale + wd * params[i];
    grads[i] = g; acc += g * g;
  }
  acc = block_sum_256(acc, red);
  if (threadIdx.x == 0) blk_partial[blk] = acc;
}

__global__ __launch_bounds__(LO_THREADS) void clip_adam_kernel(float* __restrict__ params, const float* __restrict__ grads,
                                                               float* __restrict__ m, float* __restrict__ v,
                                                               const int32_t* __restrict__ blk_seg,
                                                               const int32_t* __restrict__ blk_begin,
                                                               const int32_t* __restrict__ blk_end,
                                                               const int32_t* __restrict__ seg_first_blk,
                                                               const float* __restrict__ blk_partial, int nblk,
                                                               float* __restrict__ seg_norm2, int32_t* __restrict__ step_count,
                                                               int32_t* __restrict__ adam_t, const float* __restrict__ lr_state,
                                                               float* __restrict__ ls, imm_opt_hparams hp) {
  __shared__ double dred[LO_THREADS / 64];
  const int blk = blockIdx.x, tid = threadIdx.x;
  const bool extra = blk == nblk;                         // the workgroup that advances the state
  const int seg = extra ? 0 : blk_seg[blk];
  const int b0 = extra ? 0 : blk_begin[blk], b1 = extra ? 0 : blk_end[blk];
  const int a0 = min((b0 + 3) & ~3, b1), a1 = max(a0, b1 & ~3);
  // the first 16-byte pass of this chunk is requested before the prologue below (its latency hides the prologue's)
  const int i0 = a0 + 4 * tid;
  const bool first = i0 < a1;
  float4 g0 = {0.f, 0.f, 0.f, 0.f}, m0 = g0, v0 = g0, w0 = g0;
  if (first) { g0 = *(const float4*)(grads + i0); m0 = *(const float4*)(m + i0); v0 = *(const float4*)(v + i0); w0 = *(const float4*)(params + i0); }

  // ---- prologue: one pass over ALL chunk sums: not-a-sane-number anywhere => skip; this tensor's chunks => its squared norm ----
  const int f0 = seg_first_blk[seg], f1 = seg_first_blk[seg + 1];
  double ps = 0.0;
  int bad = 0;
  // Without a loss scale nothing reads the skip flag, and a workgroup only needs ITS tensor's chunks [f0, f1): the scan then costs
  // O(chunks of the tensor) instead of O(all chunks) per workgroup (16 MB of L2 reads per step at 2 000 chunks, growing with the
  // square of the parameter count).  Thread t still takes the chunks b = t (mod LO_THREADS), so the sums — and their order — are
  // the ones of the full scan, bit for bit.  With a loss scale (f16 storage) the decision "any chunk not finite" is global: full scan.
  const int sb1 = ls != nullptr ? nblk : f1;
  int sb0 = tid;
  if (ls == nullptr && f0 > tid) sb0 = tid + (f0 - tid + LO_THREADS - 1) / LO_THREADS * LO_THREADS;
  for (int b = sb0; b < sb1; b += LO_THREADS) {
    const float p = blk_partial[b];
    // NaN, inf, or so large that a tensor's sum of chunk sums could leave the f32 range: an f16 gradient overflowed upstream
    bad |= !(p <= IMM_NORM2_SANE);
    if (b >= f0 && b < f1) ps += (double)p;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
  if ((tid & 63) == 0) dred[tid >> 6] = ps;
  bad = __syncthreads_or(bad);
  const bool skip = ls != nullptr && bad != 0;
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < LO_THREADS / 64; ++w) t += dred[w];
  const float norm2 = (float)t;
  if (tid == 0 && !extra && blk == f0) seg_norm2[seg] = norm2;
  if (extra) {
    if (tid == 0) {
      if (ls) {
        float S = ls[0], clean = ls[1];
        if (skip) { S = fmaxf(S * 0.5f, 1.f); clean = 0.f; ls[2] += 1.f; }
        else if (hp.scale_growth_interval > 0 && clean + 1.f >= (float)hp.scale_growth_interval) {
          S = fminf(S * 2.f, hp.scale_max > 0.f ? hp.scale_max : S * 2.f); clean = 0.f;
        } else clean += 1.f;
        ls[0] = S; ls[1] = clean; ls[3] = skip ? 1.f : 0.f;
      }
      if (!skip) {                          // a skipped step leaves the weights, the slots and both counters untouched
        step_count[0] += 1;                 // TF global_step: learning-rate schedule only
        adam_t[0] += 1;                     // Adam's t (TF: beta1_power / beta2_power, independent of global_step)
      }
    }
    return;
  }
  if (skip) return;                         // overflow in this step's gradients: no update
  float factor = 1.f;
  if (hp.clip > 0.f) factor = hp.clip / fmaxf(sqrtf(norm2), hp.clip);
  const float lr_t = lr_state[0], lr = lr_state[1];
  auto one = [&](float g, float& mi, float& vi, float& w) {
    g *= factor;
    if (hp.optim == IMM_OPT_ADAM) {
      mi = hp.beta1 * mi + (1.f - hp.beta1) * g;
      vi = hp.beta2 * vi + (1.f - hp.beta2) * g * g;
      w -= lr_t * mi / (sqrtf(vi) + hp.eps);
    } else if (hp.optim == IMM_OPT_ADADELTA) {          // rho = beta1; vi = accum, mi = accum_update
      vi = hp.beta1 * vi + (1.f - hp.beta1) * g * g;
      const float u = sqrtf(mi + hp.eps) / sqrtf(vi + hp.eps) * g;
      mi = hp.beta1 * mi + (1.f - hp.beta1) * u * u;
      w -= lr * u;
    } else {                                            // Adagrad: vi = accumulator
      vi += g * g;
      w -= lr * g / sqrtf(vi);
    }
  };
  for (int i = b0 + tid; i < a0; i += LO_THREADS) one(grads[i], m[i], v[i], params[i]);
  if (first) {
    one(g0.x, m0.x, v0.x, w0.x); one(g0.y, m0.y, v0.y, w0.y); one(g0.z, m0.z, v0.z, w0.z); one(g0.w, m0.w, v0.w, w0.w);
    *(float4*)(m + i0) = m0; *(float4*)(v + i0) = v0; *(float4*)(params + i0) = w0;
  }
  for (int i = i0 + 4 * LO_THREADS; i < a1; i += 4 * LO_THREADS) {
    const float4 g = *(const float4*)(grads + i);
    float4 mi = *(const float4*)(m + i), vi = *(const float4*)(v + i), w = *(const float4*)(params + i);
    one(g.x, mi.x, vi.x, w.x); one(g.y, mi.y, vi.y, w.y); one(g.z, mi.z, vi.z, w.z); one(g.w, mi.w, vi.w, w.w);
    *(float4*)(m + i) = mi; *(float4*)(v + i) = vi; *(float4*)(params + i) = w;
  }
  for (int i = a1 + tid; i < b1; i += LO_THREADS) one(grads[i], m[i], v[i], params[i]);
}

extern "C" int imm_clip_adam_step(float* params, float* grads, float* m, float* v, const int32_t* blk_seg,
                                  const int32_t* blk_begin, const int32_t* blk_end, int nblk, int nseg,
                                  const int32_t* seg_first_blk, const float* seg_wd, float* blk_partial, float* seg_norm2,
                                  int32_t* step_count, int32_t* adam_t, float* lr_state, const imm_opt_hparams* hp,
                                  float* loss_scale_state, void* stream) {
  IMM_REQUIRE(params && grads && m && v && blk_seg && blk_begin && blk_end && seg_first_blk && seg_wd && blk_partial &&
                  seg_norm2 && step_count && adam_t && lr_state && hp, "clip_adam_step: null");
  IMM_REQUIRE(nblk > 0 && nseg > 0 && hp->lr_step > 0, "clip_adam_step: dims");
  IMM_REQUIRE(hp->optim >= IMM_OPT_ADAM && hp->optim <= IMM_OPT_ADAGRAD, "clip_adam_step: optim %d", hp->optim);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(grad_prepare_kernel, dim3(nblk + 1), dim3(LO_THREADS), 0, s, params, grads, blk_seg, blk_begin, blk_end,
                     seg_wd, hp->grad_scale, loss_scale_state, blk_partial, nblk, step_count, adam_t, lr_state, *hp);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(nblk + 1), dim3(LO_THREADS), 0, s, params, grads, m, v, blk_seg, blk_begin, blk_end,
                     seg_first_blk, blk_partial, nblk, seg_norm2, step_count, adam_t, lr_state, loss_scale_state, *hp);
  IMM_CHECK_LAUNCH("imm_clip_adam_step");
  return 0;
}
