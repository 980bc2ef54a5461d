// bottleneck.hip — IMM landmark bottleneck on gfx950: K-channel spatial soft-argmax -> (y,x) landmarks
// -> isotropic Gaussian maps, forward and backward, one workgroup per sample.
//
// Reference: imm/models/imm_model.py:252-264 (reduce_mean over the other axis, softmax over this
// axis, expectation against linspace(-1,1,n)) and :34-78 get_gaussian_maps mode 'rot'
// (exp(-((y-mu_y)^2+(x-mu_x)^2)*inv_std^2), the mode every shipped config uses,
// configs/experiments/celeba-10pts.yaml:27), 'flat' (:61, exp(-(d + 1e-5)^(1/4))) and 'ankush' (:63-72, the outer
// product of exp(-sqrt(1e-4 + |mu - coord| * inv_std)) along y and x): gauss_mode 0 / 1 / 2 (IMM_GAUSS_*).
//
// The whole heat-map of a sample (h*w*K f32: 10 KB at 16x16xK=10, 120 KB at 32x32x30) is staged in
// the CU's 160 KB LDS once; row/column means, the two softmaxes and the render then run out of LDS.
// HBM traffic per sample = the heat-map read + mu/probabilities + the s*s*K 16-bit map written
// straight into the renderer's concat buffer (channel offset given by the pointer, stride ldg).
#include "common.h"

#define BT_THREADS 256

__device__ __forceinline__ float lin_pm1(int i, int n) { return n > 1 ? -1.f + 2.f * (float)i / (float)(n - 1) : -1.f; }

// Gaussian-like map value at offset (dy, dx) = (coord - mu) and its derivative w.r.t. (mu_y, mu_x)
// (imm_model.py:48-72; d|u|/du = sign(u) with sign(0) = 0, as TF / autograd define it).
__device__ __forceinline__ float gauss_value(int mode, float dy, float dx, float inv_std) {
  const float i2 = inv_std * inv_std;
  if (mode == IMM_GAUSS_ROT) return expf(-(dy * dy + dx * dx) * i2);
  if (mode == IMM_GAUSS_FLAT) return expf(-powf((dy * dy + dx * dx) * i2 + 1e-5f, 0.25f));
  return expf(-sqrtf(1e-4f + fabsf(dy) * inv_std)) * expf(-sqrtf(1e-4f + fabsf(dx) * inv_std));
}
__device__ __forceinline__ void gauss_grad(int mode, float dy, float dx, float inv_std, float& g, float& dg_dmy, float& dg_dmx) {
  const float i2 = inv_std * inv_std;
  if (mode == IMM_GAUSS_ROT) {
    g = expf(-(dy * dy + dx * dx) * i2);
    dg_dmy = g * 2.f * i2 * dy; dg_dmx = g * 2.f * i2 * dx;
  } else if (mode == IMM_GAUSS_FLAT) {
    const float d = (dy * dy + dx * dx) * i2 + 1e-5f;
    g = expf(-powf(d, 0.25f));
    const float c = g * 0.25f * powf(d, -0.75f) * 2.f * i2;       // -dg/dd * dd/d(coord - mu)
    dg_dmy = c * dy; dg_dmx = c * dx;
  } else {
    const float ry = sqrtf(1e-4f + fabsf(dy) * inv_std), rx = sqrtf(1e-4f + fabsf(dx) * inv_std);
    const float gy = expf(-ry), gx = expf(-rx);
    g = gy * gx;
    const float sy = dy > 0.f ? 1.f : (dy < 0.f ? -1.f : 0.f), sx = dx > 0.f ? 1.f : (dx < 0.f ? -1.f : 0.f);
    dg_dmy = gx * gy * inv_std * sy / (2.f * ry);
    dg_dmx = gy * gx * inv_std * sx / (2.f * rx);
  }
}

// means over the other axis, softmax + expectation, render — from the heat-map of sample b staged in LDS (sm: [h*w][K] heat |
// [h][K] | [w][K] | [K][2])
// GM >= 0: the Gaussian mode as a compile-time constant (the launches of the bench path: 'rot'); -1: the runtime argument.  The
// three modes' expf / powf / sqrtf sequences inlined into unrolled loops were most of these kernels' 10-35 KB of code, and a
// few dozen workgroups walking that once through a cold instruction cache is latency on the pose lane's critical path.
template <typename ET, int NTHR, int GM = -1>
__device__ __forceinline__ void bt_softargmax_tail(float* sm, int b, int tid, int h, int w, int K, float inv_std, int s,
                                                   float* __restrict__ mu, float* __restrict__ py, float* __restrict__ px,
                                                   typename ET::T* __restrict__ gauss, int ldg, int mode) {
  float* sheat = sm;                      // [h*w][K]
  float* rmean = sheat + h * w * K;       // [h][K] row means -> probabilities
  float* cmean = rmean + h * K;           // [w][K]
  float* smu = cmean + w * K;             // [K][2]
  // means over the other axis (imm_model.py:254)
  for (int i = tid; i < (h + w) * K; i += NTHR) {
    if (i < h * K) {
      const int r = i / K, k = i - r * K;
      float acc = 0.f;
      for (int x = 0; x < w; ++x) acc += sheat[(r * w + x) * K + k];
      rmean[i] = acc / (float)w;
    } else {
      const int ii = i - h * K;
      const int cidx = ii / K, k = ii - cidx * K;
      float acc = 0.f;
      for (int y = 0; y < h; ++y) acc += sheat[(y * w + cidx) * K + k];
      cmean[ii] = acc / (float)h;
    }
  }
  __syncthreads();
  // softmax + expectation, one thread per (axis, k)  (imm_model.py:255-258)
  for (int i = tid; i < 2 * K; i += NTHR) {
    const int axis = i / K, k = i - axis * K;
    float* v = axis == 0 ? rmean : cmean;
    const int n = axis == 0 ? h : w;
    float mx = -INFINITY;
    for (int j = 0; j < n; ++j) mx = fmaxf(mx, v[j * K + k]);
    float den = 0.f;
    for (int j = 0; j < n; ++j) { const float e = expf(v[j * K + k] - mx); v[j * K + k] = e; den += e; }
    float ex = 0.f;
    float* pout = (axis == 0 ? py : px) + (int64_t)b * n * K;
    for (int j = 0; j < n; ++j) {
      const float p = v[j * K + k] / den;
      pout[j * K + k] = p;
      ex += p * lin_pm1(j, n);
    }
    smu[k * 2 + axis] = ex;
    mu[((int64_t)b * K + k) * 2 + axis] = ex;
  }
  __syncthreads();
  // render (imm_model.py:48-59, transposed to NHWC at :77)
  if (gauss != nullptr) {
    for (int i = tid; i < s * s * K; i += NTHR) {
      const int p = i / K, k = i - p * K;
      const int yy = p / s, xx = p - yy * s;
      const float dy = lin_pm1(yy, s) - smu[k * 2], dx = lin_pm1(xx, s) - smu[k * 2 + 1];
      gauss[((int64_t)b * s * s + p) * ldg + k] = ET::from_f32(gauss_value(GM >= 0 ? GM : mode, dy, dx, inv_std));
    }
  }
}

template <typename ET>
__global__ __launch_bounds__(BT_THREADS) void softargmax_gauss_fwd_kernel(
    const float* __restrict__ heat, int ldh, int h, int w, int K, float inv_std, int s, float* __restrict__ mu,
    float* __restrict__ py, float* __restrict__ px, typename ET::T* __restrict__ gauss, int ldg, int mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* hb = heat + (int64_t)b * h * w * ldh;
  for (int i = tid; i < h * w * K; i += BT_THREADS) {
    const int p = i / K, k = i - p * K;
    sm[i] = hb[(int64_t)p * ldh + k];
  }
  __syncthreads();
  bt_softargmax_tail<ET, BT_THREADS>(sm, b, tid, h, w, K, inv_std, s, mu, py, px, gauss, ldg, mode);
}

// The pose head in ONE launch (imm_model.py:247-264): 1x1 convolution C -> K (+ bias; no batch norm, no activation) by MFMA
// straight from global memory — both operands are K-contiguous as they lie (a pixel's channels; a packed filter row), so a lane
// loads its 8-value MFMA operand with one 16-byte load and no LDS staging is needed — into the LDS heat-map, then the soft-argmax
// and the render as above.  One workgroup of 16 waves per sample (a 16x16 heat-map = one 16-pixel tile per wave: every load of
// the convolution is in flight at once; with 4 waves the four tiles of a wave were a chain of L2 latencies, 29 us per launch);
// wave v takes the tiles v, v+16, ...
#define PH_THREADS 1024           // forward: 16 waves
#define PH_BWD_THREADS 512        // backward: 8 waves (the filter rows of the data gradient take 64-128 registers per lane)
template <typename ET, int GM = -1>
__global__ __launch_bounds__(PH_THREADS) void pose_head_fwd_kernel(
    const uint16_t* __restrict__ feat, int ldf, int C, const uint16_t* __restrict__ wt, int kpad, const float* __restrict__ bias,
    float* __restrict__ heat, int ldh, int h, int w, int K, float inv_std, int s, float* __restrict__ mu,
    float* __restrict__ py, float* __restrict__ px, uint16_t* __restrict__ gauss, int ldg, int mode) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int hw = h * w, NT = (K + 15) >> 4, KK = C >> 5;
  const int lm = lane & 15, lk = (lane >> 4) * 8;
  for (int m0 = wv * 16; m0 < hw; m0 += PH_THREADS / 4) {
    f32x4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint16_t* ap = feat + ((int64_t)b * hw + m0 + lm) * ldf + lk;
    const uint16_t* bp = wt + (int64_t)lm * kpad + lk;
    for (int k0 = 0; k0 < KK; k0 += 8) {                 // chunks of 8 k-steps (256 channels): all loads of a chunk in flight
      uint4 af[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (k0 + q < KK) af[q] = *(const uint4*)(ap + (k0 + q) * 32);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < NT) {
          uint4 bf[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (k0 + q < KK) bf[q] = *(const uint4*)(bp + (int64_t)j * 16 * kpad + (k0 + q) * 32);
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (k0 + q < KK) acc[j] = ET::mfma(bf[q], af[q], acc[j]);           // D[n = 4 (lane >> 4) + r][m = lane & 15]
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = j * 16 + 4 * (lane >> 4) + r;
          if (n < K) {
            const float v = acc[j][r] + bias[n];
            sm[(m0 + lm) * K + n] = v;
            heat[((int64_t)b * hw + m0 + lm) * ldh + n] = v;
          }
        }
      }
  }
  __syncthreads();
  bt_softargmax_tail<ET, PH_THREADS, GM>(sm, b, tid, h, w, K, inv_std, s, mu, py, px, gauss, ldg, mode);
}

// backward: dG -> dmu (through the Gaussian) -> d row/col means (through softmax-expectation) -> dheat
// HEAD: the same pass continued through the pose head's 1x1 convolution (imm_model.py:247-248) — the 16-bit heat-map gradient
// stays in LDS as the MFMA operand of the data gradient d_feat[p][c] = sum_k dheat[p][k] W[c][k] (packed filter rows straight
// from global memory), and its per-sample column sums (the bias gradient's partial row) are written for the final table-driven
// reduction: one launch instead of bottleneck backward + column sum + 1x1 data gradient.
template <typename ET, bool HEAD, int GM = -1>
__global__ __launch_bounds__(HEAD ? PH_BWD_THREADS : BT_THREADS) void softargmax_gauss_bwd_kernel(
    const typename ET::T* __restrict__ dgauss, int ldg, int h, int w, int K, float inv_std, int s,
    const float* __restrict__ mu, const float* __restrict__ py, const float* __restrict__ px,
    typename ET::T* __restrict__ dheat, int lddh, int mode, const uint16_t* __restrict__ wtd, int kpad_d, int C,
    uint16_t* __restrict__ dfeat, int lddf, float* __restrict__ bias_partial) {
  constexpr int NTHR = HEAD ? PH_BWD_THREADS : BT_THREADS;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* dmu = sm;              // [K][2]
  float* drow = dmu + 2 * K;    // [h][K]  d loss / d row-mean
  float* dcol = drow + h * K;   // [w][K]
  constexpr int KC = 10;                       // landmarks per pass: 2*KC partial sums per thread, reduced together
  __shared__ float red[NTHR / 64][2 * KC];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // dmu[k][axis] = sum_p dG * dG/dmu  ('rot': G*2*inv_std^2*(coord - mu)).  A thread walks its pixels once per pass and
  // keeps the sums of KC landmarks (their dG values are adjacent in memory); the 2*KC sums then go through ONE reduction
  // (wave butterflies + a 4-row LDS stage) instead of two block reductions per landmark (20 for K = 10: most of the
  // kernel's 22 us on the critical path of the pose encoder's backward).
  for (int k0 = 0; k0 < K; k0 += KC) {
    float ay[KC], ax[KC], my[KC], mx[KC];
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      ay[c] = 0.f; ax[c] = 0.f;
      const int k = k0 + c < K ? k0 + c : K - 1;
      my[c] = mu[((int64_t)b * K + k) * 2]; mx[c] = mu[((int64_t)b * K + k) * 2 + 1];
    }
    for (int p = tid; p < s * s; p += NTHR) {
      const int yy = p / s, xx = p - yy * s;
      const float ly = lin_pm1(yy, s), lx = lin_pm1(xx, s);
      const typename ET::T* dgp = dgauss + ((int64_t)b * s * s + p) * ldg + k0;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (k0 + c < K) {
          float g, gmy, gmx;
          gauss_grad(GM >= 0 ? GM : mode, ly - my[c], lx - mx[c], inv_std, g, gmy, gmx);
          const float dg = ET::to_f32(dgp[c]);
          ay[c] += dg * gmy;
          ax[c] += dg * gmx;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < KC; ++c) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { ay[c] += __shfl_xor(ay[c], o, 64); ax[c] += __shfl_xor(ax[c], o, 64); }
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < KC; ++c) { red[wv][2 * c] = ay[c]; red[wv][2 * c + 1] = ax[c]; }
    }
    __syncthreads();
    if (tid < 2 * KC && k0 + tid / 2 < K) {
      float t = 0.f;
#pragma unroll
      for (int wq = 0; wq < NTHR / 64; ++wq) t += red[wq][tid];
      dmu[(k0 + tid / 2) * 2 + (tid & 1)] = t;
    }
    __syncthreads();
  }
  // mu = sum_j p_j lin_j, p = softmax(r):  dr_j = p_j (lin_j - mu) dmu
  for (int i = tid; i < (h + w) * K; i += NTHR) {
    if (i < h * K) {
      const int j = i / K, k = i - j * K;
      const float p = py[(int64_t)b * h * K + i];
      drow[i] = p * (lin_pm1(j, h) - mu[((int64_t)b * K + k) * 2]) * dmu[k * 2];
    } else {
      const int ii = i - h * K;
      const int j = ii / K, k = ii - j * K;
      const float p = px[(int64_t)b * w * K + ii];
      dcol[ii] = p * (lin_pm1(j, w) - mu[((int64_t)b * K + k) * 2 + 1]) * dmu[k * 2 + 1];
    }
  }
  __syncthreads();
  // row mean = sum_x heat/w, col mean = sum_y heat/h
  // HEAD: after the floats (rounded to 16 bytes): [NTHR] column-sum scratch, then the [h*w][lddh] 16-bit copy of dheat
  float* cs = sm + (((2 + h + w) * K + 3) & ~3);
  uint16_t* sdh = (uint16_t*)(cs + NTHR);
  for (int i = tid; i < h * w * lddh; i += NTHR) {
    const int p = i / lddh, k = i - p * lddh;
    float v = 0.f;
    if (k < K) {
      const int yy = p / w, xx = p - yy * w;
      v = drow[yy * K + k] / (float)w + dcol[xx * K + k] / (float)h;
    }
    const typename ET::T q = ET::from_f32(v);
    // HEAD: gridDim.y workgroups per sample share the data gradient's output channels (round 5); each rebuilds the (tiny) heat-map
    // gradient for itself, the first one stores it and the bias partial row
    if (!HEAD || blockIdx.y == 0) dheat[(int64_t)b * h * w * lddh + i] = q;
    if constexpr (HEAD) sdh[i] = q;
  }
  if constexpr (HEAD) {
    __syncthreads();
    const int hw = h * w;
    // bias gradient partial of this sample: column sums of the STORED (16-bit) gradient, like imm_colsum; thread = (part, k)
    if (blockIdx.y == 0) {
      const int parts = NTHR / lddh;                // lddh in {32, 64}
      const int k = tid % lddh, part = tid / lddh;
      float acc = 0.f;
      for (int p = part; p < hw; p += parts) acc += ET::to_f32(sdh[p * lddh + k]);
      cs[part * lddh + k] = acc;
      __syncthreads();
      if (tid < K) {
        float t = 0.f;
        for (int q = 0; q < parts; ++q) t += cs[q * lddh + tid];
        bias_partial[(int64_t)b * K + tid] = t;
      }
    }
    // data gradient: 16-pixel tiles x 16-channel tiles, contraction over the lddh landmark channels (zeros beyond K)
    const int lm = lane & 15, lk = (lane >> 4) * 8, KS = lddh >> 5;
    // this workgroup's share of the C / 16 output-channel tiles (gridDim.y equal shares: 32 workgroups of one sample each were a
    // 31 us latency chain on the pose lane of the backward pass with 12 % of the CUs busy)
    const int nt_all = C >> 4, per = (nt_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int nt_lo = (int)blockIdx.y * per, NTc = min(nt_all, nt_lo + per);
    for (int n0 = nt_lo; n0 < NTc; n0 += 16) {            // <= 256 output channels at a time: their filter rows live in registers
      uint4 bfr[2][16];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (ks < KS && n0 + q < NTc) bfr[ks][q] = *(const uint4*)(wtd + ((int64_t)(n0 + q) * 16 + lm) * kpad_d + ks * 32 + lk);
      for (int m0 = wv * 16; m0 < hw; m0 += NTHR / 4) {
        uint4 af[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          if (ks < KS) af[ks] = *(const uint4*)(sdh + (m0 + lm) * lddh + ks * 32 + lk);
        uint16_t* op = dfeat + ((int64_t)b * hw + m0 + lm) * lddf + n0 * 16 + 4 * (lane >> 4);
#pragma unroll
        for (int q = 0; q < 16; ++q)
          if (n0 + q < NTc) {
            f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
              if (ks < KS) acc = ET::mfma(bfr[ks][q], af[ks], acc);               // D[c = 4 (lane >> 4) + r][m = lane & 15]
            *(uint2*)(op + q * 16) = make_uint2(ET::pack2(acc[0], acc[1]), ET::pack2(acc[2], acc[3]));
          }
      }
    }
  }
}

__global__ void gauss_render_f32_kernel(const float* __restrict__ mu, int K, float inv_std, int s, float* __restrict__ out,
                                        int64_t total, int mode) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx % K);
    int64_t t = idx / K;
    const int xx = (int)(t % s); t /= s;
    const int yy = (int)(t % s);
    const int64_t b = t / s;
    const float dy = lin_pm1(yy, s) - mu[(b * K + k) * 2], dx = lin_pm1(xx, s) - mu[(b * K + k) * 2 + 1];
    out[idx] = gauss_value(mode, dy, dx, inv_std);
  }
}

static const size_t kMaxDynLds = 160 * 1024 - 1024;

template <typename K_>
static int set_dyn_lds(K_ kernel, size_t bytes) {
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return imm_fail(IMM_E_HIP, "hipFuncSetAttribute(dyn LDS %zu): %s", bytes, hipGetErrorString(e));
  }
  return 0;
}

extern "C" int imm_softargmax_gauss_fwd(const float* heat, int ldh, int batch, int h, int w, int k, float inv_std, int s,
                                        float* mu, float* py, float* px, void* gauss_out, int ldg, int dtype,
                                        int gauss_mode, void* stream) {
  IMM_REQUIRE(heat && mu && py && px, "softargmax_fwd: null");
  IMM_REQUIRE(gauss_mode >= IMM_GAUSS_ROT && gauss_mode <= IMM_GAUSS_ANKUSH, "softargmax_fwd: gauss_mode %d", gauss_mode);
  IMM_REQUIRE(batch > 0 && h > 0 && w > 0 && k > 0 && ldh >= k && s > 0, "softargmax_fwd: dims");
  IMM_REQUIRE(gauss_out == nullptr || ldg >= k, "softargmax_fwd: ldg");
  const size_t lds = sizeof(float) * ((size_t)h * w * k + (size_t)(h + w) * k + 2 * (size_t)k);
  if (lds > kMaxDynLds) return imm_fail(IMM_E_UNSUPPORTED, "softargmax_fwd: heat-map %dx%dx%d needs %zu B LDS", h, w, k, lds);
  IMM_DISPATCH_DTYPE_F32(dtype, {
    if (set_dyn_lds(softargmax_gauss_fwd_kernel<ET>, lds)) return IMM_E_HIP;
    hipLaunchKernelGGL((softargmax_gauss_fwd_kernel<ET>), dim3(batch), dim3(BT_THREADS), lds, (hipStream_t)stream, heat,
                       ldh, h, w, k, inv_std, s, mu, py, px, (typename ET::T*)gauss_out, ldg, gauss_mode);
  });
  IMM_CHECK_LAUNCH("imm_softargmax_gauss_fwd");
  return 0;
}

extern "C" int imm_softargmax_gauss_bwd(const void* dgauss, int ldg, int dtype, int batch, int h, int w, int k,
                                        float inv_std, int s, const float* mu, const float* py, const float* px,
                                        void* dheat, int lddh, int gauss_mode, void* stream) {
  IMM_REQUIRE(dgauss && mu && py && px && dheat, "softargmax_bwd: null");
  IMM_REQUIRE(gauss_mode >= IMM_GAUSS_ROT && gauss_mode <= IMM_GAUSS_ANKUSH, "softargmax_bwd: gauss_mode %d", gauss_mode);
  IMM_REQUIRE(batch > 0 && h > 0 && w > 0 && k > 0 && ldg >= k && lddh >= k && s > 0, "softargmax_bwd: dims");
  const size_t lds = sizeof(float) * (2 * (size_t)k + (size_t)(h + w) * k);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((softargmax_gauss_bwd_kernel<ET, false>), dim3(batch), dim3(BT_THREADS), lds,
                                                   (hipStream_t)stream, (const typename ET::T*)dgauss, ldg, h, w, k, inv_std, s, mu,
                                                   py, px, (typename ET::T*)dheat, lddh, gauss_mode, (const uint16_t*)nullptr, 0, 0,
                                                   (uint16_t*)nullptr, 0, (float*)nullptr));
  IMM_CHECK_LAUNCH("imm_softargmax_gauss_bwd");
  return 0;
}

// ---- the pose head as one launch each way ---------------------------------------------------------------------------------
extern "C" int imm_pose_head_fwd(const void* feat, int ldf, int c, const void* wt, int kpad, const float* bias, int dtype,
                                 int batch, int h, int w, int k, float inv_std, int s, float* heat, int ldh, float* mu, float* py,
                                 float* px, void* gauss_out, int ldg, int gauss_mode, void* stream) {
  IMM_REQUIRE(feat && wt && bias && heat && mu && py && px, "pose_head_fwd: null");
  IMM_REQUIRE(gauss_mode >= IMM_GAUSS_ROT && gauss_mode <= IMM_GAUSS_ANKUSH, "pose_head_fwd: gauss_mode %d", gauss_mode);
  IMM_REQUIRE(batch > 0 && h > 0 && w > 0 && k > 0 && ldh >= k && s > 0, "pose_head_fwd: dims");
  IMM_REQUIRE(gauss_out == nullptr || ldg >= k, "pose_head_fwd: ldg");
  IMM_REQUIRE(((uintptr_t)feat % 16 == 0) && ((uintptr_t)wt % 16 == 0) && ldf % 8 == 0 && kpad % 8 == 0, "pose_head_fwd: alignment");
  if (c <= 0 || c % 32 || ldf < c || kpad < c || k > 64 || (h * w) % 16)
    return imm_fail(IMM_E_UNSUPPORTED, "pose_head_fwd: needs c %% 32 == 0, k <= 64, h*w %% 16 == 0 (c=%d k=%d h*w=%d)", c, k, h * w);
  const size_t lds = sizeof(float) * ((size_t)h * w * k + (size_t)(h + w) * k + 2 * (size_t)k);
  if (lds > kMaxDynLds) return imm_fail(IMM_E_UNSUPPORTED, "pose_head_fwd: heat-map %dx%dx%d needs %zu B LDS", h, w, k, lds);
  IMM_DISPATCH_DTYPE(dtype, {
    if (gauss_mode == IMM_GAUSS_ROT) {
      if (set_dyn_lds(pose_head_fwd_kernel<ET, IMM_GAUSS_ROT>, lds)) return IMM_E_HIP;
      hipLaunchKernelGGL((pose_head_fwd_kernel<ET, IMM_GAUSS_ROT>), dim3(batch), dim3(PH_THREADS), lds, (hipStream_t)stream,
                         (const uint16_t*)feat, ldf, c, (const uint16_t*)wt, kpad, bias, heat, ldh, h, w, k, inv_std, s, mu, py, px,
                         (uint16_t*)gauss_out, ldg, gauss_mode);
    } else {
      if (set_dyn_lds(pose_head_fwd_kernel<ET>, lds)) return IMM_E_HIP;
      hipLaunchKernelGGL((pose_head_fwd_kernel<ET>), dim3(batch), dim3(PH_THREADS), lds, (hipStream_t)stream, (const uint16_t*)feat,
                         ldf, c, (const uint16_t*)wt, kpad, bias, heat, ldh, h, w, k, inv_std, s, mu, py, px, (uint16_t*)gauss_out, ldg,
                         gauss_mode);
    }
  });
  IMM_CHECK_LAUNCH("imm_pose_head_fwd");
  return 0;
}

extern "C" int imm_pose_head_bwd(const void* dgauss, int ldg, int dtype, int batch, int h, int w, int k, float inv_std, int s,
                                 const float* mu, const float* py, const float* px, void* dheat, int lddh, int gauss_mode,
                                 const void* wt_dgrad, int kpad_d, int c, void* dfeat, int lddf, float* bias_partial, void* stream) {
  IMM_REQUIRE(dgauss && mu && py && px && dheat && wt_dgrad && dfeat && bias_partial, "pose_head_bwd: null");
  IMM_REQUIRE(gauss_mode >= IMM_GAUSS_ROT && gauss_mode <= IMM_GAUSS_ANKUSH, "pose_head_bwd: gauss_mode %d", gauss_mode);
  IMM_REQUIRE(batch > 0 && h > 0 && w > 0 && k > 0 && ldg >= k && lddh >= k && s > 0, "pose_head_bwd: dims");
  IMM_REQUIRE(((uintptr_t)wt_dgrad % 16 == 0) && ((uintptr_t)dfeat % 8 == 0) && lddf % 4 == 0 && kpad_d % 8 == 0, "pose_head_bwd: alignment");
  if ((lddh != 32 && lddh != 64) || kpad_d < lddh || c <= 0 || c % 16 || lddf < c || (h * w) % 16)
    return imm_fail(IMM_E_UNSUPPORTED, "pose_head_bwd: needs lddh in {32, 64}, c %% 16 == 0, h*w %% 16 == 0 (lddh=%d c=%d h*w=%d)",
                    lddh, c, h * w);
  // dmu | drow | dcol (floats), padded to 16 bytes, then the 16-bit heat-map gradient
  size_t fl = 2 * (size_t)k + (size_t)(h + w) * k;
  fl = (fl + 3) / 4 * 4 + PH_BWD_THREADS;                              // + the column-sum scratch
  const size_t lds = sizeof(float) * fl + (size_t)h * w * lddh * 2;
  if (lds > kMaxDynLds) return imm_fail(IMM_E_UNSUPPORTED, "pose_head_bwd: %dx%dx%d needs %zu B LDS", h, w, lddh, lds);
  IMM_DISPATCH_DTYPE(dtype, {
    // four workgroups per sample where that still leaves each at least two 16-channel tiles
    const int ysplit = (c >= 128 && batch <= 512) ? 4 : 1;
    if (gauss_mode == IMM_GAUSS_ROT) {
      if (set_dyn_lds(softargmax_gauss_bwd_kernel<ET, true, IMM_GAUSS_ROT>, lds)) return IMM_E_HIP;
      hipLaunchKernelGGL((softargmax_gauss_bwd_kernel<ET, true, IMM_GAUSS_ROT>), dim3(batch, ysplit), dim3(PH_BWD_THREADS), lds,
                         (hipStream_t)stream, (const uint16_t*)dgauss, ldg, h, w, k, inv_std, s, mu, py, px, (uint16_t*)dheat, lddh,
                         gauss_mode, (const uint16_t*)wt_dgrad, kpad_d, c, (uint16_t*)dfeat, lddf, bias_partial);
    } else {
      if (set_dyn_lds(softargmax_gauss_bwd_kernel<ET, true>, lds)) return IMM_E_HIP;
      hipLaunchKernelGGL((softargmax_gauss_bwd_kernel<ET, true>), dim3(batch, ysplit), dim3(PH_BWD_THREADS), lds, (hipStream_t)stream,
                         (const uint16_t*)dgauss, ldg, h, w, k, inv_std, s, mu, py, px, (uint16_t*)dheat, lddh, gauss_mode,
                         (const uint16_t*)wt_dgrad, kpad_d, c, (uint16_t*)dfeat, lddf, bias_partial);
    }
  });
  IMM_CHECK_LAUNCH("imm_pose_head_bwd");
  return 0;
}

extern "C" int imm_gauss_render_f32(const float* mu, int batch, int k, float inv_std, int s, float* out, int gauss_mode,
                                    void* stream) {
  IMM_REQUIRE(mu && out && batch > 0 && k > 0 && s > 0, "gauss_render: args");
  IMM_REQUIRE(gauss_mode >= IMM_GAUSS_ROT && gauss_mode <= IMM_GAUSS_ANKUSH, "gauss_render: gauss_mode %d", gauss_mode);
  const int64_t total = (int64_t)batch * s * s * k;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gauss_render_f32_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, mu, k, inv_std, s, out, total, gauss_mode);
  IMM_CHECK_LAUNCH("imm_gauss_render_f32");
  return 0;
}
