// conv_wgrad.hip — filter gradient of the implicit-GEMM convolution on gfx950 matrix cores.
//
// tf.gradients of tf.nn.conv2d w.r.t. `w` (imm/tf_utils/nn_utils.py:100; applied at
// imm/train/cnn_train_multi.py:157,231).  GEMM view:
//     dW[kk][n] = sum_p  A[p][kk] * dY[p][n],   kk = (ky*kw+kx)*ci + c,  p = output pixel
// with A the same im2col gather the forward kernel uses.  The contraction index p is the STRIDED
// index of both operands in NHWC memory, so staging transposes: each thread loads 8 channels of
// TWO neighbouring pixels (2 x 16 B), pairs them into 8 dwords {pixel p, pixel p+1} and writes
// them down 8 rows of an LDS image [row = kk or n][32 pixels]; MFMA fragments are then plain
// 16-byte row reads, exactly as in the forward kernel (same XOR swizzle).
// The pixel range is split over `nsplit` blocks per tile; each writes its f32 partial tile to
// slab[split][kpad][co]; imm_conv2d_wgrad_reduce sums slabs in a fixed order (deterministic).
#include "common.h"
#include <stdlib.h>
#include <string.h>

struct WgradArgs {
  const uint16_t* x;
  const uint16_t* dy;
  float* slab;
  int P;            // batch*ho*wo
  int hi, wi, ci8, ldx;
  int ho, wo, co, lddy;
  int kh, kw, stride, pad_t, pad_l;
  int kpad, ntaps;
  int n_kblk, n_nblk, nsplit, p_per_split;
  uint32_t x_bytes, dy_bytes;
};

// LDS image: row = kk (or n), 32*PSUB pixels = 16*PSUB dwords = 4*PSUB 16-byte chunks per row.
// chunk c of row r is stored at c ^ swz(r); swz makes the 16-row x 4-chunk ds_read_b128 fragments conflict-free
// (PSUB=1: 64-byte rows, the forward kernel's pattern; PSUB=4: 256-byte rows, every row starts on bank 0, so the
// 16 rows of a fragment must land on 16 different chunks: swz = r & 15).
template <int PSUB>
__device__ __forceinline__ int wg_swz(int row) { return PSUB == 1 ? (((row >> 3) & 1) * 3) : (row & (4 * PSUB - 1)); }
template <int PSUB>
__device__ __forceinline__ int wg_dword_idx(int row, int u, int pp) {   // pixel pair pp of sub-step u
  return row * (16 * PSUB) + (((u * 4 + (pp >> 2)) ^ wg_swz<PSUB>(row)) << 2) + (pp & 3);
}
template <int PSUB>
__device__ __forceinline__ int wg_chunk_idx(int row, int chunk) { return row * (4 * PSUB) + (chunk ^ wg_swz<PSUB>(row)); }

// FAST (ho*wo % 32 == 0 and wo | 32 or 32 | wo): a 32-pixel sub-step never crosses an image, so the image
// index and the sub-step's top-left (y0,x0) are wave-uniform; each lane adds constant pixel offsets and the
// image / pixel-block base rides in the buffer instruction's scalar offset.  PSUB sub-steps (32 pixels each) are
// staged per barrier: 4*PSUB MFMA k-steps of work and 2*PSUB independent 16-byte loads per lane in flight.
template <typename ET, int BK, int BN, int WGK, int WGN, bool FAST, int PSUB>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int TK = BK / WGK, TN = BN / WGN;
  constexpr int KT_ = TK / 16, NT = TN / 16;
  static_assert(BK == 128, "one pass of 16 k8-groups x 16 pixel pairs");
  static_assert(WGK * WGN == 4, "4 waves");
  static_assert(FAST || PSUB == 1, "multi-sub-step staging needs the uniform pixel walk");
  constexpr int BUF = (BK + BN) * 4 * PSUB;       // uint4 per stage

  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // 2 * BUF

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid / WGN, wn = wid % WGN;
  int bid = blockIdx.x;
  const int nblk = bid % a.n_nblk; bid /= a.n_nblk;
  const int kblk = bid % a.n_kblk; bid /= a.n_kblk;
  const int split = bid;
  const int k0 = kblk * BK, n0 = nblk * BN;
  const int p_begin = split * a.p_per_split;
  const int p_end = min(a.P, p_begin + a.p_per_split);

  // ---- loader state ---------------------------------------------------------------------------
  const int pp = tid & 15, g = tid >> 4;          // pixel pair / 8-channel group
  // A: group g covers kk = k0 + g*8 .. +7 -> fixed (tap, c8) per thread
  int a_ky, a_kx, a_c8;
  bool a_ok;
  {
    const int k8 = (k0 >> 3) + g;
    const int tap = k8 / a.ci8;
    a_c8 = k8 - tap * a.ci8;
    a_ky = tap / a.kw;
    a_kx = tap - a_ky * a.kw;
    a_ok = tap < a.ntaps;
  }
  const bool b_ok = (g < BN / 8) && (n0 + g * 8 < a.lddy);   // dY channel group inside the row
  uint4 ra[PSUB][2], rb[PSUB][2];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
  constexpr uint32_t OOB = 0x80000000u;

  // ---- generic pixel cursor (per lane) ---------------------------------------------------------------
  int pcur = p_begin + 2 * pp;
  int img, oy, ox;
  {
    const int hw = a.ho * a.wo;
    img = pcur / hw; const int rem = pcur - img * hw; oy = rem / a.wo; ox = rem - oy * a.wo;
  }
  // ---- fast path state: uniform sub-step origin + per-lane constant offsets ---------------------------
  int s_img = 0, s_y0 = 0, s_x0 = 0, s_p0 = p_begin;      // wave-uniform
  int cy[2], cx[2];
  uint32_t b_voff[2];
  __amdgpu_buffer_rsrc_t xr, dr;
  if constexpr (FAST) {
    const int hw = a.ho * a.wo;
    s_img = p_begin / hw;
    const int rem0 = p_begin - s_img * hw;
    s_y0 = rem0 / a.wo;
    s_x0 = rem0 - s_y0 * a.wo;
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = 2 * pp + q;
      const int jy = (a.wo >= 32) ? 0 : j / a.wo;
      const int jx = (a.wo >= 32) ? j : j - jy * a.wo;
      cy[q] = jy * a.stride - a.pad_t + a_ky;
      cx[q] = jx * a.stride - a.pad_l + a_kx;
      b_voff[q] = b_ok ? (uint32_t)((j * a.lddy + n0 + g * 8) * 2) : OOB;
    }
  }

  auto load_sub = [&](int u) {       // one 32-pixel sub-step into ra[u], rb[u]
    if constexpr (FAST) {
      const uint32_t a_soff = (uint32_t)(s_img * a.hi * a.wi) * (uint32_t)(a.ldx * 2);
      const uint32_t b_soff = (uint32_t)s_p0 * (uint32_t)(a.lddy * 2);
      const int ys = s_y0 * a.stride, xs = s_x0 * a.stride;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int iy = ys + cy[q], ix = xs + cx[q];
        const bool ok = a_ok && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
        const uint32_t vo = ok ? (uint32_t)(((iy * a.wi + ix) * a.ldx + a_c8 * 8) * 2) : OOB;
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, a_soff, 0);
        ra[u][q] = make_uint4(v.x, v.y, v.z, v.w);
        const u32x4_t w = __builtin_amdgcn_raw_buffer_load_b128(dr, b_voff[q], b_soff, 0);
        rb[u][q] = make_uint4(w.x, w.y, w.z, w.w);
      }
      // advance the uniform origin by 32 pixels
      s_p0 += 32;
      if (a.wo >= 32) { s_x0 += 32; if (s_x0 == a.wo) { s_x0 = 0; ++s_y0; } }
      else s_y0 += 32 / a.wo;
      if (s_y0 == a.ho) { s_y0 = 0; ++s_img; }
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int qi = img, qy = oy, qx = ox + q;
        if (qx >= a.wo) { qx -= a.wo; if (++qy == a.ho) { qy = 0; ++qi; } }
        const bool pok = (pcur + q) < p_end;
        const int iy = qy * a.stride - a.pad_t + a_ky, ix = qx * a.stride - a.pad_l + a_kx;
        const bool ok = pok && a_ok && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
        ra[u][q] = zero4;
        if (ok) ra[u][q] = *(const uint4*)(a.x + ((((int64_t)qi * a.hi + iy) * a.wi + ix) * a.ldx + a_c8 * 8));
        rb[u][q] = zero4;
        if (pok && b_ok) rb[u][q] = *(const uint4*)(a.dy + ((int64_t)(pcur + q) * a.lddy + n0 + g * 8));
      }
      pcur += 32;
      ox += 32;
      while (ox >= a.wo) { ox -= a.wo; if (++oy == a.ho) { oy = 0; ++img; } }
    }
  };
  auto load_tile = [&]() {
#pragma unroll
    for (int u = 0; u < PSUB; ++u) load_sub(u);
  };
  auto store_tile = [&](int buf) {
    uint32_t* Ab = (uint32_t*)(smem + buf * BUF);
    uint32_t* Bb = Ab + BK * 16 * PSUB;
#pragma unroll
    for (int u = 0; u < PSUB; ++u) {
      {
        const uint32_t lo[4] = {ra[u][0].x, ra[u][0].y, ra[u][0].z, ra[u][0].w};
        const uint32_t hi[4] = {ra[u][1].x, ra[u][1].y, ra[u][1].z, ra[u][1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = g * 8 + 2 * j;
          Ab[wg_dword_idx<PSUB>(row, u, pp)] = (lo[j] & 0xffffu) | (hi[j] << 16);
          Ab[wg_dword_idx<PSUB>(row + 1, u, pp)] = (lo[j] >> 16) | (hi[j] & 0xffff0000u);
        }
      }
      if (g < BN / 8) {
        const uint32_t lo[4] = {rb[u][0].x, rb[u][0].y, rb[u][0].z, rb[u][0].w};
        const uint32_t hi[4] = {rb[u][1].x, rb[u][1].y, rb[u][1].z, rb[u][1].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = g * 8 + 2 * j;
          Bb[wg_dword_idx<PSUB>(row, u, pp)] = (lo[j] & 0xffffu) | (hi[j] << 16);
          Bb[wg_dword_idx<PSUB>(row + 1, u, pp)] = (lo[j] >> 16) | (hi[j] & 0xffff0000u);
        }
      }
    }
  };

  f32x4_t acc[KT_][NT];
#pragma unroll
  for (int i = 0; i < KT_; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nsteps = (p_end - p_begin + 32 * PSUB - 1) / (32 * PSUB);
  if (nsteps > 0) {
    load_tile();
    store_tile(0);
  }
  __syncthreads();
  const int frow = lane & 15, fchunk = lane >> 4;
  for (int st = 0; st < nsteps; ++st) {
    const int buf = st & 1;
    const bool more = (st + 1) < nsteps;
    if (more) load_tile();
    const uint4* Ab = smem + buf * BUF;
    const uint4* Bb = Ab + BK * 4 * PSUB;
#pragma unroll
    for (int u = 0; u < PSUB; ++u) {
      uint4 af[KT_], bf[NT];
#pragma unroll
      for (int i = 0; i < KT_; ++i) af[i] = Ab[wg_chunk_idx<PSUB>(wk * TK + i * 16 + frow, u * 4 + fchunk)];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = Bb[wg_chunk_idx<PSUB>(wn * TN + j * 16 + frow, u * 4 + fchunk)];
#pragma unroll
      for (int i = 0; i < KT_; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[j], af[i], acc[i][j]);   // D[n][kk]
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  // lane holds D[n = 4*(lane>>4)+r][kk = lane&15] -> 4 consecutive n of one kk: 16-byte store
  float* out = a.slab + (int64_t)split * a.kpad * a.co;
#pragma unroll
  for (int i = 0; i < KT_; ++i) {
    const int kk = k0 + wk * TK + i * 16 + (lane & 15);
    if (kk >= a.kpad) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * TN + j * 16 + 4 * (lane >> 4);
      float* op = out + (int64_t)kk * a.co + n;
      if (n + 3 < a.co && (a.co & 3) == 0) {
        *(float4*)op = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < a.co) op[r] = acc[i][j][r];
      }
    }
  }
}

template <typename ET, int BK, int BN, int WGK, int WGN, bool FAST, int PSUB>
static void wg_launch_one(const WgradArgs& a, hipStream_t s) {
  constexpr int lds = 2 * (BK + BN) * 64 * PSUB;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<ET, BK, BN, WGK, WGN, FAST, PSUB>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_kernel<ET, BK, BN, WGK, WGN, FAST, PSUB>), dim3(a.n_kblk * a.n_nblk * a.nsplit),
                     dim3(256), lds, s, a);
}

template <typename ET, int BK, int BN, int WGK, int WGN>
static void wg_launch_cfg(const WgradArgs& a, bool fast, int psub, hipStream_t s) {
  if (fast && psub == 4) wg_launch_one<ET, BK, BN, WGK, WGN, true, 4>(a, s);
  else if (fast) wg_launch_one<ET, BK, BN, WGK, WGN, true, 1>(a, s);
  else wg_launch_one<ET, BK, BN, WGK, WGN, false, 1>(a, s);
}

static int wgrad_bn(int co) { return co > 64 ? 128 : co > 32 ? 64 : co > 16 ? 32 : 16; }

template <typename ET>
static int wgrad_launch(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit,
                        hipStream_t s) {
  WgradArgs a;
  a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.slab = slab;
  a.P = d->batch * d->ho * d->wo;
  a.hi = d->hi; a.wi = d->wi; a.ci8 = d->ci / 8; a.ldx = d->ldx;
  a.ho = d->ho; a.wo = d->wo; a.co = d->co; a.lddy = lddy;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.kpad = d->kpad; a.ntaps = d->kh * d->kw;
  const int bn = wgrad_bn(d->co);
  a.n_kblk = (d->kpad + 127) / 128;
  a.n_nblk = (d->co + bn - 1) / bn;
  a.nsplit = nsplit;
  const int hw = d->ho * d->wo;
  const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, db = (int64_t)a.P * lddy * 2;
  const bool fast = (hw % 32 == 0) && (d->wo % 32 == 0 || 32 % d->wo == 0) && xb < (1LL << 31) && db < (1LL << 31);
  constexpr int psub_env = 1;
  // 128 pixels per barrier when the pixel count allows it and each split still gets >= 4 steps
  int pps = (a.P + nsplit - 1) / nsplit;
  const int psub = (fast && psub_env == 4 && a.P % 128 == 0 && pps >= 512 && bn <= 64) ? 4 : 1;
  pps = (pps + 32 * psub - 1) / (32 * psub) * (32 * psub);
  a.p_per_split = pps;
  a.x_bytes = (uint32_t)(fast ? xb : 0);
  a.dy_bytes = (uint32_t)(fast ? db : 0);
  if (bn == 128) wg_launch_cfg<ET, 128, 128, 2, 2>(a, fast, psub, s);
  else if (bn == 64) wg_launch_cfg<ET, 128, 64, 2, 2>(a, fast, psub, s);
  else if (bn == 32) wg_launch_cfg<ET, 128, 32, 4, 1>(a, fast, psub, s);
  else wg_launch_cfg<ET, 128, 16, 4, 1>(a, fast, psub, s);
  IMM_CHECK_LAUNCH("imm_conv2d_wgrad");
  return 0;
}

bool imm_wgrad_tr_applicable(const imm_conv_desc* d, int lddy);            // conv_wgrad_tr.hip
void imm_wgrad_tr_launch(int dtype, const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit,
                         hipStream_t s);
bool imm_wgrad_halo_applicable(const imm_conv_desc* d, int lddy);          // conv_wgrad_halo.hip
int imm_wgrad_halo_splits(const imm_conv_desc* d, int lddy);
void imm_wgrad_halo_launch(int dtype, const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab,
                           int nsplit, hipStream_t s, const float* nol_scale, const float* nol_shift, int nol_relu);

int imm_conv_f32_wgrad(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit, hipStream_t s);   // conv_f32.hip

extern "C" int imm_conv2d_wgrad_splits(const imm_conv_desc* d, int lddy) {
  if (!d) return IMM_E_INVALID;
  return imm_wgrad_halo_applicable(d, lddy) ? imm_wgrad_halo_splits(d, lddy) : 0;
}

extern "C" int imm_conv2d_wgrad(const imm_conv_desc* d, int dtype, const void* x, const void* dy, int lddy,
                                float* slab, int nsplit, void* stream) {
  IMM_REQUIRE(d && x && dy && slab, "wgrad: null");
  IMM_REQUIRE(d->ci > 0 && d->ci % 8 == 0 && d->ldx % 8 == 0 && d->ldx >= d->ci, "wgrad: ci/ldx must be multiples of 8");
  IMM_REQUIRE(lddy % 8 == 0 && lddy >= d->co, "wgrad: lddy=%d must be a multiple of 8 and >= co", lddy);
  IMM_REQUIRE(d->updiv == 1, "wgrad: desc must be the forward convolution");
  IMM_REQUIRE(d->kpad % 32 == 0 && d->kpad >= d->kh * d->kw * d->ci, "wgrad: kpad");
  IMM_REQUIRE(nsplit >= 1, "wgrad: nsplit");
  IMM_REQUIRE(d->wo % 2 == 0, "wgrad: output width must be even (pixel pairs)");
  IMM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)dy % 16 == 0) && ((uintptr_t)slab % 16 == 0), "wgrad: alignment");
  if (dtype == IMM_F32) return imm_conv_f32_wgrad(d, x, dy, lddy, slab, nsplit, (hipStream_t)stream);     // the f32 witness (conv_f32.hip)
  if (imm_wgrad_halo_applicable(d, lddy) && nsplit == imm_wgrad_halo_splits(d, lddy) && (dtype == IMM_BF16 || dtype == IMM_F16)) {
    imm_wgrad_halo_launch(dtype, d, x, dy, lddy, slab, nsplit, (hipStream_t)stream, nullptr, nullptr, 0);
    IMM_CHECK_LAUNCH("imm_conv2d_wgrad(halo)");
    return 0;
  }
  if (imm_wgrad_tr_applicable(d, lddy) && (dtype == IMM_BF16 || dtype == IMM_F16)) {
    imm_wgrad_tr_launch(dtype, d, x, dy, lddy, slab, nsplit, (hipStream_t)stream);
    IMM_CHECK_LAUNCH("imm_conv2d_wgrad(tr)");
    return 0;
  }
  IMM_DISPATCH_DTYPE(dtype, return wgrad_launch<ET>(d, x, dy, lddy, slab, nsplit, (hipStream_t)stream));
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-problem launch: the filter gradients of many layers in as few launches as there are kernel variants among them.
// Nobody reads a filter gradient before the slab reduction at the end of the backward pass, so the engine collects the
// layers' jobs and issues them together: the per-layer launches (15-48 us each, most of it ramp and tail on a 256-CU part)
// leave the serial BN-backward -> data-gradient chain, a grouped launch has thousands of workgroups to balance, and — since
// the chip no longer has to be filled by ONE layer — each layer needs far fewer pixel splits (slab traffic = nsplit x |dW|).
// ---------------------------------------------------------------------------------------------------------------------
int imm_wgrad_tr_variant(const imm_conv_desc* d);
int imm_wgrad_tr_args_bytes();
int imm_wgrad_tr_fill(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit, void* out, int* steps);
void imm_wgrad_tr_launch_multi(int dtype, int bn, const void* tab_dev, const int* first_dev, int n, int blocks, hipStream_t s);
int imm_wgrad_halo_variant(const imm_conv_desc* d, int lddy);
int imm_wgrad_halo_blocks(const imm_conv_desc* d, int lddy, int* n_patches);
int imm_wgrad_halo_args_bytes();
int imm_wgrad_halo_per_cu(int variant);
int imm_wgrad_halo_fill(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit, void* out, int* steps,
                        const float* nol_scale, const float* nol_shift, int nol_relu);
void imm_wgrad_halo_launch_multi(int dtype, int variant, const void* tab_dev, const int* first_dev, int n, int blocks, hipStream_t s);

namespace {
constexpr uint32_t WGM_MAGIC = 0x57474d31u;   // "WGM1"
constexpr int WGM_MAX_LAUNCH = 16, WGM_MAX_JOBS = 64, WGM_ARG_STRIDE = 192, WGM_HEADER = 1024;
enum { WGM_SINGLE = 0, WGM_TR = 1, WGM_HALO = 2 };
struct WgmLaunch { int32_t kind, variant, n, arg_off, first_off, blocks, job0, pad; };
struct WgmHeader { uint32_t magic; int32_t dtype, n_jobs, n_launch; WgmLaunch launch[WGM_MAX_LAUNCH]; };
static_assert(sizeof(WgmHeader) <= WGM_HEADER, "header");

// kernel family + variant a job runs with (the same choice imm_conv2d_wgrad makes, except that the halo kernel takes any
// split count here)
void wgm_classify(const imm_wgrad_job* j, int dtype, int* kind, int* variant) {
  const bool fast_dt = dtype == IMM_BF16 || dtype == IMM_F16;
  const int hv = fast_dt ? imm_wgrad_halo_variant(&j->desc, j->lddy) : 0;
  if (hv) { *kind = WGM_HALO; *variant = hv; return; }
  if (fast_dt && imm_wgrad_tr_applicable(&j->desc, j->lddy)) { *kind = WGM_TR; *variant = imm_wgrad_tr_variant(&j->desc); return; }
  *kind = WGM_SINGLE; *variant = 0;
}
int wgm_check_job(const imm_wgrad_job* j) {
  const imm_conv_desc* d = &j->desc;
  IMM_REQUIRE(j->x && j->dy && j->slab, "wgrad_multi: null");
  IMM_REQUIRE(d->ci > 0 && d->ci % 8 == 0 && d->ldx % 8 == 0 && d->ldx >= d->ci, "wgrad_multi: ci/ldx must be multiples of 8");
  IMM_REQUIRE(j->lddy % 8 == 0 && j->lddy >= d->co, "wgrad_multi: lddy=%d must be a multiple of 8 and >= co", j->lddy);
  IMM_REQUIRE(d->updiv == 1 && d->kpad % 32 == 0 && d->kpad >= d->kh * d->kw * d->ci && j->nsplit >= 1 && d->wo % 2 == 0,
              "wgrad_multi: desc");
  IMM_REQUIRE(((uintptr_t)j->x % 16 == 0) && ((uintptr_t)j->dy % 16 == 0) && ((uintptr_t)j->slab % 16 == 0), "wgrad_multi: alignment");
  return 0;
}
}  // namespace

extern "C" int64_t imm_conv2d_wgrad_multi_table_bytes(int n) {
  if (n < 1 || n > WGM_MAX_JOBS) return IMM_E_INVALID;
  // header, per job one argument block + one imm_wgrad_job (single launches), per launch a first[] array
  return WGM_HEADER + (int64_t)n * (WGM_ARG_STRIDE + (int64_t)sizeof(imm_wgrad_job)) + WGM_MAX_LAUNCH * (WGM_MAX_JOBS + 1) * 4;
}

extern "C" int imm_conv2d_wgrad_variant(const imm_conv_desc* d, int lddy, int dtype, int* wg_per_split, int* units, int* wg_per_cu) {
  IMM_REQUIRE(d && lddy >= d->co, "wgrad_variant: args");
  imm_wgrad_job j; j.desc = *d; j.lddy = lddy; j.nsplit = 1; j.x = j.dy = nullptr; j.slab = nullptr;
  j.x_scale = j.x_shift = nullptr; j.x_relu = 0;
  int kind, variant;
  wgm_classify(&j, dtype, &kind, &variant);
  int wps = 1, un = 1, pcu = 2;                                // transpose-read / generic kernels: 65 KB of LDS, two per CU
  if (kind == WGM_HALO) {
    wps = imm_wgrad_halo_blocks(d, lddy, &un);                 // units: 8x16-pixel patches
    pcu = imm_wgrad_halo_per_cu(variant);
  } else {
    const int bn = d->co > 64 ? 128 : d->co > 32 ? 64 : d->co > 16 ? 32 : 16;
    wps = ((d->kpad + 127) / 128) * ((d->co + bn - 1) / bn);
    un = (d->batch * d->ho * d->wo + 31) / 32;                 // units: 32-pixel steps
  }
  if (wg_per_split) *wg_per_split = wps;
  if (units) *units = un;
  if (wg_per_cu) *wg_per_cu = pcu;
  return kind * 100000 + variant;
}

extern "C" int imm_conv2d_wgrad_multi_plan(const imm_wgrad_job* jobs, int n, int dtype, void* table_host) {
  IMM_REQUIRE(jobs && table_host && n >= 1 && n <= WGM_MAX_JOBS, "wgrad_multi_plan: args");
  IMM_REQUIRE(imm_wgrad_tr_args_bytes() <= WGM_ARG_STRIDE && imm_wgrad_halo_args_bytes() <= WGM_ARG_STRIDE, "wgrad_multi_plan: arg stride");
  char* base = (char*)table_host;
  memset(base, 0, (size_t)imm_conv2d_wgrad_multi_table_bytes(n));
  WgmHeader* h = (WgmHeader*)base;
  h->magic = WGM_MAGIC; h->dtype = dtype; h->n_jobs = n; h->n_launch = 0;
  int kind[WGM_MAX_JOBS], variant[WGM_MAX_JOBS];
  bool done[WGM_MAX_JOBS];
  for (int i = 0; i < n; ++i) {
    if (wgm_check_job(&jobs[i])) return IMM_E_INVALID;
    wgm_classify(&jobs[i], dtype, &kind[i], &variant[i]);
    IMM_REQUIRE((jobs[i].x_scale == nullptr) == (jobs[i].x_shift == nullptr), "wgrad_multi_plan: job %d: x_scale / x_shift come together", i);
    IMM_REQUIRE(jobs[i].x_scale == nullptr || kind[i] == WGM_HALO,
                "wgrad_multi_plan: job %d: normalise-on-load (x_scale) is served by the LDS-halo filter-gradient kernels only "
                "(imm_conv2d_wgrad_variant / 100000 == 2)", i);
    done[i] = false;
  }
  int64_t off = WGM_HEADER;
  int job_slot = 0;
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    IMM_REQUIRE(h->n_launch < WGM_MAX_LAUNCH, "wgrad_multi_plan: more than %d kernel variants", WGM_MAX_LAUNCH);
    WgmLaunch* L = &h->launch[h->n_launch++];
    L->kind = kind[i]; L->variant = variant[i];
    // members of this variant (a job without a multi kernel is a launch of its own)
    int idx[WGM_MAX_JOBS], m = 0;
    for (int k = i; k < n; ++k)
      if (!done[k] && kind[k] == kind[i] && variant[k] == variant[i] && (kind[i] != WGM_SINGLE || k == i)) { idx[m++] = k; done[k] = true; }
    L->n = m;
    L->arg_off = (int32_t)off;
    if (kind[i] == WGM_SINGLE) {
      memcpy(base + off, &jobs[i], sizeof(imm_wgrad_job));
      off += (int64_t)sizeof(imm_wgrad_job);
      off = (off + 15) / 16 * 16;
      L->first_off = 0; L->blocks = 0; L->job0 = job_slot++;
      continue;
    }
    // argument blocks, longest workgroups first
    int steps[WGM_MAX_JOBS], blocks[WGM_MAX_JOBS];
    char tmp[WGM_MAX_JOBS][WGM_ARG_STRIDE];
    for (int k = 0; k < m; ++k) {
      const imm_wgrad_job* j = &jobs[idx[k]];
      blocks[k] = kind[i] == WGM_TR ? imm_wgrad_tr_fill(&j->desc, j->x, j->dy, j->lddy, j->slab, j->nsplit, tmp[k], &steps[k])
                                    : imm_wgrad_halo_fill(&j->desc, j->x, j->dy, j->lddy, j->slab, j->nsplit, tmp[k], &steps[k], j->x_scale,
                                                          j->x_shift, j->x_relu);
    }
    int order[WGM_MAX_JOBS];
    for (int k = 0; k < m; ++k) order[k] = k;
    for (int a = 1; a < m; ++a)                       // insertion sort, stable, descending work per workgroup
      for (int b = a; b > 0 && steps[order[b]] > steps[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
    const int stride = kind[i] == WGM_TR ? imm_wgrad_tr_args_bytes() : imm_wgrad_halo_args_bytes();
    for (int k = 0; k < m; ++k) memcpy(base + off + (int64_t)k * stride, tmp[order[k]], (size_t)stride);
    off += (int64_t)m * stride;
    off = (off + 15) / 16 * 16;
    L->first_off = (int32_t)off;
    int32_t* first = (int32_t*)(base + off);
    first[0] = 0;
    for (int k = 0; k < m; ++k) first[k + 1] = first[k] + blocks[order[k]];
    L->blocks = first[m];
    off += (int64_t)(m + 1) * 4;
    off = (off + 15) / 16 * 16;
    L->job0 = job_slot; job_slot += m;
  }
  IMM_REQUIRE(off <= imm_conv2d_wgrad_multi_table_bytes(n), "wgrad_multi_plan: table overflow");
  return 0;
}

extern "C" int imm_conv2d_wgrad_multi(const void* table_host, const void* table_dev, void* stream) {
  IMM_REQUIRE(table_host && table_dev, "wgrad_multi: null");
  const char* hb = (const char*)table_host;
  const char* db = (const char*)table_dev;
  const WgmHeader* h = (const WgmHeader*)hb;
  IMM_REQUIRE(h->magic == WGM_MAGIC && h->n_launch >= 1 && h->n_launch <= WGM_MAX_LAUNCH, "wgrad_multi: not a planned table");
  for (int l = 0; l < h->n_launch; ++l) {
    const WgmLaunch* L = &h->launch[l];
    if (L->kind == WGM_SINGLE) {
      const imm_wgrad_job* j = (const imm_wgrad_job*)(hb + L->arg_off);
      const int rc = imm_conv2d_wgrad(&j->desc, h->dtype, j->x, j->dy, j->lddy, j->slab, j->nsplit, stream);
      if (rc) return rc;
    } else if (L->kind == WGM_TR) {
      imm_wgrad_tr_launch_multi(h->dtype, L->variant, db + L->arg_off, (const int*)(db + L->first_off), L->n, L->blocks, (hipStream_t)stream);
    } else {
      imm_wgrad_halo_launch_multi(h->dtype, L->variant, db + L->arg_off, (const int*)(db + L->first_off), L->n, L->blocks, (hipStream_t)stream);
    }
  }
  IMM_CHECK_LAUNCH("imm_conv2d_wgrad_multi");
  return 0;
}

// dw[(tap*ci_real + c)*co + n] = sum_s slab[s][(tap*ci_pad + c)][n]
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int ntaps, int ci_pad, int ci_real,
                                    int co, int kpad, float* __restrict__ dw) {
  const int64_t total = (int64_t)ntaps * ci_real * co;
  const int64_t sstride = (int64_t)kpad * co;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx % co);
    const int64_t tc = idx / co;
    const int c = (int)(tc % ci_real), tap = (int)(tc / ci_real);
    const float* sp = slab + ((int64_t)tap * ci_pad + c) * co + n;
    float acc = 0.f;
    for (int s = 0; s < nsplit; ++s) acc += sp[s * sstride];
    dw[idx] = acc;
  }
}

extern "C" int imm_conv2d_wgrad_reduce(const float* slab, int nsplit, int kh, int kw, int ci_pad, int ci_real, int co,
                                       int kpad, float* dw, void* stream) {
  IMM_REQUIRE(slab && dw && nsplit >= 1, "wgrad_reduce: args");
  IMM_REQUIRE(ci_real <= ci_pad && kpad >= kh * kw * ci_pad, "wgrad_reduce: padding");
  const int64_t total = (int64_t)kh * kw * ci_real * co;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, nsplit, kh * kw,
                     ci_pad, ci_real, co, kpad, dw);
  IMM_CHECK_LAUNCH("imm_conv2d_wgrad_reduce");
  return 0;
}
