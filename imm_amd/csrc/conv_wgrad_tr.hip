// conv_wgrad_tr.hip — general filter-gradient kernel, second generation: natural-layout LDS-DMA staging +
// hardware transpose reads.
//
//     dW[kk][n] = sum_p A[p][kk] * dY[p][n]      (A = im2col gather, kk = (ky*kw+kx)*ci + c, p = output pixel)
//
// conv_wgrad.hip transposes in software while staging (2 x 16-B loads -> 8 ds_write_b32 per lane per step): the
// ds_write + pack work, not the matrix cores, bounds it (~210 TFLOP/s on the 256-channel layers).  Here both
// operand tiles are DMA'd straight into LDS in their NATURAL layout — A as [32 pixels][128 kk] (one 256-byte row per
// pixel, gathered with the forward kernel's im2col addressing), dY as [32 pixels][BN channels] — through a ring of
// NS stages with counted vmcnt (same pipeline as conv_igemm64.hip), and ds_read_b64_tr_b16 delivers each lane 4
// consecutive PIXELS of one kk / channel: two reads = one v_mfma_f32_16x16x32 operand (semantics probed in
// tools/probes/tr_read_probe.hip).  No VGPR staging, no ds_write, no packing.
// Requires the wave-uniform pixel walk (ho*wo % 32 == 0, wo | 32 or 32 | wo) and ci % 8 == 0.
#include "common.h"
#include <stdlib.h>
#include <string.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

struct WgradTrArgs {
  const uint16_t* x; const uint16_t* dy; float* slab;
  int P, hi, wi, ci8, ldx, ho, wo, co, lddy;
  int kh, kw, stride, pad_t, pad_l, kpad, ntaps;
  int n_kblk, n_nblk, nsplit, p_per_split;
  uint32_t x_bytes, dy_bytes;
};

__device__ __forceinline__ void wt_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// chunk swizzle by pixel row for an LDS image with CB 16-byte chunks per row: the 8 pixels x 32 bytes that one
// 32-lane half of a transpose read touches must fall into 8 different 32-byte bank windows of the 256-byte bank row
// (one transpose read of a half-wave covers pixels {b..b+3} U {b+8..b+11}, b = 0,4,16,20):
//   CB = 16 (256-B rows): every pixel starts on bank 0       -> window = (p&3) | bit3(p)<<2        (8 values)
//   CB =  8 (128-B rows): parity of p picks the bank half    -> window = bit1(p) | bit3(p)<<1      (4 values)
//   CB =  4 ( 64-B rows): p&3 picks the 64-B quarter         -> window = bit3(p)                   (2 values)
//   CB =  2 ( 32-B rows): no room to swizzle, 2-way conflict accepted (co <= 16 layers only)
template <int CB>
__device__ __forceinline__ int wt_swz(int p) {
  return CB >= 16 ? (((p & 3) | (((p >> 3) & 1) << 2)) << 1)
       : CB == 8 ? ((((p >> 1) & 1) | (((p >> 3) & 1) << 1)) << 1)
       : CB == 4 ? (((p >> 3) & 1) << 1) : 0;
}
template <int CB>
__device__ __forceinline__ uint32_t wt_piece(int p, int ch) {   // byte offset of channels ch..ch+3 (ch % 4 == 0) of pixel p
  return (uint32_t)((p * CB + ((ch >> 3) ^ wt_swz<CB>(p))) * 16 + (ch & 4) * 2);
}

template <typename ET, int BN, int WGK, int WGN, int NS>
__device__ __forceinline__ void conv_wgrad_tr_body(const WgradTrArgs& a, int bid) {
  constexpr int BK = 128;
  constexpr int TK = BK / WGK, TN = BN / WGN, KT_ = TK / 16, NT = TN / 16;
  constexpr int CA = BK / 8, CB = BN / 8;                 // 16-byte chunks per pixel row
  constexpr int A_BYTES = 32 * BK * 2, B_BYTES = 32 * BN * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int A_INSTR = 8, B_INSTR = (CB / 2) > 0 ? (CB / 2) : 1;     // 1-KB DMA instructions per tile
  constexpr int A_PER_WAVE = A_INSTR / 4;                               // 2
  constexpr int B_PER_WAVE = (B_INSTR + 3) / 4;                         // 2, 1, 1, 1 (idle waves issue a dummy)
  constexpr int LPT = A_PER_WAVE + B_PER_WAVE;
  constexpr int B_PPI = 64 / CB;                                         // pixels per B instruction
  static_assert(WGK * WGN == 4, "4 waves");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];          // NS * STAGE + 1 KB dummy target

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wk = wid / WGN, wn = wid % WGN;
  const int nblk = bid % a.n_nblk; bid /= a.n_nblk;
  const int kblk = bid % a.n_kblk; bid /= a.n_kblk;
  const int split = bid;
  const int k0 = kblk * BK, n0 = nblk * BN;
  const int p_begin = split * a.p_per_split;
  const int p_end = min(a.P, p_begin + a.p_per_split);
  const int nsteps = (p_end - p_begin) / 32;                 // P and p_per_split are multiples of 32

  constexpr uint32_t OOB = 0x80000000u;
  const uint64_t xa = (uint64_t)a.x, ya = (uint64_t)a.dy;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t yr = {(uint32_t)ya, (uint32_t)(ya >> 32) & 0xffffu, a.dy_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;
  const uint32_t dummy_lds = lds_base + (uint32_t)(NS * STAGE);

  // ---- per-lane constants of this lane's DMA pieces -------------------------------------------------------
  // A instruction q covers pixels 4q..4q+3; lane -> (pixel 4q + (l>>4), stored chunk l&15)
  int a_cy[A_PER_WAVE], a_cx[A_PER_WAVE], a_coff[A_PER_WAVE];
  bool a_ok[A_PER_WAVE];
#pragma unroll
  for (int i = 0; i < A_PER_WAVE; ++i) {
    const int q = wid * A_PER_WAVE + i;
    const int p = 4 * q + (lane >> 4);
    const int c = (lane & 15) ^ wt_swz<CA>(p);              // source chunk of the 128-wide kk tile
    const int k8 = (k0 >> 3) + c;
    const int tap = k8 / a.ci8, c8 = k8 - tap * a.ci8;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    a_ok[i] = tap < a.ntaps;
    const int jy = (a.wo >= 32) ? 0 : p / a.wo;
    const int jx = (a.wo >= 32) ? p : p - jy * a.wo;
    a_cy[i] = jy * a.stride - a.pad_t + ky;
    a_cx[i] = jx * a.stride - a.pad_l + kx;
    a_coff[i] = c8 * 16;
  }
  // B instruction q covers pixels q*B_PPI..; lane -> (pixel q*B_PPI + l/CB, stored chunk l%CB)
  uint32_t b_voff[B_PER_WAVE];
  bool b_real[B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < B_PER_WAVE; ++i) {
    const int q = wid * B_PER_WAVE + i;
    b_real[i] = q < B_INSTR;
    const int p = q * B_PPI + lane / CB;
    const int c = (lane % CB) ^ wt_swz<CB>(p);
    const int n = n0 + c * 8;
    b_voff[i] = (b_real[i] && n < a.lddy) ? (uint32_t)((p * a.lddy + n) * 2) : OOB;
  }
  // wave-uniform walk of the 32-pixel step origin
  int s_img, s_y0, s_x0, s_p0 = p_begin;
  {
    const int hw = a.ho * a.wo;
    s_img = p_begin / hw;
    const int rem0 = p_begin - s_img * hw;
    s_y0 = rem0 / a.wo;
    s_x0 = rem0 - s_y0 * a.wo;
  }

  auto issue = [&](int stage) {
    const uint32_t a_soff = (uint32_t)(s_img * a.hi * a.wi) * (uint32_t)(a.ldx * 2);
    const uint32_t b_soff = (uint32_t)s_p0 * (uint32_t)(a.lddy * 2);
    const int ys = s_y0 * a.stride, xs = s_x0 * a.stride;
    const uint32_t st = lds_base + (uint32_t)(stage * STAGE);
#pragma unroll
    for (int i = 0; i < A_PER_WAVE; ++i) {
      const int iy = ys + a_cy[i], ix = xs + a_cx[i];
      const bool ok = a_ok[i] && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
      const uint32_t vo = ok ? (uint32_t)((iy * a.wi + ix) * a.ldx * 2 + a_coff[i]) : OOB;
      wt_dma16(xr, __builtin_amdgcn_readfirstlane(st + (uint32_t)((wid * A_PER_WAVE + i) * 1024)), vo, a_soff);
    }
#pragma unroll
    for (int i = 0; i < B_PER_WAVE; ++i) {
      const uint32_t vo = b_voff[i];
      const uint32_t dst = b_real[i] ? st + (uint32_t)(A_BYTES + (wid * B_PER_WAVE + i) * 1024) : dummy_lds;
      wt_dma16(yr, __builtin_amdgcn_readfirstlane(dst), vo, b_soff);
    }
    s_p0 += 32;
    if (a.wo >= 32) { s_x0 += 32; if (s_x0 == a.wo) { s_x0 = 0; ++s_y0; } }
    else s_y0 += 32 / a.wo;
    if (s_y0 == a.ho) { s_y0 = 0; ++s_img; }
  };

  f32x4_t acc[KT_][NT];
#pragma unroll
  for (int i = 0; i < KT_; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nsteps) issue(t);

  // transpose-read geometry: 16-lane group g <-> pixels 8g..8g+7 of the step; lane i16 <-> kk / channel i16 of a tile
  const int i16 = lane & 15, g = lane >> 4;
  const int pr0 = 8 * g + (i16 >> 2), ch4 = (i16 & 3) * 4;
  const char* lds_c = (const char*)smem;
  // (Round 2, measured: reading the fragments of step st+1 under the MFMAs of step st — which costs one step of DMA look-ahead,
  // the wait then being for step st+1 — is SLOWER, 3.543 -> 3.561 ms per training step, 39.9 -> 44.8 us on the 32x32 256->128
  // layer: the steps are bound by the global->LDS latency of their 16 KB, not by the LDS round trip.  A deeper ring does not
  // help either: NS = 6 / 8 instead of 4: 3.555 -> 3.655 ms, 38.7 -> 54 us on that layer.)
  int stage = 0;
  for (int st = 0; st < nsteps; ++st) {
    if (st + NS - 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      int ns = stage + NS - 1; if (ns >= NS) ns -= NS;
      if (st + NS - 1 < nsteps) issue(ns);
    }
    const char* Al = lds_c + stage * STAGE;
    const char* Bl = Al + A_BYTES;
    uint4 af[KT_], bf[NT];
#pragma unroll
    for (int i = 0; i < KT_; ++i) {
      const int kk = wk * TK + i * 16 + ch4;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Al + wt_piece<CA>(pr0, kk)));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Al + wt_piece<CA>(pr0 + 4, kk)));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      af[i] = make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nn = wn * TN + j * 16 + ch4;
      const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bl + wt_piece<CB>(pr0, nn)));
      const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Bl + wt_piece<CB>(pr0 + 4, nn)));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      bf[j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
#pragma unroll
    for (int i = 0; i < KT_; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[j], af[i], acc[i][j]);   // D[n][kk]
    if (++stage == NS) stage = 0;
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

template <typename ET, int BN, int WGK, int WGN, int NS>
__global__ __launch_bounds__(256) void conv_wgrad_tr_kernel(const WgradTrArgs a) {
  conv_wgrad_tr_body<ET, BN, WGK, WGN, NS>(a, blockIdx.x);
}

// Several filter gradients of the same tile shape in ONE launch (imm_conv2d_wgrad_multi): workgroup b belongs to member g
// with first[g] <= b < first[g+1] and runs that member's argument block (read from the device table with scalar loads)
// unchanged.  Members are ordered longest workgroups first.
template <typename ET, int BN, int WGK, int WGN, int NS>
__global__ __launch_bounds__(256) void conv_wgrad_tr_multi_kernel(const WgradTrArgs* __restrict__ tab,
                                                                  const int* __restrict__ first, int n) {
  const int b = blockIdx.x;
  int g = 0;
  while (g + 1 < n && b >= first[g + 1]) ++g;
  const WgradTrArgs a = tab[g];
  conv_wgrad_tr_body<ET, BN, WGK, WGN, NS>(a, b - first[g]);
}

template <typename ET, int BN, int WGK, int WGN, int NS>
static void wt_launch_multi_cfg(const WgradTrArgs* tab, const int* first, int n, int blocks, hipStream_t s) {
  constexpr int lds = NS * (32 * 128 * 2 + 32 * BN * 2) + 1024;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tr_multi_kernel<ET, BN, WGK, WGN, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_tr_multi_kernel<ET, BN, WGK, WGN, NS>), dim3(blocks), dim3(256), lds, s, tab, first, n);
}

template <typename ET, int BN, int WGK, int WGN, int NS>
static void wt_launch_cfg(const WgradTrArgs& a, hipStream_t s) {
  constexpr int lds = NS * (32 * 128 * 2 + 32 * BN * 2) + 1024;
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_tr_kernel<ET, BN, WGK, WGN, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_tr_kernel<ET, BN, WGK, WGN, NS>), dim3(a.n_kblk * a.n_nblk * a.nsplit), dim3(256), lds, s, a);
}

template <typename ET>
static void wt_launch(const WgradTrArgs& a, int bn, hipStream_t s) {
  if (bn == 128) wt_launch_cfg<ET, 128, 2, 2, 4>(a, s);
  else if (bn == 64) wt_launch_cfg<ET, 64, 2, 2, 4>(a, s);
  else if (bn == 32) wt_launch_cfg<ET, 32, 4, 1, 4>(a, s);
  else wt_launch_cfg<ET, 16, 4, 1, 4>(a, s);
}

bool imm_wgrad_tr_applicable(const imm_conv_desc* d, int lddy) {
  static const bool off = imm_conv_disabled("wgrad_tr");
  if (off) return false;
  const int hw = d->ho * d->wo;
  const int64_t P = (int64_t)d->batch * hw;
  const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, db = P * lddy * 2;
  return (hw % 32 == 0) && (d->wo % 32 == 0 || 32 % d->wo == 0) && xb < (1LL << 31) && db < (1LL << 31) && d->updiv == 1;
}

static int wt_bn(int co) { return co > 64 ? 128 : co > 32 ? 64 : co > 16 ? 32 : 16; }

static WgradTrArgs wt_fill(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit) {
  WgradTrArgs a;
  a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.slab = slab;
  a.P = d->batch * d->ho * d->wo;
  a.hi = d->hi; a.wi = d->wi; a.ci8 = d->ci / 8; a.ldx = d->ldx;
  a.ho = d->ho; a.wo = d->wo; a.co = d->co; a.lddy = lddy;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l;
  a.kpad = d->kpad; a.ntaps = d->kh * d->kw;
  const int bn = wt_bn(d->co);
  a.n_kblk = (d->kpad + 127) / 128;
  a.n_nblk = (d->co + bn - 1) / bn;
  a.nsplit = nsplit;
  int pps = (a.P + nsplit - 1) / nsplit;
  pps = (pps + 31) / 32 * 32;
  a.p_per_split = pps;
  a.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  a.dy_bytes = (uint32_t)((int64_t)a.P * lddy * 2);
  return a;
}

// called from conv_wgrad.hip
void imm_wgrad_tr_launch(int dtype, const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit,
                         hipStream_t s) {
  const WgradTrArgs a = wt_fill(d, x, dy, lddy, slab, nsplit);
  const int bn = wt_bn(d->co);
  if (dtype == IMM_BF16) wt_launch<BF16>(a, bn, s);
  else wt_launch<F16>(a, bn, s);
}

// ---- members of a multi-problem launch (imm_conv2d_wgrad_multi, conv_wgrad.hip) ---------------------------------------------
int imm_wgrad_tr_variant(const imm_conv_desc* d) { return wt_bn(d->co); }       // tile width = the variant of the kernel
int imm_wgrad_tr_args_bytes() { return (int)sizeof(WgradTrArgs); }
// writes the member's argument block, returns its workgroup count; *steps = 32-pixel steps of one workgroup (its length)
int imm_wgrad_tr_fill(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit, void* out,
                      int* steps) {
  const WgradTrArgs a = wt_fill(d, x, dy, lddy, slab, nsplit);
  memcpy(out, &a, sizeof(a));
  if (steps) *steps = a.p_per_split / 32;
  return a.n_kblk * a.n_nblk * a.nsplit;
}
void imm_wgrad_tr_launch_multi(int dtype, int bn, const void* tab_dev, const int* first_dev, int n, int blocks, hipStream_t s) {
  const WgradTrArgs* t = (const WgradTrArgs*)tab_dev;
#define WT_GO(ET_) \
  do { \
    if (bn == 128) wt_launch_multi_cfg<ET_, 128, 2, 2, 4>(t, first_dev, n, blocks, s); \
    else if (bn == 64) wt_launch_multi_cfg<ET_, 64, 2, 2, 4>(t, first_dev, n, blocks, s); \
    else if (bn == 32) wt_launch_multi_cfg<ET_, 32, 4, 1, 4>(t, first_dev, n, blocks, s); \
    else wt_launch_multi_cfg<ET_, 16, 4, 1, 4>(t, first_dev, n, blocks, s); \
  } while (0)
  if (dtype == IMM_BF16) WT_GO(BF16); else WT_GO(F16);
#undef WT_GO
}
