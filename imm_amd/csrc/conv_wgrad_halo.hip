// conv_wgrad_halo.hip — filter gradient of 3x3 stride-1 convolutions (maps with H % 8 == 0, W % 16 == 0):
// encoder conv_2/4/6/8, renderer conv_1..8 (reference: tf.gradients of imm/models/imm_model.py:213-300 convolutions).
// A workgroup owns a (CI x CO)-channel slice of the filter (blockIdx.y / blockIdx.z; 32- or 64-channel slices) and one
// of `nsplit` pixel ranges (blockIdx.x).  Slab traffic is nsplit x |dW| x 4 B, operand traffic per FLOP falls with the
// slice size: shallow high-resolution layers take 64-channel slices (few, large workgroup tiles, many patches each),
// deep low-resolution layers 32-channel slices (4x the channel blocks => 4x fewer pixel splits for the same number
// of workgroups).
//
//     dW[tap][c][n] = sum_p X[p + off(tap)][c] * dY[p][n]
//
// The contraction index (pixels) is the STRIDED index of both NHWC operands.  The general kernel
// (conv_wgrad.hip) transposes while staging and re-gathers X once per tap through L2 (rocprofv3 PMC: 3.5x the
// algorithmic HBM bytes at 128x128).  Here, like conv_halo.hip, a persistent workgroup DMA's the 10x18-pixel X halo
// and the 8x16-pixel dY patch ONCE in their natural [pixel][channel] layout (double buffered), the nine taps are
// nine LDS base addresses, and the transposition is done by the LDS hardware: ds_read_b64_tr_b16 hands each lane
// four CONSECUTIVE PIXELS of one channel (probe: tools/probes/tr_read_probe.hip -- within a 16-lane group lane i
// supplies the 8-byte piece {row i>>2, columns 4(i&3)..+3} of a 4x16 block and receives column i, rows 0..3).
// Two such reads = the 8 k-values of a v_mfma_f32_16x16x32 operand; X^T and dY^T use the same pixel order, so the
// contraction is consistent.  All 9*ci*co accumulators of the workgroup stay in registers over its patches and
// are written once as one f32 slab per workgroup (summed by imm_wgrad_reduce_multi, fixed order).
#include "conv_common.h"
#include <stdlib.h>
#include <string.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// -DIMM_WH_ABLATE=<bits> (diagnosis builds only, tools/ablate_build.sh; results are wrong, only the time is read):
//   1 no DMA after the prologue, 2 no X^T transpose reads inside the loop, 4 no dY^T transpose reads inside the loop, 8 no MFMAs,
//   16 no barrier / vmcnt wait inside the loop, 32 no slab stores
#ifndef IMM_WH_ABLATE
#define IMM_WH_ABLATE 0
#endif
#define WH_PH 8
#define WH_PW 16
// halo of an 8x16-pixel patch under a KH x KW stride-1 SAME filter: (8+KH-1) x (16+KW-1) pixels, padded to whole rounds of DMA
// instructions (64 pixels at 32 channels / 32 pixels at 64 channels per instruction, 4 waves): 3x3 -> 10x18 = 180 -> 192;
// 7x1 (the tap-unrolled first encoder convolution) -> 14x16 = 224 -> 256
#define WH_HROWS(KH) (WH_PH + (KH) - 1)
#define WH_HCOLS(KW) (WH_PW + (KW) - 1)
#define WH_HP(KH, KW) ((WH_HROWS(KH) * WH_HCOLS(KW) + 63) / 64 * 64)
// 3x3 STRIDE 2 (encoder conv_3; TF SAME on an even side pads bottom / right only): the 17 x 33 input halo of an 8x16 output patch is
// stored parity-de-interleaved exactly as in conv_halo.hip's stride-2 forward — four planes (row parity, column parity) of 9 x 17
// pixels padded to 160 — so tap (ky,kx) of output pixel (y,x) is pixel (y + ky/2, x + kx/2) of plane (ky&1, kx&1) and the four
// consecutive pixels a transpose read delivers are consecutive in LDS
#define WH_S2_PW 17
#define WH_S2_PP 160
#define WH_S2_HP 640

__device__ __forceinline__ void wh_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// Chunk swizzle of an LDS image [pixel][C8 chunks of 16 B], for the 8-byte transpose reads.  A 32-lane half of such a read takes
// EIGHT pixels of one image row — x..x+3 and x+8..x+11 — times 32 bytes (16 channels), and the LDS serves 256 B = its 64 banks
// per pass: the eight 32-byte windows must differ modulo 256 B.  Pixel stride 128 B (64 channels): even and odd pixels already
// split the 256 B in halves, the chunk pair is XORed with (bit 1 of x) | (bit 3 of x) << 1; 64 B (32 channels): with bit 3 of x.
// (Round 3.  Measured with SQ_LDS_BANK_CONFLICT: the swizzle of the forward halo kernels, ((pixel >> 1) & 3), and a first
// replacement, pixel & 3, both give pixels x and x + 8 the same window — two passes per read, conflict cycles = half of
// SQ_LDS_IDX_ACTIVE.)  It depends on the pixel's COLUMN only: a lane keeps one base address per (horizontal tap offset, first /
// second read) and every read of the loop is base + immediate offset — the address arithmetic (3.3 VALU instructions per MFMA at
// one wave per SIMD) leaves the loop.
template <int C8>
__device__ __forceinline__ int wh_swzx(int x) { return C8 == 8 ? ((((x >> 1) & 1) | (((x >> 3) & 1) << 1)) << 1) : (((x >> 3) & 1) << 1); }

struct WgradHaloArgs {
  const uint16_t* x; const uint16_t* dy; float* slab;
  int batch, h, w, ldx, lddy, co, kpad, ci_total;
  int n_patches, patches_x, patches_y;
  int nsplit, nci, nco;          // grid of the member (multi-problem launches)
  uint32_t x_bytes, dy_bytes;
  // normalise on load (round 4): x is the RAW output y of the conv + batch-norm + ReLU block in front of this convolution; the
  // tensor the forward pass convolved, relu(scale[c] * y + shift[c]), is rebuilt in the LDS halo (conv_halo.hip does the same on
  // the forward side), so it never exists in HBM.  NULL = x is the stored activation.
  const float* nol_scale; const float* nol_shift; int nol_relu;
};

// CI input channels, CO = channels of the dY rows that are staged (32 or 64; co <= CO real ones).
// NS-stage ring (round 2): a patch is ~20-40 KB of fresh HBM data for ~0.5 us of MFMA work, so with the two-stage ring of
// round 1 (wait vmcnt(0), then prefetch ONE patch ahead) every patch exposed most of its fetch latency (renderer conv_5:
// 1.5 us per patch for 0.55 us of matrix work).  Now NS-1 patches are in flight and the wait is counted: every wave issues
// exactly LPP DMA instructions per patch, so "at most (NS-2)*LPP outstanding" == "this patch has landed".
// bx = pixel split of this workgroup, G = number of splits, by / bz = its ci / co slice
template <typename ET, int CI, int CO, int NS, int KH = 3, int KW = 3, int ST = 1>
__device__ __forceinline__ void conv_wgrad_halo_body(const WgradHaloArgs& a, const int bx, const int G, const int by, const int bz) {
  static_assert(ST == 1 || (ST == 2 && KH == 3 && KW == 3), "stride 2: 3x3 only");
  constexpr int NTAP = KH * KW, PT = (KH - 1) / 2, PL = (KW - 1) / 2;
  constexpr int WH_HW = WH_HCOLS(KW), HPIX = WH_HROWS(KH) * WH_HCOLS(KW), WH_HP_ = ST == 2 ? WH_S2_HP : WH_HP(KH, KW);
  constexpr int XC8 = CI / 8, YC8 = CO / 8;
  constexpr int NCT = CI / 16, NNT = CO / 16;          // 16-channel tiles
  constexpr int WC = NCT, WN = 4 / WC;                 // wave grid: wc = ci tile, wn = slice of the co tiles
  constexpr int TNT = NNT / WN;                        // co tiles per wave
  static_assert(WC * WN == 4 && TNT >= 1, "tile split");
  constexpr int X_PIX_PER_DMA = 64 / XC8, Y_PIX_PER_DMA = 64 / YC8;
  constexpr int X_DMA = WH_HP_ / X_PIX_PER_DMA, Y_DMA = 128 / Y_PIX_PER_DMA;
  constexpr int X_BYTES = WH_HP_ * CI * 2, Y_BYTES = 128 * CO * 2, STAGE = X_BYTES + Y_BYTES;
  constexpr int LPP = X_DMA / 4 + Y_DMA / 4;           // DMA instructions per wave and patch
  static_assert(X_DMA % 4 == 0 && Y_DMA % 4 == 0 && NS >= 2 && (NS - 2) * LPP < 64, "uniform per-wave DMA count");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // NS stages of [X halo | dY patch] | [CI/8][16] f32 normalise-on-load table
  const bool nol = a.nol_scale != nullptr;
  float* coef = (float*)((char*)smem + NS * STAGE);

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wc = wid % WC, wn = wid / WC;
  const int ci0 = by * CI, co0 = bz * CO;     // channel slice of this workgroup
  constexpr uint32_t OOB = 0x80000000u;
  const uint64_t xa = (uint64_t)a.x, ya = (uint64_t)a.dy;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t yr = {(uint32_t)ya, (uint32_t)(ya >> 32) & 0xffffu, a.dy_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;
  const int per_img = a.patches_x * a.patches_y;

  // Per-lane halves of the DMA addresses, computed ONCE (they were ~250 VALU instructions per patch and wave, ahead of the patch's
  // first MFMA): piece u of this wave is DMA instruction wid + 4u; a patch adds a scalar offset, and for X the image-border test.
  constexpr int XU = X_DMA / 4, YU = Y_DMA / 4;
  int x_iy[XU], x_ix[XU];            // input row / column of the piece's pixel relative to the patch's input origin
  int x_rel[XU];                     // its byte offset relative to that origin (+ channel slice, swizzled chunk)
  int x_sc[XU];                      // the 8-channel chunk of the slice that lands in this lane's 16-byte slot (normalise on load)
  uint32_t y_rel[YU];
  const int w_in = ST * a.w, h_in = ST * a.h;
#pragma unroll
  for (int u = 0; u < XU; ++u) {
    const int hp = (wid + 4 * u) * X_PIX_PER_DMA + lane / XC8;
    int iy, ix, col;
    bool valid;
    if constexpr (ST == 2) {
      const int plane = hp / WH_S2_PP, rem = hp - plane * WH_S2_PP;
      const int pi = rem / WH_S2_PW, pj = rem - pi * WH_S2_PW;
      iy = 2 * pi + (plane >> 1); ix = 2 * pj + (plane & 1);
      valid = rem < 9 * WH_S2_PW;
      col = pj;
    } else {
      const int hy = hp / WH_HW, hx = hp - hy * WH_HW;
      iy = hy - PT; ix = hx - PL;
      valid = hp < HPIX;
      col = hx;
    }
    const int sc = (lane % XC8) ^ wh_swzx<XC8>(col);
    x_iy[u] = valid ? iy : (1 << 24);          // never inside an image
    x_ix[u] = ix;
    x_rel[u] = ((iy * w_in + ix) * a.ldx + ci0) * 2 + sc * 16;
    x_sc[u] = sc;
  }
  if (nol && tid < CI) {
    coef[(tid >> 3) * 16 + (tid & 7)] = a.nol_scale[ci0 + tid];
    coef[(tid >> 3) * 16 + 8 + (tid & 7)] = a.nol_shift[ci0 + tid];
  }
#pragma unroll
  for (int u = 0; u < YU; ++u) {
    const int p = (wid + 4 * u) * Y_PIX_PER_DMA + lane / YC8;       // patch pixel 0..127
    const int ty = p >> 4, tx = p & 15;
    const int sc = (lane % YC8) ^ wh_swzx<YC8>(tx);
    y_rel[u] = (uint32_t)(((ty * a.w + tx) * a.lddy + co0) * 2 + sc * 16);
  }

  auto issue = [&](int patch, int stage) {
    const int img = patch / per_img, pr = patch - img * per_img;
    const int y0 = (pr / a.patches_x) * WH_PH, x0 = (pr % a.patches_x) * WH_PW;
    const uint32_t xs = (uint32_t)(img * h_in * w_in) * (uint32_t)(a.ldx * 2);   // a.h, a.w = OUTPUT map (input = ST x)
    const uint32_t ys = (uint32_t)(img * a.h * a.w) * (uint32_t)(a.lddy * 2) + (uint32_t)((y0 * a.w + x0) * a.lddy * 2);
    const int py = ST * y0, px = ST * x0;
    const int poff = (py * w_in + px) * a.ldx * 2;
    const uint32_t lds_stage = lds_base + (uint32_t)(stage * STAGE + wid * 1024);
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const bool ok = ((unsigned)(py + x_iy[u]) < (unsigned)h_in) && ((unsigned)(px + x_ix[u]) < (unsigned)w_in);
      const uint32_t vo = ok ? (uint32_t)(x_rel[u] + poff) : OOB;
      wh_dma16(xr, __builtin_amdgcn_readfirstlane(lds_stage + (uint32_t)(u * 4096)), vo, xs);
    }
#pragma unroll
    for (int u = 0; u < YU; ++u)
      wh_dma16(yr, __builtin_amdgcn_readfirstlane(lds_stage + (uint32_t)(X_BYTES + u * 4096)), y_rel[u], ys);
  };

  f32x4_t acc[NTAP][TNT];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int j = 0; j < TNT; ++j) acc[t][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transpose-read geometry of this lane: 16-lane group g covers k = 8g..8g+7 of a 32-pixel k-step
  // (= 2 patch rows x 16 px): row g>>1, x = (g&1)*8 + (i>>2) (+4 for the second read); lane i <-> channel i
  const int i16 = lane & 15, g = lane >> 4;
  const int k_row = g >> 1, k_x = (g & 1) * 8 + (i16 >> 2), ch4 = (i16 & 3) * 4;
  const char* lds_c = (const char*)smem;
  // per-lane byte offsets inside a stage (see wh_swzx): X^T piece for horizontal tap offset e and first / second read (+4 pixels),
  // dY^T piece of tile j; everything else of a read's address is a compile-time offset
  constexpr int KXE = ST == 2 ? 2 : KW;                      // horizontal offsets inside a plane / the halo: kx >> 1 or kx
  const int lx = k_row * (ST == 2 ? WH_S2_PW : WH_HW) + k_x, ly = k_row * WH_PW + k_x;
  uint32_t xb[KXE][2], yb[TNT][2];
#pragma unroll
  for (int e = 0; e < KXE; ++e)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      xb[e][h] = (uint32_t)((lx * XC8 + (((wc * 16 + ch4) >> 3) ^ wh_swzx<XC8>(k_x + e + 4 * h))) * 16 + (ch4 & 4) * 2);
#pragma unroll
  for (int j = 0; j < TNT; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      yb[j][h] = (uint32_t)(X_BYTES + (ly * YC8 + (((((wn * TNT + j) * 16) + ch4) >> 3) ^ wh_swzx<YC8>(k_x + 4 * h))) * 16 + (ch4 & 4) * 2);

  uint4 af_keep = make_uint4(0, 0, 0, 0);
  const int n_mine = (bx < a.n_patches) ? (a.n_patches - bx + G - 1) / G : 0;
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < n_mine) issue(bx + t * G, t);
  int stage = 0;
  for (int it = 0; it < n_mine; ++it) {
    // patches it .. min(it+NS-2, n_mine-1) are in flight; patch `it` must have landed (this wave's share, then everyone's)
    constexpr int ABL = IMM_WH_ABLATE;
    if (!(ABL & 16) || it == 0) {
      if (it + NS - 2 < n_mine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (nol) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (first patch: the coefficient table)
      __builtin_amdgcn_s_barrier();
    }
    if (it + NS - 1 < n_mine && !(ABL & 1)) {      // into the stage patch it-1 occupied: every wave is past its reads (the barrier above)
      int ns = stage + NS - 1; if (ns >= NS) ns -= NS;
      issue(bx + (it + NS - 1) * G, ns);
    }
    const char* Xl = lds_c + stage * STAGE;     // (the dY patch follows the X halo inside the stage: yb includes X_BYTES)
    if (nol) {
      // affine + ReLU of this patch's X halo, in place: every thread transforms the 16-byte pieces its own DMA instructions wrote
      // (pixel and channel chunk known); pieces outside the image were zero-filled and stay zero
      const int patch_ = bx + it * G;
      const int img_ = patch_ / per_img, pr_ = patch_ - img_ * per_img;
      const int py_ = ST * (pr_ / a.patches_x) * WH_PH, px_ = ST * (pr_ % a.patches_x) * WH_PW;
      char* Xw = (char*)smem + stage * STAGE + wid * 1024 + lane * 16;
      const bool nrelu = a.nol_relu != 0;
#pragma unroll
      for (int u = 0; u < XU; ++u) {
        if (((unsigned)(py_ + x_iy[u]) < (unsigned)h_in) && ((unsigned)(px_ + x_ix[u]) < (unsigned)w_in)) {
          uint4* slot = (uint4*)(Xw + u * 4096);
          float f[8];
          unpack8<ET>(*slot, f);
          const float4* cf = (const float4*)(coef + x_sc[u] * 16);
          const float4 s0 = cf[0], s1 = cf[1], h0 = cf[2], h1 = cf[3];
          const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
          const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            f[e] = f[e] * sc[e] + sh[e];                   // the arithmetic of bn_apply_fused_kernel (elementwise.hip)
            if (nrelu) f[e] = fmaxf(f[e], 0.f);
          }
          *slot = pack8<ET>(f);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }

    // dY^T fragments of this wave's co tiles for the 4 k-steps of the patch: reused by all nine taps
    uint4 bfr[4][TNT];
    if (!(ABL & 4) || it == 0)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int j = 0; j < TNT; ++j) {
        // pixel = ly + ks * 32 (+ 4)
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Xl + yb[j][0] + ks * 32 * YC8 * 16));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Xl + yb[j][1] + (ks * 32 + 4) * YC8 * 16));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        bfr[ks][j] = make_uint4(l2.x, l2.y, h2.x, h2.y);
      }
#pragma unroll
    for (int tap = 0; tap < NTAP; ++tap) {
      const int ky = tap / KW, kx = tap % KW;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // pixel = lx (this lane's part) + cp (compile-time part); its column = k_x + e (+ 4)
        const int e = ST == 2 ? (kx >> 1) : kx;
        const int cp = ST == 2 ? ((ky & 1) * 2 + (kx & 1)) * WH_S2_PP + (ks * 2 + (ky >> 1)) * WH_S2_PW + (kx >> 1)
                               : (ks * 2 + ky) * WH_HW + kx;
        uint4 af;
        if (!(ABL & 2) || (it == 0 && tap == 0 && ks == 0)) {
          const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Xl + xb[e][0] + cp * XC8 * 16));
          const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(Xl + xb[e][1] + (cp + 4) * XC8 * 16));
          const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
          af = make_uint4(l2.x, l2.y, h2.x, h2.y);
          af_keep = af;
        } else af = af_keep;
        if (!(ABL & 8))
#pragma unroll
        for (int j = 0; j < TNT; ++j) acc[tap][j] = ET::mfma(bfr[ks][j], af, acc[tap][j]);   // D[n][c]
      }
    }
    if (++stage == NS) stage = 0;
  }

  // lane holds D[n = 4*(lane>>4)+r][c = lane&15] -> slab[split][kk = tap*ci + ci0 + wc*16 + c][co0 + n .. +3]
  float* out = a.slab + (int64_t)bx * a.kpad * a.co;
#pragma unroll
  for (int tap = 0; tap < NTAP; ++tap) {
    const int kk = tap * a.ci_total + ci0 + wc * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < TNT; ++j) {
      const int n = co0 + (wn * TNT + j) * 16 + 4 * (lane >> 4);
      float* op = out + (int64_t)kk * a.co + n;
      if (IMM_WH_ABLATE & 32) {
        if (acc[tap][j][0] == 123456.789f) op[0] = acc[tap][j][1] + acc[tap][j][2] + acc[tap][j][3];
      } else if (n + 3 < a.co && (a.co & 3) == 0) {
        *(float4*)op = make_float4(acc[tap][j][0], acc[tap][j][1], acc[tap][j][2], acc[tap][j][3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n + r < a.co) op[r] = acc[tap][j][r];
      }
    }
  }
}

template <typename ET, int CI, int CO, int NS, int KH = 3, int KW = 3, int ST = 1>
__global__ __launch_bounds__(256) void conv_wgrad_halo_kernel(const WgradHaloArgs a) {
  conv_wgrad_halo_body<ET, CI, CO, NS, KH, KW, ST>(a, blockIdx.x, gridDim.x, blockIdx.y, blockIdx.z);
}

// Several filter gradients with the same slice shape in ONE launch (imm_conv2d_wgrad_multi): workgroup b belongs to member g
// with first[g] <= b < first[g+1]; inside a member the workgroups are numbered split-fastest, (ci slice, co slice) slowest.
template <typename ET, int CI, int CO, int NS, int KH = 3, int KW = 3, int ST = 1>
__global__ __launch_bounds__(256) void conv_wgrad_halo_multi_kernel(const WgradHaloArgs* __restrict__ tab,
                                                                    const int* __restrict__ first, int n) {
  const int b = blockIdx.x;
  int g = 0;
  while (g + 1 < n && b >= first[g + 1]) ++g;
  const WgradHaloArgs a = tab[g];
  int r = b - first[g];
  const int bx = r % a.nsplit; r /= a.nsplit;
  const int by = r % a.nci, bz = r / a.nci;
  conv_wgrad_halo_body<ET, CI, CO, NS, KH, KW, ST>(a, bx, a.nsplit, by, bz);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int wh_num_cu() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t p; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// Slice plan of a layer: channel slice widths, number of slices and pixel splits.
struct WhPlan { int cs, ns, nci, nco, nsplit; bool k71, s2; };

static bool wh_plan(const imm_conv_desc* d, int lddy, WhPlan* pl) {
  const bool k33 = d->kh == 3 && d->kw == 3 && d->pad_t == 1 && d->pad_l == 1;
  // the tap-unrolled first encoder convolution (7x1 over the 32-channel unrolled image -> 32 channels, imm_model.py:190)
  const bool k71 = d->kh == 7 && d->kw == 1 && d->pad_t == 3 && d->pad_l == 0 && d->ci == 32 && lddy == 32 && d->co <= 32;
  pl->k71 = k71;
  // encoder conv_3: 3x3 stride 2, 32 -> 64 channels (whole filter = one slice; its halo + dY stage is 57 KB: two stages)
  const bool s2 = d->kh == 3 && d->kw == 3 && d->stride == 2 && d->pad_t == 0 && d->pad_l == 0 && d->updiv == 1 && d->ci == 32 &&
                  lddy == 64 && d->co > 32 && d->co <= 64 && d->hi == 2 * d->ho && d->wi == 2 * d->wo && d->ho % WH_PH == 0 &&
                  d->wo % WH_PW == 0 && d->ho * d->wo >= 32 * 32 && d->kpad == 9 * d->ci;
  pl->s2 = s2;
  if (s2) {
    const int64_t xb2 = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, yb2 = (int64_t)d->batch * d->ho * d->wo * lddy * 2;
    if (xb2 >= (1LL << 31) || yb2 >= (1LL << 31)) return false;
    pl->cs = 32; pl->ns = 64; pl->nci = 1; pl->nco = 1;
    const int n_patches2 = d->batch * (d->ho / WH_PH) * (d->wo / WH_PW);
    int ns2 = wh_num_cu();
    if (ns2 > n_patches2 / 4) ns2 = n_patches2 / 4;
    pl->nsplit = ns2 < 1 ? 1 : ns2;
    return true;
  }
  if ((!k33 && !k71) || d->stride != 1 || d->updiv != 1) return false;
  if (d->hi != d->ho || d->wi != d->wo || d->ho % WH_PH || d->wo % WH_PW) return false;
  if (d->kpad != d->kh * d->kw * d->ci || d->ci % 32) return false;
  const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, yb = (int64_t)d->batch * d->ho * d->wo * lddy * 2;
  if (xb >= (1LL << 31) || yb >= (1LL << 31)) return false;
  const int n_patches = d->batch * (d->ho / WH_PH) * (d->wo / WH_PW);
  int cs, ns;
  if (d->ci <= 64 && lddy <= 64 && (lddy == 32 || lddy == 64) && d->co <= lddy) {
    // the whole filter is one slice (dY rows staged at their padded width): encoder conv_2/4, renderer conv_6..8
    if (d->ho * d->wo < 64 * 64) return false;
    cs = d->ci; ns = lddy;
    pl->nci = 1; pl->nco = 1;
  } else {
    // Channel slices of deeper filters: X is fetched co/ns times and dY ci/cs times.  64 x 64 slices wherever both channel
    // counts allow (per launch they were always the faster ones: renderer conv_5 45 -> 31 us, 32^2 128->128 29.7 -> 20.3, 16^2
    // 256->256 23.2 -> 20.4; round 2 kept 32-slices because a layer alone needed 4x the pixel splits to fill the chip with
    // 64-slices and the slab traffic ate the gain — in a multi-problem launch the layers share the chip and the splits are few:
    // 3.474 -> 3.436 ms per step, round 3), 32 x 32 slices otherwise, and then only at >= 64x64 maps (below, the
    // transpose-read kernel with its larger tiles is faster).
    if (d->co % 32 || lddy < d->co) return false;
    if (d->ci % 64 == 0 && d->co % 64 == 0 && d->ho * d->wo >= 16 * 16) {
      cs = 64; ns = 64;
    } else {
      if (d->ho * d->wo < 64 * 64) return false;
      cs = 32; ns = 32;
    }
    pl->nci = d->ci / cs; pl->nco = d->co / ns;
  }
  pl->cs = cs; pl->ns = ns;
  const int blocks = pl->nci * pl->nco;
  // one workgroup per CU: two per CU (round 1) run the launches no faster and double the slab traffic of the whole-filter
  // layers (64->64 at 64x64: 75 MB of slabs each); measured 3.83 -> 3.80 ms per step (round 2, same box)
  constexpr int per_cu = 1;
  int grid = per_cu * wh_num_cu();
  if (blocks > 1) grid = wh_num_cu();                  // sliced layers: slab bytes = nsplit x |dW|, keep nsplit small
  int nsplit = (grid + blocks - 1) / blocks;
  const int min_patches = blocks > 1 ? 2 : 4;          // patches per workgroup that amortise its slab write
  if (nsplit > n_patches / min_patches) nsplit = n_patches / min_patches;
  if (nsplit < 1) nsplit = 1;
  pl->nsplit = nsplit;
  return true;
}

bool imm_wgrad_halo_applicable(const imm_conv_desc* d, int lddy) {
  static const bool off = imm_conv_disabled("wgrad_halo");
  if (off) return false;
  WhPlan pl;
  return wh_plan(d, lddy, &pl);
}

// number of pixel splits == slab copies the caller must allocate / reduce
int imm_wgrad_halo_splits(const imm_conv_desc* d, int lddy) {
  WhPlan pl;
  return wh_plan(d, lddy, &pl) ? pl.nsplit : 0;
}

// Ring depth.  32 x 32 slices (20-25 KB per stage): 3 stages = 61-74 KB, so that TWO workgroups share a CU — their patches are
// short chains of 8-byte transpose reads and 4 MFMAs per fragment with one wave per SIMD, and a second resident workgroup fills
// the issue gaps (measured in the multi-problem launch: 74 -> 52 us for encoder conv_2 x 2 + renderer conv_8; 40 -> 36 us for the
// 7x1 layers).  The wider slices keep one workgroup per CU and the deepest ring that fits: 64x64 3 x 40 KB (two-per-CU with 2
// stages: 188 vs 187 us), 64x32 4 x 32 KB (two-per-CU with 2 stages: 46 vs 42 us).
template <int CI, int CO, int KH, int KW, int ST = 1>
struct WhRing {
  static constexpr int stage = (ST == 2 ? WH_S2_HP : WH_HP(KH, KW)) * CI * 2 + 128 * CO * 2;
  static constexpr int NS = ST == 2 ? 2 : (CI == 32 && CO == 32) ? 3 : (stage > 36 * 1024 ? 3 : 4);
  static constexpr int per_cu = (CI == 32 && CO == 32) ? 2 : 1;
};

template <typename ET, int CI, int CO, int KH = 3, int KW = 3, int ST = 1>
static void wh_launch_cfg(const WgradHaloArgs& a, dim3 grid, hipStream_t s) {
  constexpr int stage = WhRing<CI, CO, KH, KW, ST>::stage;
  constexpr int NS = WhRing<CI, CO, KH, KW, ST>::NS;
  constexpr int lds = NS * stage + CI * 8;             // + the normalise-on-load coefficient table
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_halo_kernel<ET, CI, CO, NS, KH, KW, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_halo_kernel<ET, CI, CO, NS, KH, KW, ST>), grid, dim3(256), lds, s, a);
}

static WgradHaloArgs wh_fill(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit,
                             const WhPlan& pl, const float* nol_scale = nullptr, const float* nol_shift = nullptr, int nol_relu = 0) {
  WgradHaloArgs a;
  a.nol_scale = nol_scale; a.nol_shift = nol_shift; a.nol_relu = nol_relu;
  a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.slab = slab;
  a.batch = d->batch; a.h = d->ho; a.w = d->wo; a.ldx = d->ldx; a.lddy = lddy; a.co = d->co; a.kpad = d->kpad;
  a.ci_total = d->ci;
  a.patches_x = d->wo / WH_PW; a.patches_y = d->ho / WH_PH;
  a.n_patches = d->batch * a.patches_x * a.patches_y;
  a.nsplit = nsplit; a.nci = pl.nci; a.nco = pl.nco;
  a.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  a.dy_bytes = (uint32_t)((int64_t)d->batch * d->ho * d->wo * lddy * 2);
  return a;
}

void imm_wgrad_halo_launch(int dtype, const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab,
                           int nsplit, hipStream_t s, const float* nol_scale, const float* nol_shift, int nol_relu) {
  WhPlan pl;
  (void)wh_plan(d, lddy, &pl);
  const WgradHaloArgs a = wh_fill(d, x, dy, lddy, slab, nsplit, pl, nol_scale, nol_shift, nol_relu);
  const dim3 grid(nsplit, pl.nci, pl.nco);
#define WH_GO(ET_) \
  do { \
    if (pl.s2) wh_launch_cfg<ET_, 32, 64, 3, 3, 2>(a, grid, s); \
    else if (pl.k71) wh_launch_cfg<ET_, 32, 32, 7, 1>(a, grid, s); \
    else if (pl.cs == 64 && pl.ns == 64) wh_launch_cfg<ET_, 64, 64>(a, grid, s); \
    else if (pl.cs == 64) wh_launch_cfg<ET_, 64, 32>(a, grid, s); \
    else if (pl.ns == 64) wh_launch_cfg<ET_, 32, 64>(a, grid, s); \
    else wh_launch_cfg<ET_, 32, 32>(a, grid, s); \
  } while (0)
  if (dtype == IMM_BF16) WH_GO(BF16); else WH_GO(F16);
#undef WH_GO
}

// ---- members of a multi-problem launch (imm_conv2d_wgrad_multi, conv_wgrad.hip) ---------------------------------------------
// variant = slice shape: 64*100+64, 64*100+32, 32*100+64, 32*100+32; 0 = this kernel does not take the layer
int imm_wgrad_halo_variant(const imm_conv_desc* d, int lddy) {
  WhPlan pl;
  return wh_plan(d, lddy, &pl) ? (pl.s2 ? 20000 : pl.k71 ? 7100 : 0) + pl.cs * 100 + pl.ns : 0;   // 7x1: 7100 + 3232; stride 2: 20000 + 3264
}
// workgroups per pixel split (channel-slice blocks) and the number of 8x16 patches of the layer
int imm_wgrad_halo_blocks(const imm_conv_desc* d, int lddy, int* n_patches) {
  WhPlan pl;
  if (!wh_plan(d, lddy, &pl)) return 0;
  if (n_patches) *n_patches = d->batch * (d->ho / WH_PH) * (d->wo / WH_PW);
  return pl.nci * pl.nco;
}
int imm_wgrad_halo_args_bytes() { return (int)sizeof(WgradHaloArgs); }
int imm_wgrad_halo_fill(const imm_conv_desc* d, const void* x, const void* dy, int lddy, float* slab, int nsplit, void* out,
                        int* steps, const float* nol_scale, const float* nol_shift, int nol_relu) {
  WhPlan pl;
  (void)wh_plan(d, lddy, &pl);
  const WgradHaloArgs a = wh_fill(d, x, dy, lddy, slab, nsplit, pl, nol_scale, nol_shift, nol_relu);
  memcpy(out, &a, sizeof(a));
  if (steps) *steps = (a.n_patches + nsplit - 1) / nsplit * 4 * (pl.cs / 16);     // ~ matrix work per workgroup
  return nsplit * pl.nci * pl.nco;
}

// resident workgroups per CU of a variant's kernel (the engine sizes a launch to one round of them)
int imm_wgrad_halo_per_cu(int variant) { return (variant == 3232 || variant == 7100 + 3232) ? 2 : 1; }

template <typename ET, int CI, int CO, int KH = 3, int KW = 3, int ST = 1>
static void wh_launch_multi_cfg(const WgradHaloArgs* tab, const int* first, int n, int blocks, hipStream_t s) {
  constexpr int stage = WhRing<CI, CO, KH, KW, ST>::stage;
  constexpr int NS = WhRing<CI, CO, KH, KW, ST>::NS;
  constexpr int lds = NS * stage + CI * 8;             // + the normalise-on-load coefficient table
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_halo_multi_kernel<ET, CI, CO, NS, KH, KW, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_wgrad_halo_multi_kernel<ET, CI, CO, NS, KH, KW, ST>), dim3(blocks), dim3(256), lds, s, tab, first, n);
}

void imm_wgrad_halo_launch_multi(int dtype, int variant, const void* tab_dev, const int* first_dev, int n, int blocks,
                                 hipStream_t s) {
  const WgradHaloArgs* t = (const WgradHaloArgs*)tab_dev;
#define WH_GO(ET_) \
  do { \
    if (variant == 20000 + 3264) wh_launch_multi_cfg<ET_, 32, 64, 3, 3, 2>(t, first_dev, n, blocks, s); \
    else if (variant == 7100 + 3232) wh_launch_multi_cfg<ET_, 32, 32, 7, 1>(t, first_dev, n, blocks, s); \
    else if (variant == 6464) wh_launch_multi_cfg<ET_, 64, 64>(t, first_dev, n, blocks, s); \
    else if (variant == 6432) wh_launch_multi_cfg<ET_, 64, 32>(t, first_dev, n, blocks, s); \
    else if (variant == 3264) wh_launch_multi_cfg<ET_, 32, 64>(t, first_dev, n, blocks, s); \
    else wh_launch_multi_cfg<ET_, 32, 32>(t, first_dev, n, blocks, s); \
  } while (0)
  if (dtype == IMM_BF16) WH_GO(BF16); else WH_GO(F16);
#undef WH_GO
}
