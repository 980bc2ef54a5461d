// conv_halo.hip — 3x3 stride-1 convolution for the high-resolution, low-channel layers (ci in {32,64}, co <= 64):
// encoder conv_2 / conv_4, renderer conv_6..8, VGG conv1_2, and the data gradients of the same layers.
//
// Why a second convolution kernel: in the im2col formulation every input pixel is gathered once per filter tap
// (9x) through L2; at 128x128 / 64x64 with 32-64 channels that re-gather traffic, not the matrix cores, sets the
// time (rocprofv3 PMC: 3.5x the algorithmic HBM bytes; tools/bench_conv.py ablation).  Here a persistent
// workgroup keeps ALL nine taps of the filter resident in LDS (<= 72 KB) and walks 8x16-pixel output patches:
// the 10x18-pixel input halo of a patch is DMA'd once (buffer_load ... lds, double buffered, zero fill outside
// the image through the out-of-range buffer offset) and the nine taps are just nine different LDS base
// addresses of the same tile.  HBM/L2 sees each input pixel 1.4x (halo overlap) instead of 9x.
//
// MFMA mapping: one 16-pixel patch row = one 16-row MFMA operand; lane (l&15) = pixel x, (l>>4) = 8-channel chunk.
// LDS images are [pixel or filter row][ci/8 chunks of 16 B], chunk XOR-swizzled on the DMA source side.
// Epilogue = bias / ReLU / ReLU-backward mask / 16-bit or f32 NHWC store; BN partial sums are accumulated in
// registers over all patches of the workgroup and written once (one partial row per workgroup).
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define HALO_PH 8                 // patch rows
#define HALO_PW 16                // patch cols (= MFMA operand rows)
#define HALO_HW (HALO_PW + 2)     // halo width 18
#define HALO_HP 192               // halo pixels (10*18 = 180) padded to a whole number of DMA instructions

__device__ __forceinline__ void halo_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// Chunk swizzle that keeps a ds_read_b128 of 16 CONSECUTIVE rows (any start offset: the nine taps shift the pixel
// index by 0/1/2 and by the halo pitch) x 4 chunks conflict-free.  16-byte slot of (row r, stored chunk c') is
// (r*C8 + c') mod 16; the instruction is served in 16-lane groups {rows 0-3,12-15 @ chunk c, rows 4-11 @ chunk c^1}.
//   C8 = 8 (128-B rows): rows of equal parity share 8 slots; XOR bits 1-2 of the chunk with (r>>1)&3: the four
//     rows {0,1,6,7}+q of one parity/chunk get 4 different values, bit 0 separates chunk c from c^1.
//   C8 = 4 (64-B rows): rows equal mod 4 share 4 slots; XOR bit 1 with (r>>2)&1: the two rows (r, r+12) resp.
//     (r+4, r+8) of a class differ, bit 0 again separates the two chunks.
template <int C8>
__device__ __forceinline__ int halo_swz(int row) { return C8 == 8 ? (((row >> 1) & 3) << 1) : (((row >> 2) & 1) << 1); }

struct HaloArgs {
  ConvArgs c;
  int n_patches, patches_x, patches_y;    // per image: patches_x * patches_y
  // NOL (normalise on load, round 4): x is the RAW 16-bit output y of the preceding convolution of a conv + batch-norm + ReLU block
  // (tf.layers.batch_normalization + relu, nn_utils.py:201-209); the input of THIS convolution is relu(scale[c] * y + shift[c]),
  // which is never stored: the affine + ReLU is applied once per halo pixel to the LDS tile after its DMA has landed (out-of-image
  // halo pixels stay zero: the zero padding is of the normalised tensor).  scale / shift: f32 [CI] from imm_bn_finalize.
  const float* nol_scale; const float* nol_shift; int nol_relu;
};

// S2: 3x3 STRIDE-2 forward (encoder conv_3: 32 -> 64 channels at 128^2 -> 64^2, imm_model.py:197; TF SAME on an even side pads
// bottom / right only).  An 8x16 output patch reads a 17x33 input halo; it is stored PARITY-DE-INTERLEAVED — four
// planes (row parity, column parity) of 9 x 17 pixels, padded to 160 — so that tap (ky,kx) of output pixel (y,x), source pixel
// (2y+ky, 2x+kx), is pixel (y + ky/2, x + kx/2) of plane (ky&1, kx&1): the 16 lanes of an MFMA operand row read 16 CONSECUTIVE
// LDS pixels exactly as in the stride-1 form (a stride of two 64-byte pixels would put all of them on the same banks).  The
// im2col kernel it replaces gathered every input pixel 2.25x through L2 and spent its 9 short K-steps in load latency
// (30 us per encoder for 4.8 GFLOP); here the halo arrives once by DMA, two patches ahead.
#define HALO_S2_PW 17            // plane width  (33 columns -> 17 even + 16 odd)
#define HALO_S2_PP 160           // plane pitch in pixels (9 x 17 = 153, padded)
#define HALO_S2_HP 640           // 4 planes
// S2D (round 4): the DATA GRADIENT of a 3x3 stride-2 SAME convolution whose whole flipped filter fits the LDS (encoder conv_3:
// dy 64 channels -> dx 32 channels at 128x128, imm_model.py:197) as one launch: four accumulator sets (the input-pixel parity
// classes, see conv_hdeep.hip S2D for the index algebra: tap t' = (ky', kx') of the mode-1 filter image accumulates into class
// (ky' & 1, kx' & 1) from halo offset (min(ky', 1), min(kx', 1))), scattered to pixel (2i + py, 2j + px) of dx.  The deep-K
// kernel serves the same operation for the wider layers; here the layer is HBM-bound (50 MB for 4.8 GFLOP) and a persistent
// workgroup with the filter resident and the next halo in flight is what it wants (hdeep S2D: 1024 one-tile workgroups in four
// rounds, 32 us).
template <typename ET, int CI, int BN, bool S2 = false, bool NOL = false, bool S2D = false>
__global__ __launch_bounds__(256) void conv_halo_kernel(const HaloArgs ha) {
  const ConvArgs& a = ha.c;
  static_assert(!S2D || (!S2 && !NOL), "stride-2 data gradient: plain stride-1 halo of dy");
  constexpr int NCLS = S2D ? 4 : 1;
  constexpr int HP = S2 ? HALO_S2_HP : HALO_HP;    // halo pixels per stage
  constexpr int C8 = CI / 8;                       // 16-byte chunks per pixel / filter row
  constexpr int KS = CI / 32;                      // MFMA k-steps per tap
  constexpr int WGM = (BN == 64) ? 2 : 4, WGN = 4 / WGM;
  constexpr int TM = 128 / WGM, TN = BN / WGN, MT = TM / 16, NT = TN / 16;
  constexpr int PIX_PER_DMA = 64 / C8;             // pixels (or filter rows) per 1-KB DMA instruction
  constexpr int HALO_DMA = HP / PIX_PER_DMA;       // instructions per halo tile (24 / 12; stride 2: 40)
  constexpr int W_DMA = 9 * BN / PIX_PER_DMA;      // instructions for the whole filter
  constexpr int W_U4 = 9 * BN * C8;                // uint4 of the filter image
  constexpr int H_U4 = HP * C8;                    // uint4 per halo stage
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [W_U4] filter | [2][H_U4] halo tiles | NOL: [C8][16] f32 coefficients

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  constexpr uint32_t OOB = 0x80000000u;
  const uint64_t xa = (uint64_t)a.x, wa = (uint64_t)a.wt;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t wr = {(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, a.wt_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;

  // ---- filter: all nine taps resident for the lifetime of the workgroup -----------------------------------
  // LDS row (tap*BN + n) holds Wt[n][tap*CI .. +CI) ; one DMA instruction = PIX_PER_DMA consecutive rows
  for (int i = wid; i < W_DMA; i += 4) {
    const int row = i * PIX_PER_DMA + lane / C8;          // tap*BN + n
    const int tap = row / BN, n = row - tap * BN;
    const int sc = (lane % C8) ^ halo_swz<C8>(row);
    const uint32_t vo = (n < a.co) ? (uint32_t)((n * a.kpad + tap * CI) * 2 + sc * 16) : OOB;
    halo_dma16(wr, __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(i * 1024)), vo, 0u);
  }

  // ---- normalise on load: per-chunk coefficient table + this thread's halo pieces (DMA instruction wid + 4k, 16 bytes each) ----
  constexpr int NPC = HALO_DMA / 4;
  static_assert(HALO_DMA % 4 == 0, "every wave issues the same number of halo pieces");
  float* coef = (float*)(smem + W_U4 + 2 * H_U4);          // [C8][16]: scale[8] | shift[8] of an 8-channel chunk
  int p_dy[NOL ? NPC : 1], p_dx[NOL ? NPC : 1], p_sc[NOL ? NPC : 1];
  if constexpr (NOL) {
    if (tid < CI) {
      coef[(tid >> 3) * 16 + (tid & 7)] = ha.nol_scale[tid];
      coef[(tid >> 3) * 16 + 8 + (tid & 7)] = ha.nol_shift[tid];
    }
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int hp = (wid + 4 * k) * PIX_PER_DMA + lane / C8;
      bool valid;
      if constexpr (S2) {
        const int plane = hp / HALO_S2_PP, rem = hp - plane * HALO_S2_PP;
        const int pi = rem / HALO_S2_PW, pj = rem - pi * HALO_S2_PW;
        p_dy[k] = 2 * pi + (plane >> 1); p_dx[k] = 2 * pj + (plane & 1);       // relative to input pixel (2 y0, 2 x0)
        valid = rem < 9 * HALO_S2_PW;
      } else {
        const int hy = hp / HALO_HW, hx = hp - hy * HALO_HW;
        p_dy[k] = hy - 1; p_dx[k] = hx - 1;                                    // relative to input pixel (y0, x0)
        valid = hp < (HALO_PH + 2) * HALO_HW;
      }
      if (!valid) p_dy[k] = 1 << 24;                                           // padding slot of the tile: never inside an image
      p_sc[k] = (lane % C8) ^ halo_swz<C8>(hp);
    }
  }

  // ---- halo loader ----------------------------------------------------------------------------------------
  const int per_img = ha.patches_x * ha.patches_y;
  auto issue_halo = [&](int patch, int stage) {
    const int img = patch / per_img, pr = patch - img * per_img;
    const int y0 = (pr / ha.patches_x) * HALO_PH, x0 = (pr % ha.patches_x) * HALO_PW;
    const uint32_t soff = (uint32_t)(img * a.hi * a.wi) * (uint32_t)(a.ldx * 2);
    for (int i = wid; i < HALO_DMA; i += 4) {
      const int hp = i * PIX_PER_DMA + lane / C8;        // halo pixel index 0..191 (stride 2: 0..639)
      int iy, ix;
      bool ok;
      if constexpr (S2) {
        const int plane = hp / HALO_S2_PP, rem = hp - plane * HALO_S2_PP;
        const int pi = rem / HALO_S2_PW, pj = rem - pi * HALO_S2_PW;
        iy = 2 * (y0 + pi) + (plane >> 1); ix = 2 * (x0 + pj) + (plane & 1);
        ok = rem < 9 * HALO_S2_PW && iy < a.hi && ix < a.wi;
      } else {
        const int hy = hp / HALO_HW, hx = hp - hy * HALO_HW;
        iy = y0 - 1 + hy; ix = x0 - 1 + hx;
        ok = (hp < (HALO_PH + 2) * HALO_HW) && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
      }
      const int sc = (lane % C8) ^ halo_swz<C8>(hp);
      const uint32_t vo = ok ? (uint32_t)((iy * a.wi + ix) * a.ldx * 2 + sc * 16) : OOB;
      halo_dma16(xr, __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((W_U4 + stage * H_U4) * 16 + i * 1024)), vo, soff);
    }
  };

  // ---- epilogue state -------------------------------------------------------------------------------------
  const bool f_bias = a.flags & IMM_CONV_BIAS, f_relu = a.flags & IMM_CONV_RELU;
  const bool f_stats = a.flags & IMM_CONV_STATS, f_mask = a.flags & IMM_CONV_MASK;
  const bool f_f32 = a.flags & IMM_CONV_OUT_F32;
  float s1[NT][4], s2[NT][4], bv[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[j][r] = 0.f; s2[j][r] = 0.f;
      const int n = wn * TN + j * 16 + 4 * (lane >> 4) + r;
      bv[j][r] = (f_bias && n < a.co) ? a.bias[n] : 0.f;
    }

  int patch = blockIdx.x, stage = 0;
  if (patch < ha.n_patches) issue_halo(patch, 0);
  const int frow = lane & 15, fchunk = lane >> 4;
  const uint4* Wl = smem;
  // Each wave issues, in order: [halo DMA of the next patch] ... [MT*NT output stores of this patch].  vmcnt retires
  // in order, so at the top of the next iteration "at most MT*NT operations outstanding" means the DMA has landed
  // while the stores (HBM write latency) are still allowed in flight.  Only valid when every tile is ONE store
  // instruction (all 4 channels in range, no mask loads in between); otherwise drain completely.
  const bool counted = (a.co % 4 == 0) && (a.co == BN) && !f_mask;
  constexpr int N_STORES = MT * NT * NCLS;                 // store instructions per wave and patch
  bool first = true;
  for (; patch < ha.n_patches; patch += gridDim.x) {
    if (counted && !first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_STORES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the filter / this patch's halo landed
    first = false;
    if constexpr (NOL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (first patch: the coefficient table)
    __builtin_amdgcn_s_barrier();                          // => everyone's share; everyone is done with the other stage
    if (patch + (int)gridDim.x < ha.n_patches && !(a.flags & IMM_DBG_NO_GLOAD)) issue_halo(patch + gridDim.x, stage ^ 1);
    const uint4* Hl = smem + W_U4 + stage * H_U4;
    if constexpr (NOL) {
      // affine + ReLU of this patch's halo, in place: every thread transforms the 16-byte pieces its own DMA instructions wrote
      // (it knows their pixel and channel chunk); pieces outside the image were zero-filled and stay zero
      const int img_ = patch / per_img, pr_ = patch - img_ * per_img;
      const int by = (S2 ? 2 : 1) * (pr_ / ha.patches_x) * HALO_PH, bx = (S2 ? 2 : 1) * (pr_ % ha.patches_x) * HALO_PW;
      uint4* Hw = smem + W_U4 + stage * H_U4;
      const bool nrelu = ha.nol_relu != 0;
#pragma unroll
      for (int k = 0; k < NPC; ++k) {
        if (((unsigned)(by + p_dy[k]) < (unsigned)a.hi) && ((unsigned)(bx + p_dx[k]) < (unsigned)a.wi)) {
          uint4* slot = Hw + (wid + 4 * k) * 64 + lane;
          float f[8];
          unpack8<ET>(*slot, f);
          const float4* cf = (const float4*)(coef + p_sc[k] * 16);
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
      __builtin_amdgcn_s_barrier();                        // the normalised tile is complete
    }

    f32x4_t acc[NCLS][MT][NT];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[c][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (!(a.flags & IMM_DBG_NO_MFMA))
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
      const int cls = S2D ? ((ky & 1) * 2 + (kx & 1)) : 0;               // parity class this tap feeds
      const int rky = S2D ? (ky > 0 ? 1 : 0) : ky, rkx = S2D ? (kx > 0 ? 1 : 0) : kx;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        uint4 af[MT], bf[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int hp = S2 ? ((ky & 1) * 2 + (kx & 1)) * HALO_S2_PP + (wm * MT + i + (ky >> 1)) * HALO_S2_PW + frow + (kx >> 1)
                            : (wm * MT + i + rky) * HALO_HW + frow + rkx;
          af[i] = Hl[hp * C8 + ((ks * 4 + fchunk) ^ halo_swz<C8>(hp))];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int row = tap * BN + wn * TN + j * 16 + frow;
          bf[j] = Wl[row * C8 + ((ks * 4 + fchunk) ^ halo_swz<C8>(row))];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[cls][i][j] = ET::mfma(bf[j], af[i], acc[cls][i][j]);   // D[n][pixel]
      }
    }

    if (a.flags & IMM_DBG_NO_EPILOGUE) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) t += acc[0][i][j][0] + acc[0][i][j][1] + acc[0][i][j][2] + acc[0][i][j][3];
      if (t == 123456.789f) ((float*)a.y)[0] = t;
      stage ^= 1;
      continue;
    }
    if constexpr (S2D) {
      // class (py, px) of tile pixel (row wm*MT + i, column lane & 15) -> pixel (2 row + py, 2 column + px) of the [2 ho, 2 wo] map
      const int img2 = patch / per_img, pr2 = patch - img2 * per_img;
      const int ty0 = (pr2 / ha.patches_x) * HALO_PH, tx0 = (pr2 % ha.patches_x) * HALO_PW;
      const int W2 = 2 * a.wo;
#pragma unroll
      for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int64_t m = ((int64_t)img2 * (2 * a.ho) + 2 * (ty0 + wm * MT + i) + (c >> 1)) * W2 + 2 * (tx0 + (lane & 15)) + (c & 1);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = wn * TN + j * 16 + 4 * (lane >> 4);
            uint16_t* yp = (uint16_t*)a.y + m * a.ldy + n;
            const uint2 v2 = make_uint2(ET::pack2(acc[c][i][j][0], acc[c][i][j][1]), ET::pack2(acc[c][i][j][2], acc[c][i][j][3]));
            if (n + 3 < a.co) *(uint2*)yp = v2;
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r) if (n + r < a.co) yp[r] = ET::from_f32(acc[c][i][j][r]);
            }
          }
        }
      stage ^= 1;
      continue;
    }
    // ---- store this patch: lane = pixel (wm*MT+i, lane&15), 4 consecutive channels -------------------------
    const int img = patch / per_img, pr = patch - img * per_img;
    const int y0 = (pr / ha.patches_x) * HALO_PH, x0 = (pr % ha.patches_x) * HALO_PW;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int yy = y0 + wm * MT + i, xx = x0 + (lane & 15);
      const bool mok = yy < a.ho && xx < a.wo;
      const int64_t m = ((int64_t)img * a.ho + yy) * a.wo + xx;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = wn * TN + j * 16 + 4 * (lane >> 4);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[0][i][j][r] + bv[j][r];
          if (f_relu) v[r] = fmaxf(v[r], 0.f);
        }
        if (!mok) continue;
        if (f_mask) {
          const uint16_t* mp = a.mask + m * a.ldmask + n;
          if (n + 3 < a.co && (a.ldmask & 3) == 0) {
            const uint2 mw = *(const uint2*)mp;
            const uint16_t mh[4] = {(uint16_t)(mw.x & 0xffffu), (uint16_t)(mw.x >> 16), (uint16_t)(mw.y & 0xffffu), (uint16_t)(mw.y >> 16)};
#pragma unroll
            for (int r = 0; r < 4; ++r) if (!(ET::to_f32(mh[r]) > 0.f)) v[r] = 0.f;
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.co && !(ET::to_f32(mp[r]) > 0.f)) v[r] = 0.f;
          }
        }
        if (f_stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
        }
        if (f_f32) {
          float* yp = (float*)a.y + m * a.ldy + n;
          if (n + 3 < a.co) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.co) yp[r] = v[r];
          }
        } else {
          uint16_t* yp = (uint16_t*)a.y + m * a.ldy + n;
          if (n + 3 < a.co) *(uint2*)yp = make_uint2(ET::pack2(v[0], v[1]), ET::pack2(v[2], v[3]));
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.co) yp[r] = ET::from_f32(v[r]);
          }
        }
      }
    }
    stage ^= 1;
  }

  if (f_stats) {
    // per-workgroup partial sums: 16 pixel lanes -> wave rows -> one row of stats per workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = (float*)(smem + W_U4);     // halo stages are free now: [WGM][2][BN]
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1[j][r] += __shfl_xor(s1[j][r], o, 64);
          s2[j][r] += __shfl_xor(s2[j][r], o, 64);
        }
      }
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = wn * TN + j * 16 + 4 * (lane >> 4) + r;
          red[(wm * 2 + 0) * BN + nl] = s1[j][r];
          red[(wm * 2 + 1) * BN + nl] = s2[j][r];
        }
    }
    __syncthreads();
    if (tid < BN && tid < a.co) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
      a.stats[((int64_t)blockIdx.x * 2 + 0) * a.co + tid] = t1;
      a.stats[((int64_t)blockIdx.x * 2 + 1) * a.co + tid] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int g_halo_cu = 0;
static int halo_num_cu() {
  if (g_halo_cu == 0) {
    hipDeviceProp_t p; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_halo_cu = p.multiProcessorCount;
    if (g_halo_cu <= 0) g_halo_cu = 256;
  }
  return imm_limit_cus(g_halo_cu);
}

static bool halo_is_s2(const imm_conv_desc* d) {
  return d->kh == 3 && d->kw == 3 && d->stride == 2 && d->updiv == 1 && d->pad_t == 0 && d->pad_l == 0 && d->ci == 32 &&
         d->co > 32 && d->co <= 64 && d->hi == 2 * d->ho && d->wi == 2 * d->wo && d->ho % HALO_PH == 0 && d->wo % HALO_PW == 0 &&
         d->ho * d->wo >= 32 * 32 && d->out_scale <= 1 && !(d->flags & (0xf00 | IMM_CONV_MASK)) && d->kpad == 9 * d->ci;
}

bool imm_halo_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("halo");
  if (off) return false;
  if (halo_is_s2(d)) return (int64_t)d->batch * d->hi * d->wi * d->ldx * 2 < (1LL << 31);
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->updiv != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
  if (d->ci != 32 && d->ci != 64) return false;
  if (d->out_scale > 1) return false;
  if (d->co > 64) return false;
  if (d->hi != d->ho || d->wi != d->wo || d->ho % HALO_PH || d->wo % HALO_PW) return false;
  if (d->ho * d->wo < 64 * 64) return false;      // deep layers: the im2col kernels have more parallelism
  if (d->flags & 0xf00) return false;   // debug ablation bits select the im2col kernel
  const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2;
  return xb < (1LL << 31);
}

static int halo_bn(int co) { return co > 32 ? 64 : co > 16 ? 32 : 16; }
static size_t halo_lds(int ci, int bn, bool s2 = false, bool nol = false) {
  return (size_t)(9 * bn * (ci / 8) + 2 * (s2 ? HALO_S2_HP : HALO_HP) * (ci / 8)) * 16 + (nol ? (size_t)(ci / 8) * 64 : 0);
}

int imm_halo_grid(const imm_conv_desc* d) {
  const int n_patches = d->batch * (d->ho / HALO_PH) * (d->wo / HALO_PW);
  const size_t lds = halo_lds(d->ci, halo_bn(d->co), halo_is_s2(d), true);     // (with the normalise-on-load table: one grid for both forms)
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  const int grid = halo_num_cu() * per_cu;
  return n_patches < grid ? n_patches : grid;
}

template <typename ET, int CI, int BN, bool S2 = false, bool NOL = false, bool S2D = false>
static void halo_launch_cfg(const HaloArgs& ha, int grid, hipStream_t s) {
  const size_t lds = halo_lds(CI, BN, S2, NOL);
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    (void)hipFuncSetAttribute((const void*)conv_halo_kernel<ET, CI, BN, S2, NOL, S2D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_halo_kernel<ET, CI, BN, S2, NOL, S2D>), dim3(grid), dim3(256), lds, s, ha);
}

template <typename ET>
static void halo_launch(const imm_conv_desc* d, const ConvArgs& a, hipStream_t s, const float* nol_scale = nullptr,
                        const float* nol_shift = nullptr, int nol_relu = 0) {
  HaloArgs ha;
  ha.c = a;
  ha.nol_scale = nol_scale; ha.nol_shift = nol_shift; ha.nol_relu = nol_relu;
  ha.patches_x = d->wo / HALO_PW; ha.patches_y = d->ho / HALO_PH;
  ha.n_patches = d->batch * ha.patches_x * ha.patches_y;
  ha.c.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  const int grid = imm_halo_grid(d), bn = halo_bn(d->co);
  if (nol_scale) {
    // normalise-on-load instantiations: the shapes whose input is a batch-norm output in the encoders / the renderer
    if (halo_is_s2(d)) halo_launch_cfg<ET, 32, 64, true, true>(ha, grid, s);
    else if (d->ci == 64 && bn == 64) halo_launch_cfg<ET, 64, 64, false, true>(ha, grid, s);
    else if (d->ci == 64 && bn == 32) halo_launch_cfg<ET, 64, 32, false, true>(ha, grid, s);
    else if (d->ci == 32 && bn == 64) halo_launch_cfg<ET, 32, 64, false, true>(ha, grid, s);
    else if (d->ci == 32 && bn == 32) halo_launch_cfg<ET, 32, 32, false, true>(ha, grid, s);
    else halo_launch_cfg<ET, 32, 16, false, true>(ha, grid, s);
    return;
  }
  if (halo_is_s2(d)) { halo_launch_cfg<ET, 32, 64, true>(ha, grid, s); return; }
  if (d->ci == 64) {
    if (bn == 64) halo_launch_cfg<ET, 64, 64>(ha, grid, s);
    else if (bn == 32) halo_launch_cfg<ET, 64, 32>(ha, grid, s);
    else halo_launch_cfg<ET, 64, 16>(ha, grid, s);
  } else {
    if (bn == 64) halo_launch_cfg<ET, 32, 64>(ha, grid, s);
    else if (bn == 32) halo_launch_cfg<ET, 32, 32>(ha, grid, s);
    else halo_launch_cfg<ET, 32, 16>(ha, grid, s);
  }
}

void imm_conv_halo_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  if (dtype == IMM_BF16) halo_launch<BF16>(d, a, s);
  else halo_launch<F16>(d, a, s);
}

// normalise on load: served for every shape this kernel takes except (ci = 64, co <= 16), which no layer has
bool imm_halo_nol_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("nol");
  if (off || !imm_halo_applicable(d)) return false;
  return halo_is_s2(d) || !(d->ci == 64 && halo_bn(d->co) == 16);
}

void imm_conv_halo_nol_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, const float* scale, const float* shift, int relu,
                              hipStream_t s) {
  if (dtype == IMM_BF16) halo_launch<BF16>(d, a, s, scale, shift, relu);
  else halo_launch<F16>(d, a, s, scale, shift, relu);
}

// ---- stride-2 data gradient with the whole flipped filter in LDS (S2D): dy with 64 channels -> dx with <= 32 channels ----
// d = the class grid as a same-size 3x3 convolution of dy (see imm_hdeep_s2d_applicable); wt = imm_pack_weights mode 1 image.
bool imm_halo_s2d_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("s2d_halo");
  if (off) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->updiv != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
  if (d->ci != 64 || d->ldx != 64 || d->kpad != 9 * 64 || d->co > 32 || d->co < 4 || d->co % 4 || d->ldy % 4 || d->ldy < d->co) return false;
  if (d->hi != d->ho || d->wi != d->wo || d->ho % HALO_PH || d->wo % HALO_PW || d->ho * d->wo < 32 * 32) return false;
  if (d->flags || d->out_scale > 1) return false;
  return (int64_t)d->batch * d->hi * d->wi * d->ldx * 2 < (1LL << 31);
}

void imm_conv_halo_s2d_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  HaloArgs ha;
  ha.c = a;
  ha.nol_scale = nullptr; ha.nol_shift = nullptr; ha.nol_relu = 0;
  ha.patches_x = d->wo / HALO_PW; ha.patches_y = d->ho / HALO_PH;
  ha.n_patches = d->batch * ha.patches_x * ha.patches_y;
  ha.c.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  const int cus = halo_num_cu();                       // 84 KB of LDS: one persistent workgroup per CU
  const int grid = ha.n_patches < cus ? ha.n_patches : cus;
  if (dtype == IMM_BF16) halo_launch_cfg<BF16, 64, 32, false, false, true>(ha, grid, s);
  else halo_launch_cfg<F16, 64, 32, false, false, true>(ha, grid, s);
}
