// conv_hdeep.hip — 3x3 stride-1 convolution for the DEEP layers (ci % 64 == 0, co % 64 == 0, 16x16 .. 64x64 maps):
// VGG conv2_1 .. conv4_3 and their data gradients (imm/models/selfsup/vgg16.py:349-362), encoder conv_6 / conv_8 and
// renderer conv_2 .. conv_5 when the grid is large enough (imm/models/imm_model.py:213-241, :258-300).
//
// Why not the im2col kernel (conv_igemm64.hip) for these: rocprofv3 PMC on VGG conv4_2 shows its waves parked on
// s_waitcnt/s_barrier for 30 % of their cycles with the matrix pipe 49 % busy — every 128x128x64 K-tile pulls 32 KB
// through L2 for 2.1 MFLOP (65 FLOP/B; at the MFMA peak that would be 38 TB/s against 34 TB/s of L2), and the input
// pixel tile is fetched nine times, once per filter tap.  Here a workgroup owns a 16x16-pixel output patch x BN
// channels and walks K as (64-channel slice) x (9 taps): the 18x18-pixel input halo of a slice is DMA'd into LDS ONCE
// and all nine taps read it at shifted addresses (immediate offsets), only the 64 x BN filter slice of each tap
// streams through a 4-stage ring.  L2 -> LDS traffic per MFLOP drops 3x (204 FLOP/B at BN = 128).
//
// 512 threads = 8 waves (2 per SIMD): wave (wm, wn) owns patch rows 4wm..4wm+3 x BN/2 channels.  All loop DMA is
// issued from inline asm with exactly (BN/64 + 1) instructions per wave and tap (halo pieces of the next slice ride on
// taps 0-5, the rest are no-op pieces into a dump slot), so one counted s_waitcnt vmcnt per tap is exact.
#include "conv_common.h"
#include <stdlib.h>
#include <stdio.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define HD_PW 16                         // patch width (= MFMA operand rows)
#define HD_HW (HD_PW + 2)                // halo width 18
// patch height PH = 16 (8 waves, 512 threads) or 8 (4 waves, 256 threads: half the tile, for grids that would
// otherwise leave CUs idle; 78 KB of LDS at BN = 64, so two workgroups share a CU and cover each other's prologue,
// epilogue and barriers)
#define HD_SLOTS(PH) (((PH) + 2) * HD_HW)            // halo pixels: 324 / 180
#define HD_HINSTR(PH) ((HD_SLOTS(PH) + 7) / 8)       // DMA instructions (8 pixels x 128 B each): 41 / 23
#define HD_HSTAGE(PH) (HD_HINSTR(PH) * 64)           // uint4 per halo stage
#define HD_OOB 0x80000000u

struct HdArgs {
  ConvArgs c;
  int n_patches, patches_x, patches_y;
  int n_wg;                              // n_patches * n_nblk
  int n_img;                             // batch (MAP8 tiles hold two images)
  unsigned long long* prof;              // IMM_HDEEP_PROF=1: wall-clock stamps of workgroup 0 / wave 0 (diagnosis only)
};

__device__ __forceinline__ void hd_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// halo pixel -> chunk swizzle (same as conv_halo2.hip): a ds_read_b128 of 16 consecutive pixels is conflict-free
__device__ __forceinline__ int hd_swz(int hx) { return ((hx >> 1) & 3) << 1; }
// Filter stage in LDS: row = output channel of the BN block (128 B = 8 chunks), chunk XOR-swizzled by hd_bswz.  The MFMA
// B-operand row rho of tile j is NOT channel 16j + rho but channel (rho>>2)*4NT + 4j + (rho&3) of the wave's TN = 16 NT
// channels: then a lane (which holds D rows 4q..4q+3 of every tile) owns 4NT CONSECUTIVE channels of its pixel — 16-byte
// stores / mask loads in the epilogue instead of 8-byte ones.  A ds_read_b128 lane group therefore touches rows
// {0-3, 12NT..12NT+3} at chunk c and {4NT..4NT+3, 8NT..8NT+3} at chunk c^1 (plus 4j): the swizzle spreads exactly those.
template <int NT>
__device__ __forceinline__ int hd_bswz(int row) { return ((row >> 1) & 1) | (((row >> (NT == 4 ? 4 : 3)) & 3) << 1); }
template <int NT>
__device__ __forceinline__ int hd_bidx(int row, int chunk) { return row * 8 + (chunk ^ hd_bswz<NT>(row)); }

template <int V> struct HdInt { static constexpr int value = V; };

// total over each row of 16 lanes, in every lane: x += row_ror(x, 8), 4, 2, 1 — the rotation rides on the add (one VALU
// instruction per step; s_nop 1 = the two wait states a DPP read needs after the VALU write of its source)
__device__ __forceinline__ float hd_row_sum16(float x) {
  float y;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
               : "=&v"(y) : "v"(x));
  return y;
}

// MAP8: the maps are 8x8 (VGG conv5 at 128x128 inputs): a workgroup tile is TWO whole images (each with its own 10x10
// zero-padded halo), wave wm owns image wm, and an MFMA operand row of 16 pixels is two image rows of 8.
// PERSIST (round 2): the grid is one workgroup per CU and a workgroup walks tiles w = blockIdx.x, +gridDim.x, ...  The
// loader simply keeps going across the tile boundary — during the last taps of a tile the ring receives the first filter
// taps of the NEXT tile and the free halo stage its first slice — so the prologue burst (every CU fetching ~100 KB at once:
// ~4 us, MI355X_MICROARCH.md "prologue HBM burst") and the store burst of the epilogue overlap with matrix work instead of
// adding ~6 us per tile (conv2_2: 24 of 78 us).  The counted vmcnt waits stay exact for the DMA loads: the epilogue's stores
// and mask loads only ADD to the counter (waits become conservative, never early).
//
// ROW3 (round 2, the 4-wave variants on small grids): a barrier interval is one FILTER ROW (3 taps x 64 channels = 6 k-steps,
// 48 MFMAs per wave) instead of one tap.  Measured from inside: a tap of the 4-wave variant costs 0.30 us for 0.11 us of MFMA
// time — 0.19 us of barrier, DMA issue and exposed LDS round trip per interval with one wave per SIMD — and 16 more MFMAs in
// the same interval cost only their own 0.11 us.  The ring is the nine tap stages of a slice (tap (ky, kx) of every slice lands
// in stage 3 ky + kx; a row's three stages are refilled for the next slice right after the row's barrier), the next slice's halo
// is requested after the FIRST row's barrier so that it has landed when the last row's barrier is passed.
//
// S2D (round 4): the DATA GRADIENT of a 3x3 STRIDE-2 SAME convolution (encoder conv_3 / conv_5 / conv_7, imm_model.py:197,204,211;
// padding 0 top / left, 1 bottom / right) as ONE launch over ONE dy halo.  Input pixel (2i+py, 2j+px) only receives
// the filter taps ky = py (mod 2), kx = px (mod 2): four parity classes with 4 / 2 / 2 / 1 taps, each a dense sub-convolution of
// dy.  The tile is an 8x16 patch of CLASS pixels (i, j) — i.e. a 16x32 patch of dx — with FOUR accumulator sets; the "input" is
// dy with the ordinary 10x18 halo; the filter is the ordinary flipped data-gradient image (imm_pack_weights mode 1: tap t' =
// (ky', kx') holds W[2-ky'][2-kx']), and tap t' simply accumulates into class (ky' & 1, kx' & 1) from halo offset
// (min(ky', 1), min(kx', 1)): dy[i - (ky >> 1)][j - (kx >> 1)] with ky = 2 - ky'.  Nine taps of matrix work per tile — the
// algorithmic count; the im2col forms it replaces ran four launches' worth of gathers at 1-7 % matrix duty with 6-39 VALU
// instructions per MFMA (profiles/r03_v3_pmc_sq_ratios.txt).  The epilogue scatters class (py, px) to pixel (2i+py, 2j+px).
// IMM_HD_ROLL (round 6, default 1; conv_hdeep6.hip's IMM_H6_ROLL for the row-at-a-time variants): the barrier interval of a ROW3
// kernel becomes one filter COLUMN — taps (0, kx), (1, kx), (2, kx) — and the A fragments a rolling window of six halo rows per
// channel half: tap (ky, kx) multiplies rows ky .. ky + 3, so ky = 1 / 2 need one new row each: 12 instead of 24 A reads per
// interval (with NT = 2: 24 instead of 36 ds_read_b128).  Ring stage 3 kx + ky holds memory tap 3 ky + kx (the DMA's choice; the
// packed filter image is unchanged).  Not for MAP8 (an operand row is two image rows: another reuse pattern) and S2D (halo offsets
// min(k', 1)).  -DIMM_HD_ROLL=0: the row walk (A/B builds).
#ifndef IMM_HD_ROLL
#define IMM_HD_ROLL 1
#endif
template <typename ET, int BN, int NW, int PH, bool MAP8 = false, int NSB_ = 0, bool PERSIST = false, bool ROW3 = false, bool S2D = false>
__global__ __launch_bounds__(NW * 64) void conv_hdeep_kernel(const HdArgs ha) {
  const ConvArgs& a = ha.c;
  constexpr bool COL = IMM_HD_ROLL && ROW3 && !S2D && !MAP8;   // column-at-a-time intervals with a rolling A window
  static_assert(!S2D || (!MAP8 && !PERSIST && NW == 4 && BN == 64), "stride-2 data gradient: 8x16 class patches x 64 channels, one tile per workgroup");
  constexpr int NCLS = S2D ? 4 : 1;                    // accumulator sets (parity classes)
  static_assert((NW == 8 && PH == 16) || (NW == 4 && PH == 8), "wave (wm, wn) owns patch rows 4wm..4wm+3 x BN/2 channels");
  static_assert(!MAP8 || NW == 4, "two 8x8 images per 4-wave workgroup");
  static_assert(!(PERSIST && MAP8), "persistent tiles: 16x16 / 8x16 patches only");
  static_assert(!ROW3 || (NSB_ == 9 && !PERSIST), "row-at-a-time schedule: nine tap stages, one tile per workgroup");
  constexpr int HD_NSB = NSB_ ? NSB_ : MAP8 ? 3 : 4;   // filter-slice ring depth (MAP8: 3 keeps two workgroups per CU)
  constexpr int NPIECE = MAP8 ? 7 : 6;                 // halo DMA pieces per wave and slice
  constexpr int TN = BN / 2, MT = 4, NT = TN / 16;
  constexpr int B_I = BN / (8 * NW);                   // filter DMA instructions per wave and tap (BN rows / 8 / NW waves)
  constexpr int SLOTS = MAP8 ? 200 : HD_SLOTS(PH);     // halo pixels
  constexpr int HINSTR = (SLOTS + 7) / 8, HSTAGE = HINSTR * 64;
  static_assert(B_I >= 1 && NPIECE * NW >= HINSTR && (ROW3 || NPIECE <= 11 - HD_NSB), "halo pieces per wave cover the halo and land in time");
  // fragment row strides (uint4 units): output tile row i, vertical tap ky
  constexpr int STEP_I = MAP8 ? 160 : (HD_HW * 128) / 16, STEP_KY = MAP8 ? 80 : (HD_HW * 128) / 16;
  constexpr int B_U4 = BN * 8;                         // uint4 per filter stage
  constexpr int WN_STEADY = (HD_NSB - 2) * (B_I + 1);  // outstanding VMEM allowed at the top of a tap (see header)
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [2][HSTAGE] halo | [64] dump | [HD_NSB][B_U4] filter
  constexpr int DUMP_U4 = 2 * HSTAGE, BRING_U4 = DUMP_U4 + 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int frow = lane & 15, q = lane >> 4;
  const int per_img = ha.patches_x * ha.patches_y;
  const uint64_t xa = (uint64_t)a.x, wa = (uint64_t)a.wt;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t wr = {(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, a.wt_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;

  // ---- tile state ----------------------------------------------------------------------------------------------------
  // ONE set of loader registers (the DMA offsets of the tile whose data is being fetched: the current tile, and from the
  // last slice on the next one) + the coordinates of the tile being computed (e_*, for the epilogue) and of the next (n_*).
  int e_patch, e_img, e_y0, e_x0, e_n0, n_patch = 0, n_img = 0, n_y0 = 0, n_x0 = 0, n_n0 = 0;
  uint32_t h_voff[NPIECE];       // halo piece k of this wave = instruction wid + NW*k (instructions >= HINSTR: dump slot)
  uint32_t h_soff;
  uint32_t b_voff[B_I], nb_voff[B_I];
  auto locate = [&](int w, int& patch, int& img, int& y0, int& x0, int& n0) {
    int bid = w;
    {   // XCD-contiguous order: the n-blocks of a patch and neighbouring patches share one L2
      const int xq = ha.n_wg >> 3, xr_ = ha.n_wg & 7, xcd = bid & 7;
      bid = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + (bid >> 3);
    }
    const int nblk = bid % a.n_nblk;
    patch = bid / a.n_nblk;
    // MAP8: `patch` is a pair of images (2*patch, 2*patch + 1), origin (0, 0)
    img = MAP8 ? 2 * patch : patch / per_img;
    const int pr = MAP8 ? 0 : patch - img * per_img;
    y0 = MAP8 ? 0 : (pr / ha.patches_x) * PH; x0 = MAP8 ? 0 : (pr % ha.patches_x) * HD_PW;
    n0 = nblk * BN;
  };
  auto halo_offsets = [&](int img, int y0, int x0) {
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      const int hp = (wid + NW * k) * 8 + (lane >> 3);
      int hy, hx, il = 0;                                // halo row / column (, image of the pair)
      if (MAP8) { il = hp / 100; const int rr = hp - il * 100; hy = rr / 10; hx = rr - hy * 10; }
      else { hy = hp / HD_HW; hx = hp - hy * HD_HW; }
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < SLOTS && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi && (!MAP8 || img + il < ha.n_img);
      h_voff[k] = ok ? (uint32_t)(((il * a.hi + iy) * a.wi + ix) * a.ldx * 2 + (((lane & 7) ^ hd_swz(hx)) * 16)) : HD_OOB;
    }
    h_soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(img * a.hi * a.wi) * (uint32_t)(a.ldx * 2)));
  };
  auto filter_offsets = [&](int n0, uint32_t (&bv)[B_I]) {
#pragma unroll
    for (int j = 0; j < B_I; ++j) {
      const int r = (wid * B_I + j) * 8 + (lane >> 3);
      bv[j] = (n0 + r < a.co) ? (uint32_t)((n0 + r) * a.kpad * 2 + (((lane & 7) ^ hd_bswz<NT>(r)) * 16)) : HD_OOB;
    }
  };
  const int ncc = a.ci8 >> 3;                          // 64-channel slices
  const int T = ncc * 9;                               // taps per tile
  // filter tap t (< T) of the loader's tile -> ring stage; real = false: a no-op piece that keeps the per-tap DMA count exact
  auto issue_b = [&](int t, int stage, bool real) {
    // tap t = slice cc, tap index tp: filter columns [tp*ci + cc*64, +64) of Wt[n][kpad]
    const int cc = t / 9, tp = t - cc * 9;
    const uint32_t soff = (uint32_t)((tp * (a.ci8 << 3) + cc * 64) * 2);
#pragma unroll
    for (int j = 0; j < B_I; ++j)
      hd_dma16(wr, lds_base + (uint32_t)((BRING_U4 + stage * B_U4 + (wid * B_I + j) * 64) * 16), real ? b_voff[j] : HD_OOB, soff);
  };
  // halo piece k of slice cc of the loader's tile -> halo stage hs
  auto issue_halo_piece = [&](int cc, int hs, int k, bool real) {
    const int i = wid + NW * k;
    const bool exists = real && i < HINSTR;
    const uint32_t dst = exists ? (uint32_t)((hs * HSTAGE + i * 64) * 16) : (uint32_t)(DUMP_U4 * 16);
    hd_dma16(xr, lds_base + dst, exists ? h_voff[k < NPIECE ? k : 0] : HD_OOB, h_soff + (uint32_t)(cc * 128));
  };

  const int G = PERSIST ? (int)gridDim.x : 1;
  const int n_mine = PERSIST ? (ha.n_wg - (int)blockIdx.x + G - 1) / G : 1;
  locate(blockIdx.x, e_patch, e_img, e_y0, e_x0, e_n0);
  halo_offsets(e_img, e_y0, e_x0);
  filter_offsets(e_n0, b_voff);

  // ---- prologue: halo of slice 0, filter taps 0 .. NSB-1 (every ring stage) ---------------------------------------
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) issue_halo_piece(0, 0, k, true);
#pragma unroll
  for (int t = 0; t < HD_NSB; ++t) issue_b(COL ? (t % 3) * 3 + t / 3 : t, t, true);   // COL: stage 3 kx + ky <- memory tap 3 ky + kx

  // The accumulators START at the bias: its loads are issued here, behind the prologue DMA (an epilogue that begins with a
  // dependent global load costs its whole latency: ~0.6 us of the 2.2 us measured per tile), and the epilogue has no adds.
  const bool f_bias = a.flags & IMM_CONV_BIAS;
  f32x4_t acc[NCLS][MT][NT], bias4[NT];
  auto load_bias = [&](int n0_) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float4 b4 = f_bias ? *(const float4*)(a.bias + n0_ + wn * TN + q * (4 * NT) + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      bias4[j] = f32x4_t{b4.x, b4.y, b4.z, b4.w};
    }
  };
  load_bias(e_n0);
#pragma unroll
  for (int c = 0; c < NCLS; ++c)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[c][i][j] = bias4[j];

  // per-lane fragment offsets (uint4 units): A = halo pixel (row wm*4 + i + ky, col frow + kx), B = filter row.  Persistent
  // tiles recompute them after every epilogue (11 registers that need not live through it).
  int aoff[3], boff[NT][2];
  auto frag_offsets = [&]() {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      if (MAP8) aoff[kx] = (wm * 100 + (frow >> 3) * 10 + (frow & 7) + kx) * 8 + (q ^ hd_swz((frow & 7) + kx));
      else aoff[kx] = (wm * 4 * HD_HW + frow + kx) * 8 + (q ^ hd_swz(frow + kx));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        boff[j][ks] = hd_bidx<NT>(wn * TN + (frow >> 2) * (4 * NT) + j * 4 + (frow & 3), ks * 4 + q);
  };
  frag_offsets();

  // Software pipeline at k-step granularity (a tap = two k-steps of 32 channels): the fragments of the next k-step are
  // read from LDS while the 16 MFMAs of the current one run, and a k-step's MFMAs are already queued when the wave
  // reaches the barrier — the matrix pipe keeps working through the barrier / DMA-issue / ds_read window that the two
  // waves of a SIMD (same workgroup, same barrier) would otherwise both sit in.  Per tap t:
  //     lgkmcnt(0)            fragments (t, k-step 0) are in registers
  //     ds_read (t, 1)        same ring stage / halo
  //     MFMA (t, 0)
  //     lgkmcnt(0)            => this wave no longer reads ring stage t % NSB
  //     vmcnt(N); s_barrier   filter tap t+1 (and the next slice's halo, when due) landed everywhere
  //     DMA filter tap t+NSB -> stage t % NSB, one halo piece of the next slice
  //     ds_read (t+1, 0)
  //     MFMA (t, 1)
  uint4 af[2][COL ? MT + 2 : MT], bf[2][NT];          // [k-step][tile]  (COL: [channel half][halo row of the column's window])
  auto read_frags = [&](const int ks, const uint4* Hs, const uint4* Bs, const int ky_, const int kx_) __attribute__((always_inline)) {
    const int ky = S2D ? (ky_ > 0 ? 1 : 0) : ky_, kx = S2D ? (kx_ > 0 ? 1 : 0) : kx_;   // S2D: halo offset min(k', 1), see the header
    if constexpr (COL) {
      // ky = 0 opens the column: rows 0..3; ky = 1 / 2 add row 4 / 5 (a k-step multiplies rows ky .. ky + 3)
      if (ky == 0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) af[ks][i] = Hs[(aoff[kx] ^ (ks * 4)) + i * STEP_I];
      } else {
        af[ks][MT - 1 + ky] = Hs[(aoff[kx] ^ (ks * 4)) + (MT - 1 + ky) * STEP_I];
      }
    } else {
#pragma unroll
    for (int i = 0; i < MT; ++i) af[ks][i] = Hs[(aoff[kx] ^ (ks * 4)) + i * STEP_I + ky * STEP_KY];
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[ks][j] = Bs[boff[j][ks]];
  };
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ROW3 ? 6 : HD_NSB - 1) * B_I) : "memory");   // halo(0) and filter tap 0 (ROW3: row 0)
  __builtin_amdgcn_s_barrier();
  read_frags(0, smem, smem + BRING_U4, 0, 0);

  int bs = 0, gs = 0;                                  // ring stage of the current tap; slices done so far (halo stage = parity)
  // -DIMM_HDEEP_PROFILE + IMM_HDEEP_PROF=1: workgroup 0 / wave 0 stamps the wall clock at tile / tap boundaries (how the
  // 2.2 us epilogue, the 0.84 us tile setup and the 0.84 us taps of DESIGN.md were measured); compiled out otherwise
#ifdef IMM_HDEEP_PROFILE
  int pidx = 1;
  const bool prof = ha.prof != nullptr && blockIdx.x == 0 && wid == 0;
#define HD_STAMP(code) do { if (prof && pidx < 4000) { if (lane == 0) ha.prof[pidx] = ((unsigned long long)(code) << 56) | (wall_clock64() & 0xffffffffffffffull); ++pidx; } } while (0)
#else
#define HD_STAMP(code) do { } while (0)
#endif
  for (int it = 0; it < n_mine; ++it) {
  const bool have_next = PERSIST && it + 1 < n_mine;
  int t = 0;
  if constexpr (ROW3) {
    for (int cc = 0; cc < ncc; ++cc, ++gs) {
      const bool next_slice = cc + 1 < ncc;
      const uint4* Hc = smem + (gs & 1) * HSTAGE;
      const uint4* Hnx = smem + ((gs + 1) & 1) * HSTAGE;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {                  // k-step j of the row: tap (ky, kx = j / 2), channels 32 (j & 1) ..
          const int cur = j & 1;                      // tap (ky, j >> 1), channel half j & 1 = fragment buffer
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // fragments of k-step j are in registers
          __builtin_amdgcn_sched_barrier(0);
          // (COL: the outer index `ky` of this loop nest is the filter COLUMN kx and (j >> 1) the row; stage = 3 outer + inner either way)
          if (j < 5) {
            const int nkx = (j + 1) >> 1;
            if constexpr (COL) read_frags(cur ^ 1, Hc, smem + BRING_U4 + (ky * 3 + nkx) * B_U4, nkx, ky);
            else read_frags(cur ^ 1, Hc, smem + BRING_U4 + (ky * 3 + nkx) * B_U4, ky, nkx);
          } else {
            // every fragment of this row has been read: its three stages are free; the next row's taps (requested one
            // slice = three rows ago) and, before the last row, the next slice's halo (requested after the first) must be in
            if (ky == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * B_I + NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * B_I) : "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int x3 = 0; x3 < 3; ++x3) issue_b((cc + 1) * 9 + (COL ? x3 * 3 + ky : ky * 3 + x3), ky * 3 + x3, next_slice);
            if (ky == 0) {
#pragma unroll
              for (int k = 0; k < NPIECE; ++k) issue_halo_piece(cc + 1, (gs + 1) & 1, k, next_slice);
            }
            if (ky < 2) {
              if constexpr (COL) read_frags(0, Hc, smem + BRING_U4 + ((ky + 1) * 3) * B_U4, 0, ky + 1);
              else read_frags(0, Hc, smem + BRING_U4 + ((ky + 1) * 3) * B_U4, ky + 1, 0);
            } else read_frags(0, Hnx, smem + BRING_U4, 0, 0);         // past the last slice: landed no-op data, never used
          }
          const int cls = S2D ? ((ky & 1) * 2 + ((j >> 1) & 1)) : 0;     // parity class of tap (ky, kx = j >> 1)
          const int arow = COL ? (j >> 1) : 0;                            // COL: tap row = first window row of this k-step
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) acc[cls][i][jj] = ET::mfma(bf[cur][jj], af[cur][i + arow], acc[cls][i][jj]);
#pragma unroll
          for (int m = 0; m < MT * NT; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
          }
        }
      }
    }
  } else
  for (int cc = 0; cc < ncc; ++cc, ++gs) {
    const bool next_slice = cc + 1 < ncc;
    if (PERSIST && !next_slice && have_next) {         // the loader turns to the next tile during this slice: its halo now,
      HD_STAMP(1);                                     // its filter rows once this tile's last tap has been requested
      locate(blockIdx.x + (it + 1) * G, n_patch, n_img, n_y0, n_x0, n_n0);
      halo_offsets(n_img, n_y0, n_x0);
      filter_offsets(n_n0, nb_voff);
      HD_STAMP(2);
    }
    const uint4* Hc = smem + (gs & 1) * HSTAGE;
    const int hs_next = (gs + 1) & 1;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int ky = tp / 3, kx = tp % 3;
      // next tap: (tp+1) of this slice, or tap 0 of the next slice / the next tile's first slice (other halo stage)
      const int ntp = tp == 8 ? 0 : tp + 1;
      const uint4* Hn = smem + ((tp == 8 ? gs + 1 : gs) & 1) * HSTAGE;
      const uint4* Bc = smem + BRING_U4 + bs * B_U4;
      int nbs = bs + 1; if (nbs == HD_NSB) nbs = 0;
      const uint4* Bn = smem + BRING_U4 + nbs * B_U4;

      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      read_frags(1, Hc, Bc, ky, kx);
      const int cls = S2D ? ((ky & 1) * 2 + (kx & 1)) : 0;                                 // parity class of this tap
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[cls][i][j] = ET::mfma(bf[0][j], af[0][i], acc[cls][i][j]);   // D[n][pixel]
      // issue order inside the region: 1 MFMA, 1 ds_read, <= 2 address ops, ... (non-MFMA issues ride in the pipe's shadow)
#pragma unroll
      for (int m = 0; m < MT * NT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // first tile only: the prologue issued its filter taps back to back (no halo pieces in between)
      if (it == 0 && t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((HD_NSB - 2) * B_I) : "memory");          // taps 2.. in flight
      else if (it == 0 && t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((HD_NSB - 2) * B_I + 1) : "memory");
      else if (it > 0 && t < HD_NSB - 1) { /* taps 1 .. NSB-1 landed before the previous tile's epilogue (drain below): no wait, so
                                              that its output stores get NSB-1 taps to be acknowledged before a counted wait sees them */ }
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN_STEADY) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      HD_STAMP(3);
      // Both waves of a SIMD leave the barrier together: whatever comes first here runs with the matrix pipe idle.  The
      // fragments of k-step 1 are in registers, so a quarter of its MFMAs goes first and the loader's scalar address
      // arithmetic + DMA issue (~250 cycles when it led the block, measured as 0.77 us per tap against 0.51 us of MFMA
      // time) rides in their shadow; the second quarter covers the halo piece, the rest the next tap's fragment reads.
      constexpr int QM = MT * NT / 4;
#pragma unroll
      for (int m = 0; m < QM; ++m) acc[cls][m % MT][m / MT] = ET::mfma(bf[1][m / MT], af[1][m % MT], acc[cls][m % MT][m / MT]);
      __builtin_amdgcn_sched_barrier(0);
      if (PERSIST && tp == 9 - HD_NSB && !next_slice && have_next) {          // t + NSB == T: the ring runs on into the next tile
#pragma unroll
        for (int j = 0; j < B_I; ++j) b_voff[j] = nb_voff[j];
      }
      if (t + HD_NSB < T) issue_b(t + HD_NSB, bs, true);
      else issue_b(t + HD_NSB - T, bs, have_next);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = QM; m < 2 * QM; ++m) acc[cls][m % MT][m / MT] = ET::mfma(bf[1][m / MT], af[1][m % MT], acc[cls][m % MT][m / MT]);
      __builtin_amdgcn_sched_barrier(0);
      issue_halo_piece(next_slice ? cc + 1 : 0, hs_next, tp, (next_slice || have_next) && tp < NPIECE);   // ... and so does the halo
      // (k-step 0 of the next tap; not across a tile boundary: 32 fragment registers would stay live through the epilogue)
      if (!(PERSIST && tp == 8 && !next_slice)) read_frags(0, Hn, Bn, ntp / 3, ntp % 3);
#pragma unroll
      for (int m = 2 * QM; m < MT * NT; ++m) acc[cls][m % MT][m / MT] = ET::mfma(bf[1][m / MT], af[1][m % MT], acc[cls][m % MT][m / MT]);
#pragma unroll
      for (int m = 0; m < 2 * QM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
      }
      bs = nbs;
      ++t;
    }
  }
  HD_STAMP(4);
  if (!have_next) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing no-op pieces: no LDS-DMA may outlive the workgroup
    __syncthreads();
  } else {
    // the next tile's halo and first NSB filter taps (issued 0.5 .. NSB-0.5 taps ago) have landed: from here on only this
    // tile's output stores are outstanding, and the next tile's first NSB-1 taps need no counted wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  HD_STAMP(5);
  const int img = e_img, y0 = e_y0, x0 = e_x0, n0 = e_n0, patch = e_patch;

  // ---- epilogue: lane = pixel (row wm*4 + i, col frow), 4*NT consecutive channels (see hd_bswz) ----------------------
  // Two waves per SIMD run this at the same time with no matrix work to hide behind: it is VALU-bound (measured with the
  // in-kernel clock: 2.2 us per tile for ~450 VALU instructions per wave in the generic form).  Hence one uniform branch per
  // variant instead of per-value selects, packed f32 adds, ReLU and the ReLU-backward mask as PACKED 16-bit integer operations
  // on the converted outputs (bf16 / f16 are sign-magnitude: max_i16(x, 0) == relu(x); (max_i16(mask, 0) != 0) == (mask > 0)).
  const bool f_relu = a.flags & IMM_CONV_RELU;
  // (persistent tiles: launches without batch-norm sums only — the host falls back to one workgroup per tile otherwise;
  // the sums' 32 extra registers do not fit beside the loop state that stays live through this epilogue)
  const bool f_stats = !PERSIST && (a.flags & IMM_CONV_STATS), f_mask = a.flags & IMM_CONV_MASK;
  const int nb = n0 + wn * TN + q * (4 * NT);          // first channel of this lane
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
  if (have_next) load_bias(n_n0);                      // the next tile's accumulators start from it (below)
  const int64_t m_first = MAP8 ? ((int64_t)(img + wm) * 8 + (frow >> 3)) * 8 + (frow & 7)
                               : ((int64_t)img * a.ho + y0 + wm * 4) * a.wo + x0 + frow;
  const int64_t m_step = MAP8 ? 16 : a.wo;             // pixel index of tile row i = m_first + i * m_step
  const bool rows_exist = !MAP8 || img + wm < ha.n_img;   // odd batch: the second image of the last pair does not exist
  typedef short s16x2_t __attribute__((ext_vector_type(2)));
  typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
  const s16x2_t zero2 = {0, 0}, one2 = {1, 1};
  if constexpr (S2D) {
    // ---- stride-2 data gradient: class (py, px) of tile pixel (row wm*4 + i, col frow) -> dx pixel (2 row + py, 2 col + px) ----
    const int W2 = 2 * a.wo;
#pragma unroll
    for (int c = 0; c < NCLS; ++c) {
      const int64_t mc = ((int64_t)img * (2 * a.ho) + 2 * (y0 + wm * 4) + (c >> 1)) * W2 + 2 * (x0 + frow) + (c & 1);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int64_t m = mc + (int64_t)i * 2 * W2;
#pragma unroll
        for (int h = 0; h < NT / 2; ++h) {
          uint32_t w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 2 * h + (e >> 1), r = (e & 1) * 2;
            w[e] = ET::pack2(acc[c][i][j][r], acc[c][i][j][r + 1]);
          }
          if (nb + 8 * h < a.co) *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
  } else if ((a.flags & IMM_CONV_TAP_) && rows_exist) {
    // ---- perceptual tap (imm_conv2d_tap): v = acc + c_k * lossmask[pixel] * (a_pred - a_gt), zero where a_pred <= 0 -----
    const float ck = a.tap_coef[a.tap_idx];
    const int hw = a.ho * a.wo, rr = a.tap_lmask ? a.tap_S / a.ho : 1;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int64_t m = m_first + i * m_step;
      float cm = ck;
      if (a.tap_lmask) {
        const int im = (int)(m / hw), rem = (int)(m - (int64_t)im * hw);
        const int yy = rem / a.wo, xx = rem - yy * a.wo;
        cm *= a.tap_lmask[((int64_t)im * a.tap_S + (int64_t)yy * rr) * a.tap_S + (int64_t)xx * rr];
      }
#pragma unroll
      for (int h = 0; h < NT / 2; ++h) {
        float v[8], fp[8], fg[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 2 * h + (e >> 1), r = (e & 1) * 2;
          v[2 * e] = acc[0][i][j][r]; v[2 * e + 1] = acc[0][i][j][r + 1];
        }
        unpack8<ET>(*(const uint4*)(a.mask + m * a.ldmask + nb + 8 * h), fp);
        unpack8<ET>(*(const uint4*)(a.tap_gt + m * a.ldmask + nb + 8 * h), fg);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float d = fp[e] - fg[e];
          if (a.tap_l1) d = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
          // the separate pass rounds the incoming gradient to 16 bits before adding the tap term: same here (bitwise equal)
          float t = ET::to_f32(ET::from_f32(v[e])) + cm * d;
          if (!(fp[e] > 0.f)) t = 0.f;
          v[e] = t;
        }
        *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = pack8<ET>(v);
      }
    }
  } else if (!f_stats && rows_exist) {
    // ---- plain / masked store: ~12 (+12 with the mask) VALU per 8 outputs --------------------------------------------
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int64_t m = m_first + i * m_step;
#pragma unroll
      for (int h = 0; h < NT / 2; ++h) {               // 8 channels = one 16-byte store
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                  // pair e = channels 2e, 2e+1 of the eight
          const int j = 2 * h + (e >> 1), r = (e & 1) * 2;
          w[e] = ET::pack2(acc[0][i][j][r], acc[0][i][j][r + 1]);
        }
        if (f_relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            w[e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, w[e]), zero2));
        }
        if (f_mask) {
          const uint4 mk = *(const uint4*)(a.mask + m * a.ldmask + nb + 8 * h);
          const uint32_t mw[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const s16x2_t pos = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2_t, mw[e]), zero2), one2);   // 1 where mask > 0
            w[e] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, w[e]) * __builtin_bit_cast(u16x2_t, pos));
          }
        }
        *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
  } else if (rows_exist) {
    // ---- batch-norm statistics (+ optional mask): the sums need the f32 values ----------------------------------------
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int64_t m = m_first + i * m_step;
#pragma unroll
      for (int h = 0; h < NT / 2; ++h) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = 2 * h + (e >> 1), r = (e & 1) * 2;
          v[2 * e] = acc[0][i][j][r]; v[2 * e + 1] = acc[0][i][j][r + 1];
        }
        if (f_relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (f_mask) {
          // STATS | MASK: second sum = sum(v * mask_ref) — the batch-norm backward sums of the layer this gradient enters
          float mf[8];
          unpack8<ET>(*(const uint4*)(a.mask + m * a.ldmask + nb + 8 * h), mf);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if (!(mf[e] > 0.f)) v[e] = 0.f;
            s1[2 * h + (e >> 2)][e & 3] += v[e];
            s2[2 * h + (e >> 2)][e & 3] += v[e] * mf[e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            s1[2 * h + (e >> 2)][e & 3] += v[e];
            s2[2 * h + (e >> 2)][e & 3] = fmaf(v[e], v[e], s2[2 * h + (e >> 2)][e & 3]);
          }
        }
        *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = pack8<ET>(v);
      }
    }
  }
  if (f_stats) {
    // [NW/2 wm][2][BN]; persistent tiles: behind the filter ring (the halo stages are receiving the next tile's data)
    // (the previous tile's sums were read at least a tile's worth of barriers ago)
    float* red = (float*)(PERSIST ? smem + BRING_U4 + HD_NSB * B_U4 : smem);
    // sum over the 16 pixel lanes of a row: DPP row rotations (plain VALU; the ds_bpermute shuffles went through the LDS)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[j][r] = hd_row_sum16(s1[j][r]); s2[j][r] = hd_row_sum16(s2[j][r]); }
    if (frow == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = wn * TN + q * (4 * NT) + j * 4 + r;
          red[(wm * 2 + 0) * BN + nl] = s1[j][r];
          red[(wm * 2 + 1) * BN + nl] = s2[j][r];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS only: a __syncthreads would also drain the next tile's DMA
    __builtin_amdgcn_s_barrier();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW / 2; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
      a.stats[((int64_t)patch * 2 + 0) * a.co + n0 + tid] = t1;
      a.stats[((int64_t)patch * 2 + 1) * a.co + n0 + tid] = t2;
    }
  }
  HD_STAMP(6);
  if (have_next) {
    frag_offsets();
    read_frags(0, smem + (gs & 1) * HSTAGE, smem + BRING_U4 + bs * B_U4, 0, 0);   // next tile, tap 0 (landed before the epilogue)
    e_patch = n_patch; e_img = n_img; e_y0 = n_y0; e_x0 = n_x0; e_n0 = n_n0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[0][i][j] = bias4[j];
  }
  }   // tiles of this workgroup
#ifdef IMM_HDEEP_PROFILE
  if (prof && lane == 0) ha.prof[0] = (unsigned long long)pidx;
#endif
#undef HD_STAMP
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int hd_num_cu() {
  static int cu = 0;
  if (cu == 0) {
    hipDeviceProp_t p; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cu = p.multiProcessorCount;
    else (void)hipGetLastError();
    if (cu <= 0) cu = 256;
  }
  return imm_limit_cus(cu);
}

// Tile plan of a layer: patch height (16 = 8-wave kernel, 8 = 4-wave kernel) and channel-block width.
//   16 x 16 patch x 128 channels when that still gives every CU a workgroup;
//   16 x 16 x 64 when THAT does;
//   else 8 x 16 x 64 (4 waves, two workgroups per CU) when the map height allows — the small-grid layers (16x16 maps of
//   a 32-image batch: 32 patches) otherwise leave half the chip idle.
struct HdPlan { int ph, bn, n_patches, n_wg; bool map8, persist, row3; };

static HdPlan hd_plan(const imm_conv_desc* d) {
  constexpr bool no_small = false;
  const int cus = hd_num_cu();
  constexpr int small_below = 2;   // x CUs (round 3: 4 -> 2.  128 -> 64 channels at 64x64 maps, batch 32 — renderer conv_5, VGG conv2_1's data
                                   // gradient — as 512 tiles of 16x16x64 (8 waves) instead of 1024 of 8x16x64: -12 us per step, same box)
  constexpr bool no_big = false;
  HdPlan p;
  p.map8 = false; p.persist = false; p.row3 = false;
  if (d->ho == 8 && d->wo == 8) {                      // two whole 8x8 images per (4-wave) workgroup
    p.ph = 8; p.bn = 64; p.map8 = true;
    p.n_patches = (d->batch + 1) / 2;
    p.n_wg = p.n_patches * (d->co / 64);
    return p;
  }
  const int np16 = (d->ho % 16 == 0) ? d->batch * (d->ho / 16) * (d->wo / HD_PW) : 0;
  if (!no_big && np16 > 0 && d->co % 128 == 0 && np16 * (d->co / 128) >= cus) { p.ph = 16; p.bn = 128; p.n_patches = np16; }
  else if (np16 > 0 && (np16 * (d->co / 64) >= small_below * cus || no_small)) { p.ph = 16; p.bn = 64; p.n_patches = np16; }
  else if (!no_small) { p.ph = 8; p.bn = 64; p.n_patches = d->batch * (d->ho / 8) * (d->wo / HD_PW); }
  else { p.ph = 16; p.bn = 64; p.n_patches = np16; }
  p.n_wg = p.n_patches * (d->co / p.bn);
  // 8x16 tiles that would be two rounds of 4-wave workgroups (e.g. 32x32 maps, 128 channels: 512) as ONE round of 8-wave
  // 16x16x64 workgroups with the row-at-a-time schedule
  constexpr bool row3_big = true;
  if (row3_big && !no_small && p.ph == 8 && p.bn == 64 && p.n_wg > cus && np16 > 0 && np16 * (d->co / 64) <= cus &&
      np16 * (d->co / 64) >= cus / 2) {
    p.ph = 16; p.n_patches = np16; p.n_wg = np16 * (d->co / 64); p.row3 = true;
  }
  // more tiles than CUs: one persistent workgroup per CU walks them
  constexpr bool persist = true;
  p.persist = persist && p.ph == 16 && p.n_wg > cus && (cus % 8) == 0 && !(d->flags & IMM_CONV_STATS);
  return p;
}

bool imm_hdeep_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("hdeep");
  if (off) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->updiv != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
  if (d->ci % 64 || d->co % 64 || d->ci < 64) return false;
  if (d->out_scale > 1 || (d->flags & (IMM_CONV_OUT_F32 | 0xf00))) return false;   // (bit 0x20 = IMM_CONV_TAP_ is this kernel's own)
  if (d->hi != d->ho || d->wi != d->wo || d->ho % 8 || (d->wo % HD_PW && !(d->ho == 8 && d->wo == 8))) return false;
  if (d->ldy % 8 || ((d->flags & IMM_CONV_MASK) && d->ldmask % 8)) return false;
  const int64_t px = (int64_t)d->batch * d->hi * d->wi;
  if (px * d->ldx * 2 >= (1LL << 31) || (int64_t)d->co * d->kpad * 2 >= (1LL << 31)) return false;
  // small grids keep the im2col kernel (64x64 tiles give it 4x the workgroups)
  constexpr int min_wg = 100;
  const HdPlan p = hd_plan(d);
  return p.n_patches > 0 && p.n_wg >= min_wg;
}

int imm_hdeep_stats_blocks(const imm_conv_desc* d) { return hd_plan(d).n_patches; }

static bool hd_takes_hdeep6(const imm_conv_desc* d, const HdPlan& p);
// imm_conv2d_variant: 6 * 100000 + persist for the six-k-steps-per-barrier kernel (conv_hdeep6.hip), else
// 5 * 100000 + map8 * 40000 + row-at-a-time * 20000 + persistent * 10000 + channel block * 10 + waves
int imm_hdeep_variant(const imm_conv_desc* d) {
  const HdPlan p = hd_plan(d);
  if (hd_takes_hdeep6(d, p)) return 600000 + (p.persist ? 1 : 0);
  const bool one_wave = p.n_wg <= hd_num_cu();
  const bool row3 = (p.map8 && one_wave) || (p.ph == 8 && p.bn == 64 && !p.map8 && one_wave) || (p.ph == 16 && p.bn == 64 && p.row3);
  const bool persist = !row3 && !p.map8 && p.ph == 16 && p.persist;
  return 500000 + (p.map8 ? 40000 : 0) + (row3 ? 20000 : 0) + (persist ? 10000 : 0) + p.bn * 10 + (p.ph == 16 ? 8 : 4);
}

template <typename ET, int BN, int NW, int PH, bool MAP8 = false, int NSB_ = 0, bool PERSIST = false, bool ROW3 = false, bool S2D = false>
static void hd_launch_cfg(const HdArgs& ha, hipStream_t s) {
  constexpr int hstage = MAP8 ? 25 * 64 : HD_HSTAGE(PH), nsb = NSB_ ? NSB_ : MAP8 ? 3 : 4;
  constexpr int lds = (2 * hstage + 64 + nsb * BN * 8) * 16 + (PERSIST ? NW * BN * 4 : 0);   // + [NW/2][2][BN] f32 stats scratch
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_hdeep_kernel<ET, BN, NW, PH, MAP8, NSB_, PERSIST, ROW3, S2D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const int grid = PERSIST ? (ha.n_wg < hd_num_cu() ? ha.n_wg : hd_num_cu()) : ha.n_wg;
  hipLaunchKernelGGL((conv_hdeep_kernel<ET, BN, NW, PH, MAP8, NSB_, PERSIST, ROW3, S2D>), dim3(grid), dim3(NW * 64), lds, s, ha);
}

template <typename ET>
static void hd_launch(const HdPlan& p, const HdArgs& ha, hipStream_t s) {
  constexpr bool row3 = true;   // one barrier per filter row (DESIGN item 27)
  // (119 KB of LDS: one workgroup per CU — only where the grid leaves it at one per CU anyway; with more tiles than CUs the
  // 78 KB tap-at-a-time form keeps two co-resident: 32x32 128->128, 512 tiles, 14.2 vs 17.1 us)
  const bool one_wave = p.n_wg <= hd_num_cu();
  if (p.map8 && row3 && one_wave) hd_launch_cfg<ET, 64, 4, 8, true, 9, false, true>(ha, s);
  else if (p.ph == 8 && p.bn == 64 && !p.map8 && row3 && one_wave) hd_launch_cfg<ET, 64, 4, 8, false, 9, false, true>(ha, s);
  else if (p.map8) hd_launch_cfg<ET, 64, 4, 8, true>(ha, s);
  else if (p.ph == 16 && p.bn == 128 && p.persist) hd_launch_cfg<ET, 128, 8, 16, false, 0, true>(ha, s);
  else if (p.ph == 16 && p.bn == 128) hd_launch_cfg<ET, 128, 8, 16>(ha, s);
  else if (p.ph == 16 && p.bn == 64 && p.row3) hd_launch_cfg<ET, 64, 8, 16, false, 9, false, true>(ha, s);
  else if (p.ph == 16 && p.persist) hd_launch_cfg<ET, 64, 8, 16, false, 0, true>(ha, s);
  else if (p.ph == 16) hd_launch_cfg<ET, 64, 8, 16>(ha, s);
  else hd_launch_cfg<ET, 64, 4, 8>(ha, s);
}

// IMM_HDEEP_PROF=1: workgroup 0 / wave 0 of every hdeep launch stamps the device wall clock (100 MHz) at its tile / tap
// boundaries into one buffer (the last launch wins); printed at process exit.  Diagnosis only.
static unsigned long long* hd_prof_buf = nullptr;
static void hd_prof_dump() {
  if (!hd_prof_buf) return;
  static unsigned long long host[4000];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, hd_prof_buf, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return;
  const int n = (int)host[0] < 4000 ? (int)host[0] : 4000;
  const unsigned long long mask = 0xffffffffffffffull;
  for (int i = 1; i < n; ++i)
    fprintf(stderr, "HDPROF %4d code %d  t %9.2f us  (+%7.2f)\n", i, (int)(host[i] >> 56), (double)((host[i] & mask) - (host[1] & mask)) / 100.0,
            i > 1 ? (double)((host[i] & mask) - (host[i - 1] & mask)) / 100.0 : 0.0);
}

bool imm_hdeep6_enabled();                                                           // conv_hdeep6.hip
void imm_conv_hdeep6_launch(int dtype, const ConvArgs& a, int n_patches, int patches_x, int patches_y, int n_wg, bool persist, int cus,
                            hipStream_t s);
// the 16x16x128 tile without batch-norm partial sums runs the six-k-steps-per-barrier form (conv_hdeep6.hip)
static bool hd_takes_hdeep6(const imm_conv_desc* d, const HdPlan& p) {
  return p.ph == 16 && p.bn == 128 && !p.map8 && !(d->flags & IMM_CONV_STATS) && imm_hdeep6_enabled();
}

void imm_conv_hdeep_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  HdArgs ha;
  ha.c = a;
  static const bool prof = getenv("IMM_HDEEP_PROF") != nullptr;
  if (prof && !hd_prof_buf && hipMalloc((void**)&hd_prof_buf, 4000 * sizeof(unsigned long long)) == hipSuccess) {
    (void)hipMemset(hd_prof_buf, 0, 4000 * sizeof(unsigned long long));
    atexit(hd_prof_dump);
  }
  ha.prof = prof ? hd_prof_buf : nullptr;
  const HdPlan p = hd_plan(d);
  ha.patches_x = p.map8 ? 1 : d->wo / HD_PW; ha.patches_y = p.map8 ? 1 : d->ho / p.ph;
  ha.n_img = d->batch;
  ha.n_patches = p.n_patches;
  ha.c.n_nblk = d->co / p.bn;
  ha.n_wg = p.n_wg;
  ha.c.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  if (hd_takes_hdeep6(d, p)) {
    imm_conv_hdeep6_launch(dtype, ha.c, ha.n_patches, ha.patches_x, ha.patches_y, ha.n_wg, p.persist, hd_num_cu(), s);
    return;
  }
  if (dtype == IMM_BF16) hd_launch<BF16>(p, ha, s); else hd_launch<F16>(p, ha, s);
}

// ---- stride-2 data gradient (S2D) ---------------------------------------------------------------------------------------------
// d describes the CLASS grid as a same-size 3x3 convolution of dy: batch, hi = ho = dy rows, wi = wo = dy columns, ci = dy channels
// entering the K loop (its pixel stride: padding channels hold zeros), co = dx channels, ldy = dx pixel stride; the output tensor
// is [batch, 2 ho, 2 wo].  wt = the flipped data-gradient image (imm_pack_weights mode 1), rows >= co, row length kpad = 9 ci.
bool imm_hdeep_s2d_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("s2d");
  if (off) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->updiv != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
  if (d->ci % 64 || d->ci < 64 || d->ldx != d->ci || d->kpad != 9 * d->ci) return false;
  if (d->co % 8 || d->co < 8 || d->ldy % 8 || d->ldy < d->co) return false;
  if (d->hi != d->ho || d->wi != d->wo || d->ho % 8 || d->wo % HD_PW) return false;
  if (d->flags || d->out_scale > 1) return false;
  const int64_t px = (int64_t)d->batch * d->hi * d->wi;
  if (px * d->ldx * 2 >= (1LL << 31) || (int64_t)d->co * d->kpad * 2 >= (1LL << 31)) return false;
  if (px * 4 * d->ldy * 2 >= (1LL << 40)) return false;
  return true;
}

void imm_conv_hdeep_s2d_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  HdArgs ha;
  ha.c = a;
  ha.prof = nullptr;
  ha.patches_x = d->wo / HD_PW; ha.patches_y = d->ho / 8;
  ha.n_img = d->batch;
  ha.n_patches = d->batch * ha.patches_x * ha.patches_y;
  ha.c.n_nblk = (d->co + 63) / 64;
  ha.n_wg = ha.n_patches * ha.c.n_nblk;
  ha.c.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  // one round of workgroups or less: the row-at-a-time schedule (119 KB of LDS, one workgroup per CU anyway); larger grids keep
  // the 78 KB tap-at-a-time form
  const bool row3 = ha.n_wg <= hd_num_cu();
  if (dtype == IMM_BF16) {
    if (row3) hd_launch_cfg<BF16, 64, 4, 8, false, 9, false, true, true>(ha, s);
    else hd_launch_cfg<BF16, 64, 4, 8, false, 0, false, false, true>(ha, s);
  } else {
    if (row3) hd_launch_cfg<F16, 64, 4, 8, false, 9, false, true, true>(ha, s);
    else hd_launch_cfg<F16, 64, 4, 8, false, 0, false, false, true>(ha, s);
  }
}
