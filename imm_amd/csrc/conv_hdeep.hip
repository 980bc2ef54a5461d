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

// MAP8: the maps are 8x8 (VGG conv5 at 128x128 inputs): a workgroup tile is TWO whole images (each with its own 10x10
// zero-padded halo), wave wm owns image wm, and an MFMA operand row of 16 pixels is two image rows of 8.
template <typename ET, int BN, int NW, int PH, bool MAP8 = false>
__global__ __launch_bounds__(NW * 64) void conv_hdeep_kernel(const HdArgs ha) {
  const ConvArgs& a = ha.c;
  static_assert((NW == 8 && PH == 16) || (NW == 4 && PH == 8), "wave (wm, wn) owns patch rows 4wm..4wm+3 x BN/2 channels");
  static_assert(!MAP8 || NW == 4, "two 8x8 images per 4-wave workgroup");
  constexpr int HD_NSB = MAP8 ? 3 : 4;                 // filter-slice ring depth (MAP8: 3 keeps two workgroups per CU)
  constexpr int NPIECE = MAP8 ? 7 : 6;                 // halo DMA pieces per wave and slice
  constexpr int TN = BN / 2, MT = 4, NT = TN / 16;
  constexpr int B_I = BN / (8 * NW);                   // filter DMA instructions per wave and tap (BN rows / 8 / NW waves)
  constexpr int SLOTS = MAP8 ? 200 : HD_SLOTS(PH);     // halo pixels
  constexpr int HINSTR = (SLOTS + 7) / 8, HSTAGE = HINSTR * 64;
  static_assert(B_I >= 1 && NPIECE * NW >= HINSTR && NPIECE <= 11 - HD_NSB, "halo pieces per wave cover the halo and land in time");
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
  int bid = blockIdx.x;
  {   // XCD-contiguous order: the n-blocks of a patch and neighbouring patches share one L2
    const int xq = ha.n_wg >> 3, xr = ha.n_wg & 7, xcd = bid & 7;
    bid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  }
  const int nblk = bid % a.n_nblk, patch = bid / a.n_nblk;
  const int per_img = ha.patches_x * ha.patches_y;
  // MAP8: `patch` is a pair of images (2*patch, 2*patch + 1), origin (0, 0)
  const int img = MAP8 ? 2 * patch : patch / per_img, pr = MAP8 ? 0 : patch - img * per_img;
  const int y0 = MAP8 ? 0 : (pr / ha.patches_x) * PH, x0 = MAP8 ? 0 : (pr % ha.patches_x) * HD_PW;
  const int n0 = nblk * BN;

  const uint64_t xa = (uint64_t)a.x, wa = (uint64_t)a.wt;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t wr = {(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, a.wt_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;

  // ---- loader state ------------------------------------------------------------------------------------------
  // halo piece k of this wave = instruction wid + NW*k (k < 6; instructions >= HINSTR do not exist -> dump slot)
  uint32_t h_voff[NPIECE];
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
  const uint32_t img_soff = (uint32_t)(img * a.hi * a.wi) * (uint32_t)(a.ldx * 2);
  uint32_t b_voff[B_I];
#pragma unroll
  for (int j = 0; j < B_I; ++j) {
    const int r = (wid * B_I + j) * 8 + (lane >> 3);
    b_voff[j] = (n0 + r < a.co) ? (uint32_t)((n0 + r) * a.kpad * 2 + (((lane & 7) ^ hd_bswz<NT>(r)) * 16)) : HD_OOB;
  }
  const int ncc = a.ci8 >> 3;                          // 64-channel slices
  const int T = ncc * 9;                               // taps in total
  auto issue_b = [&](int t, int stage) {
    // tap t = slice cc, tap index tp: filter columns [tp*ci + cc*64, +64) of Wt[n][kpad]
    const int cc = t / 9, tp = t - cc * 9;
    const uint32_t soff = (uint32_t)((tp * (a.ci8 << 3) + cc * 64) * 2);
    const bool real = t < T;
#pragma unroll
    for (int j = 0; j < B_I; ++j)
      hd_dma16(wr, lds_base + (uint32_t)((BRING_U4 + stage * B_U4 + (wid * B_I + j) * 64) * 16), real ? b_voff[j] : HD_OOB, soff);
  };
  auto issue_halo_piece = [&](int cc, int k, bool real) {
    const int i = wid + NW * k;
    const bool exists = real && i < HINSTR;
    const uint32_t dst = exists ? (uint32_t)(((cc & 1) * HSTAGE + i * 64) * 16) : (uint32_t)(DUMP_U4 * 16);
    hd_dma16(xr, lds_base + dst, exists ? h_voff[k < NPIECE ? k : 0] : HD_OOB, img_soff + (uint32_t)(cc * 128));
  };

  // ---- prologue: halo of slice 0, filter taps 0 .. NSB-1 (every ring stage) ---------------------------------------
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) issue_halo_piece(0, k, true);
#pragma unroll
  for (int t = 0; t < HD_NSB; ++t) issue_b(t, t);

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // per-lane fragment offsets (uint4 units): A = halo pixel (row wm*4 + i + ky, col frow + kx), B = filter row
  int aoff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    if (MAP8) aoff[kx] = (wm * 100 + (frow >> 3) * 10 + (frow & 7) + kx) * 8 + (q ^ hd_swz((frow & 7) + kx));
    else aoff[kx] = (wm * 4 * HD_HW + frow + kx) * 8 + (q ^ hd_swz(frow + kx));
  }
  int boff[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      boff[j][ks] = hd_bidx<NT>(wn * TN + (frow >> 2) * (4 * NT) + j * 4 + (frow & 3), ks * 4 + q);

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
  uint4 af[2][MT], bf[2][NT];                          // [k-step][tile]
  auto read_frags = [&](const int ks, const uint4* Hs, const uint4* Bs, const int ky, const int kx) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i) af[ks][i] = Hs[(aoff[kx] ^ (ks * 4)) + i * STEP_I + ky * STEP_KY];
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[ks][j] = Bs[boff[j][ks]];
  };
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((HD_NSB - 1) * B_I) : "memory");   // halo(0) and filter tap 0
  __builtin_amdgcn_s_barrier();
  read_frags(0, smem, smem + BRING_U4, 0, 0);

  int t = 0, bs = 0;
  for (int cc = 0; cc < ncc; ++cc) {
    const bool next_slice = cc + 1 < ncc;
    const uint4* Hc = smem + (cc & 1) * HSTAGE;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int ky = tp / 3, kx = tp % 3;
      // next tap: (tp+1) of this slice, or tap 0 of the next slice (other halo stage)
      const int ntp = tp == 8 ? 0 : tp + 1;
      const uint4* Hn = smem + ((tp == 8 ? cc + 1 : cc) & 1) * HSTAGE;
      const uint4* Bc = smem + BRING_U4 + bs * B_U4;
      int nbs = bs + 1; if (nbs == HD_NSB) nbs = 0;
      const uint4* Bn = smem + BRING_U4 + nbs * B_U4;

      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      read_frags(1, Hc, Bc, ky, kx);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[0][j], af[0][i], acc[i][j]);   // D[n][pixel]
      // issue order inside the region: 1 MFMA, 1 ds_read, <= 2 address ops, ... (non-MFMA issues ride in the pipe's shadow)
#pragma unroll
      for (int m = 0; m < MT * NT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((HD_NSB - 2) * B_I) : "memory");          // prologue: taps 2.. in flight
      else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((HD_NSB - 2) * B_I + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN_STEADY) : "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue_b(t + HD_NSB, bs);
      issue_halo_piece(cc + 1, tp, next_slice && tp < NPIECE);
      read_frags(0, Hn, Bn, ntp / 3, ntp % 3);         // past the last tap: reads of landed no-op data, never used
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[1][j], af[1][i], acc[i][j]);
#pragma unroll
      for (int m = 0; m < MT * NT; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);
      }
      bs = nbs;
      ++t;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing no-op pieces: no LDS-DMA may outlive the workgroup
  __syncthreads();

  // ---- epilogue: lane = pixel (row wm*4 + i, col frow), 4*NT consecutive channels (see hd_bswz) ----------------------
  const bool f_bias = a.flags & IMM_CONV_BIAS, f_relu = a.flags & IMM_CONV_RELU;
  const bool f_stats = a.flags & IMM_CONV_STATS, f_mask = a.flags & IMM_CONV_MASK;
  const int nb = n0 + wn * TN + q * (4 * NT);          // first channel of this lane
  float s1[NT][4], s2[NT][4], bv[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[j][r] = f_bias ? a.bias[nb + j * 4 + r] : 0.f; s1[j][r] = 0.f; s2[j][r] = 0.f; }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int64_t m = MAP8 ? ((int64_t)(img + wm) * 8 + 2 * i + (frow >> 3)) * 8 + (frow & 7)
                           : ((int64_t)img * a.ho + y0 + wm * 4 + i) * a.wo + x0 + frow;
    if (MAP8 && img + wm >= ha.n_img) continue;        // odd batch: the second image of the last pair does not exist
#pragma unroll
    for (int h = 0; h < NT / 2; ++h) {                 // 8 channels = one 16-byte store
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] = acc[i][2 * h + (e >> 2)][e & 3] + bv[2 * h + (e >> 2)][e & 3];
        if (f_relu) v[e] = fmaxf(v[e], 0.f);
      }
      float mf[8];
      if (f_mask) {
        unpack8<ET>(*(const uint4*)(a.mask + m * a.ldmask + nb + 8 * h), mf);
#pragma unroll
        for (int e = 0; e < 8; ++e) if (!(mf[e] > 0.f)) v[e] = 0.f;
      }
      if (f_stats) {
        // STATS | MASK: second sum = sum(v * mask_ref) — the batch-norm backward sums of the layer this gradient enters
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[2 * h + (e >> 2)][e & 3] += v[e];
          s2[2 * h + (e >> 2)][e & 3] += v[e] * (f_mask ? mf[e] : v[e]);
        }
      }
      *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = pack8<ET>(v);
    }
  }
  if (f_stats) {
    float* red = (float*)smem;                         // [4 wm][2][BN]
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
    __syncthreads();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW / 2; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
      a.stats[((int64_t)patch * 2 + 0) * a.co + n0 + tid] = t1;
      a.stats[((int64_t)patch * 2 + 1) * a.co + n0 + tid] = t2;
    }
  }
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
  return cu;
}

// Tile plan of a layer: patch height (16 = 8-wave kernel, 8 = 4-wave kernel) and channel-block width.
//   16 x 16 patch x 128 channels when that still gives every CU a workgroup;
//   16 x 16 x 64 when THAT does;
//   else 8 x 16 x 64 (4 waves, two workgroups per CU) when the map height allows — the small-grid layers (16x16 maps of
//   a 32-image batch: 32 patches) otherwise leave half the chip idle.
struct HdPlan { int ph, bn, n_patches, n_wg; bool map8; };

static HdPlan hd_plan(const imm_conv_desc* d) {
  static const bool no_small = getenv("IMM_HDEEP_NO_SMALL") != nullptr;
  const int cus = hd_num_cu();
  static const int small_below = getenv("IMM_HDEEP_SMALL_BELOW") ? atoi(getenv("IMM_HDEEP_SMALL_BELOW")) : 4;   // x CUs
  static const bool no_big = getenv("IMM_HDEEP_NO_BIG") != nullptr;
  HdPlan p;
  p.map8 = false;
  if (d->ho == 8 && d->wo == 8) {                      // two whole 8x8 images per (4-wave) workgroup
    static const bool no_map8 = getenv("IMM_HDEEP_NO_MAP8") != nullptr;
    p.ph = 8; p.bn = 64; p.map8 = true;
    p.n_patches = no_map8 ? 0 : (d->batch + 1) / 2;
    p.n_wg = p.n_patches * (d->co / 64);
    return p;
  }
  const int np16 = (d->ho % 16 == 0) ? d->batch * (d->ho / 16) * (d->wo / HD_PW) : 0;
  if (!no_big && np16 > 0 && d->co % 128 == 0 && np16 * (d->co / 128) >= cus) { p.ph = 16; p.bn = 128; p.n_patches = np16; }
  else if (np16 > 0 && (np16 * (d->co / 64) >= small_below * cus || no_small)) { p.ph = 16; p.bn = 64; p.n_patches = np16; }
  else if (!no_small) { p.ph = 8; p.bn = 64; p.n_patches = d->batch * (d->ho / 8) * (d->wo / HD_PW); }
  else { p.ph = 16; p.bn = 64; p.n_patches = np16; }
  p.n_wg = p.n_patches * (d->co / p.bn);
  return p;
}

bool imm_hdeep_applicable(const imm_conv_desc* d) {
  static const bool off = getenv("IMM_NO_HDEEP") != nullptr;
  if (off) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 1 || d->updiv != 1 || d->pad_t != 1 || d->pad_l != 1) return false;
  if (d->ci % 64 || d->co % 64 || d->ci < 64) return false;
  if (d->out_scale > 1 || (d->flags & (IMM_CONV_OUT_F32 | 0xf00))) return false;
  if (d->hi != d->ho || d->wi != d->wo || d->ho % 8 || (d->wo % HD_PW && !(d->ho == 8 && d->wo == 8))) return false;
  if (d->ldy % 8 || ((d->flags & IMM_CONV_MASK) && d->ldmask % 8)) return false;
  const int64_t px = (int64_t)d->batch * d->hi * d->wi;
  if (px * d->ldx * 2 >= (1LL << 31) || (int64_t)d->co * d->kpad * 2 >= (1LL << 31)) return false;
  // small grids keep the im2col kernel (64x64 tiles give it 4x the workgroups)
  static const int min_wg = getenv("IMM_HDEEP_MIN_WG") ? atoi(getenv("IMM_HDEEP_MIN_WG")) : 100;
  const HdPlan p = hd_plan(d);
  return p.n_patches > 0 && p.n_wg >= min_wg;
}

int imm_hdeep_stats_blocks(const imm_conv_desc* d) { return hd_plan(d).n_patches; }

template <typename ET, int BN, int NW, int PH, bool MAP8 = false>
static void hd_launch_cfg(const HdArgs& ha, hipStream_t s) {
  constexpr int hstage = MAP8 ? 25 * 64 : HD_HSTAGE(PH), nsb = MAP8 ? 3 : 4;
  constexpr int lds = (2 * hstage + 64 + nsb * BN * 8) * 16;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_hdeep_kernel<ET, BN, NW, PH, MAP8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_hdeep_kernel<ET, BN, NW, PH, MAP8>), dim3(ha.n_wg), dim3(NW * 64), lds, s, ha);
}

template <typename ET>
static void hd_launch(const HdPlan& p, const HdArgs& ha, hipStream_t s) {
  if (p.map8) hd_launch_cfg<ET, 64, 4, 8, true>(ha, s);
  else if (p.ph == 16 && p.bn == 128) hd_launch_cfg<ET, 128, 8, 16>(ha, s);
  else if (p.ph == 16) hd_launch_cfg<ET, 64, 8, 16>(ha, s);
  else hd_launch_cfg<ET, 64, 4, 8>(ha, s);
}

void imm_conv_hdeep_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  HdArgs ha;
  ha.c = a;
  const HdPlan p = hd_plan(d);
  ha.patches_x = p.map8 ? 1 : d->wo / HD_PW; ha.patches_y = p.map8 ? 1 : d->ho / p.ph;
  ha.n_img = d->batch;
  ha.n_patches = p.n_patches;
  ha.c.n_nblk = d->co / p.bn;
  ha.n_wg = p.n_wg;
  ha.c.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  if (dtype == IMM_BF16) hd_launch<BF16>(p, ha, s); else hd_launch<F16>(p, ha, s);
}
