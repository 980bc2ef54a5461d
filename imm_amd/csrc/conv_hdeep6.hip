// conv_hdeep6.hip — the 16x16-pixel x 128-channel tile of conv_hdeep.hip with SIX k-steps (96 MFMAs per wave) per barrier
// interval instead of two: VGG conv2_1 .. conv4_3 forward and the data gradients that take the 128-wide tile
// (imm/models/selfsup/vgg16.py:182-189,349-362; renderer data gradients, imm/models/imm_model.py:154-179).
//
// Why (round 5; profiles/r04_v2_pmc_sq_ratios.txt): the tap-at-a-time form runs 32 MFMAs per wave between two
// barriers — 512 cycles of matrix work per wave against ~1 700 cycles per tap measured from inside (barrier, DMA issue, the LDS
// round trip of the next fragments): 25-30 % matrix duty, 38 % of wave cycles parked.  The row-at-a-time schedule of the 4-wave
// tiles (three taps per interval) cut that 3x, but nine 16 KB tap stages + two 41 KB halo stages do not fit 160 KB, and the
// 16x32-pixel tile that would have fitted spilled (128 accumulators).  Here the K loop runs over 32-CHANNEL slices:
//   * a halo stage is 324 pixels x 64 B = 21 KB (two stages, slice parity = stage: static addresses),
//   * a filter stage is one tap x 32 channels x 128 output channels = 8 KB,
//   * the ring is TWO GROUPS of six stages (96 KB): interval g reads group g & 1 while the DMA fills the other one,
//   * a barrier interval is six consecutive k-steps of the stream (slice, tap) — 18 k-steps = 3 intervals per 64 input channels,
//     so the code is unrolled over one 64-channel "super-slice" and every LDS offset except the group base is an immediate.
// 64 accumulator + 64 fragment registers per wave as before (no spills), 138 KB of LDS, one workgroup (8 waves) per CU.
//
// Loader timeline (per super-slice ss; A/B/C = its three intervals; "after X" = between X's barrier and the next one):
//     after C(ss-1): halo (ss, slice 1) + filters of B(ss)        after A(ss): filters of C(ss)
//     after B(ss):   halo (ss+1, slice 0) + filters of A(ss+1)
// i.e. everything is requested one interval (~1.5 us of matrix work) ahead and every barrier waits with vmcnt(0); the pieces of a
// slot are issued 3 right after the barrier and 3 + 3 inside the first two k-steps of the following interval, between MFMA
// quarters.  Persistent workgroups (more tiles than CUs) simply let the loader run on into the next tile.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

// compile-time loop: the k-step index decides fragment buffers, LDS immediates and which DMA pieces ride on the step
template <int I, int N, typename F>
__device__ __forceinline__ void h6_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    h6_static_for<I + 1, N>(f);
  }
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define H6_PW 16
#define H6_HW 18
#define H6_SLOTS (18 * 18)
#define H6_HINSTR 21                          // halo DMA instructions per stage: 16 pixels x 64 B each
#define H6_HSTAGE (H6_HINSTR * 1024)          // bytes per halo stage (21 504)
#define H6_BSTAGE 8192                        // bytes per filter stage: 128 rows x 64 B
#define H6_GROUP (6 * H6_BSTAGE)
#define H6_RING (2 * H6_HSTAGE)               // byte offset of the filter ring
#define H6_LDS (H6_RING + 2 * H6_GROUP)       // 141 312
#define H6_ROWB (H6_HW * 64)                  // bytes per halo row (1 152)
#define H6_OOB 0x80000000u
// -DIMM_H6_ABLATE=<bits> (diagnosis builds only, tools/ablate_build.sh; results are wrong, only the time is read):
//   1 no DMA inside the loop, 2 no fragment reads inside the loop, 4 no barrier / vmcnt wait inside the loop, 8 no MFMAs,
//   16 no output stores
#ifndef IMM_H6_ABLATE
#define IMM_H6_ABLATE 0
#endif
// IMM_H6_ROLL (round 6, default 1): the nine taps of a slice are walked COLUMN by column (kx outer, ky inner) and the A fragments are
// a rolling window of six halo rows per column: tap (ky, kx) multiplies rows ky .. ky+3, so ky = 1 / 2 each need ONE new row instead
// of four — 6 instead of 12 A reads per column, 18 instead of 24 ds_read_b128 per three k-steps.  The LDS pipeline runs about as
// long as the matrix pipeline in this kernel (MFMA-only 41.5 us, everything but the MFMAs 36.7 us, together
// 63.4 for conv4_2), so a quarter fewer operand reads is time.  The filter image is unchanged: the DMA picks memory tap 3 ky + kx
// for stream position 3 kx + ky.  -DIMM_H6_ROLL=0: the row-major walk with four fresh A rows per k-step (A/B builds).
#ifndef IMM_H6_ROLL
#define IMM_H6_ROLL 1
#endif


struct H6Args {
  ConvArgs c;
  int n_patches, patches_x, patches_y, n_wg;
};

__device__ __forceinline__ void h6_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// 64-byte LDS rows (4 chunks of 16 B): a ds_read_b128 is served in groups of 16 lanes ({0-3, 12-15, 20-27}, ... —
// MI355X_MICROARCH.md, LDS table), conflict-free when the group's 16 chunks are distinct mod 256 B.
//   halo pixel x of a row: chunk ^= 2 * bit 2 of x (conv_halo.hip's swizzle for 32-channel pixels);
//   filter row r: the lanes frow = 0..15 of an operand read rows (frow >> 2) * 16 + 4 j + (frow & 3): chunk ^= 2 * bit 5 of r.
__device__ __forceinline__ int h6_aswz(int hx) { return ((hx >> 2) & 1) << 1; }
__device__ __forceinline__ int h6_bswz(int r) { return ((r >> 5) & 1) << 1; }

template <typename ET, bool PERSIST>
__global__ __launch_bounds__(512) void conv_hdeep6_kernel(const H6Args ha) {
  const ConvArgs& a = ha.c;
  constexpr int BN = 128, TN = 64, MT = 4, NT = 4;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [2] halo stages | [2][6] filter stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wid >= 0 && wid < 8);
  const int wm = wid >> 1, wn = wid & 1;
  const int frow = lane & 15, q = lane >> 4;
  const int per_img = ha.patches_x * ha.patches_y;
  const uint64_t xa = (uint64_t)a.x, wa = (uint64_t)a.wt;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t wr = {(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, a.wt_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;
  const int ci = a.ci8 << 3;
  const int ncc = ci >> 6;                             // 64-channel super-slices

  // ---- tile coordinates and loader registers ------------------------------------------------------------------------------
  int e_patch, e_img, e_y0, e_x0, e_n0, n_patch = 0, n_img = 0, n_y0 = 0, n_x0 = 0, n_n0 = 0;
  uint32_t h_voff[3], nh_voff[3], b_voff, nb_voff = H6_OOB;
  uint32_t h_soff, nh_soff = 0;
  auto locate = [&](int w, int& patch, int& img, int& y0, int& x0, int& n0) {
    int bid = w;
    {   // XCD-contiguous order: the n-blocks of a patch and neighbouring patches share one L2
      const int xq = ha.n_wg >> 3, xr_ = ha.n_wg & 7, xcd = bid & 7;
      bid = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + (bid >> 3);
    }
    const int nblk = bid % a.n_nblk;
    patch = bid / a.n_nblk;
    img = patch / per_img;
    const int pr = patch - img * per_img;
    y0 = (pr / ha.patches_x) * 16; x0 = (pr % ha.patches_x) * H6_PW;
    n0 = nblk * BN;
  };
  // halo piece k of this wave = DMA instruction wid + 8 k (16 pixels x 4 chunks); instructions >= 21 do not exist
  auto halo_offsets = [&](int img, int y0, int x0, uint32_t (&hv)[3], uint32_t& hs) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int hp = (wid + 8 * k) * 16 + (lane >> 2);
      const int hy = hp / H6_HW, hx = hp - hy * H6_HW;
      const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
      const bool ok = hp < H6_SLOTS && (unsigned)iy < (unsigned)a.hi && (unsigned)ix < (unsigned)a.wi;
      hv[k] = ok ? (uint32_t)((iy * a.wi + ix) * a.ldx * 2 + (((lane & 3) ^ h6_aswz(hx)) * 16)) : H6_OOB;
    }
    hs = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(img * a.hi * a.wi) * (uint32_t)(a.ldx * 2)));
  };
  auto filter_offsets = [&](int n0, uint32_t& bv) {
    const int r = wid * 16 + (lane >> 2);
    bv = (n0 + r < a.co) ? (uint32_t)((n0 + r) * a.kpad * 2 + (((lane & 3) ^ h6_bswz(r)) * 16)) : H6_OOB;
  };
  // one filter piece: this wave's 16 rows of k-step u (slice u / 9, tap u % 9) of super-slice ssx -> group grp, position pos
  constexpr int ABL = IMM_H6_ABLATE;
  bool in_loop = false;
  auto dma_b = [&](int ssx, int u, int grp, int pos) {
    if ((ABL & 1) && in_loop) return;
    const int h = u / 9, tq = u - h * 9;
    const int tp = IMM_H6_ROLL ? (tq % 3) * 3 + tq / 3 : tq;      // memory tap (3 ky + kx) of stream position tq
    const uint32_t soff = (uint32_t)((tp * ci + ssx * 64 + h * 32) * 2);
    h6_dma16(wr, lds_base + (uint32_t)(H6_RING + grp * H6_GROUP + pos * H6_BSTAGE + wid * 1024), b_voff, soff);
  };
  // halo piece k of 32-channel slice h of super-slice ssx -> halo stage h
  auto dma_h = [&](int ssx, int h, int k) {
    if ((ABL & 1) && in_loop) return;
    const int i = wid + 8 * k;
    if (i < H6_HINSTR) h6_dma16(xr, lds_base + (uint32_t)(h * H6_HSTAGE + i * 1024), h_voff[k], h_soff + (uint32_t)((ssx * 64 + h * 32) * 2));
  };

  const int G = PERSIST ? (int)gridDim.x : 1;
  const int n_mine = PERSIST ? (ha.n_wg - (int)blockIdx.x + G - 1) / G : 1;
  locate(blockIdx.x, e_patch, e_img, e_y0, e_x0, e_n0);
  halo_offsets(e_img, e_y0, e_x0, h_voff, h_soff);
  filter_offsets(e_n0, b_voff);

  // ---- prologue ------------------------------------------------------------------------------------------------------------
  // The first tile's bias comes through inline-asm loads issued AHEAD of the DMA: a compiler-visible load would make hipcc drain
  // vmcnt to zero before the first MFMA (it cannot see the asm DMA behind it), i.e. wait for the whole 69 KB prologue.  Being the
  // oldest operations in flight they are covered by every counted wait below (a count allows the YOUNGEST n to be outstanding).
  const bool f_bias = a.flags & IMM_CONV_BIAS;
  f32x4_t acc[MT][NT], bias4[NT];
  auto load_bias = [&](int n0_) {                      // later tiles (persistent): ordinary loads during the epilogue
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float4 b4 = f_bias ? *(const float4*)(a.bias + n0_ + wn * TN + q * (4 * NT) + j * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      bias4[j] = f32x4_t{b4.x, b4.y, b4.z, b4.w};
    }
  };
#pragma unroll
  for (int j = 0; j < NT; ++j) bias4[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (f_bias) {
    const float* bp = a.bias + e_n0 + wn * TN + q * (4 * NT);
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4[j]) : "v"(bp + 4 * j) : "memory");
  }
  // halo slice 0 of super-slice 0 and the six filter stages of interval A -> group 0; the tile starts as soon as the halo and
  // stage 0 are in, stages 1..5 are waited for one k-step at a time inside the first interval (RAMP below), halo slice 1 is
  // requested there too
#pragma unroll
  for (int k = 0; k < 3; ++k) dma_h(0, 0, k);
#pragma unroll
  for (int p = 0; p < 6; ++p) dma_b(0, p, 0, p);

  // per-lane fragment offsets (bytes).  A: halo pixel (row wm*4 + i + ky, col frow + kx), logical chunk q; B: filter row
  // wn*64 + (frow >> 2)*16 + 4 j + (frow & 3) (conv_hdeep.hip's permutation: a lane ends up with 16 consecutive channels of
  // its pixel), logical chunk q — the swizzle term does not depend on j, so tile j is an immediate offset of 256 B.
  uint32_t aoff[3], boff;
  auto frag_offsets = [&]() {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) aoff[kx] = (uint32_t)(((wm * 4 * H6_HW + frow + kx) * 4 + (q ^ h6_aswz(frow + kx))) * 16);
    const int r = wn * TN + (frow >> 2) * 16 + (frow & 3);
    boff = (uint32_t)(H6_RING + (r * 4 + (q ^ h6_bswz(r))) * 16);
  };
  frag_offsets();

#if IMM_H6_ROLL
  uint4 aw[2][MT + 2], bf[2][NT];                      // [column parity][halo row of the column's window], [k-step parity][tile]
#else
  uint4 af[2][MT], bf[2][NT];                          // [fragment buffer][tile]
#endif
  const char* const lds = (const char*)smem;
  // fragments of k-step u (of the super-slice stream) from filter group base vb (bytes)
  auto read_frags = [&](const int buf, const int u, const uint32_t vb) __attribute__((always_inline)) {
    if ((ABL & 2) && in_loop) return;
    const int h = u / 9, tp = u - h * 9, pos = u % 6;
#if IMM_H6_ROLL
    // stream position tp = 3 kx + ky; column (u / 3) keeps its six halo rows in aw[(u / 3) & 1]: ky = 0 brings rows 0..3, ky = 1 row 4,
    // ky = 2 row 5 (the rows a k-step multiplies are ky .. ky + 3)
    const int kx = tp / 3, ky = tp - kx * 3, set = (u / 3) & 1;
    if (ky == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) aw[set][i] = *(const uint4*)(lds + aoff[kx] + (h * H6_HSTAGE + i * H6_ROWB));
    } else {
      aw[set][MT - 1 + ky] = *(const uint4*)(lds + aoff[kx] + (h * H6_HSTAGE + (MT - 1 + ky) * H6_ROWB));
    }
#else
    const int ky = tp / 3, kx = tp - ky * 3;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[buf][i] = *(const uint4*)(lds + aoff[kx] + (h * H6_HSTAGE + (i + ky) * H6_ROWB));
#endif
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[buf][j] = *(const uint4*)(lds + vb + (pos * H6_BSTAGE + j * 256));
  };
  // MFMAs m0 .. m1-1 of the 16 of a k-step (tile m: pixel row m % 4, channel tile m / 4); uu = the k-step's stream position
#if IMM_H6_ROLL
#define H6_MFMA(buf, m0, m1)                                                                             \
  if (!(ABL & 8)) _Pragma("unroll") for (int m = (m0); m < (m1); ++m)                                      \
    acc[m % MT][m / MT] = ET::mfma(bf[(ABL & 2) ? 0 : buf][m / MT],                                       \
                                   aw[(ABL & 2) ? 0 : (u / 3) & 1][(ABL & 2) ? m % MT : m % MT + u % 3], acc[m % MT][m / MT])
#else
#define H6_MFMA(buf, m0, m1)                                                                             \
  if (!(ABL & 8)) _Pragma("unroll") for (int m = (m0); m < (m1); ++m)                                      \
    acc[m % MT][m / MT] = ET::mfma(bf[(ABL & 2) ? 0 : buf][m / MT], af[(ABL & 2) ? 0 : buf][m % MT], acc[m % MT][m / MT])
#endif
#define H6_FENCE() __builtin_amdgcn_sched_barrier(0)

  // this wave's pieces of halo slice 0 and filter stage 0 are in (stages 1..5 may be outstanding), then everybody's; the wait
  // "modifies" the bias registers so that nothing reads them before their loads have returned
  asm volatile("s_waitcnt vmcnt(5)" : "+v"(bias4[0]), "+v"(bias4[1]), "+v"(bias4[2]), "+v"(bias4[3]) :: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = bias4[j];
  int rd_grp = 0;                                      // filter group the current interval reads
  uint32_t vB = boff;
  read_frags(0, 0, vB);
  in_loop = true;
  int ld_ss = 0;                                       // super-slice the loader's "next" pieces belong to
  bool ld_real = true;                                 // false: past the last tile of this workgroup

  for (int it = 0; it < n_mine; ++it) {
    const bool have_next = PERSIST && it + 1 < n_mine;
    for (int ss = 0; ss < ncc; ++ss) {
      const bool last_ss = ss + 1 == ncc;
      const bool first = it == 0 && ss == 0;           // RAMP: the first interval of the workgroup overlaps its own prologue
      if (PERSIST && last_ss && have_next) {           // the loader turns to the next tile after this super-slice's interval B
        locate(blockIdx.x + (it + 1) * G, n_patch, n_img, n_y0, n_x0, n_n0);
        halo_offsets(n_img, n_y0, n_x0, nh_voff, nh_soff);
        filter_offsets(n_n0, nb_voff);
      }
      h6_static_for<0, 18>([&](auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        constexpr int v = u / 6, pos = u % 6, cur = u & 1;
        constexpr int vp = (v + 2) % 3;                // type of the slot that is still being issued: after the previous interval
        H6_FENCE();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments of k-step u are in registers
        H6_FENCE();
        if constexpr (pos < 5) {
          if constexpr (v == 0) {
            if (first) {
              // filter stage u + 1 of the prologue: younger than it are the stages behind it, the 3 + 3 pieces k-steps 0 and 1
              // issue and (from k-step 2 on) at least two pieces of halo slice 1
              constexpr int allow = u == 0 ? 4 : u == 1 ? 6 : u == 2 ? 8 : u == 3 ? 9 : 8;
              asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow) : "memory");
              __builtin_amdgcn_s_barrier();
              H6_FENCE();
            }
          }
          read_frags(cur ^ 1, u + 1, vB);
          H6_MFMA(cur, 0, 4);
          {
            // the next k-step's operand reads between this one's first MFMAs: 8 (a new column: four A rows) or 5 (one A row)
            constexpr int nrd = (IMM_H6_ROLL && (u + 1) % 3 != 0) ? 5 : 8;
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, nrd >= 6 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if constexpr (nrd >= 8) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
          if constexpr (pos < 2) {
            // pieces 3 + 3 pos .. 5 + 3 pos of the slot opened at the last barrier = filter stages 3 pos .. 3 pos + 2 of the interval
            // after next: A-slot -> C(ss) = k-steps 12.., B-slot -> A(ld_ss) = 0.., C-slot -> B(ld_ss) = 6..
            constexpr int ub = vp == 0 ? 12 : vp == 1 ? 0 : 6;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
              H6_FENCE();
              if (ld_real) dma_b(ld_ss, ub + 3 * pos + p, rd_grp ^ 1, 3 * pos + p);
              H6_FENCE();
              H6_MFMA(cur, 4 + 4 * p, 8 + 4 * p);
            }
          } else {
            if constexpr (u == 2) {
              if (first) {                             // halo slice 1 of super-slice 0 (later ones: the C-slot of the super-slice before)
                H6_FENCE();
#pragma unroll
                for (int k = 0; k < 3; ++k) dma_h(0, 1, k);
                H6_FENCE();
              }
            }
            H6_MFMA(cur, 4, 16);
          }
        } else {
          // every fragment of this interval has been read: its group is free; the other group (and a halo slice) were requested
          // one interval ago
          if (!(ABL & 4)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
          }
          H6_FENCE();
          rd_grp ^= 1;
          vB = boff + (uint32_t)(rd_grp * H6_GROUP);
          H6_MFMA(cur, 0, 4);
          H6_FENCE();
          if constexpr (v == 1) {                      // the loader moves on to the next super-slice / tile
            if (!last_ss) ld_ss = ss + 1;
            else if (have_next) {
#pragma unroll
              for (int k = 0; k < 3; ++k) h_voff[k] = nh_voff[k];
              h_soff = nh_soff; b_voff = nb_voff; ld_ss = 0;
            } else ld_real = false;
          }
          if constexpr (u < 17) read_frags(cur ^ 1, u + 1, vB);
          else if (!last_ss) read_frags(cur ^ 1, 0, vB);
          H6_MFMA(cur, 4, 8);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
          // pieces 0..2 of the slot this barrier opens: the halo slice (B-slot: slice 0, C-slot: slice 1 of super-slice ld_ss)
          if constexpr (v != 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              H6_FENCE();
              if (ld_real) dma_h(ld_ss, v - 1, k);
              H6_FENCE();
              H6_MFMA(cur, 8 + 4 * k, 12 + 4 * k);
            }
            H6_FENCE();
            if (ld_real) dma_h(ld_ss, v - 1, 2);
          } else {
            H6_MFMA(cur, 8, 16);
          }
        }
      });
    }
    const int img = e_img, y0 = e_y0, x0 = e_x0, n0 = e_n0;

    // ---- epilogue (conv_hdeep.hip's: lane = pixel (row wm*4 + i, col frow), 16 consecutive channels) -----------------------
    const bool f_relu = a.flags & IMM_CONV_RELU, f_mask = a.flags & IMM_CONV_MASK;
    const int nb = n0 + wn * TN + q * (4 * NT);
    if (have_next) load_bias(n_n0);
    const int64_t m_first = ((int64_t)img * a.ho + y0 + wm * 4) * a.wo + x0 + frow;
    const int64_t m_step = a.wo;
    typedef short s16x2_t __attribute__((ext_vector_type(2)));
    typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
    const s16x2_t zero2 = {0, 0}, one2 = {1, 1};
    if (ABL & 16) {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (t == 123456.789f) ((float*)a.y)[0] = t;      // keeps the accumulators live
    } else if (a.flags & IMM_CONV_TAP_) {
      // perceptual tap (imm_conv2d_tap): v = round16(acc) + c_k * lossmask[pixel] * (a_pred - a_gt), zero where a_pred <= 0
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
            v[2 * e] = acc[i][j][r]; v[2 * e + 1] = acc[i][j][r + 1];
          }
          unpack8<ET>(*(const uint4*)(a.mask + m * a.ldmask + nb + 8 * h), fp);
          unpack8<ET>(*(const uint4*)(a.tap_gt + m * a.ldmask + nb + 8 * h), fg);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float d = fp[e] - fg[e];
            if (a.tap_l1) d = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            float t = ET::to_f32(ET::from_f32(v[e])) + cm * d;
            if (!(fp[e] > 0.f)) t = 0.f;
            v[e] = t;
          }
          *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = pack8<ET>(v);
        }
      }
    } else {
      // plain / ReLU / masked store: packed 16-bit integer max / multiply on the converted outputs (sign-magnitude formats)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int64_t m = m_first + i * m_step;
#pragma unroll
        for (int h = 0; h < NT / 2; ++h) {
          uint32_t w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int j = 2 * h + (e >> 1), r = (e & 1) * 2;
            w[e] = ET::pack2(acc[i][j][r], acc[i][j][r + 1]);
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
              const s16x2_t ps = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(s16x2_t, mw[e]), zero2), one2);
              w[e] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2_t, w[e]) * __builtin_bit_cast(u16x2_t, ps));
            }
          }
          *(uint4*)((uint16_t*)a.y + m * a.ldy + nb + 8 * h) = make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
    }
    if (have_next) {
      // the next tile's interval A and halo slice 0 were waited for at this tile's last barrier
      read_frags(0, 0, vB);
      e_patch = n_patch; e_img = n_img; e_y0 = n_y0; e_x0 = n_x0; e_n0 = n_n0;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = bias4[j];
    }
  }   // tiles of this workgroup
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
#undef H6_MFMA
#undef H6_FENCE
}

// ---------------------------------------------------------------------------------------------
// host side (called from conv_hdeep.hip's tile plan: 16x16 patches x 128 channels, no batch-norm partial sums)
// ---------------------------------------------------------------------------------------------
bool imm_hdeep6_enabled() {
  static const bool off = imm_conv_disabled("hdeep6");
  return !off;
}

template <typename ET, bool PERSIST>
static void h6_launch_cfg(const H6Args& ha, int cus, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_hdeep6_kernel<ET, PERSIST>, hipFuncAttributeMaxDynamicSharedMemorySize, H6_LDS);
    attr_set = true;
  }
  const int grid = PERSIST ? (ha.n_wg < cus ? ha.n_wg : cus) : ha.n_wg;
  hipLaunchKernelGGL((conv_hdeep6_kernel<ET, PERSIST>), dim3(grid), dim3(512), H6_LDS, s, ha);
}

void imm_conv_hdeep6_launch(int dtype, const ConvArgs& a, int n_patches, int patches_x, int patches_y, int n_wg, bool persist, int cus,
                            hipStream_t s) {
  H6Args ha;
  ha.c = a;
  ha.n_patches = n_patches; ha.patches_x = patches_x; ha.patches_y = patches_y; ha.n_wg = n_wg;
  if (dtype == IMM_BF16) {
    if (persist) h6_launch_cfg<BF16, true>(ha, cus, s); else h6_launch_cfg<BF16, false>(ha, cus, s);
  } else {
    if (persist) h6_launch_cfg<F16, true>(ha, cus, s); else h6_launch_cfg<F16, false>(ha, cus, s);
  }
}
