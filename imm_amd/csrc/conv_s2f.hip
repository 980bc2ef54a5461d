// conv_s2f.hip — FORWARD 3x3 STRIDE-2 SAME convolution through an LDS halo: encoder conv_5 (64 -> 128 at 64x64 -> 32x32) and
// conv_7 (128 -> 256 at 32x32 -> 16x16), imm/models/imm_model.py:204,211 (tf.nn.conv2d(strides=2, 'SAME') at nn_utils.py:100;
// on even sides TF pads 0 rows / columns before and 1 after).  Until round 5 these two ran on the im2col kernel
// (conv_igemm64: 6-8 % matrix duty, 6-10 VALU instructions per MFMA, 19.2 / 15.7 us for 4.8 GFLOP each —
// profiles/r04_v2_pmc_sq_ratios.txt): every K tile re-gathers its pixels with per-lane address arithmetic.
//
// Here the stride-2 data gradient's trick (conv_hdeep.hip S2D) runs in reverse.  A workgroup owns an 8x16 patch of OUTPUT pixels
// x BN channels; its input footprint is 17 rows x 33 columns, DMA'd once per 32-channel slice into LDS with the columns
// DE-INTERLEAVED BY PARITY (the per-lane global offsets of the DMA do the permutation for free):
//     LDS slot of input pixel (2 y0 + r, 2 x0 + 2 j + p)  =  (2 r + p) * 17 + j            r < 17, p < 2, j < 17
// so tap (ky, kx) of output pixel (y0 + i, x0 + f) reads slot (2 (2 i + ky) + (kx & 1)) * 17 + f + (kx >> 1): the 16 lanes of an
// MFMA operand row read 16 CONSECUTIVE 64-byte slots, exactly the stride-1 access pattern (same chunk swizzle, no bank
// conflicts), and every tap is an immediate offset.  The K loop runs over 32-channel slices with the row-at-a-time schedule of
// conv_hdeep.hip (ROW3): the nine filter taps of a slice are resident (stage = tap, BN x 64 B each), a barrier interval is one
// filter row (3 k-steps), the row's three stages are refilled with the next slice's taps in the three k-steps after its barrier,
// the next slice's halo (its stage has been free since the slice before ended) rides on the window of a slice's first barrier.  Every
// wave issues the same number of DMA pieces per window (missing ones are out-of-range pieces into a dump slot), so the counted
// s_waitcnt vmcnt at a barrier — "everything issued before the previous barrier has landed" — is exact.  Out-of-image input
// (the one row / column of bottom / right padding) is an out-of-range buffer offset: the DMA writes zeros.
// Wave (wm, wn) of the NW = BN / 16 waves owns patch rows 4 wm .. 4 wm + 3 x 32 channels (8 MFMAs per k-step); a lane ends up with 8
// consecutive channels of its pixel (conv_hdeep.hip's filter-row permutation): one 16-byte store per pixel.  Epilogue: bias and
// the batch-norm partial sums (one row per patch) — the stride-2 layers are conv + BN + ReLU blocks (imm_model.py:182-217).
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define S2_PH 8
#define S2_PW 16
#define S2_ROWS 17                              // input rows of a patch: 2 * 8 + 1
#define S2_RP 17                                // slots per (row, parity) run: 16 + 1
#define S2_SLOTS (S2_ROWS * 2 * S2_RP)          // 578
#define S2_HINSTR 37                            // DMA instructions per halo stage (16 slots x 64 B each)
#define S2_HSTAGE (S2_HINSTR * 1024)            // 37 888 B
#define S2_RUNB (S2_RP * 64)                    // bytes per (row, parity) run: 1 088
#define S2_OOB 0x80000000u

struct S2fArgs {
  ConvArgs c;
  int n_patches, patches_x, patches_y, n_wg;
};

template <int I, int N, typename F>
__device__ __forceinline__ void s2_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    s2_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void s2_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// 64-byte LDS rows: chunk ^= 2 * bit 2 of the slot's column (halo) / 2 * bit 4 of the filter row — a ds_read_b128 lane group
// ({0-3, 12-15, 20-27}, ...; MI355X_MICROARCH.md) then hits 16 distinct 16-byte slots mod 256 B (checked by enumeration)
__device__ __forceinline__ int s2_aswz(int j) { return ((j >> 2) & 1) << 1; }
__device__ __forceinline__ int s2_bswz(int r) { return ((r >> 4) & 1) << 1; }

// total over each row of 16 lanes, in every lane (DPP row rotations; conv_hdeep.hip)
__device__ __forceinline__ float s2_row_sum16(float x) {
  float y;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
               : "=&v"(y) : "v"(x));
  return y;
}

template <typename ET, int BN>
__global__ __launch_bounds__(BN * 4) void conv_s2f_kernel(const S2fArgs ha) {
  const ConvArgs& a = ha.c;
  constexpr int NW = BN / 16;                          // 4 (BN 64) / 8 (BN 128) waves
  constexpr int MT = 4, NT = 2;
  constexpr int NPIECE = (S2_HINSTR + NW - 1) / NW;    // halo DMA pieces per wave and slice: 10 / 5
  static_assert(NPIECE <= 12, "halo pieces of a slice ride on the three k-steps after its first barrier, four per k-step at most");
  constexpr int BSTAGE = BN * 64;
  constexpr int DUMP = 2 * S2_HSTAGE, RING = DUMP + 1024;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [2] halo stages | dump slot | [9] filter stages

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_assume(wid >= 0 && wid < NW);
  const int wm = wid & 1, wn = wid >> 1;
  const int frow = lane & 15, q = lane >> 4;
  const uint64_t xa = (uint64_t)a.x, wa = (uint64_t)a.wt;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t wr = {(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, a.wt_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;
  const int ci = a.ci8 << 3;
  const int nsl = ci >> 5;                             // 32-channel slices

  // ---- tile ------------------------------------------------------------------------------------------------------------------
  int bid = blockIdx.x;
  {   // XCD-contiguous order: the channel blocks of a patch and neighbouring patches share one L2
    const int xq = ha.n_wg >> 3, xr_ = ha.n_wg & 7, xcd = bid & 7;
    bid = (xcd < xr_ ? xcd * (xq + 1) : xr_ * (xq + 1) + (xcd - xr_) * xq) + (bid >> 3);
  }
  const int nblk = bid % a.n_nblk, patch = bid / a.n_nblk;
  const int per_img = ha.patches_x * ha.patches_y;
  const int img = patch / per_img, pr = patch - img * per_img;
  const int y0 = (pr / ha.patches_x) * S2_PH, x0 = (pr % ha.patches_x) * S2_PW;
  const int n0 = nblk * BN;

  // halo piece k of this wave = DMA instruction wid + NW k: 16 slots x 4 chunks; slot s = (2 r + p) * 17 + j.  Instructions >= 37
  // (and every piece of a slice that does not exist) are out-of-range loads into the dump slot: the per-window count stays uniform.
  uint32_t h_voff[NPIECE];
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) {
    const int s = (wid + NW * k) * 16 + (lane >> 2);
    const int rp = s / S2_RP, j = s - rp * S2_RP;
    const int r = rp >> 1, p = rp & 1;
    const int iy = 2 * y0 + r, ix = 2 * x0 + 2 * j + p;
    const bool ok = s < S2_SLOTS && (p == 0 || j < 16) && iy < a.hi && ix < a.wi;
    h_voff[k] = ok ? (uint32_t)((iy * a.wi + ix) * a.ldx * 2 + (((lane & 3) ^ s2_aswz(j)) * 16)) : S2_OOB;
  }
  const uint32_t h_soff = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(img * a.hi * a.wi) * (uint32_t)(a.ldx * 2)));
  uint32_t b_voff;
  {
    const int r = wid * 16 + (lane >> 2);              // filter row of the BN block: one DMA instruction per wave and stage
    b_voff = (n0 + r < a.co) ? (uint32_t)((n0 + r) * a.kpad * 2 + (((lane & 3) ^ s2_bswz(r)) * 16)) : S2_OOB;
  }
  auto dma_b = [&](int sl, int tp, bool real) {        // filter tap tp of slice sl -> stage tp
    const uint32_t soff = (uint32_t)((tp * ci + sl * 32) * 2);
    s2_dma16(wr, lds_base + (uint32_t)(real ? RING + tp * BSTAGE + wid * 1024 : DUMP), real ? b_voff : S2_OOB, soff);
  };
  auto dma_h = [&](int sl, int hs, int k, bool real) {   // halo piece k of slice sl -> halo stage hs
    const int i = wid + NW * k;
    const bool ex = real && i < S2_HINSTR;
    s2_dma16(xr, lds_base + (uint32_t)(ex ? hs * S2_HSTAGE + i * 1024 : DUMP), ex ? h_voff[k] : S2_OOB, h_soff + (uint32_t)(sl * 64));
  };

  // ---- prologue: bias (inline asm, ahead of the DMA: see conv_hdeep6.hip), halo of slice 0, the nine taps of slice 0 ------------
  const bool f_bias = a.flags & IMM_CONV_BIAS;
  f32x4_t acc[MT][NT], bias4[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bias4[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  if (f_bias) {
    const float* bp = a.bias + n0 + wn * 32 + q * 8;
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4[j]) : "v"(bp + 4 * j) : "memory");
  }
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) dma_h(0, 0, k, true);
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) dma_b(0, tp, true);

  // per-lane fragment offsets (bytes).  A: slot (2 (2 (4 wm + i) + ky) + (kx & 1)) * 17 + frow + (kx >> 1), logical chunk q — one base
  // per column shift (kx >> 1), everything else immediate; B: filter row wn*32 + (frow >> 2)*8 + 4 j + (frow & 3), chunk q
  uint32_t aoff[2], boff;
#pragma unroll
  for (int sh = 0; sh < 2; ++sh)
    aoff[sh] = (uint32_t)(((16 * wm * S2_RP + frow + sh) * 4 + (q ^ s2_aswz(frow + sh))) * 16);
  {
    const int r = wn * 32 + (frow >> 2) * 8 + (frow & 3);
    boff = (uint32_t)(RING + (r * 4 + (q ^ s2_bswz(r))) * 16);
  }
  uint4 af[2][MT], bf[2][NT];
  const char* const lds = (const char*)smem;
  // fragments of tap (ky, kx) of a slice whose halo sits in stage hs; filter stage 3 ky + kx
  auto read_frags = [&](const int buf, const int hs, const int ky, const int kx) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
      af[buf][i] = *(const uint4*)(lds + aoff[kx >> 1] + (hs * S2_HSTAGE + (2 * (2 * i + ky) + (kx & 1)) * S2_RUNB));
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[buf][j] = *(const uint4*)(lds + boff + ((3 * ky + kx) * BSTAGE + j * 256));
  };
#define S2_MFMA(buf, m0, m1)                                                                              \
  _Pragma("unroll") for (int m = (m0); m < (m1); ++m)                                                       \
    acc[m % MT][m / MT] = ET::mfma(bf[buf][m / MT], af[buf][m % MT], acc[m % MT][m / MT])
#define S2_FENCE() __builtin_amdgcn_sched_barrier(0)

  // halo of slice 0 and filter row 0 are in (rows 1 and 2 — six pieces — may still be on their way)
  asm volatile("s_waitcnt vmcnt(6)" : "+v"(bias4[0]), "+v"(bias4[1]) :: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = bias4[j];
  read_frags(0, 0, 0, 0);

  // Unrolled over PAIRS of slices (ci % 64 == 0): nine k-steps per slice would flip the parity of the fragment double buffer from
  // slice to slice; over 18 the buffers, the halo stage (= slice parity) and every LDS offset are compile-time constants.
  const int nss = nsl >> 1;
  for (int ss = 0; ss < nss; ++ss) {
    const bool more_pairs = ss + 1 < nss;
    s2_static_for<0, 18>([&](auto uc) __attribute__((always_inline)) {
      constexpr int u = decltype(uc)::value;
      constexpr int h = u / 9, t = u % 9, ky = t / 3, kx = t % 3, cur = u & 1;
      const int sl = 2 * ss + h;                                   // this k-step's slice; its halo sits in stage h
      const bool more = h == 0 ? true : more_pairs;                // a slice after this one exists
      // DMA window slot this k-step carries: the window opened by the LAST barrier — B(sl, ky - 1) for kx < 2 (slots 1, 2), by this
      // k-step's own barrier B(sl, ky) for kx == 2 (slot 0, issued after it).  The window of B(sl', r) refills filter row r with
      // slice sl' + 1; the window of a slice's FIRST barrier also carries the whole halo of slice sl' + 1.
      constexpr int wslot = kx == 2 ? 0 : kx + 1;
      constexpr int wrow = kx == 2 ? ky : (ky + 2) % 3;            // the row the window's barrier freed
      S2_FENCE();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // fragments of this tap are in registers
      S2_FENCE();
      if constexpr (kx < 2) {
        read_frags(cur ^ 1, h, ky, kx + 1);
      } else {
        // row ky has been read by this wave; everything issued before the previous barrier has landed when at most the pieces
        // issued since are outstanding: 3 filter stages, plus the next slice's whole halo in the window of a slice's FIRST barrier
        // (it — and the refilled row 0 — must be in at the slice's LAST barrier: the next slice starts right behind it)
        constexpr int allow = ky == 1 ? 3 + NPIECE : 3;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(allow) : "memory");
        __builtin_amdgcn_s_barrier();
        S2_FENCE();
        if constexpr (ky == 2) {
          if (more) read_frags(cur ^ 1, h ^ 1, 0, 0);
        } else {
          read_frags(cur ^ 1, h, ky + 1, 0);
        }
      }
      S2_MFMA(cur, 0, 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
      S2_FENCE();
      // the window's pieces of this slot.  Windows opened in this slice load slice sl + 1; the window of the previous slice's last
      // barrier (k-steps (0, 0), (0, 1)) loads row 2 of THIS slice — in slice 0 the prologue has done that.
      constexpr bool prev_window = ky == 0 && kx < 2;
      if (!(prev_window && sl == 0)) {
        const bool real = prev_window || more;
        dma_b(prev_window ? sl : sl + 1, 3 * wrow + wslot, real);
        S2_FENCE();
        S2_MFMA(cur, 4, 6);
        S2_FENCE();
        if constexpr (wrow == 0) {
          if constexpr (wslot < NPIECE) dma_h(sl + 1, h ^ 1, wslot, more);
          if constexpr (wslot + 3 < NPIECE) dma_h(sl + 1, h ^ 1, wslot + 3, more);
          S2_FENCE();
          S2_MFMA(cur, 6, 8);
          S2_FENCE();
          if constexpr (wslot + 6 < NPIECE) dma_h(sl + 1, h ^ 1, wslot + 6, more);
          if constexpr (wslot + 9 < NPIECE) dma_h(sl + 1, h ^ 1, wslot + 9, more);
        } else {
          S2_MFMA(cur, 6, 8);
        }
      } else {
        S2_MFMA(cur, 4, 8);
      }
    });
  }
#undef S2_MFMA
#undef S2_FENCE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // trailing dummy pieces: no LDS-DMA may outlive the workgroup
  __syncthreads();

  // ---- epilogue: lane = pixel (row 4 wm + i, col frow), 8 consecutive channels; bias is in the accumulators ----------------------
  const bool f_stats = a.flags & IMM_CONV_STATS;
  const int nb = n0 + wn * 32 + q * 8;
  const int64_t m_first = ((int64_t)img * a.ho + y0 + wm * 4) * a.wo + x0 + frow;
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int64_t m = m_first + (int64_t)i * a.wo;
    float v[8];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[4 * j + r] = acc[i][j][r];
        s1[j][r] += acc[i][j][r];
        s2[j][r] = fmaf(acc[i][j][r], acc[i][j][r], s2[j][r]);
      }
    if (nb < a.co) *(uint4*)((uint16_t*)a.y + m * a.ldy + nb) = pack8<ET>(v);
  }
  if (f_stats) {
    float* red = (float*)smem;                         // [2 wm][2][BN]: the halo stages are dead
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[j][r] = s2_row_sum16(s1[j][r]); s2[j][r] = s2_row_sum16(s2[j][r]); }
    if (frow == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nl = wn * 32 + q * 8 + j * 4 + r;
          red[(wm * 2 + 0) * BN + nl] = s1[j][r];
          red[(wm * 2 + 1) * BN + nl] = s2[j][r];
        }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.co) {
      a.stats[((int64_t)patch * 2 + 0) * a.co + n0 + tid] = red[0 * BN + tid] + red[2 * BN + tid];
      a.stats[((int64_t)patch * 2 + 1) * a.co + n0 + tid] = red[1 * BN + tid] + red[3 * BN + tid];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int s2f_num_cu() {
  static int cu = 0;
  if (cu == 0) {
    hipDeviceProp_t p; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cu = p.multiProcessorCount;
    else (void)hipGetLastError();
    if (cu <= 0) cu = 256;
  }
  return cu;
}

// channel-block width: 128 (8 waves) when that still gives every CU a workgroup, else 64 (4 waves)
static int s2f_bn(const imm_conv_desc* d) {
  const int np = d->batch * (d->ho / S2_PH) * (d->wo / S2_PW);
  return (d->co % 128 == 0 && np * (d->co / 128) >= s2f_num_cu()) ? 128 : 64;
}

bool imm_s2f_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("s2f");
  if (off) return false;
  if (d->kh != 3 || d->kw != 3 || d->stride != 2 || d->updiv != 1 || d->pad_t != 0 || d->pad_l != 0) return false;
  if ((d->hi & 1) || (d->wi & 1) || d->ho != d->hi / 2 || d->wo != d->wi / 2) return false;
  if (d->ci % 64 || d->co % 64 || d->kpad != 9 * d->ci) return false;     // pairs of 32-channel slices (32 -> 64, encoder conv_3: conv_halo.hip)
  if (d->ho % S2_PH || d->wo % S2_PW || d->ldy % 8) return false;
  if (d->out_scale > 1 || (d->flags & ~(IMM_CONV_BIAS | IMM_CONV_STATS))) return false;
  const int64_t px = (int64_t)d->batch * d->hi * d->wi;
  if (px * d->ldx * 2 >= (1LL << 31) || (int64_t)d->co * d->kpad * 2 >= (1LL << 31)) return false;
  return true;
}

int imm_s2f_stats_blocks(const imm_conv_desc* d) { return d->batch * (d->ho / S2_PH) * (d->wo / S2_PW); }
int imm_s2f_variant(const imm_conv_desc* d) { return 700000 + s2f_bn(d); }

template <typename ET, int BN>
static void s2f_launch_cfg(const S2fArgs& ha, hipStream_t s) {
  constexpr int lds = 2 * S2_HSTAGE + 1024 + 9 * BN * 64;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_s2f_kernel<ET, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_s2f_kernel<ET, BN>), dim3(ha.n_wg), dim3(BN * 4), lds, s, ha);
}

void imm_conv_s2f_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  S2fArgs ha;
  ha.c = a;
  const int bn = s2f_bn(d);
  ha.patches_x = d->wo / S2_PW; ha.patches_y = d->ho / S2_PH;
  ha.n_patches = d->batch * ha.patches_x * ha.patches_y;
  ha.c.n_nblk = d->co / bn;
  ha.n_wg = ha.n_patches * ha.c.n_nblk;
  ha.c.x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  if (dtype == IMM_BF16) { if (bn == 128) s2f_launch_cfg<BF16, 128>(ha, s); else s2f_launch_cfg<BF16, 64>(ha, s); }
  else { if (bn == 128) s2f_launch_cfg<F16, 128>(ha, s); else s2f_launch_cfg<F16, 64>(ha, s); }
}
