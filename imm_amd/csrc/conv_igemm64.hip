// conv_igemm64.hip — implicit-GEMM convolution, deep-K variant for layers with ci % 64 == 0 (all of VGG16 and
// the 64..256-channel encoder / renderer layers): BK = 64 (one filter tap x 64 channels per K tile), operand
// tiles DMA'd straight from HBM/L2 into LDS (buffer_load_dwordx4 ... lds: no VGPR round trip, no ds_write),
// two LDS stages, 32 MFMAs per wave between barriers.
//
// LDS image: row = 128 bytes = 8 chunks of 16 B; chunk c of row r is stored at slot c ^ ((r>>1)&7), which makes
// every ds_read_b128 of a 16-row x 4-chunk MFMA fragment conflict-free.  LDS-DMA writes lane-linear
// (wave base + lane*16), so the permutation is applied on the SOURCE side: lane l of a wave instruction
// covers row (l>>3) of an 8-row group and fetches source chunk (l&7) ^ swz(row).  Out-of-image taps and
// rows beyond M / channels beyond co use an out-of-range buffer offset, for which the DMA writes zeros.
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

__device__ __forceinline__ int lds64_idx(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

// One LDS-DMA instruction: 64 lanes x 16 B from buffer `rsrc` (+ per-lane voff + scalar soff; out-of-range lanes
// deliver zeros) to LDS bytes [lds_addr, lds_addr + 1024).  Inline asm on purpose: hipcc models the builtin form
// as an LDS write and drains vmcnt(0) in front of the next ds_read, which serialises the ring; an asm statement is
// invisible to that bookkeeping, so the counted s_waitcnt below is the only wait.  M0 (LDS base of the DMA) is
// written in the same statement that uses it (the compiler does not preserve M0 across statements).
__device__ __forceinline__ void lds_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// NS-stage LDS ring: tile kt+NS-1 is DMA'd while tile kt feeds the MFMAs.  Ordering per K step (raw
// s_barrier, counted vmcnt -- a __syncthreads() would drain the DMA queue to zero):
//   s_waitcnt vmcnt((NS-2)*LPT)   this wave's share of tile kt has landed (LPT = DMA instructions per tile per wave)
//   s_barrier                     => every wave's share has landed, and every wave finished reading tile kt-1
//   issue tile kt+NS-1            into the stage tile kt-1 occupied
//   ds_read + MFMA on tile kt
// NW waves per workgroup (4 or 8): 8 waves halve each wave's tile (more waves per SIMD to hide DMA / LDS latency, at
// 1.5x the LDS bytes per MFMA).
template <typename ET, int BM, int BN, int NS, int NW>
__device__ __forceinline__ void conv_igemm64_body(const ConvArgs& a, int bid_in) {
  constexpr int WGM = NW / 2, WGN = 2;
  constexpr int TM = BM / WGM, TN = BN / WGN, MT = TM / 16, NT = TN / 16;
  constexpr int A_INSTR = BM / (8 * NW), B_INSTR = BN / (8 * NW);   // 8-row wave instructions per wave per tile
  static_assert(A_INSTR >= 1 && B_INSTR >= 1 && MT >= 1 && NT >= 1, "tile too small for the wave count");
  constexpr int BUF = (BM + BN) * 8;                    // uint4 per stage
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // NS * BUF
  constexpr int LPT = A_INSTR + B_INSTR;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  int bid = bid_in;
  {
    const int q = a.n_blocks >> 3, r = a.n_blocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nblk = bid % a.n_nblk, mblk = bid / a.n_nblk;
  const int m0 = mblk * BM, n0 = nblk * BN;

  constexpr uint32_t OOB = 0x80000000u;
  const uint64_t xa = (uint64_t)a.x, wa = (uint64_t)a.wt;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t wr = {(uint32_t)wa, (uint32_t)(wa >> 32) & 0xffffu, a.wt_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;

  // ---- loader state: this lane's rows and source chunk --------------------------------------------------
  const int lrow = lane >> 3, lchunk = lane & 7;
  int by[A_INSTR], bx[A_INSTR], pbase[A_INSTR], a_coff[A_INSTR];
  uint32_t a_voff[A_INSTR], b_voff[B_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int r = (wid * A_INSTR + i) * 8 + lrow;
    const int m = m0 + r;
    a_coff[i] = (lchunk ^ ((r >> 1) & 7)) * 16;
    if (m < a.M) {
      const int hw = a.ho * a.wo;
      const int img = m / hw, rem = m - img * hw;
      const int oy = rem / a.wo, ox = rem - oy * a.wo;
      by[i] = oy * a.stride - a.pad_t;
      bx[i] = ox * a.stride - a.pad_l;
      pbase[i] = img * a.hi * a.wi;
    } else {
      by[i] = -(1 << 28); bx[i] = -(1 << 28); pbase[i] = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < B_INSTR; ++j) {
    const int r = (wid * B_INSTR + j) * 8 + lrow;
    const int n = n0 + r;
    b_voff[j] = (n < a.co) ? (uint32_t)(n * a.kpad * 2 + (lchunk ^ ((r >> 1) & 7)) * 16) : OOB;
  }
  int ky = 0, kx = 0, cs = 0;           // wave-uniform tap walk; cs = 64-channel slice
  const int ncs = a.ci8 >> 3;
  auto tap_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      int iy = by[i] + ky, ix = bx[i] + kx;
      bool ok = true;
      if (a.updiv == 2) { ok = (((iy | ix) & 1) == 0); iy >>= 1; ix >>= 1; }
      ok = ok && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
      a_voff[i] = ok ? (uint32_t)((pbase[i] + iy * a.wi + ix) * a.ldx * 2 + a_coff[i]) : OOB;
    }
  };
  tap_offsets();

  auto issue_tile = [&](int kt, int buf) {
    const uint32_t a_lds = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((buf * BUF + (wid * A_INSTR) * 64) * 16));
    const uint32_t b_lds = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)((buf * BUF + BM * 8 + (wid * B_INSTR) * 64) * 16));
    const uint32_t a_soff = (uint32_t)(cs * 128), b_soff = (uint32_t)(kt * 128);
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) lds_dma16(xr, a_lds + i * 1024, a_voff[i], a_soff);
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) lds_dma16(wr, b_lds + j * 1024, b_voff[j], b_soff);
    if (++cs == ncs) {
      cs = 0;
      if (++kx == a.kw) { kx = 0; ++ky; }
      tap_offsets();
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < a.KT) issue_tile(t, t);

  const int frow = lane & 15, fchunk = lane >> 4;
  int stage = 0;
  for (int kt = 0; kt < a.KT; ++kt) {
    // tiles kt .. min(kt+NS-2, KT-1) are in flight; tile kt must have landed
    if (kt + NS - 2 < a.KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * LPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    {
      const int nt = kt + NS - 1;
      int ns = stage + NS - 1; if (ns >= NS) ns -= NS;
      if (nt < a.KT) issue_tile(nt, ns);
    }
    const uint4* Ab = smem + stage * BUF;
    const uint4* Bb = Ab + BM * 8;
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      uint4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = Ab[lds64_idx(wm * TM + i * 16 + frow, kh * 4 + fchunk)];
#pragma unroll
      for (int j = 0; j < NT; ++j) bf[j] = Bb[lds64_idx(wn * TN + j * 16 + frow, kh * 4 + fchunk)];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[j], af[i], acc[i][j]);
    }
    if (++stage == NS) stage = 0;
  }
  __syncthreads();   // all waves done with LDS before the epilogue reuses it for the BN partial sums
  conv_epilogue<ET, BM, BN, WGM, WGN, MT, NT>(a, acc, tid, wm, wn, m0, n0, mblk, (float*)smem);
}

template <typename ET, int BM, int BN, int NS, int NW>
__global__ __launch_bounds__(NW * 64) void conv_igemm64_kernel(const ConvArgs a) {
  conv_igemm64_body<ET, BM, BN, NS, NW>(a, blockIdx.x);
}

// Several convolutions of the same tile shape in ONE launch (the four parity classes of a stride-2 data gradient: each
// alone is a grid of 64-512 workgroups a few microseconds long).  Workgroup b belongs to member g with
// first[g] <= b < first[g+1] and runs that member's argument block unchanged.

template <typename ET, int BM, int BN, int NS, int NW>
__global__ __launch_bounds__(NW * 64) void conv_igemm64_group_kernel(const ConvArgsGroup g) {
  int m = 0;
#pragma unroll
  for (int i = 1; i < IMM_CONV_GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.first[i]) m = i;
  conv_igemm64_body<ET, BM, BN, NS, NW>(g.a[m], (int)blockIdx.x - g.first[m]);
}

template <typename ET, int BM, int BN, int NS, int NW = 4>
static void launch64_cfg(const ConvArgs& a, hipStream_t s) {
  constexpr int lds = NS * (BM + BN) * 128;
  static bool attr_set = false;   // per instantiation; > 64 KB of dynamic LDS needs the opt-in once
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm64_kernel<ET, BM, BN, NS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_igemm64_kernel<ET, BM, BN, NS, NW>), dim3(a.n_blocks), dim3(NW * 64), lds, s, a);
}

template <typename ET>
static void launch64(ConvArgs& a, int bm, int bn, hipStream_t s) {
  const int mblk = (a.M + bm - 1) / bm;
  a.n_blocks = mblk * a.n_nblk;
  a.KT = a.kpad / 64;
  // stage count: big tiles are throughput-bound and want 2-3 co-resident workgroups per CU (2 stages = 64 / 48 KB);
  // the 64x64 tile is used when the grid is small (deep layers), is latency-bound and wants a deeper ring instead
  if (bm == 128 && bn == 128) launch64_cfg<ET, 128, 128, 2>(a, s);
  else if (bm == 128 && bn == 64) launch64_cfg<ET, 128, 64, 2>(a, s);
  else launch64_cfg<ET, 64, 64, 4>(a, s);
}

template <typename ET>
static bool launch64_group(ConvArgs* args, int n, int bm, int bn, hipStream_t s) {
  if (!(bm == 64 && bn == 64) || n < 1 || n > IMM_CONV_GROUP_MAX) return false;
  constexpr int NS = 4, lds = NS * (64 + 64) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv_igemm64_group_kernel<ET, 64, 64, NS, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  ConvArgsGroup g;
  g.n = n;
  int total = 0;
  for (int i = 0; i < n; ++i) {
    args[i].n_blocks = ((args[i].M + 63) / 64) * args[i].n_nblk;
    args[i].KT = args[i].kpad / 64;
    g.a[i] = args[i];
    g.first[i] = total;
    total += args[i].n_blocks;
  }
  for (int i = n; i <= IMM_CONV_GROUP_MAX; ++i) g.first[i] = total;
  hipLaunchKernelGGL((conv_igemm64_group_kernel<ET, 64, 64, NS, 4>), dim3(total), dim3(256), lds, s, g);
  return true;
}

// grouped launch of up to 4 deep-K convolutions that all take the 64x64 tile; false = not applicable (caller falls back)
bool imm_conv64_group_launch(int dtype, ConvArgs* args, int n, int bm, int bn, hipStream_t s) {
  return dtype == IMM_BF16 ? launch64_group<BF16>(args, n, bm, bn, s) : launch64_group<F16>(args, n, bm, bn, s);
}

// called from conv_igemm.hip's dispatcher
void imm_conv64_launch(int dtype, ConvArgs& a, int bm, int bn, hipStream_t s) {
  if (dtype == IMM_BF16) launch64<BF16>(a, bm, bn, s);
  else launch64<F16>(a, bm, bn, s);
}
