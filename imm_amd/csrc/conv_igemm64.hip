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

typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ int lds64_idx(int row, int chunk) { return row * 8 + (chunk ^ ((row >> 1) & 7)); }

template <typename ET, int BM, int BN>
__global__ __launch_bounds__(256) void conv_igemm64_kernel(const ConvArgs a) {
  constexpr int WGM = 2, WGN = 2;
  constexpr int TM = BM / WGM, TN = BN / WGN, MT = TM / 16, NT = TN / 16;
  constexpr int A_INSTR = BM / 32, B_INSTR = BN / 32;   // 8-row wave instructions per wave per tile
  constexpr int BUF = (BM + BN) * 8;                    // uint4 per stage
  __shared__ uint4 smem[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  int bid = blockIdx.x;
  {
    const int q = a.n_blocks >> 3, r = a.n_blocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nblk = bid % a.n_nblk, mblk = bid / a.n_nblk;
  const int m0 = mblk * BM, n0 = nblk * BN;

  constexpr uint32_t OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);

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
    const int a_base = buf * BUF + (wid * A_INSTR) * 64;
    const int b_base = buf * BUF + BM * 8 + (wid * B_INSTR) * 64;
    const int a_soff = cs * 128, b_soff = kt * 128;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i) {
      const uint32_t vo = a_voff[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_t*)(smem + a_base + i * 64), 16, vo, a_soff, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
      const uint32_t vo = b_voff[j];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void_t*)(smem + b_base + j * 64), 16, vo, b_soff, 0, 0);
    }
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

  issue_tile(0, 0);
  __syncthreads();          // (an LDS-DMA in flight makes this wait vmcnt(0) first)

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < a.KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < a.KT) issue_tile(kt + 1, buf ^ 1);
    const uint4* Ab = smem + buf * BUF;
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
    __syncthreads();
  }
  conv_epilogue<ET, BM, BN, WGM, WGN, MT, NT>(a, acc, tid, wm, wn, m0, n0, mblk, (float*)smem);
}

template <typename ET>
static void launch64(ConvArgs& a, int bm, int bn, hipStream_t s) {
  const int mblk = (a.M + bm - 1) / bm;
  a.n_blocks = mblk * a.n_nblk;
  a.KT = a.kpad / 64;
  if (bm == 128 && bn == 128) hipLaunchKernelGGL((conv_igemm64_kernel<ET, 128, 128>), dim3(a.n_blocks), dim3(256), 0, s, a);
  else if (bm == 128 && bn == 64) hipLaunchKernelGGL((conv_igemm64_kernel<ET, 128, 64>), dim3(a.n_blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_igemm64_kernel<ET, 64, 64>), dim3(a.n_blocks), dim3(256), 0, s, a);
}

// called from conv_igemm.hip's dispatcher
void imm_conv64_launch(int dtype, ConvArgs& a, int bm, int bn, hipStream_t s) {
  if (dtype == IMM_BF16) launch64<BF16>(a, bm, bn, s);
  else launch64<F16>(a, bm, bn, s);
}
