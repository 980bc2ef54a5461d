// conv_igemm.hip — implicit-GEMM convolution on gfx950 matrix cores (forward + data gradient).
//
// Replaces tf.nn.conv2d(SAME)+bias_add (imm/tf_utils/nn_utils.py:100,108), the frozen VGG
// conv+bias+relu (imm/models/selfsup/vgg16.py:182-189,230) and tf.gradients of both w.r.t. their
// input.  GEMM view: M = batch*ho*wo output pixels, N = output channels, K = kh*kw*ci with the
// channel index fastest (NHWC), so one 16-byte vector = 8 consecutive channels of one filter tap.
//
// Tile: BM pixels x BN channels x BK=32, 256 threads = 4 waves.  Both operand tiles live in LDS as
// [row][32 k] (64-byte rows, 16-byte chunks XOR-swizzled so that ds_read_b128 of 16 rows x 4 chunks
// is conflict-free on the 64-bank LDS).  Global->register->LDS staging is double-buffered: the next
// K tile's vectors are in flight while the current tile feeds v_mfma_f32_16x16x32_{bf16,f16}.
// The MFMA is issued as D[n][m] (weights as the row operand) so every lane ends up holding 4
// CONSECUTIVE channels of one pixel -> 8-byte (16-bit out) / 16-byte (f32 out) NHWC stores.
// Epilogue: bias, ReLU, ReLU-backward mask, and deterministic per-M-block batch-norm partial sums.
#include "conv_common.h"
#include <stdlib.h>

bool imm_halo_applicable(const imm_conv_desc* d);                                  // conv_halo.hip
int imm_halo_grid(const imm_conv_desc* d);
void imm_conv_halo_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s);
bool imm_hdeep_applicable(const imm_conv_desc* d);                                 // conv_hdeep.hip
int imm_hdeep_stats_blocks(const imm_conv_desc* d);
int imm_hdeep_variant(const imm_conv_desc* d);
void imm_conv_hdeep_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s);
bool imm_halo_nol_applicable(const imm_conv_desc* d);
void imm_conv_halo_nol_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, const float* scale, const float* shift, int relu,
                              hipStream_t s);
bool imm_halo_s2d_applicable(const imm_conv_desc* d);
void imm_conv_halo_s2d_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s);
bool imm_s2f_applicable(const imm_conv_desc* d);                                   // conv_s2f.hip
int imm_s2f_stats_blocks(const imm_conv_desc* d);
int imm_s2f_variant(const imm_conv_desc* d);
void imm_conv_s2f_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s);
bool imm_hdeep_s2d_applicable(const imm_conv_desc* d);
void imm_conv_hdeep_s2d_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s);
bool imm_halo2_applicable(const imm_conv_desc* d);                                 // conv_halo2.hip
int imm_halo2_grid(const imm_conv_desc* d);
bool imm_halo2_x32(const imm_conv_desc* d);
void imm_conv_halo2_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s);

__device__ __forceinline__ int lds_chunk_idx(int row, int chunk) {
  // 64-byte rows; swizzle so the four 16-lane service groups of ds_read_b128 hit 16 distinct slots
  return row * 4 + (chunk ^ (((row >> 3) & 1) * 3));
}

// FAST: ci % 32 == 0, so a K tile is ONE filter tap x 32 channels: the tap walk is wave-uniform (SGPRs),
// per-lane gather offsets are recomputed once per tap and the channel slice rides in the buffer
// instruction's scalar offset -> a handful of VALU instructions per K step instead of ~150.
template <typename ET, int BM, int BN, int WGM, int WGN, bool FAST>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const int block) {
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MT = TM / 16, NT = TN / 16;
  constexpr int A_PASSES = BM / 64;
  constexpr int B_CHUNKS = BN * 4;
  constexpr int B_PASSES = (B_CHUNKS + 255) / 256;
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(BM % 64 == 0 && TM % 16 == 0 && TN % 16 == 0, "tile");

  __shared__ uint4 smem[2 * (BM + BN) * 4];
  constexpr int BUF = (BM + BN) * 4;   // uint4 per stage: A rows then B rows

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WGN, wn = wid % WGN;
  // XCD-aware remap (workgroup id -> XCD id%8 is the observed dispatch rule; speed only): give every
  // XCD a contiguous run of logical tiles so the N-blocks of one pixel tile and its halo neighbours
  // share that XCD's L2.  Bijective for any grid size.
  int bid = block;
  {
    const int q = a.n_blocks >> 3, r = a.n_blocks & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nblk = bid % a.n_nblk, mblk = bid / a.n_nblk;
  const int m0 = mblk * BM, n0 = nblk * BN;

  // ---- per-thread loader state ---------------------------------------------------------------
  const int chunk = tid & 3, r0 = tid >> 2;
  int by[A_PASSES], bx[A_PASSES];
  int64_t pbase[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + i * 64 + r0;
    if (m < a.M) {
      const int hw = a.ho * a.wo;
      const int img = m / hw, rem = m - img * hw;
      const int oy = rem / a.wo, ox = rem - oy * a.wo;
      by[i] = oy * a.stride - a.pad_t;
      bx[i] = ox * a.stride - a.pad_l;
      pbase[i] = (int64_t)img * a.hi * a.wi;
    } else {
      by[i] = -(1 << 28); bx[i] = -(1 << 28); pbase[i] = 0;
    }
  }
  uint4 ra[A_PASSES], rb[B_PASSES];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

  // ---- generic cursor (ci = 8 or 16): k8 = kt*4 + chunk -> (tap=(ky,kx), c8), per lane ----------------
  int c8 = chunk, tap = 0, ky = 0, kx = 0;
  // ---- fast path state --------------------------------------------------------------------------------
  constexpr uint32_t OOB = 0x80000000u;
  uint32_t a_voff[A_PASSES], b_voff[B_PASSES];
  int cs = 0;                               // channel slice of the current tap (uniform)
  const int ncs = a.ci8 >> 2;
  __amdgpu_buffer_rsrc_t xr, wr;
  auto tap_offsets = [&]() {
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
      int iy = by[i] + ky, ix = bx[i] + kx;
      bool ok = true;
      if (a.updiv == 2) { ok = (((iy | ix) & 1) == 0); iy >>= 1; ix >>= 1; }
      ok = ok && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
      a_voff[i] = ok ? (uint32_t)((((int)pbase[i] + iy * a.wi + ix) * a.ldx + chunk * 8) * 2) : OOB;
    }
  };
  if constexpr (FAST) {
    xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
    wr = __builtin_amdgcn_make_buffer_rsrc((void*)a.wt, 0, a.wt_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const int cidx = j * 256 + tid;
      const int n = n0 + (cidx >> 2);
      b_voff[j] = (cidx < B_CHUNKS && n < a.co) ? (uint32_t)((n * a.kpad + chunk * 8) * 2) : OOB;
    }
    if (a.flags & IMM_DBG_NO_GLOAD) {
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) b_voff[j] = OOB;
    }
    tap_offsets();
  } else {
    while (c8 >= a.ci8) { c8 -= a.ci8; ++tap; if (++kx == a.kw) { kx = 0; ++ky; } }
  }

  auto load_tile = [&](int kt) {
    if constexpr (FAST) {
      const int a_soff = cs * 64, b_soff = kt * 64;
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        const uint32_t vo = (a.flags & IMM_DBG_NO_GLOAD) ? OOB : a_voff[i];
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xr, vo, a_soff, 0);
        ra[i] = make_uint4(v.x, v.y, v.z, v.w);
      }
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) {
        const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(wr, b_voff[j], b_soff, 0);
        rb[j] = make_uint4(v.x, v.y, v.z, v.w);
      }
      if (++cs == ncs) {          // next filter tap (uniform branch)
        cs = 0;
        if (++kx == a.kw) { kx = 0; ++ky; }
        tap_offsets();
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PASSES; ++i) {
        int iy = by[i] + ky, ix = bx[i] + kx;
        bool ok = tap < a.ntaps;
        if (a.updiv == 2) { ok = ok && (((iy | ix) & 1) == 0); iy >>= 1; ix >>= 1; }
        ok = ok && ((unsigned)iy < (unsigned)a.hi) && ((unsigned)ix < (unsigned)a.wi);
        ra[i] = zero4;
        if (ok && !(a.flags & IMM_DBG_NO_GLOAD)) ra[i] = *(const uint4*)(a.x + ((pbase[i] + (int64_t)iy * a.wi + ix) * a.ldx + c8 * 8));
      }
#pragma unroll
      for (int j = 0; j < B_PASSES; ++j) {
        const int cidx = j * 256 + tid;
        rb[j] = zero4;
        if (cidx < B_CHUNKS) {
          const int n = n0 + (cidx >> 2);
          if (n < a.co && !(a.flags & IMM_DBG_NO_GLOAD)) rb[j] = *(const uint4*)(a.wt + ((int64_t)n * a.kpad + kt * 32 + chunk * 8));
        }
      }
      // advance the cursor by one K tile (4 chunks)
      c8 += 4;
      while (c8 >= a.ci8) { c8 -= a.ci8; ++tap; if (++kx == a.kw) { kx = 0; ++ky; } }
    }
  };
  auto store_tile = [&](int buf) {
    if (a.flags & IMM_DBG_NO_LDS_STORE) return;
    uint4* Ab = smem + buf * BUF;
    uint4* Bb = Ab + BM * 4;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) Ab[lds_chunk_idx(i * 64 + r0, chunk)] = ra[i];
#pragma unroll
    for (int j = 0; j < B_PASSES; ++j) {
      const int cidx = j * 256 + tid;
      if (cidx < B_CHUNKS) Bb[lds_chunk_idx(cidx >> 2, chunk)] = rb[j];
    }
  };

  f32x4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int frow = lane & 15, fchunk = lane >> 4;
  for (int kt = 0; kt < a.KT; ++kt) {
    const int buf = kt & 1;
    const bool more = (kt + 1) < a.KT;
    if (more) load_tile(kt + 1);
    uint4 af[MT], bf[NT];
    if (a.flags & IMM_DBG_NO_MFMA) { if (more) store_tile(buf ^ 1); __syncthreads(); continue; }
    const uint4* Ab = smem + buf * BUF;
    const uint4* Bb = Ab + BM * 4;
#pragma unroll
    for (int i = 0; i < MT; ++i) af[i] = Ab[lds_chunk_idx(wm * TM + i * 16 + frow, fchunk)];
#pragma unroll
    for (int j = 0; j < NT; ++j) bf[j] = Bb[lds_chunk_idx(wn * TN + j * 16 + frow, fchunk)];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[j], af[i], acc[i][j]);
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

  conv_epilogue<ET, BM, BN, WGM, WGN, MT, NT>(a, acc, tid, wm, wn, m0, n0, mblk, (float*)smem);
}

template <typename ET, int BM, int BN, int WGM, int WGN, bool FAST>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  conv_igemm_body<ET, BM, BN, WGM, WGN, FAST>(a, blockIdx.x);
}

// The four parity classes of a stride-2 data gradient with <= 32 output channels (encoder conv_3: 64 -> 32 at 128x128) in
// ONE launch of the 128x32 tile: four launches of 8-14 us each otherwise.
template <typename ET, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_igemm_group_kernel(const ConvArgsGroup g) {
  int m = 0;
#pragma unroll
  for (int i = 1; i < IMM_CONV_GROUP_MAX; ++i)
    if (i < g.n && (int)blockIdx.x >= g.first[i]) m = i;
  conv_igemm_body<ET, BM, BN, WGM, WGN, true>(g.a[m], (int)blockIdx.x - g.first[m]);
}

// ---------------------------------------------------------------------------------------------
// host side: tile selection + launch
// ---------------------------------------------------------------------------------------------
struct TileCfg { int bm, bn; };

static int g_num_cu = 0;
static int num_cu() {
  if (g_num_cu == 0) {
    hipDeviceProp_t p;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      g_num_cu = p.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

static TileCfg pick_tile(int64_t M, int co) {
  // candidates in order of preference (largest tile first); take the first that fills the chip
  TileCfg cands[3];
  int nc = 0;
  if (co > 64) { cands[nc++] = {128, 128}; cands[nc++] = {128, 64}; cands[nc++] = {64, 64}; }
  else if (co > 32) { cands[nc++] = {128, 64}; cands[nc++] = {64, 64}; }
  else if (co > 16) { cands[nc++] = {128, 32}; }
  else { cands[nc++] = {128, 16}; }
  const int64_t want = 2LL * num_cu();
  for (int i = 0; i < nc; ++i) {
    const int64_t blocks = ((M + cands[i].bm - 1) / cands[i].bm) * ((co + cands[i].bn - 1) / cands[i].bn);
    if (blocks >= want) return cands[i];
  }
  return cands[nc - 1];
}

static int validate_desc(const imm_conv_desc* d) {
  IMM_REQUIRE(d != nullptr, "conv: null desc");
  IMM_REQUIRE(d->batch > 0 && d->hi > 0 && d->wi > 0 && d->ho > 0 && d->wo > 0 && d->co > 0, "conv: dims");
  IMM_REQUIRE(d->ci > 0 && d->ci % 8 == 0, "conv: ci=%d must be a positive multiple of 8", d->ci);
  IMM_REQUIRE(d->ldx >= d->ci && d->ldx % 8 == 0, "conv: ldx=%d (ci=%d) must be a multiple of 8", d->ldx, d->ci);
  IMM_REQUIRE(d->ldy >= d->co && d->ldy % 4 == 0, "conv: ldy=%d (co=%d) must be a multiple of 4", d->ldy, d->co);
  IMM_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0, "conv: kernel/stride");
  IMM_REQUIRE(d->updiv == 1 || d->updiv == 2, "conv: updiv must be 1 or 2");
  IMM_REQUIRE(d->kpad % 32 == 0 && d->kpad >= d->kh * d->kw * d->ci, "conv: kpad=%d too small / unaligned", d->kpad);
  IMM_REQUIRE((int64_t)d->batch * d->ho * d->wo < (1LL << 31), "conv: M overflow");
  IMM_REQUIRE(d->out_scale >= 0 && d->out_scale <= 2 && d->out_off_y >= 0 && d->out_off_x >= 0 &&
                  d->out_off_y < (d->out_scale > 1 ? d->out_scale : 1) && d->out_off_x < (d->out_scale > 1 ? d->out_scale : 1),
              "conv: output scatter");
  IMM_REQUIRE(!(d->flags & IMM_CONV_OUT_F32) || !(d->flags & IMM_CONV_MASK), "conv: f32 output excludes the mask");
  return 0;
}

extern "C" int imm_conv_stats_blocks(const imm_conv_desc* d) {
  if (validate_desc(d)) return IMM_E_INVALID;
  if (imm_halo2_applicable(d)) return imm_halo2_grid(d);
  if (imm_halo_applicable(d)) return imm_halo_grid(d);
  if (imm_hdeep_applicable(d)) return imm_hdeep_stats_blocks(d);
  if (imm_s2f_applicable(d)) return imm_s2f_stats_blocks(d);
  const int64_t M = (int64_t)d->batch * d->ho * d->wo;
  const TileCfg t = pick_tile(M, d->co);
  return (int)((M + t.bm - 1) / t.bm);
}

template <typename ET, int BM, int BN, int WGM, int WGN>
static void launch_cfg(ConvArgs& a, bool fast, hipStream_t s) {
  const int mblk = (a.M + BM - 1) / BM;
  a.n_blocks = mblk * a.n_nblk;
  if (fast) hipLaunchKernelGGL((conv_igemm_kernel<ET, BM, BN, WGM, WGN, true>), dim3(a.n_blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((conv_igemm_kernel<ET, BM, BN, WGM, WGN, false>), dim3(a.n_blocks), dim3(256), 0, s, a);
}

void imm_conv64_launch(int dtype, ConvArgs& a, int bm, int bn, hipStream_t s);   // conv_igemm64.hip
bool imm_conv64_group_launch(int dtype, ConvArgs* args, int n, int bm, int bn, hipStream_t s);

static void fill_args(ConvArgs& a, const imm_conv_desc* d, const void* x, const void* wt, const float* bias, void* y,
                      float* stats, const void* mask) {
  a.x = (const uint16_t*)x; a.wt = (const uint16_t*)wt; a.bias = bias; a.y = y; a.stats = stats;
  a.mask = (const uint16_t*)mask;
  a.M = d->batch * d->ho * d->wo;
  a.hi = d->hi; a.wi = d->wi; a.ci8 = d->ci / 8; a.ldx = d->ldx;
  a.ho = d->ho; a.wo = d->wo; a.co = d->co; a.ldy = d->ldy;
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad_t = d->pad_t; a.pad_l = d->pad_l; a.updiv = d->updiv;
  a.kpad = d->kpad; a.KT = d->kpad / 32; a.ntaps = d->kh * d->kw;
  a.flags = d->flags; a.ldmask = d->ldmask;
  a.oscale = d->out_scale > 1 ? d->out_scale : 1; a.ooff_y = d->out_off_y; a.ooff_x = d->out_off_x;
  a.tap_gt = nullptr; a.tap_lmask = nullptr; a.tap_coef = nullptr; a.tap_idx = 0; a.tap_S = 0; a.tap_l1 = 0;
}

template <typename ET>
static int conv_launch(const imm_conv_desc* d, const void* x, const void* wt, const float* bias, void* y,
                       float* stats, const void* mask, hipStream_t s) {
  ConvArgs a;
  fill_args(a, d, x, wt, bias, y, stats, mask);
  if (imm_halo2_applicable(d)) {
    imm_conv_halo2_launch(ET::kEnum, d, a, s);
    IMM_CHECK_LAUNCH("imm_conv2d(halo2)");
    return 0;
  }
  if (imm_halo_applicable(d)) {
    imm_conv_halo_launch(ET::kEnum, d, a, s);
    IMM_CHECK_LAUNCH("imm_conv2d(halo)");
    return 0;
  }
  if (imm_hdeep_applicable(d)) {
    imm_conv_hdeep_launch(ET::kEnum, d, a, s);
    IMM_CHECK_LAUNCH("imm_conv2d(hdeep)");
    return 0;
  }
  if (imm_s2f_applicable(d)) {
    imm_conv_s2f_launch(ET::kEnum, d, a, s);
    IMM_CHECK_LAUNCH("imm_conv2d(s2f)");
    return 0;
  }
  const TileCfg t = pick_tile(a.M, a.co);
  a.n_nblk = (a.co + t.bn - 1) / t.bn;
  const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, wb = (int64_t)d->co * d->kpad * 2;
  const bool fast = (d->ci % 32 == 0) && xb < (1LL << 31) && wb < (1LL << 31);
  a.x_bytes = (uint32_t)(fast ? xb : 0);
  a.wt_bytes = (uint32_t)(fast ? wb : 0);
  const bool deep = fast && (d->ci % 64 == 0) && t.bn >= 64 && !(d->flags & 0xf00) && !imm_conv_disabled("deepk");
  if (deep) imm_conv64_launch(ET::kEnum, a, t.bm, t.bn, s);
  else if (t.bm == 128 && t.bn == 128) launch_cfg<ET, 128, 128, 2, 2>(a, fast, s);
  else if (t.bm == 128 && t.bn == 64) launch_cfg<ET, 128, 64, 2, 2>(a, fast, s);
  else if (t.bm == 64 && t.bn == 64) launch_cfg<ET, 64, 64, 2, 2>(a, fast, s);
  else if (t.bm == 128 && t.bn == 32) launch_cfg<ET, 128, 32, 4, 1>(a, fast, s);
  else launch_cfg<ET, 128, 16, 4, 1>(a, fast, s);
  IMM_CHECK_LAUNCH("imm_conv2d");
  return 0;
}

// Which kernel imm_conv2d runs for a descriptor (the twin of imm_conv2d_wgrad_variant): family * 100000 + tile variant, following
// conv_launch's order exactly.  Families: 1 conv_igemm (BK = 32, register-staged), 2 conv_igemm64 (BK = 64, LDS-DMA ring),
// 3 conv_halo (filter in LDS), 4 conv_halo2 (filter in registers), 5 conv_hdeep (LDS halo, tap / row at a time), 6 conv_hdeep6,
// 7 conv_s2f (stride-2 forward through a parity-de-interleaved LDS halo).
extern "C" int imm_conv2d_variant(const imm_conv_desc* d, int dtype) {
  if (validate_desc(d)) return IMM_E_INVALID;
  if (dtype == IMM_F32) return 800000;                 // family 8: the plain f32 kernels of the witness engine (conv_f32.hip)
  IMM_REQUIRE(dtype == IMM_BF16 || dtype == IMM_F16, "unknown dtype %d", dtype);
  if (imm_halo2_applicable(d)) return 400000 + (d->kw == 1 ? 10000 : 0) + (imm_halo2_x32(d) ? 20000 : 0) + d->ci * 100 + (d->co > 32 ? 64 : 32);   // + 20000: the 32x32x16 form
  if (imm_halo_applicable(d)) return 300000 + (d->stride == 2 ? 10000 : 0) + d->ci * 100 + (d->co > 32 ? 64 : d->co > 16 ? 32 : 16);
  if (imm_hdeep_applicable(d)) return imm_hdeep_variant(d);
  if (imm_s2f_applicable(d)) return imm_s2f_variant(d);
  const TileCfg t = pick_tile((int64_t)d->batch * d->ho * d->wo, d->co);
  const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, wb = (int64_t)d->co * d->kpad * 2;
  const bool fast = (d->ci % 32 == 0) && xb < (1LL << 31) && wb < (1LL << 31);
  const bool deep = fast && (d->ci % 64 == 0) && t.bn >= 64 && !(d->flags & 0xf00) && !imm_conv_disabled("deepk");
  return (deep ? 200000 : 100000) + (fast ? 10000 : 0) + (t.bm / 16) * 100 + t.bn / 16;
}

int imm_conv_f32(const imm_conv_desc* d, const void* x, const void* wt, const float* bias, void* y, float* stats_partial,
                 const void* mask_ref, hipStream_t s);                                  // conv_f32.hip

extern "C" int imm_conv2d(const imm_conv_desc* d, int dtype, const void* x, const void* wt, const float* bias,
                          void* y, float* stats_partial, const void* mask_ref, void* stream) {
  if (validate_desc(d)) return IMM_E_INVALID;
  IMM_REQUIRE(x && wt && y, "conv: null tensor");
  IMM_REQUIRE(!(d->flags & IMM_CONV_BIAS) || bias, "conv: bias flag without bias");
  IMM_REQUIRE(!(d->flags & IMM_CONV_STATS) || stats_partial, "conv: stats flag without buffer");
  IMM_REQUIRE(!(d->flags & IMM_CONV_MASK) || (mask_ref && d->ldmask >= d->co), "conv: mask flag without mask/ldmask");
  IMM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)wt % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv: 16-byte alignment");
  if (dtype == IMM_F32) return imm_conv_f32(d, x, wt, bias, y, stats_partial, mask_ref, (hipStream_t)stream);   // the f32 witness (conv_f32.hip)
  IMM_DISPATCH_DTYPE(dtype, return conv_launch<ET>(d, x, wt, bias, y, stats_partial, mask_ref, (hipStream_t)stream));
  return 0;
}

// Data gradient entering a TAPPED activation of the frozen VGG16 (imm_model.py:142-147: conv3_2, conv4_2 sit in the middle of the
// network): the feature-loss term and the tapped layer's ReLU backward in the epilogue of the data gradient that produces the
// incoming gradient, instead of a separate pass over it (imm_tap_grad: read da, a_pred, a_gt, write da).  LDS-halo deep-K kernel only.
extern "C" int imm_conv2d_tap_supported(const imm_conv_desc* d) {
  if (!d || validate_desc(d)) return 0;
  if (d->flags & (IMM_CONV_BIAS | IMM_CONV_RELU | IMM_CONV_STATS | IMM_CONV_MASK | IMM_CONV_OUT_F32 | 0xfe0)) return 0;
  if (imm_halo2_applicable(d) || imm_halo_applicable(d)) return 0;
  return imm_hdeep_applicable(d) && d->co % 8 == 0 ? 1 : 0;
}

extern "C" int imm_conv2d_tap(const imm_conv_desc* d, int dtype, const void* x, const void* wt, void* y, const void* a_pred,
                              const void* a_gt, int lda, const float* loss_mask, int S, const float* coef, int idx, int l1,
                              void* stream) {
  IMM_REQUIRE(d && x && wt && y && a_pred && a_gt && coef && idx >= 0, "conv_tap: null");
  if (!imm_conv2d_tap_supported(d)) return imm_fail(IMM_E_UNSUPPORTED, "conv_tap: shape not served by the LDS-halo deep-K kernel");
  IMM_REQUIRE(lda >= d->co && lda % 8 == 0, "conv_tap: lda=%d (co=%d) must be a multiple of 8", lda, d->co);
  IMM_REQUIRE(loss_mask == nullptr || (S >= d->ho && S % d->ho == 0 && d->ho == d->wo), "conv_tap: mask side");
  IMM_REQUIRE(((uintptr_t)x % 16 == 0) && ((uintptr_t)wt % 16 == 0) && ((uintptr_t)y % 16 == 0) && ((uintptr_t)a_pred % 16 == 0) &&
                  ((uintptr_t)a_gt % 16 == 0), "conv_tap: 16-byte alignment");
  IMM_REQUIRE(dtype == IMM_BF16 || dtype == IMM_F16, "unknown dtype %d", dtype);
  imm_conv_desc dd = *d;
  dd.ldmask = lda;
  ConvArgs a;
  fill_args(a, &dd, x, wt, nullptr, y, nullptr, a_pred);
  a.flags |= IMM_CONV_TAP_;
  a.tap_gt = (const uint16_t*)a_gt; a.tap_lmask = loss_mask; a.tap_coef = coef; a.tap_idx = idx; a.tap_S = S; a.tap_l1 = l1;
  imm_conv_hdeep_launch(dtype, &dd, a, (hipStream_t)stream);
  IMM_CHECK_LAUNCH("imm_conv2d_tap");
  return 0;
}

// Normalise on load (conv_halo.hip): the batch-norm apply pass of the block in front of this convolution folded into its halo tile.
extern "C" int imm_conv2d_nol_supported(const imm_conv_desc* d) {
  if (!d || validate_desc(d)) return 0;
  if (d->flags & (IMM_CONV_MASK | IMM_CONV_RELU | 0xfe0)) return 0;
  return imm_halo_nol_applicable(d) ? 1 : 0;
}

extern "C" int imm_conv2d_nol_stats_blocks(const imm_conv_desc* d) {
  if (!imm_conv2d_nol_supported(d)) return imm_fail(IMM_E_UNSUPPORTED, "conv_nol: shape not served by the LDS-halo kernel");
  return imm_halo_grid(d);
}

extern "C" int imm_conv2d_nol(const imm_conv_desc* d, int dtype, const void* x_raw, const float* x_scale, const float* x_shift,
                              int x_relu, const void* wt, const float* bias, void* y, float* stats_partial, void* stream) {
  if (validate_desc(d)) return IMM_E_INVALID;
  IMM_REQUIRE(x_raw && x_scale && x_shift && wt && y, "conv_nol: null tensor");
  IMM_REQUIRE(dtype == IMM_BF16 || dtype == IMM_F16, "unknown dtype %d", dtype);
  IMM_REQUIRE(!(d->flags & IMM_CONV_BIAS) || bias, "conv_nol: bias flag without bias");
  IMM_REQUIRE(!(d->flags & IMM_CONV_STATS) || stats_partial, "conv_nol: stats flag without buffer");
  IMM_REQUIRE(((uintptr_t)x_raw % 16 == 0) && ((uintptr_t)wt % 16 == 0) && ((uintptr_t)y % 16 == 0), "conv_nol: 16-byte alignment");
  if (!imm_conv2d_nol_supported(d)) return imm_fail(IMM_E_UNSUPPORTED, "conv_nol: shape not served by the LDS-halo kernel");
  ConvArgs a;
  fill_args(a, d, x_raw, wt, bias, y, stats_partial, nullptr);
  imm_conv_halo_nol_launch(dtype, d, a, x_scale, x_shift, x_relu, (hipStream_t)stream);
  IMM_CHECK_LAUNCH("imm_conv2d_nol");
  return 0;
}

// Data gradient of a 3x3 stride-2 SAME convolution in ONE launch over ONE dy halo (conv_hdeep.hip, S2D): the class grid is described
// to the kernel as a same-size 3x3 convolution of dy whose four accumulator sets are scattered to the four parities of dx.
static void s2d_desc(imm_conv_desc* d, int batch, int h, int w, int lddy, int c_dx, int lddx) {
  *d = imm_conv_desc{};
  d->batch = batch; d->hi = d->ho = h; d->wi = d->wo = w; d->ci = lddy; d->ldx = lddy; d->co = c_dx; d->ldy = lddx;
  d->kh = d->kw = 3; d->stride = 1; d->pad_t = d->pad_l = 1; d->updiv = 1; d->kpad = 9 * lddy; d->flags = 0; d->ldmask = 0;
  d->out_scale = 0; d->out_off_y = d->out_off_x = 0;
}

extern "C" int imm_conv2d_dgrad_s2_supported(int batch, int h, int w, int lddy, int c_dx, int lddx) {
  if (batch <= 0 || h <= 0 || w <= 0 || lddy <= 0 || c_dx <= 0 || lddx <= 0 || lddy % 32) return 0;
  imm_conv_desc d;
  s2d_desc(&d, batch, h, w, lddy, c_dx, lddx);
  return (imm_halo_s2d_applicable(&d) || imm_hdeep_s2d_applicable(&d)) ? 1 : 0;
}

extern "C" int imm_conv2d_dgrad_s2(const void* dy, int lddy, const void* wt, int kpad, void* dx, int lddx, int c_dx, int dtype,
                                   int batch, int h, int w, void* stream) {
  IMM_REQUIRE(dy && wt && dx, "conv_dgrad_s2: null tensor");
  IMM_REQUIRE(dtype == IMM_BF16 || dtype == IMM_F16, "unknown dtype %d", dtype);
  IMM_REQUIRE(((uintptr_t)dy % 16 == 0) && ((uintptr_t)wt % 16 == 0) && ((uintptr_t)dx % 16 == 0), "conv_dgrad_s2: 16-byte alignment");
  if (!imm_conv2d_dgrad_s2_supported(batch, h, w, lddy, c_dx, lddx))
    return imm_fail(IMM_E_UNSUPPORTED, "conv_dgrad_s2: batch %d, dy %dx%d x %d -> dx x %d (stride %d): needs h %% 8 == 0, w %% 16 == 0, "
                    "lddy %% 64 == 0, c_dx %% 8 == 0, lddx %% 8 == 0", batch, h, w, lddy, c_dx, lddx);
  IMM_REQUIRE(kpad == 9 * lddy, "conv_dgrad_s2: kpad=%d must be 9 * lddy (imm_pack_weights mode 1 with c_pad = lddy)", kpad);
  imm_conv_desc d;
  s2d_desc(&d, batch, h, w, lddy, c_dx, lddx);
  ConvArgs a;
  fill_args(a, &d, dy, wt, nullptr, dx, nullptr, nullptr);
  // the whole flipped filter in LDS where it fits (dy 64 channels -> dx <= 32: encoder conv_3, HBM-bound), the deep-K kernel otherwise
  if (imm_halo_s2d_applicable(&d)) imm_conv_halo_s2d_launch(dtype, &d, a, (hipStream_t)stream);
  else imm_conv_hdeep_s2d_launch(dtype, &d, a, (hipStream_t)stream);
  IMM_CHECK_LAUNCH("imm_conv2d_dgrad_s2");
  return 0;
}

// Up to 4 convolutions reading the same x and writing (disjoint parts of) the same y in ONE launch — the four parity
// classes of a stride-2 data gradient (ops.dgrad_s2_class_descs).  Members carry no bias; all must take the deep-K
// 64x64-tile kernel to share a launch, otherwise they are launched one after the other (same results either way).
// With IMM_CONV_STATS | IMM_CONV_MASK on the members (batch-norm backward sums of the layer the gradient enters) the
// partial-sum rows of member i follow those of members 0..i-1; imm_conv2d_group_stats_blocks = total row count.
struct GroupPlan { bool grouped, grouped32; int bm, bn; int rows[4]; int total_rows; };

static int group_plan(const imm_conv_desc* descs, int n, GroupPlan* gp) {
  static const bool off = imm_conv_disabled("group");
  gp->grouped = !off; gp->grouped32 = !off && !imm_conv_disabled("group32"); gp->bm = gp->bn = 0; gp->total_rows = 0;
  for (int i = 0; i < n; ++i) {
    const imm_conv_desc* d = descs + i;
    if (validate_desc(d)) return IMM_E_INVALID;
    IMM_REQUIRE(!(d->flags & IMM_CONV_BIAS), "conv_group: members carry no bias");
    if (imm_halo2_applicable(d) || imm_halo_applicable(d) || imm_hdeep_applicable(d)) gp->grouped = gp->grouped32 = false;
    const int64_t M = (int64_t)d->batch * d->ho * d->wo;
    const TileCfg t = pick_tile(M, d->co);
    const int64_t xb = (int64_t)d->batch * d->hi * d->wi * d->ldx * 2, wb = (int64_t)d->co * d->kpad * 2;
    const bool deep = (d->ci % 64 == 0) && xb < (1LL << 31) && wb < (1LL << 31) && t.bn >= 64 && !(d->flags & 0xf00) &&
                      !imm_conv_disabled("deepk");
    if (!deep || (i > 0 && (t.bm != gp->bm || t.bn != gp->bn))) gp->grouped = false;
    // the general kernel's 128x32 tile, one tap x 32 channels per K tile (ci % 32 == 0)
    const bool fast32 = (d->ci % 32 == 0) && xb < (1LL << 31) && wb < (1LL << 31) && t.bm == 128 && t.bn == 32 && !(d->flags & 0xf00);
    if (!fast32) gp->grouped32 = false;
    gp->bm = t.bm; gp->bn = t.bn;
  }
  if (gp->grouped) gp->grouped32 = false;
  if (gp->grouped && !(gp->bm == 64 && gp->bn == 64)) gp->grouped = false;   // the grouped kernel exists for the 64x64 tile only
  for (int i = 0; i < n; ++i) {
    const int64_t M = (int64_t)descs[i].batch * descs[i].ho * descs[i].wo;
    gp->rows[i] = gp->grouped ? (int)((M + 63) / 64) : gp->grouped32 ? (int)((M + 127) / 128) : imm_conv_stats_blocks(descs + i);
    gp->total_rows += gp->rows[i];
  }
  return 0;
}

extern "C" int imm_conv2d_group_stats_blocks(const imm_conv_desc* descs, int n) {
  if (!descs || n < 1 || n > 4) return IMM_E_INVALID;
  GroupPlan gp;
  if (group_plan(descs, n, &gp)) return IMM_E_INVALID;
  return gp.total_rows;
}

extern "C" int imm_conv2d_group(const imm_conv_desc* descs, int n, int dtype, const void* x, const void* const* wts, void* y,
                                float* stats_partial, const void* mask_ref, void* stream) {
  IMM_REQUIRE(descs && wts && x && y && n >= 1 && n <= 4, "conv_group: 1..4 members");
  IMM_REQUIRE(dtype == IMM_BF16 || dtype == IMM_F16 || dtype == IMM_F32, "unknown dtype %d", dtype);
  GroupPlan gp;
  if (group_plan(descs, n, &gp)) return IMM_E_INVALID;
  if (dtype == IMM_F32) gp.grouped = gp.grouped32 = false;       // f32 witness: the members one after the other (imm_conv2d)
  ConvArgs args[4];
  int row0 = 0;
  for (int i = 0; i < n; ++i) {
    const imm_conv_desc* d = descs + i;
    IMM_REQUIRE(wts[i] && ((uintptr_t)wts[i] % 16 == 0), "conv_group: filter pointer %d", i);
    IMM_REQUIRE(!(d->flags & IMM_CONV_STATS) || stats_partial, "conv_group: stats flag without buffer");
    IMM_REQUIRE(!(d->flags & IMM_CONV_MASK) || (mask_ref && d->ldmask >= d->co), "conv_group: mask flag without mask/ldmask");
    float* st = (d->flags & IMM_CONV_STATS) ? stats_partial + (int64_t)row0 * 2 * d->co : nullptr;
    fill_args(args[i], d, x, wts[i], nullptr, y, st, mask_ref);
    args[i].n_nblk = (d->co + gp.bn - 1) / gp.bn;
    args[i].x_bytes = (uint32_t)((int64_t)d->batch * d->hi * d->wi * d->ldx * 2);
    args[i].wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
    row0 += gp.rows[i];
  }
  if (gp.grouped && imm_conv64_group_launch(dtype, args, n, gp.bm, gp.bn, (hipStream_t)stream)) {
    IMM_CHECK_LAUNCH("imm_conv2d_group");
    return 0;
  }
  if (gp.grouped32) {
    ConvArgsGroup g;
    g.n = n;
    int total = 0;
    for (int i = 0; i < n; ++i) {
      args[i].n_blocks = ((args[i].M + 127) / 128) * args[i].n_nblk;
      g.a[i] = args[i];
      g.first[i] = total;
      total += args[i].n_blocks;
    }
    for (int i = n; i <= IMM_CONV_GROUP_MAX; ++i) g.first[i] = total;
    if (dtype == IMM_BF16) hipLaunchKernelGGL((conv_igemm_group_kernel<BF16, 128, 32, 4, 1>), dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((conv_igemm_group_kernel<F16, 128, 32, 4, 1>), dim3(total), dim3(256), 0, (hipStream_t)stream, g);
    IMM_CHECK_LAUNCH("imm_conv2d_group(128x32)");
    return 0;
  }
  row0 = 0;
  for (int i = 0; i < n; ++i) {
    float* st = (descs[i].flags & IMM_CONV_STATS) ? stats_partial + (int64_t)row0 * 2 * descs[i].co : nullptr;
    const int rc = imm_conv2d(descs + i, dtype, x, wts[i], nullptr, y, st, mask_ref, stream);
    if (rc) return rc;
    row0 += gp.rows[i];
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// weight packing: f32 HWIO master -> 16-bit Wt[rows][kpad]
// ---------------------------------------------------------------------------------------------
template <typename ET>
__global__ void pack_weights_kernel(const float* __restrict__ w, typename ET::T* __restrict__ wt, int mode, int kh, int kw,
                                    int ci_real, int co_real, int c_pad, int rows, int kpad) {
  const int64_t total = (int64_t)rows * kpad;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(idx / kpad), k = (int)(idx - (int64_t)n * kpad);
    const int tap = k / c_pad, c = k - tap * c_pad;
    float v = 0.f;
    if (tap < kh * kw) {
      if (mode == 0) {
        if (n < co_real && c < ci_real) v = w[((int64_t)tap * ci_real + c) * co_real + n];
      } else if (mode == 1) {
        const int kyf = kh - 1 - tap / kw, kxf = kw - 1 - tap % kw;
        if (n < ci_real && c < co_real) v = w[(((int64_t)kyf * kw + kxf) * ci_real + n) * co_real + c];
      } else {      // parity class (py,px) of the stride-2 data gradient
        const int py = ((mode - 4) >> 1) & 1, px = (mode - 4) & 1;
        const int ny = (kh - py + 1) / 2, nx = (kw - px + 1) / 2;
        if (tap < ny * nx) {
          const int jy = tap / nx, jx = tap - jy * nx;
          const int kyf = py + 2 * (ny - 1 - jy), kxf = px + 2 * (nx - 1 - jx);
          if (n < ci_real && c < co_real) v = w[(((int64_t)kyf * kw + kxf) * ci_real + n) * co_real + c];
        }
      }
    }
    wt[idx] = ET::from_f32(v);
  }
}

extern "C" int imm_pack_weights(const float* w, void* wt, int dtype, int mode, int kh, int kw, int ci_real,
                                int co_real, int c_pad, int rows, int kpad, void* stream) {
  IMM_REQUIRE(w && wt, "pack_weights: null");
  IMM_REQUIRE(mode == 0 || mode == 1 || (mode >= 4 && mode <= 7), "pack_weights: mode");
  IMM_REQUIRE(c_pad % 8 == 0 && kpad % 32 == 0, "pack_weights: padding");
  if (mode < 4) IMM_REQUIRE(kpad >= kh * kw * c_pad, "pack_weights: kpad too small");
  else {
    const int py = ((mode - 4) >> 1) & 1, px = (mode - 4) & 1;
    IMM_REQUIRE(kpad >= ((kh - py + 1) / 2) * ((kw - px + 1) / 2) * c_pad, "pack_weights: kpad too small for the parity class");
  }
  IMM_REQUIRE(c_pad >= (mode == 0 ? ci_real : co_real), "pack_weights: c_pad too small");
  IMM_REQUIRE(rows >= (mode == 0 ? co_real : ci_real), "pack_weights: rows too small");
  const int64_t total = (int64_t)rows * kpad;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  IMM_DISPATCH_DTYPE_F32(dtype, hipLaunchKernelGGL((pack_weights_kernel<ET>), dim3(blocks), dim3(256), 0,
                                                   (hipStream_t)stream, w, (typename ET::T*)wt, mode, kh, kw, ci_real,
                                                   co_real, c_pad, rows, kpad));
  IMM_CHECK_LAUNCH("imm_pack_weights");
  return 0;
}
