// conv_first.hip — the first encoder convolution (7x7, 3 -> 32 channels, stride 1, SAME; imm/models/imm_model.py:190 through
// nn_utils.py:100,108) straight from the f32 image.
//
// Since round 1 this layer runs as a 7x1 convolution over the image with its seven horizontal taps unrolled into 21 (+11 zero)
// channels (imm_pack_image_taps), which made it an ordinary 32-channel layer for the LDS-halo kernels — at the price of a 16-bit
// [B,S,S,32] copy of the image written by a pass of its own and read back by the convolution: per encoder 12 us of packing in front
// of the 128x128 layer and 67 MB of HBM traffic for a 6 MB image, at the head of both forward lanes.  Here the tap-unrolled tile is
// built in LDS instead: a persistent workgroup loads the (8+6) x (16+6) f32 halo of an 8x16-pixel patch (3.7 KB, coalesced rows,
// requested one patch ahead), rounds it to 16 bits, gathers the [14 x 16 pixels][32 channels] operand tile from it and runs the
// seven vertical taps as 7 k-steps of v_mfma_f32_16x16x32 against the filter image resident in LDS (the same packed image
// Wt[n][ky*32 + kx*3 + c] the 7x1 form uses, so the arithmetic is unchanged: same 16-bit operands, f32 accumulation).  Epilogue =
// conv_halo.hip's: bias, 16-bit NHWC store, batch-norm partial sums accumulated over the workgroup's patches (one row per
// workgroup).  The packed copy is still produced for the layer's filter gradient, but off the forward chain.
#include "conv_common.h"

#define CF_PH 8
#define CF_PW 16
#define CF_K 7
#define CF_PAD 3
#define CF_TR (CF_PH + CF_K - 1)          // tile rows 14
#define CF_SW (CF_PW + CF_K - 1)          // staging columns 22
#define CF_BN 32

struct ConvFirstArgs {
  const float* img; const uint16_t* wt; const float* bias; uint16_t* y; float* stats;
  int batch, s, co, ldy, kpad, flags;
  int n_patches, patches_x, patches_y;
};

// chunk swizzle of 64-byte rows (4 chunks): a ds_read_b128 of 16 consecutive rows x 4 chunks is conflict-free (conv_halo.hip)
__device__ __forceinline__ int cf_swz(int row) { return ((row >> 2) & 1) << 1; }

// FULL: co == 32 (every shipped configuration) — compile-time: every lane then issues exactly MT*NT output stores per patch, which is
// what makes the counted s_waitcnt of the loop exact.
//
// Memory pipeline (what three earlier versions of this loop taught, 33 -> 27 us for the layer against the 33 us of the two launches
// it replaces): a patch is 3.7 KB in, 8 KB out and 28 MFMAs per wave, so the loop is a latency chain unless the halo of patch k+2
// is in flight while patch k is computed and nobody waits for output stores.  With ordinary loads hipcc decides the waits, and it
// drains vmcnt to zero wherever loads and stores are both outstanding (its counter model treats mixed event types as unordered) or
// a branch sits in between.  So the halo loads are issued from inline asm (buffer_load_dword into registers the compiler does not
// track), EXACTLY four per lane and patch (beyond the last patch: out-of-range offsets, which return zeros), into two register sets
// that alternate, and the loop waits with a counted vmcnt: after the four loads of patch k come the stores of patch k-2 (4), the
// loads of patch k+1 (4) and the stores of patch k-1 (4) -> vmcnt(12) (8 / 4 for the first two patches).  (A version with the
// halo by 4-byte LDS-DMA pieces was slower, 37 us: that path wants 16-byte pieces.)
typedef __attribute__((ext_vector_type(4))) unsigned int cf_u32x4_t;
// asynchronous: `dst` is valid only after an s_waitcnt that covers this load AND a cf_arrived() on it (which stops the compiler from
// reading the register any earlier)
__device__ __forceinline__ void cf_load(float& dst, cf_u32x4_t rsrc, uint32_t voff) {
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void cf_arrived(float& v) { asm volatile("" : "+v"(v) :: "memory"); }

template <typename ET, bool FULL>
__global__ __launch_bounds__(256) void conv_first_kernel(const ConvFirstArgs a) {
  constexpr int MT = 2, NT = 2;                      // wave w: patch rows 2w, 2w+1 x 32 channels
  constexpr int W_U4 = CF_K * CF_BN * 4;             // filter image: row = ky*32 + n, 4 chunks
  constexpr int T_U4 = CF_TR * CF_PW * 4;            // operand tile: row = pixel (r*16 + x), 4 chunks
  constexpr int FROW = CF_SW * 3;                    // f32 values per halo row (66): pixel-major, channels packed
  constexpr int N_ELEM = CF_TR * FROW;               // f32 values of a halo (924)
  constexpr int N_LD = 4;                            // loads per lane and patch (256 lanes x 4 >= 924)
  constexpr int SROW = FROW + 6;                     // 72: 16-bit halo row pitch (the over-read of a row's last chunks stays inside)
  __shared__ __attribute__((aligned(16))) uint4 Wl[W_U4];
  __shared__ __attribute__((aligned(16))) uint4 Tl[T_U4];
  // 16-bit halo, rows of 22 pixels x 3 channels PACKED: the unrolled channels kx*3 + c of tile pixel (r, x) are the 21 CONSECUTIVE
  // values S[r][3x .. 3x+20] (channel kx*3 + c of pixel x = channel c of pixel x + kx = element 3(x+kx) + c)
  __shared__ __attribute__((aligned(16))) uint16_t Sl[CF_TR * SROW + 8];
  __shared__ float red[4 * 2 * CF_BN];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fchunk = lane >> 4;
  const int S = a.s;
  const int per_img = a.patches_x * a.patches_y;

  // ---- filter image -> LDS (once) ------------------------------------------------------------------------------------
  for (int idx = tid; idx < W_U4; idx += 256) {
    const int row = idx >> 2, q = idx & 3;
    const int tap = row / CF_BN, n = row - tap * CF_BN;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < a.co) v = *(const uint4*)(a.wt + (int64_t)n * a.kpad + tap * 32 + q * 8);
    Wl[row * 4 + (q ^ cf_swz(row))] = v;
  }
  for (int idx = tid; idx < T_U4; idx += 256) Tl[idx] = make_uint4(0, 0, 0, 0);     // (chunk 3 of every pixel stays zero)
  for (int idx = tid; idx < (CF_TR * SROW + 8) / 2; idx += 256) ((uint32_t*)Sl)[idx] = 0u;
  __syncthreads();              // (the zero fill is ordered before the first patch's halo values)

  // ---- halo loader: this lane's element u of the [14][22][3] f32 halo is e = tid + 256 u ------------------------------
  int e_r[N_LD], e_px[N_LD], e_off[N_LD], e_lds[N_LD];
#pragma unroll
  for (int u = 0; u < N_LD; ++u) {
    const int e = tid + 256 * u;
    const int r = e / FROW, t = e - r * FROW;
    e_r[u] = e < N_ELEM ? r : (1 << 24);             // (elements beyond the halo: never inside an image -> zeros)
    e_px[u] = t / 3;
    e_off[u] = (r * S * 3 + t) * 4;                   // byte offset relative to the f32 element of halo pixel (0, 0)
    e_lds[u] = e < N_ELEM ? r * SROW + t : CF_TR * SROW + (tid & 7);      // (beyond the halo: a dump slot nobody reads)
  }
  const uint64_t ia = (uint64_t)a.img;
  const cf_u32x4_t ir = {(uint32_t)ia, (uint32_t)(ia >> 32) & 0xffffu, (uint32_t)((int64_t)a.batch * S * S * 12), 0x00020000u};
  float bufA[N_LD], bufB[N_LD];                      // the halos of the next two patches of this workgroup, in flight
  auto issue_halo = [&](int patch, float (&buf)[N_LD]) __attribute__((always_inline)) {
    const bool real = patch < a.n_patches;
    const int img = patch / per_img, pr = patch - img * per_img;
    const int y0 = (pr / a.patches_x) * CF_PH - CF_PAD, x0 = (pr % a.patches_x) * CF_PW - CF_PAD;
    const int pbase = ((img * S + y0) * S + x0) * 12;             // byte offset of halo pixel (0, 0); may be negative
#pragma unroll
    for (int u = 0; u < N_LD; ++u) {
      const bool ok = real && (unsigned)(y0 + e_r[u]) < (unsigned)S && (unsigned)(x0 + e_px[u]) < (unsigned)S;
      cf_load(buf[u], ir, ok ? (uint32_t)(pbase + e_off[u]) : 0x80000000u);
    }
  };
  // operand-tile builder: this thread's chunks are (pixel (tid >> 2) + 64 u, q = tid & 3), u = 0 .. 3 (pixel < 224); source = 8
  // consecutive 16-bit values from S[(r0 + 4u)*SROW + 3x + 8q]; q = 3 (channels 24..31) is all zero, q = 2: channels 16..20 + zeros
  const int t_q = tid & 3, t_pix0 = tid >> 2;
  const int t_src0 = (t_pix0 >> 4) * SROW + 3 * (t_pix0 & 15) + 8 * t_q;

  const bool f_bias = a.flags & IMM_CONV_BIAS, f_stats = a.flags & IMM_CONV_STATS;
  float s1[NT][4], s2[NT][4], bv[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[j][r] = 0.f; s2[j][r] = 0.f;
      const int n = j * 16 + 4 * fchunk + r;
      bv[j][r] = (f_bias && n < a.co) ? a.bias[n] : 0.f;
    }
  // everything loaded so far (filter image, bias) has ARRIVED before the loop: its waits must not land inside it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[j][r]));

  const int G = (int)gridDim.x;
  const int n_mine = ((int)blockIdx.x < a.n_patches) ? (a.n_patches - (int)blockIdx.x + G - 1) / G : 0;
  issue_halo(blockIdx.x, bufA);
  issue_halo(blockIdx.x + G, bufB);
  // one patch: `buf` holds its halo (requested TWO patches ago) and is refilled for the patch two ahead; the two register sets alternate
  auto do_patch = [&](const int it, float (&buf)[N_LD]) __attribute__((always_inline)) {
    const int patch = blockIdx.x + it * G;
    // the four loads of this patch have arrived (see the header for the counts)
    if (!FULL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // (co < 32: the store count per patch varies — drain)
    else if (it == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LD) : "memory");
    else if (it == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LD + MT * NT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LD + 2 * MT * NT) : "memory");
#pragma unroll
    for (int u = 0; u < N_LD; ++u) cf_arrived(buf[u]);
#pragma unroll
    for (int u = 0; u < N_LD; ++u) Sl[e_lds[u]] = ET::from_f32(buf[u]);         // 16-bit halo (outside the image: zeros)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // halo complete; every wave is past the previous patch's reads of the operand tile
    if (t_q < 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pix = t_pix0 + 64 * u;
        if (pix < CF_TR * CF_PW) {
          const uint16_t* src = Sl + t_src0 + 4 * u * SROW;
          const uint32_t w0 = (uint32_t)src[0] | ((uint32_t)src[1] << 16), w1 = (uint32_t)src[2] | ((uint32_t)src[3] << 16);
          uint32_t w2 = (uint32_t)src[4] | ((uint32_t)src[5] << 16), w3 = (uint32_t)src[6] | ((uint32_t)src[7] << 16);
          if (t_q == 2) { w2 &= 0xffffu; w3 = 0u; }      // channels 21, 22, 23 do not exist
          Tl[pix * 4 + (t_q ^ cf_swz(pix))] = make_uint4(w0, w1, w2, w3);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // tile complete (also: the filter image, first patch)
    issue_halo(patch + 2 * G, buf);          // two patches ahead (out-of-range offsets beyond the last patch: the count stays exact)

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < CF_K; ++ky) {
      uint4 af[MT], bf[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int hp = (wid * MT + i + ky) * CF_PW + frow;
        af[i] = Tl[hp * 4 + (fchunk ^ cf_swz(hp))];
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int row = ky * CF_BN + j * 16 + frow;
        bf[j] = Wl[row * 4 + (fchunk ^ cf_swz(row))];
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = ET::mfma(bf[j], af[i], acc[i][j]);   // D[n][pixel]
    }

    // ---- store: lane = pixel (row wid*MT + i, column lane & 15), 4 consecutive channels -----------------------------
    const int img = patch / per_img, pr = patch - img * per_img;
    const int y0 = (pr / a.patches_x) * CF_PH, x0 = (pr % a.patches_x) * CF_PW;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int64_t m = ((int64_t)img * S + y0 + wid * MT + i) * S + x0 + frow;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = j * 16 + 4 * fchunk;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bv[j][r];
        if (f_stats) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
        }
        uint16_t* yp = a.y + m * a.ldy + n;
        if constexpr (FULL) *(uint2*)yp = make_uint2(ET::pack2(v[0], v[1]), ET::pack2(v[2], v[3]));
        else if (n + 3 < a.co) *(uint2*)yp = make_uint2(ET::pack2(v[0], v[1]), ET::pack2(v[2], v[3]));
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n + r < a.co) yp[r] = ET::from_f32(v[r]);
        }
      }
    }
  };
  for (int it = 0; it < n_mine; it += 2) {
    do_patch(it, bufA);
    if (it + 1 < n_mine) do_patch(it + 1, bufB);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the trailing look-ahead loads
#pragma unroll
  for (int u = 0; u < N_LD; ++u) { cf_arrived(bufA[u]); cf_arrived(bufB[u]); }

  if (f_stats) {
    // per-workgroup partial sums: 16 pixel lanes -> the four waves -> one row of (sum, sum of squares) per workgroup
    __syncthreads();
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
          const int nl = j * 16 + 4 * fchunk + r;
          red[(wid * 2 + 0) * CF_BN + nl] = s1[j][r];
          red[(wid * 2 + 1) * CF_BN + nl] = s2[j][r];
        }
    }
    __syncthreads();
    if (tid < CF_BN && tid < a.co) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += red[(w * 2 + 0) * CF_BN + tid]; t2 += red[(w * 2 + 1) * CF_BN + tid]; }
      a.stats[((int64_t)blockIdx.x * 2 + 0) * a.co + tid] = t1;
      a.stats[((int64_t)blockIdx.x * 2 + 1) * a.co + tid] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int cf_num_cu() {
  static int n = 0;
  if (n == 0) {
    hipDeviceProp_t p; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    else (void)hipGetLastError();
    if (n <= 0) n = 256;
  }
  return n;
}

static int cf_grid(int batch, int s) {
  const int n_patches = batch * (s / CF_PH) * (s / CF_PW);
  const int grid = 4 * cf_num_cu();                 // ~31 KB of LDS and 4 waves per workgroup: four per CU keep enough patches in flight
  return n_patches < grid ? n_patches : grid;
}

extern "C" int imm_conv_first_supported(int batch, int s, int co, int ldy) {
  static const bool off = imm_conv_disabled("first");
  if (off) return 0;
  if (batch <= 0 || s < 16 || s % CF_PW || co < 4 || co > CF_BN || co % 4 || ldy < co || ldy % 4) return 0;
  return ((int64_t)batch * s * s * 12 < (1LL << 31)) ? 1 : 0;      // (the image is addressed through a buffer descriptor)
}

extern "C" int imm_conv_first_stats_blocks(int batch, int s) { return (batch > 0 && s >= 16 && s % CF_PW == 0) ? cf_grid(batch, s) : IMM_E_INVALID; }

extern "C" int imm_conv_first(const float* image, const void* wt, int kpad, const float* bias, void* y, int ldy, float* stats_partial,
                              int dtype, int batch, int s, int co, int flags, void* stream) {
  IMM_REQUIRE(image && wt && y, "conv_first: null tensor");
  IMM_REQUIRE(dtype == IMM_BF16 || dtype == IMM_F16, "unknown dtype %d", dtype);
  if (!imm_conv_first_supported(batch, s, co, ldy))
    return imm_fail(IMM_E_UNSUPPORTED, "conv_first: batch %d side %d co %d ldy %d: needs side %% 16 == 0, co <= 32, co %% 4 == 0", batch, s, co, ldy);
  IMM_REQUIRE(kpad >= CF_K * 32 && kpad % 8 == 0, "conv_first: kpad=%d must hold 7 taps x 32 unrolled channels (imm_pack_weights, kh 7, kw 1, c_pad 32)", kpad);
  IMM_REQUIRE(!(flags & ~(IMM_CONV_BIAS | IMM_CONV_STATS)), "conv_first: flags 0x%x (bias and batch-norm sums only)", flags);
  IMM_REQUIRE(!(flags & IMM_CONV_BIAS) || bias, "conv_first: bias flag without bias");
  IMM_REQUIRE(!(flags & IMM_CONV_STATS) || stats_partial, "conv_first: stats flag without buffer");
  IMM_REQUIRE(((uintptr_t)wt % 16 == 0) && ((uintptr_t)y % 8 == 0) && ((uintptr_t)image % 4 == 0), "conv_first: alignment");
  ConvFirstArgs a;
  a.img = image; a.wt = (const uint16_t*)wt; a.bias = bias; a.y = (uint16_t*)y; a.stats = stats_partial;
  a.batch = batch; a.s = s; a.co = co; a.ldy = ldy; a.kpad = kpad; a.flags = flags;
  a.patches_x = s / CF_PW; a.patches_y = s / CF_PH; a.n_patches = batch * a.patches_x * a.patches_y;
  const int grid = cf_grid(batch, s);
  const bool full = co == CF_BN;
  if (dtype == IMM_BF16) {
    if (full) hipLaunchKernelGGL((conv_first_kernel<BF16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_first_kernel<BF16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  } else {
    if (full) hipLaunchKernelGGL((conv_first_kernel<F16, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_first_kernel<F16, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  }
  IMM_CHECK_LAUNCH("imm_conv_first");
  return 0;
}
