// conv_halo2.hip — second-generation 3x3 stride-1 convolution for the high-resolution, low-channel layers
// (ci in {32,64}, co in {32,64}): encoder conv_2 / conv_4, renderer conv_6 / conv_7, VGG conv1_2 and the data
// gradients of the same layers (reference call sites: imm/models/imm_model.py:213-241 encoder, :258-300 renderer,
// imm/models/selfsup/vgg16.py:346 conv1_2).  conv_halo.hip (v1) stays for co < 32, f32 output and the ablation bits.
//
// What the v1 ablation showed (tools/bench_conv.py --ablate, VGG conv1_2, 129 us): LDS reads + MFMA alone 67 us,
// epilogue +40 us, halo DMA +20 us — the three phases of a patch ran back to back because one workgroup per CU
// (72 KB of LDS-resident filter) walks them in lock step, and the LDS pipe (A and B fragments for every wave)
// was 1.6x busier than the matrix pipe.  This kernel changes the three things that follow from that:
//   * the filter lives in REGISTERS: a wave owns 32 output channels, so its nine taps are 9 x (ci/32) x 2 MFMA B
//     operands = 72/144 VGPRs loaded once per workgroup; LDS only carries the input halo (A operand), which
//     halves the LDS traffic per MFMA;
//   * the epilogue of patch p-1 (bias / ReLU / ReLU-backward mask / BN partial sums / 16-byte stores) is
//     interleaved tile row by tile row into the MFMA loop of patch p (two accumulator sets, loop unrolled by two so
//     the sets are addressed statically) — VALU and store issue ride in the shadow of the matrix pipe;
//   * the halo ring is NS deep (prefetch distance NS-1 patches instead of 1), and every global access of the loop
//     — halo DMA, mask DMA, output stores — is issued from inline asm so that ONE counted s_waitcnt vmcnt(N) per
//     patch is exact: each iteration issues the same number of VMEM instructions (out-of-range buffer offsets turn
//     the ones without a patch into no-ops), so "at most N outstanding" always means "the DMA of this patch landed".
// Output channel order inside a wave is permuted (MFMA row 4q+r of tile j <-> channel 8q+4j+r) so that a lane owns 8
// consecutive channels of its pixel: one 16-byte store (and one 16-byte mask read) per pixel and tile row.
// Results are bit-identical to v1 except for the order of the BN partial sums (deterministic either way).
#include "conv_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define H2_PH 8                 // patch rows
#define H2_PW 16                // patch cols (= MFMA operand rows)
#define H2_HW (H2_PW + 2)       // halo width 18
#define H2_HP 192               // halo pixels (10*18 = 180) padded to a whole number of DMA instructions
#define H2_OOB 0x80000000u      // buffer offset beyond every extent: loads return zero, stores are dropped

struct Halo2Args {
  ConvArgs c;
  int n_patches, patches_x, patches_y;
  uint32_t y_bytes, mask_bytes;
  int lg_px, lg_pi;                       // log2(patches_x), log2(patches per image), or -1 when not powers of two
};

__device__ __forceinline__ void h2_dma16(u32x4_t rsrc, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void h2_store16(u32x4_t rsrc, u32x4_t data, uint32_t voff, uint32_t soff) {
  // The trailing s_nop 1 is REQUIRED (round 5): a VMEM store of more than 64 bits reads its data registers over the cycles after
  // issue, and hipcc — which does not know what is inside an asm statement — pads no wait states behind it; without them its next
  // VALU instruction may overwrite v[data] before the store has read it (cdna_hip_programming.md §5.7 "Stores").  Seen as a
  // 0.3 % rate of corrupted 16-byte outputs of the ci = 64 instantiations (358 registers, the data registers are reused at once)
  // whenever a second process shared the GPU — the flaky two-rank test of rounds 3-5 (tools/det_graph.py reproduces it).
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

// same chunk swizzle as conv_halo.hip: a ds_read_b128 of 16 consecutive halo pixels (any start) is conflict-free
template <int C8>
__device__ __forceinline__ int h2_swz(int row) { return C8 == 8 ? (((row >> 1) & 3) << 1) : (((row >> 2) & 1) << 1); }

template <int V> struct H2Int { static constexpr int value = V; };

// halo pixel slots of one LDS stage: (8 + kh - 1) x (16 + kw - 1) pixels, rounded up so that the tile is a whole number
// of 1-KB DMA instructions per wave (4 waves)
__host__ __device__ constexpr int h2_halo_slots(int ci, int kh, int kw) {
  const int per_round = 4 * (64 / (ci / 8));
  return ((H2_PH + kh - 1) * (H2_PW + kw - 1) + per_round - 1) / per_round * per_round;
}

// sched_group_barrier masks (LLVM AMDGPU): the MFMA loop of a halo row is laid out as "1 MFMA + up to 3 other issues"
#define H2_SG_VALU 0x002
#define H2_SG_SALU 0x004
#define H2_SG_MFMA 0x008
#define H2_SG_DSREAD 0x100

template <typename ET, int CI, int BN, bool MASK, bool STATS, int NS, int KH = 3, int KW = 3>
__global__ __launch_bounds__(256) void conv_halo2_kernel(const Halo2Args ha) {
  const ConvArgs& a = ha.c;
  constexpr int C8 = CI / 8;                       // 16-byte chunks per input pixel
  constexpr int KS = CI / 32;                      // MFMA k-steps per tap
  constexpr int WGN = BN / 32, WGM = 4 / WGN;      // a wave owns 32 channels x MT patch rows
  constexpr int MT = H2_PH / WGM, NT = 2, NR = MT + KH - 1;   // NR halo rows per wave
  constexpr int HW = H2_PW + KW - 1, HROWS = H2_PH + KH - 1;   // halo tile (KH x KW filter; 3x3 or the 7x1 first layer)
  constexpr int PIXB = CI * 2, ROWB = HW * PIXB;         // bytes per halo pixel / halo row in LDS
  constexpr int PIX_PER_DMA = 64 / C8;             // halo pixels per 1-KB DMA instruction
  constexpr int HP = h2_halo_slots(CI, KH, KW);    // halo pixel slots, padded to 4 whole DMA instructions per wave step
  constexpr int HALO_DMA = HP / PIX_PER_DMA;       // 24 / 12 (3x3)
  constexpr int DH = HALO_DMA / 4;                 // per wave
  constexpr int H_U4 = HP * C8;                    // uint4 per halo stage
  constexpr int O8 = BN / 8;                       // 16-byte chunks per output (= mask) pixel
  constexpr int M_U4 = MASK ? H2_PH * H2_PW * O8 : 0;
  constexpr int MASK_DMA = M_U4 / 64, DM = MASK_DMA / 4;
  // VMEM instructions per wave and iteration: DH + DM DMA pieces and MT stores, spread over the halo rows.  "At most
  // (NS-2) whole iterations outstanding" => everything issued NS-1 iterations ago (halo of this patch, mask of the
  // previous one) has landed, whatever the order inside an iteration.
  constexpr int WAITN = (NS - 2) * (DH + DM + MT);
  static_assert(HALO_DMA % 4 == 0 && MASK_DMA % 4 == 0 && WAITN < 64, "per-wave VMEM schedule");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [NS][H_U4] halo ring | [NS][M_U4] mask ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: LDS-DMA destinations are wave-uniform
  const int wm = wid / WGN, wn = wid % WGN;
  const int frow = lane & 15, q = lane >> 4;
  const uint64_t xa = (uint64_t)a.x, ya = (uint64_t)a.y, ma = (uint64_t)a.mask;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t yr = {(uint32_t)ya, (uint32_t)(ya >> 32) & 0xffffu, ha.y_bytes, 0x00020000u};
  const u32x4_t mr = {(uint32_t)ma, (uint32_t)(ma >> 32) & 0xffffu, ha.mask_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;
  const int G = gridDim.x, per_img = ha.patches_x * ha.patches_y;
  // Patch order.  Workgroups are dealt round-robin to the 8 XCDs (private L2 each): give every XCD one contiguous
  // band of patches and let its workgroups walk the band side by side, so the halo rows that neighbouring patches
  // share are fetched once per L2 instead of once per XCD.  seq = this workgroup's sequence index; -1 = no patch.
  const bool xcd_mode = (G & 7) == 0;
  const int xq = ha.n_patches >> 3, xrem = ha.n_patches & 7, xid = blockIdx.x & 7;
  const int band0 = xid * xq + (xid < xrem ? xid : xrem), band_n = xq + (xid < xrem ? 1 : 0);
  auto seq_patch = [&](int seq) -> int {
    if (xcd_mode) {
      const int local = (int)(blockIdx.x >> 3) + seq * (G >> 3);
      return local < band_n ? band0 + local : -1;
    }
    const int pt = blockIdx.x + seq * G;
    return pt < ha.n_patches ? pt : -1;
  };
  // decoded patch: image and pixel origin (scalars).  Invalid patches get an origin far outside every image, which
  // turns all their DMA / store offsets into out-of-range no-ops.
  struct Pd { int img, y0, x0; };
  const bool pow2 = ha.lg_px >= 0;
  auto decode = [&](int patch) -> Pd {
    Pd d;
    if (patch < 0) { d.img = 0; d.y0 = 0x4000; d.x0 = 0; return d; }
    int pr, py;
    if (pow2) { d.img = patch >> ha.lg_pi; pr = patch & (per_img - 1); py = pr >> ha.lg_px; d.x0 = (pr & (ha.patches_x - 1)) * H2_PW; }
    else { d.img = patch / per_img; pr = patch - d.img * per_img; py = pr / ha.patches_x; d.x0 = (pr - py * ha.patches_x) * H2_PW; }
    d.y0 = py * H2_PH;
    return d;
  };

  // ---- filter -> registers (MFMA B operand layout; channel permutation described in the header) ---------------
  u32x4_t bw[KH * KW][KS][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = wn * 32 + (frow >> 2) * 8 + j * 4 + (frow & 3);
    const uint16_t* wrow = a.wt + (size_t)n * a.kpad + q * 8;
#pragma unroll
    for (int tap = 0; tap < KH * KW; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bw[tap][ks][j] = *(const u32x4_t*)(wrow + tap * CI + ks * 32);
  }
  const bool f_bias = a.flags & IMM_CONV_BIAS;
  const float relu_floor = (a.flags & IMM_CONV_RELU) ? 0.f : -__builtin_huge_valf();
  float bv[NT][4], s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[j][r] = 0.f; s2[j][r] = 0.f;
      bv[j][r] = f_bias ? a.bias[wn * 32 + q * 8 + j * 4 + r] : 0.f;
    }
  // Every compiler-visible global load ends here: pass the values through empty asm statements so the compiler's
  // own s_waitcnt for them is placed before the loop, not (conservatively, as vmcnt(0)) inside it.
#pragma unroll
  for (int tap = 0; tap < KH * KW; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(bw[tap][ks][j]));
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(bv[j][r]));

  // ---- DMA pieces (piece k of a wave = instruction wid + 4k of the workgroup's batch) ---------------------------
  // patch-invariant part of a lane's halo addresses: halo pixel (hy, hx), byte offset relative to the halo origin;
  // the 12 padding slots get a row far outside every image.  LDS image: [halo row][pixel][chunk ^ swz(pixel)].
  int hyx[DH];
  uint32_t hrel[DH];
#pragma unroll
  for (int k = 0; k < DH; ++k) {
    const int hp = (wid + 4 * k) * PIX_PER_DMA + lane / C8;        // halo slot 0..191
    const int hy = hp / HW, hx = hp - hy * HW;
    const int sc = (lane % C8) ^ h2_swz<C8>(hx);
    hyx[k] = ((hp < HROWS * HW ? hy : 0x4000) << 16) | hx;
    hrel[k] = (uint32_t)((hy * a.wi + hx) * a.ldx * 2 + sc * 16);
  }
  auto halo_piece = [&](const Pd& d, int stage, int k) {
    const int y0 = d.y0 - KH / 2, x0 = d.x0 - KW / 2;
    const uint32_t soff = (uint32_t)(d.img * a.hi * a.wi) * (uint32_t)(a.ldx * 2);
    const uint32_t base = (uint32_t)((y0 * a.wi + x0) * a.ldx * 2);     // may be "negative": only used when in range
    const int iy = y0 + (hyx[k] >> 16), ix = x0 + (hyx[k] & 0xffff);
    const bool ok = ((unsigned)iy < (unsigned)a.hi) & ((unsigned)ix < (unsigned)a.wi);
    uint32_t in_range = base + hrel[k];
    asm volatile("" : "+v"(in_range));        // keep the add unconditional: a select, not an exec-masked block
#ifdef H2_ABLATE   // diagnosis builds only (tools/ablate_build.sh): bit 0 = no halo bytes move, bit 1 = no output bytes move
    const uint32_t vo = (H2_ABLATE & 1) ? H2_OOB : (ok ? in_range : H2_OOB);
#else
    const uint32_t vo = ok ? in_range : H2_OOB;
#endif
    h2_dma16(xr, lds_base + (uint32_t)((stage * H_U4) * 16 + (wid + 4 * k) * 1024), vo, soff);
  };
  uint32_t mrel[DM > 0 ? DM : 1];
  if constexpr (MASK) {
#pragma unroll
    for (int k = 0; k < DM; ++k) {
      const int px = (wid + 4 * k) * (64 / O8) + lane / O8;          // patch pixel 0..127
      const int row = px >> 4, x = px & 15;
      const int c = (lane % O8) ^ (x & (O8 - 1));                    // stored slot (lane % O8) holds source chunk c
      mrel[k] = (uint32_t)((row * a.wo + x) * a.ldmask * 2 + c * 16);
    }
  }
  auto mask_piece = [&](const Pd& d, int stage, int k) {
    if constexpr (MASK) {
      const uint32_t soff = (uint32_t)(d.img * a.ho * a.wo) * (uint32_t)(a.ldmask * 2);
      const uint32_t vo = d.y0 < 0x4000 ? (uint32_t)((d.y0 * a.wo + d.x0) * a.ldmask * 2) + mrel[k] : H2_OOB;
      h2_dma16(mr, lds_base + (uint32_t)((NS * H_U4 + stage * M_U4) * 16 + (wid + 4 * k) * 1024), vo, soff);
    }
  };

  // per-lane byte offset of the A fragments inside a halo stage, for the three horizontal taps (k-step 1 = XOR 64;
  // halo row rr = + rr * ROWB, an immediate of the ds_read)
  int lrel[KW];
#pragma unroll
  for (int kx = 0; kx < KW; ++kx) lrel[kx] = wm * MT * ROWB + (frow + kx) * PIXB + (q ^ h2_swz<C8>(frow + kx)) * 16;
  const int moff = frow * O8 + ((wn * 4 + q) ^ (frow & (O8 - 1)));   // + patch row * 16 * O8
  const uint32_t ovoff = (uint32_t)((frow * a.ldy + wn * 32 + q * 8) * 2);   // lane part of the output address

  // ---- prologue: NS-1 batches in flight; dummy stores keep the per-iteration VMEM count uniform ----------------
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  Pd dq[NS];                                   // decoded patches it .. it+NS-1 (shift register)
#pragma unroll
  for (int s = 0; s < NS; ++s) dq[s] = decode(seq_patch(s));
  const Pd none = decode(-1);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
    for (int k = 0; k < DH; ++k) halo_piece(dq[s], s, k);
#pragma unroll
    for (int k = 0; k < DM; ++k) mask_piece(s >= 1 ? dq[s - 1] : none, s >= 1 ? s - 1 : NS - 1, k);
#pragma unroll
    for (int i = 0; i < MT; ++i) h2_store16(yr, zero4, H2_OOB, 0u);
  }

  f32x4_t acc[2][MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[1][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int it = 0, hs = 0;                        // sequence index of this workgroup's patch, its halo stage (it % NS)
  uint32_t prev_voff = H2_OOB, prev_soff = 0u;
  float prev_w = 0.f;                        // 1 when the previous patch exists (BN partial sums)

  // One patch: MFMA loop of patch `it` into acc[PH] with the epilogue of patch it-1 (acc[PH^1]) interleaved.
  // MMA = 0 is the pass after the last patch: only the deferred epilogue runs.
  auto step = [&](auto phase, auto mma) {
    constexpr int PH = decltype(phase)::value;
    constexpr bool MMA = decltype(mma)::value;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");   // this wave's share of halo(it) and mask(it-1) landed
    __builtin_amdgcn_s_barrier();                                  // => everyone's; stage it-1 is free for reuse
    __builtin_amdgcn_sched_barrier(0);
    const int hprev = hs == 0 ? NS - 1 : hs - 1;                   // (it-1) % NS
    const int hprev2 = hprev == 0 ? NS - 1 : hprev - 1;            // (it-2) % NS
    const uint4* Ml = smem + NS * H_U4 + hprev * M_U4 + moff;
    const uint4* rowp[KW][KS];                                     // lane's fragment address in halo row 0 of this stage
#pragma unroll
    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        rowp[kx][ks] = (const uint4*)((const char*)smem + hs * (H_U4 * 16) + (lrel[kx] ^ (ks * 64)));
    uint4 fa[2][KW][KS];
    if constexpr (MMA) {
#pragma unroll
      for (int kx = 0; kx < KW; ++kx)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) fa[0][kx][ks] = rowp[kx][ks][0];
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[PH][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);

    // Halo-row major: the A fragments of halo row rr (3 horizontal shifts x KS k-steps) are read ONCE and feed the
    // output rows rr, rr-1, rr-2 (vertical taps 0, 1, 2) — NR*3*KS ds_read_b128 per patch instead of 9*MT*KS.  One
    // wave per SIMD means nothing else hides this wave's non-MFMA issues, so each row region is laid out as: next
    // row's fragments first, then 1 MFMA : <= 3 other issues — the epilogue of the previous patch (tile row rr-1),
    // this row's share of the next halo / mask DMA and the scalar bookkeeping ride in the matrix pipe's shadow.
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
      int n_mfma = 0;
      if constexpr (MMA) {
        if (rr + 1 < NR) {
#pragma unroll
          for (int kx = 0; kx < KW; ++kx)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) fa[(rr + 1) & 1][kx][ks] = rowp[kx][ks][(rr + 1) * (ROWB / 16)];
        }
#pragma unroll
        for (int k = rr; k < DH; k += NR) halo_piece(dq[NS - 1], hprev, k);
#pragma unroll
        for (int k = rr; k < DM; k += NR) mask_piece(dq[NS - 2], hprev2, k);
#pragma unroll
        for (int kx = 0; kx < KW; ++kx)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int ky = 0; ky < KH; ++ky) {
              const int i = rr - ky;
              if (i >= 0 && i < MT) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                  acc[PH][i][j] = ET::mfma(__builtin_bit_cast(uint4, bw[ky * KW + kx][ks][j]), fa[rr & 1][kx][ks], acc[PH][i][j]);   // D[n][pixel]
                n_mfma += NT;
              }
            }
      }
      // ---- epilogue of the previous patch, tile row (rr - 1) ----------------------------------------------------
      if (rr >= 1 && rr <= MT) {
        const int i = rr - 1;
        float v[8];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) v[j * 4 + r] = fmaxf(acc[PH ^ 1][i][j][r] + bv[j][r], relu_floor);
        float mf[8];
        if constexpr (MASK) {
          const uint4 mk = Ml[(wm * MT + i) * 16 * O8];
          unpack8<ET>(mk, mf);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (mf[e] > 0.f) ? v[e] : 0.f;
        }
        if constexpr (STATS) {
          // MASK && STATS: second sum = sum(v * mask_ref), the batch-norm backward sums of the layer this gradient enters
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float t = v[j * 4 + r] * prev_w;
              s1[j][r] += t; s2[j][r] += t * (MASK ? mf[j * 4 + r] : t);
            }
        }
        const uint4 o = pack8<ET>(v);
        const u32x4_t od = {o.x, o.y, o.z, o.w};
#ifdef H2_ABLATE
        h2_store16(yr, od, (H2_ABLATE & 2) ? H2_OOB : prev_voff + (uint32_t)((wm * MT + i) * a.wo * a.ldy * 2), prev_soff);
#else
        h2_store16(yr, od, prev_voff + (uint32_t)((wm * MT + i) * a.wo * a.ldy * 2), prev_soff);
#endif
      }
      if (rr == NR - 1) {
        // bookkeeping for the next iteration (scalar): hand this patch to the next epilogue, shift the decode queue
        if constexpr (MMA) {
          prev_voff = (uint32_t)((dq[0].y0 * a.wo + dq[0].x0) * a.ldy * 2) + ovoff;
          prev_soff = (uint32_t)(dq[0].img * a.ho * a.wo) * (uint32_t)(a.ldy * 2);
          prev_w = 1.f;
#pragma unroll
          for (int s = 0; s + 1 < NS; ++s) dq[s] = dq[s + 1];
          dq[NS - 1] = decode(seq_patch(it + NS));
        }
      }
      if constexpr (MMA) {
        __builtin_amdgcn_sched_group_barrier(H2_SG_DSREAD, KW * KS, 0);
#pragma unroll
        for (int m = 0; m < 36; ++m) {
          if (m < n_mfma) {
            __builtin_amdgcn_sched_group_barrier(H2_SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(H2_SG_VALU | H2_SG_SALU, 3, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    hs = hs + 1 == NS ? 0 : hs + 1;
    ++it;
  };
  // patches p0 .. p_last (every workgroup has at least one), then one more pass that only stores p_last
  for (;;) {
    step(H2Int<0>(), H2Int<1>());
    if (dq[0].y0 >= 0x4000) { step(H2Int<1>(), H2Int<0>()); break; }
    step(H2Int<1>(), H2Int<1>());
    if (dq[0].y0 >= 0x4000) { step(H2Int<0>(), H2Int<0>()); break; }
  }

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
  if constexpr (STATS) {
    __syncthreads();
    float* red = (float*)smem;                         // [WGM][2][BN]
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
          const int nl = wn * 32 + q * 8 + j * 4 + r;
          red[(wm * 2 + 0) * BN + nl] = s1[j][r];
          red[(wm * 2 + 1) * BN + nl] = s2[j][r];
        }
    }
    __syncthreads();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
      a.stats[((int64_t)blockIdx.x * 2 + 0) * a.co + tid] = t1;
      a.stats[((int64_t)blockIdx.x * 2 + 1) * a.co + tid] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// conv_halo2x_kernel — the 64 -> 64 channel instantiation on v_mfma_f32_32x32x16 (round 6).
//
// Why: this kernel family runs ONE wave per SIMD (352-400 registers: nine filter taps + two accumulator sets), and at one wave per
// SIMD v_mfma_f32_16x16x32 sustains 1.09-1.25 PFLOP/s on this part against 1.58-1.70 for v_mfma_f32_32x32x16 (two waves per SIMD:
// 1.7-1.9 either way; tools/probes/mfma_rate_probe.hip, profiles/r06_mfma_rate_probe.txt).  VGG conv1_2 (77 GFLOP) sat exactly on
// that ceiling: 62 us = 1.25 PFLOP/s with no input bytes moving (profiles/r06_halo2_ablation.txt), 213 MB of HBM traffic = ~45 us.
//
// Same program as conv_halo2_kernel<64, 64> (persistent workgroup, filter in registers, NS-deep halo ring, epilogue of patch p-1
// interleaved into the MFMA loop of patch p, every loop VMEM op from inline asm with a fixed count per step); what changes is the
// tile algebra.  Wave (wm, wn) still owns patch rows 4wm .. 4wm+3 x channels 32wn .. 32wn+31, now as TWO 32 x 32 tiles:
//   tile t = output rows (t, t + 2) of the wave's four  x 16 columns  = 32 pixels;   lane: pixel p = lane & 31 (row half p >> 4,
//   column p & 15), k group h5 = lane >> 5 (8 of the 16 channels of a k-step).
//   A operand of "fragment row" r (r = 0..3) = halo rows (r, r + 2) x 16 columns: tile 0 uses it for vertical tap ky = r, tile 1
//   for ky = r - 1 — four fragment rows x 3 column shifts x 4 k-steps = 48 ds_read_b128 per patch and wave (the 16x16 form: 36).
//   B operand (registers): MFMA row rho <-> channel 32wn + 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3), so that a lane, which
//   holds D rows (reg & 3) + 8 (reg >> 2) + 4 h5, owns the 16 CONSECUTIVE channels 32wn + 16 h5 + reg: two 16-byte stores per tile.
//   LDS halo image: [halo row][pixel][chunk ^ ((hx >> 1) & 7)] — the 16 lanes a ds_read_b128 serves together ({0-3, 12-15, 20-27}:
//   columns 0-3 and 12-15 of one halo row, 4-11 of the other) then hit 16 distinct 16-byte slots mod 256 B (checked by enumeration
//   for every wave, fragment row, column shift and k-step).
// Accumulation order over the taps differs from the 16x16 form (results to rounding).  No batch-norm partial sums (the 64 -> 64
// layers that need them keep the 16x16 form).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <typename ET> struct H2xMfma;
template <> struct H2xMfma<BF16> {
  __device__ static __forceinline__ f32x16_t mfma(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct H2xMfma<F16> {
  __device__ static __forceinline__ f32x16_t mfma(uint4 a, uint4 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
__device__ __forceinline__ int h2x_swz(int hx) { return (hx >> 1) & 7; }

#ifndef H2X_ABLATE   // diagnosis builds: 1 no halo bytes move, 2 no output bytes move, 4 no MFMAs, 8 no fragment reads inside the row loop
#define H2X_ABLATE 0
#endif
template <typename ET, bool MASK, int NS>
__global__ __launch_bounds__(256) void conv_halo2x_kernel(const Halo2Args ha) {
  const ConvArgs& a = ha.c;
  constexpr int CI = 64, BN = 64, C8 = 8, KS = 4, KH = 3, KW = 3;
  constexpr int MT = 4, NF = 4;                    // patch rows per wave; fragment rows per wave and patch
  constexpr int HW = H2_PW + 2, HROWS = H2_PH + 2;
  constexpr int PIXB = CI * 2, ROWB = HW * PIXB;
  constexpr int PIX_PER_DMA = 64 / C8;
  constexpr int HP = h2_halo_slots(CI, KH, KW);
  constexpr int HALO_DMA = HP / PIX_PER_DMA, DH = HALO_DMA / 4;
  constexpr int H_U4 = HP * C8;
  constexpr int O8 = BN / 8;
  constexpr int M_U4 = MASK ? H2_PH * H2_PW * O8 : 0;
  constexpr int MASK_DMA = M_U4 / 64, DM = MASK_DMA / 4;
  constexpr int WAITN = (NS - 2) * (DH + DM + MT);
  static_assert(HALO_DMA % 4 == 0 && MASK_DMA % 4 == 0 && WAITN < 64, "per-wave VMEM schedule");
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];   // [NS][H_U4] halo ring | [NS][M_U4] mask ring

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int px = lane & 15, half = (lane >> 4) & 1, h5 = lane >> 5;
  const uint64_t xa = (uint64_t)a.x, ya = (uint64_t)a.y, ma = (uint64_t)a.mask;
  const u32x4_t xr = {(uint32_t)xa, (uint32_t)(xa >> 32) & 0xffffu, a.x_bytes, 0x00020000u};
  const u32x4_t yr = {(uint32_t)ya, (uint32_t)(ya >> 32) & 0xffffu, ha.y_bytes, 0x00020000u};
  const u32x4_t mr = {(uint32_t)ma, (uint32_t)(ma >> 32) & 0xffffu, ha.mask_bytes, 0x00020000u};
  const uint32_t lds_base = (uint32_t)(size_t)(lds_void_t*)smem;
  const int G = gridDim.x, per_img = ha.patches_x * ha.patches_y;
  const bool xcd_mode = (G & 7) == 0;
  const int xq = ha.n_patches >> 3, xrem = ha.n_patches & 7, xid = blockIdx.x & 7;
  const int band0 = xid * xq + (xid < xrem ? xid : xrem), band_n = xq + (xid < xrem ? 1 : 0);
  auto seq_patch = [&](int seq) -> int {
    if (xcd_mode) {
      const int local = (int)(blockIdx.x >> 3) + seq * (G >> 3);
      return local < band_n ? band0 + local : -1;
    }
    const int pt = blockIdx.x + seq * G;
    return pt < ha.n_patches ? pt : -1;
  };
  struct Pd { int img, y0, x0; };
  const bool pow2 = ha.lg_px >= 0;
  auto decode = [&](int patch) -> Pd {
    Pd d;
    if (patch < 0) { d.img = 0; d.y0 = 0x4000; d.x0 = 0; return d; }
    int pr, py;
    if (pow2) { d.img = patch >> ha.lg_pi; pr = patch & (per_img - 1); py = pr >> ha.lg_px; d.x0 = (pr & (ha.patches_x - 1)) * H2_PW; }
    else { d.img = patch / per_img; pr = patch - d.img * per_img; py = pr / ha.patches_x; d.x0 = (pr - py * ha.patches_x) * H2_PW; }
    d.y0 = py * H2_PH;
    return d;
  };

  // ---- filter -> registers: B operand of a 32x32x16 MFMA = 32 channels x 16 k; lane: row rho = lane & 31, k group h5 ---------
  u32x4_t bw[KH * KW][KS];
  {
    const int rho = lane & 31;
    const int n = wn * 32 + 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3);
    const uint16_t* wrow = a.wt + (size_t)n * a.kpad + h5 * 8;
#pragma unroll
    for (int tap = 0; tap < KH * KW; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) bw[tap][ks] = *(const u32x4_t*)(wrow + tap * CI + ks * 16);
  }
  const bool f_bias = a.flags & IMM_CONV_BIAS;
  const float relu_floor = (a.flags & IMM_CONV_RELU) ? 0.f : -__builtin_huge_valf();
  float bv[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bv[r] = f_bias ? a.bias[wn * 32 + 16 * h5 + r] : 0.f;
#pragma unroll
  for (int tap = 0; tap < KH * KW; ++tap)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(bw[tap][ks]));
#pragma unroll
  for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(bv[r]));

  // ---- DMA pieces (conv_halo2_kernel's, with this kernel's chunk swizzle) ---------------------------------------------
  int hyx[DH];
  uint32_t hrel[DH];
#pragma unroll
  for (int k = 0; k < DH; ++k) {
    const int hp = (wid + 4 * k) * PIX_PER_DMA + lane / C8;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int sc = (lane % C8) ^ h2x_swz(hx);
    hyx[k] = ((hp < HROWS * HW ? hy : 0x4000) << 16) | hx;
    hrel[k] = (uint32_t)((hy * a.wi + hx) * a.ldx * 2 + sc * 16);
  }
  auto halo_piece = [&](const Pd& d, int stage, int k) {
    const int y0 = d.y0 - 1, x0 = d.x0 - 1;
    const uint32_t soff = (uint32_t)(d.img * a.hi * a.wi) * (uint32_t)(a.ldx * 2);
    const uint32_t base = (uint32_t)((y0 * a.wi + x0) * a.ldx * 2);
    const int iy = y0 + (hyx[k] >> 16), ix = x0 + (hyx[k] & 0xffff);
    const bool ok = ((unsigned)iy < (unsigned)a.hi) & ((unsigned)ix < (unsigned)a.wi);
    uint32_t in_range = base + hrel[k];
    asm volatile("" : "+v"(in_range));
    const uint32_t vo = (H2X_ABLATE & 1) ? H2_OOB : (ok ? in_range : H2_OOB);
    h2_dma16(xr, lds_base + (uint32_t)((stage * H_U4) * 16 + (wid + 4 * k) * 1024), vo, soff);
  };
  uint32_t mrel[DM > 0 ? DM : 1];
  if constexpr (MASK) {
#pragma unroll
    for (int k = 0; k < DM; ++k) {
      const int mp = (wid + 4 * k) * (64 / O8) + lane / O8;
      const int row = mp >> 4, x = mp & 15;
      const int c = (lane % O8) ^ (x & (O8 - 1));
      mrel[k] = (uint32_t)((row * a.wo + x) * a.ldmask * 2 + c * 16);
    }
  }
  auto mask_piece = [&](const Pd& d, int stage, int k) {
    if constexpr (MASK) {
      const uint32_t soff = (uint32_t)(d.img * a.ho * a.wo) * (uint32_t)(a.ldmask * 2);
      const uint32_t vo = d.y0 < 0x4000 ? (uint32_t)((d.y0 * a.wo + d.x0) * a.ldmask * 2) + mrel[k] : H2_OOB;
      h2_dma16(mr, lds_base + (uint32_t)((NS * H_U4 + stage * M_U4) * 16 + (wid + 4 * k) * 1024), vo, soff);
    }
  };

  // per-lane byte offset of the A fragment of fragment row 0 (halo rows wm*4 + 2*half), column shift kx, k-step 0; k-step ks = XOR
  // ks * 32 (chunk 2 ks + h5 = (2 ks) ^ h5, the swizzle is an XOR too), fragment row r = + r * ROWB
  int lrel[KW];
#pragma unroll
  for (int kx = 0; kx < KW; ++kx) lrel[kx] = ((wm * MT + 2 * half) * HW + px + kx) * PIXB + ((h5 ^ h2x_swz(px + kx)) << 4);
  // mask ring: pixel (row, x) of the patch, chunk c at slot c ^ (x & 7); this lane's 16 channels = chunks 4 wn + 2 h5, + 1
  const int moff0 = px * O8 + ((wn * 4 + 2 * h5) ^ (px & (O8 - 1))), moff1 = px * O8 + ((wn * 4 + 2 * h5 + 1) ^ (px & (O8 - 1)));
  const uint32_t ovoff = (uint32_t)(((2 * half * a.wo + px) * a.ldy + wn * 32 + 16 * h5) * 2);   // lane part of the output address

  // ---- prologue ------------------------------------------------------------------------------------------------------
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  Pd dq[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) dq[s] = decode(seq_patch(s));
  const Pd none = decode(-1);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
#pragma unroll
    for (int k = 0; k < DH; ++k) halo_piece(dq[s], s, k);
#pragma unroll
    for (int k = 0; k < DM; ++k) mask_piece(s >= 1 ? dq[s - 1] : none, s >= 1 ? s - 1 : NS - 1, k);
#pragma unroll
    for (int i = 0; i < MT; ++i) h2_store16(yr, zero4, H2_OOB, 0u);
  }

  f32x16_t acc[2][2];                          // [accumulator set][tile]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[1][t][r] = 0.f;
  int it = 0, hs = 0;
  uint32_t prev_voff = H2_OOB, prev_soff = 0u;

  auto step = [&](auto phase, auto mma) {
    constexpr int PH = decltype(phase)::value;
    constexpr bool MMA = decltype(mma)::value;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int hprev = hs == 0 ? NS - 1 : hs - 1;
    const int hprev2 = hprev == 0 ? NS - 1 : hprev - 1;
    const uint4* Ml = smem + NS * H_U4 + hprev * M_U4;
    const uint4* rowp[KW][KS];
#pragma unroll
    for (int kx = 0; kx < KW; ++kx)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        rowp[kx][ks] = (const uint4*)((const char*)smem + hs * (H_U4 * 16) + (lrel[kx] ^ (ks * 32)));
    // The 48 fragment reads of a patch as TWELVE units (fragment row rr = u / 3, column shift kx = u % 3) of four k-steps each, in a
    // four-deep register ring: unit u + 3 is requested while unit u's 4-8 MFMAs run (the row-at-a-time double buffer of the
    // 16x16 form — twelve reads in front of 12-24 MFMAs — left this wave's in-order issue behind its own LDS queue: 86 us against
    // 50 with the reads compiled out and 44 with the MFMAs compiled out, profiles/r06_halo2x_ablation.txt)
    constexpr int NU = NF * KW, PD = 3;
    uint4 fa[4][KS];
    auto load_unit = [&](const int u) __attribute__((always_inline)) {
      const int rr = u / KW, kx = u - rr * KW;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) fa[u & 3][ks] = rowp[kx][ks][rr * (ROWB / 16)];
    };
    if constexpr (MMA) {
#pragma unroll
      for (int u = 0; u < PD; ++u) load_unit(u);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[PH][t][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int rr = u / KW, kx = u - rr * KW;
      int n_mfma = 0;
      if constexpr (MMA) {
        if (u + PD < NU && !(H2X_ABLATE & 8)) load_unit(u + PD);
        if (u < DH) halo_piece(dq[NS - 1], hprev, u);
        if (u >= DH && u - DH < DM) mask_piece(dq[NS - 2], hprev2, u - DH);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int ky = rr - t;                     // fragment row rr = halo rows (rr, rr + 2): tile t's vertical tap rr - t
            if (ky >= 0 && ky < KH && !(H2X_ABLATE & 4)) {
              acc[PH][t] = H2xMfma<ET>::mfma(__builtin_bit_cast(uint4, bw[ky * KW + kx][ks]), fa[(H2X_ABLATE & 8) ? (u < PD ? u : 0) : (u & 3)][ks], acc[PH][t]);   // D[n][pixel]
              n_mfma += 1;
            }
          }
      }
      // ---- epilogue of the previous patch, one of its four 16-byte stores per fragment row: tile e >> 1, channel half e & 1 ------
      if (kx == 0) {
        const int e = rr, t = e >> 1, ch = e & 1;
        float v[8];
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) v[q8] = fmaxf(acc[PH ^ 1][t][ch * 8 + q8] + bv[ch * 8 + q8], relu_floor);
        if constexpr (MASK) {
          const uint4 mk = Ml[(wm * MT + t + 2 * half) * 16 * O8 + (ch ? moff1 : moff0)];   // this lane's pixel: patch row wm*4 + t + 2*half
          float mf[8];
          unpack8<ET>(mk, mf);
#pragma unroll
          for (int q8 = 0; q8 < 8; ++q8) v[q8] = (mf[q8] > 0.f) ? v[q8] : 0.f;
        }
        const uint4 o = pack8<ET>(v);
        const u32x4_t od = {o.x, o.y, o.z, o.w};
        h2_store16(yr, od, (H2X_ABLATE & 2) ? H2_OOB : prev_voff + (uint32_t)((wm * MT + t) * a.wo * a.ldy * 2 + ch * 16), prev_soff);
      }
      if (u == NU - 1) {
        if constexpr (MMA) {
          prev_voff = (uint32_t)((dq[0].y0 * a.wo + dq[0].x0) * a.ldy * 2) + ovoff;
          prev_soff = (uint32_t)(dq[0].img * a.ho * a.wo) * (uint32_t)(a.ldy * 2);
#pragma unroll
          for (int s_ = 0; s_ + 1 < NS; ++s_) dq[s_] = dq[s_ + 1];
          dq[NS - 1] = decode(seq_patch(it + NS));
        }
      }
      if constexpr (MMA) {
#pragma unroll
        for (int m = 0; m < 2 * KS; ++m) {
          if (m < n_mfma) {
            __builtin_amdgcn_sched_group_barrier(H2_SG_MFMA, 1, 0);
            if (m < KS) __builtin_amdgcn_sched_group_barrier(H2_SG_DSREAD, 1, 0);
            __builtin_amdgcn_sched_group_barrier(H2_SG_VALU | H2_SG_SALU, 4, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    hs = hs + 1 == NS ? 0 : hs + 1;
    ++it;
  };
  for (;;) {
    step(H2Int<0>(), H2Int<1>());
    if (dq[0].y0 >= 0x4000) { step(H2Int<1>(), H2Int<0>()); break; }
    step(H2Int<1>(), H2Int<1>());
    if (dq[0].y0 >= 0x4000) { step(H2Int<0>(), H2Int<0>()); break; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // no LDS-DMA may outlive the workgroup
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Halo ring depth (prefetch distance NS-1 patches): bytes in flight per CU, not arithmetic, set the speed of these
// layers (one workgroup per CU at ci = 64).  ci = 64: 24 KB per stage, +16 KB for the mask stage.
static int h2_ns(bool mask) {
  static int v[2] = {0, 0};
  if (v[mask] == 0) {
    v[mask] = mask ? 3 : 4;
  }
  return v[mask];
}

static int h2_num_cu() {
  static int cu = 0;
  if (cu == 0) {
    hipDeviceProp_t p; int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cu = p.multiProcessorCount;
    if (cu <= 0) cu = 256;
  }
  return imm_limit_cus(cu);
}

bool imm_halo2_applicable(const imm_conv_desc* d) {
  static const bool off = imm_conv_disabled("halo2") || imm_conv_disabled("halo");
  if (off) return false;
  const bool k33 = d->kh == 3 && d->kw == 3 && d->pad_t == 1 && d->pad_l == 1;
  // the tap-unrolled first layer (imm_model.py:215 7x7 conv over the 7x3 horizontally unrolled image): 7x1 over 32 channels
  const bool k71 = d->kh == 7 && d->kw == 1 && d->pad_t == 3 && d->pad_l == 0 && d->ci == 32 && d->co == 32 &&
                   !(d->flags & IMM_CONV_MASK);
  if ((!k33 && !k71) || d->stride != 1 || d->updiv != 1) return false;
  if (d->ci != 32 && d->ci != 64) return false;
  if (d->co != 32 && d->co != 64) return false;
  if (d->kpad != d->kh * d->kw * d->ci) return false;
  if (d->out_scale > 1 || (d->flags & (IMM_CONV_OUT_F32 | 0xf00))) return false;
  if (d->hi != d->ho || d->wi != d->wo || d->ho % H2_PH || d->wo % H2_PW) return false;
  if (d->ho * d->wo < 64 * 64) return false;
  if (d->ldy % 8) return false;
  if ((d->flags & IMM_CONV_MASK) && (d->ci != d->co || d->ldmask % 8)) return false;
  const int64_t px = (int64_t)d->batch * d->hi * d->wi;
  return px * d->ldx * 2 < (1LL << 31) && px * d->ldy * 2 < (1LL << 31) && px * (int64_t)d->ldmask * 2 < (1LL << 31);
}

// IMM_HALO2X=1: the 64 -> 64 launches without batch-norm partial sums take conv_halo2x_kernel (32x32x16 tiles).  OFF by default:
// measured 79-81 us against 75-78 for VGG conv1_2 (profiles/r06_halo2x_ablation.txt)
bool imm_halo2_x32(const imm_conv_desc* d) {
  static const bool on = getenv("IMM_HALO2X") && getenv("IMM_HALO2X")[0] == '1';
  return on && imm_halo2_applicable(d) && d->ci == 64 && d->co == 64 && d->kh == 3 && d->kw == 3 && !(d->flags & IMM_CONV_STATS);
}

static size_t h2_lds(int ci, int bn, bool mask, int ns, int kh = 3, int kw = 3) {
  return (size_t)ns * ((size_t)h2_halo_slots(ci, kh, kw) * (ci / 8) + (mask ? (size_t)H2_PH * H2_PW * (bn / 8) : 0)) * 16;
}

template <typename ET, int CI, int BN, bool MASK, bool STATS, int NS, int KH = 3, int KW = 3>
static int h2_occupancy() {
  static int occ = 0;
  if (occ == 0) {
    const size_t lds = h2_lds(CI, BN, MASK, NS, KH, KW);
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)conv_halo2_kernel<ET, CI, BN, MASK, STATS, NS, KH, KW>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv_halo2_kernel<ET, CI, BN, MASK, STATS, NS, KH, KW>, 256, lds) != hipSuccess || n < 1) {
      (void)hipGetLastError();
      n = 1;
    }
    occ = n > 4 ? 4 : n;
  }
  return occ;
}

// dispatch over the instantiated (ci, co, mask, stats, ring depth, filter shape) combinations
template <typename F>
static void h2_dispatch(const imm_conv_desc* d, F&& f) {
  const int ci = d->ci, co = d->co;
  const bool mask = d->flags & IMM_CONV_MASK, stats = d->flags & IMM_CONV_STATS;
  const int ns = h2_ns(mask);
  if (d->kh == 7) {   // 7x1 first layer: 32 -> 32 channels
    if (stats) f(H2Int<32>(), H2Int<32>(), H2Int<0>(), H2Int<1>(), H2Int<4>(), H2Int<7>(), H2Int<1>());
    else f(H2Int<32>(), H2Int<32>(), H2Int<0>(), H2Int<0>(), H2Int<4>(), H2Int<7>(), H2Int<1>());
    return;
  }
  auto with_ns = [&](auto c, auto b, auto st) {
    if (ns == 3) f(c, b, H2Int<0>(), st, H2Int<3>(), H2Int<3>(), H2Int<3>());
    else f(c, b, H2Int<0>(), st, H2Int<4>(), H2Int<3>(), H2Int<3>());
  };
  auto with_stats = [&](auto c, auto b) {
    if (stats) with_ns(c, b, H2Int<1>()); else with_ns(c, b, H2Int<0>());
  };
  if (mask) {      // ci == co in {32, 64}: ReLU-backward mask (VGG conv1_2) and, with stats, the batch-norm backward sums
    auto with_mask = [&](auto c, auto st) {
      if (ns == 3) f(c, c, H2Int<1>(), st, H2Int<3>(), H2Int<3>(), H2Int<3>());
      else f(c, c, H2Int<1>(), st, H2Int<4>(), H2Int<3>(), H2Int<3>());
    };
    if (ci == 64) { if (stats) with_mask(H2Int<64>(), H2Int<1>()); else with_mask(H2Int<64>(), H2Int<0>()); }
    else { if (stats) with_mask(H2Int<32>(), H2Int<1>()); else with_mask(H2Int<32>(), H2Int<0>()); }
  }
  else if (ci == 64 && co == 64) with_stats(H2Int<64>(), H2Int<64>());
  else if (ci == 64) with_stats(H2Int<64>(), H2Int<32>());
  else if (co == 64) with_stats(H2Int<32>(), H2Int<64>());
  else with_stats(H2Int<32>(), H2Int<32>());
}

// number of persistent workgroups (= rows of BN partial sums).  Queried on the bf16 instantiation: both element types
// have the same register / LDS footprint, and the count must not depend on the dtype of a later launch.
int imm_halo2_grid(const imm_conv_desc* d) {
  const int n_patches = d->batch * (d->ho / H2_PH) * (d->wo / H2_PW);
  int occ = 1;
  h2_dispatch(d, [&](auto ci, auto bn, auto mk, auto st, auto ns, auto kh, auto kw) {
    occ = h2_occupancy<BF16, decltype(ci)::value, decltype(bn)::value, (bool)decltype(mk)::value, (bool)decltype(st)::value,
                       decltype(ns)::value, decltype(kh)::value, decltype(kw)::value>();
  });
  const int grid = h2_num_cu() * occ;
  return n_patches < grid ? n_patches : grid;
}

template <typename ET>
static void h2_launch(const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  Halo2Args ha;
  ha.c = a;
  ha.patches_x = d->wo / H2_PW; ha.patches_y = d->ho / H2_PH;
  ha.n_patches = d->batch * ha.patches_x * ha.patches_y;
  const int64_t px = (int64_t)d->batch * d->hi * d->wi;
  ha.c.x_bytes = (uint32_t)(px * d->ldx * 2);
  ha.c.wt_bytes = (uint32_t)((int64_t)d->co * d->kpad * 2);
  ha.y_bytes = (uint32_t)(px * d->ldy * 2);
  ha.mask_bytes = (uint32_t)(px * d->ldmask * 2);
  const int per_img = ha.patches_x * ha.patches_y;
  ha.lg_px = ha.lg_pi = -1;
  if ((ha.patches_x & (ha.patches_x - 1)) == 0 && (per_img & (per_img - 1)) == 0) {
    ha.lg_px = __builtin_ctz(ha.patches_x); ha.lg_pi = __builtin_ctz(per_img);
  }
  const int grid = imm_halo2_grid(d);
  if (imm_halo2_x32(d)) {     // 64 -> 64 channels without batch-norm sums: the 32x32x16 form (one workgroup per CU either way)
    const bool mask = d->flags & IMM_CONV_MASK;
    const size_t lds = h2_lds(64, 64, mask, h2_ns(mask));
    static bool attr_set[2] = {false, false};       // (per element type: this function is a template)
    if (!attr_set[mask]) {
      if (mask) (void)hipFuncSetAttribute((const void*)conv_halo2x_kernel<ET, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      else (void)hipFuncSetAttribute((const void*)conv_halo2x_kernel<ET, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set[mask] = true;
    }
    if (mask) hipLaunchKernelGGL((conv_halo2x_kernel<ET, true, 3>), dim3(grid), dim3(256), lds, s, ha);
    else hipLaunchKernelGGL((conv_halo2x_kernel<ET, false, 4>), dim3(grid), dim3(256), lds, s, ha);
    return;
  }
  h2_dispatch(d, [&](auto ci, auto bn, auto mk, auto st, auto ns, auto kh, auto kw) {
    constexpr int CI = decltype(ci)::value, BN = decltype(bn)::value, NS = decltype(ns)::value;
    constexpr int KH = decltype(kh)::value, KW = decltype(kw)::value;
    constexpr bool MASK = decltype(mk)::value, STATS = decltype(st)::value;
    (void)h2_occupancy<ET, CI, BN, MASK, STATS, NS, KH, KW>();  // sets the dynamic-LDS attribute on first use
    hipLaunchKernelGGL((conv_halo2_kernel<ET, CI, BN, MASK, STATS, NS, KH, KW>), dim3(grid), dim3(256),
                       h2_lds(CI, BN, MASK, NS, KH, KW), s, ha);
  });
}

void imm_conv_halo2_launch(int dtype, const imm_conv_desc* d, const ConvArgs& a, hipStream_t s) {
  if (dtype == IMM_BF16) h2_launch<BF16>(d, a, s);
  else h2_launch<F16>(d, a, s);
}
