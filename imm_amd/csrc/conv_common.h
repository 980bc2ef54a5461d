// conv_common.h — argument block and epilogue shared by the implicit-GEMM convolution kernels.
#pragma once
#include "common.h"

struct ConvArgs {
  const uint16_t* x;
  const uint16_t* wt;
  const float* bias;
  void* y;
  float* stats;
  const uint16_t* mask;
  int M, hi, wi, ci8, ldx;
  int ho, wo, co, ldy;
  int kh, kw, stride, pad_t, pad_l, updiv;
  int kpad, KT, ntaps;
  int flags, ldmask;
  int n_nblk, n_blocks;
  int oscale, ooff_y, ooff_x;   // output scatter (oscale 1 = dense)
  uint32_t x_bytes, wt_bytes;   // buffer-descriptor extents (fast path: out-of-range lanes read zeros)
  // perceptual tap folded into a data gradient's epilogue (IMM_CONV_TAP_, conv_hdeep.hip only; imm_conv2d_tap)
  const uint16_t* tap_gt;       // gt half of the tapped activation (geometry of y / mask)
  const float* tap_lmask;       // loss mask f32 [batch, tap_S, tap_S] or NULL
  const float* tap_coef;        // device table of gradient coefficients
  int tap_idx, tap_S, tap_l1;
};
#define IMM_CONV_TAP_ 0x20      // internal flag bit (set by imm_conv2d_tap only)


// Several convolutions of the same tile shape in ONE launch (conv_igemm64.hip / conv_igemm.hip group kernels): workgroup b
// belongs to member g with first[g] <= b < first[g+1] and runs that member's argument block unchanged.
#define IMM_CONV_GROUP_MAX 4
struct ConvArgsGroup {
  ConvArgs a[IMM_CONV_GROUP_MAX];
  int first[IMM_CONV_GROUP_MAX + 1];
  int n;
};

// Epilogue of one BM x BN tile: bias, ReLU, ReLU-backward mask, NHWC store (lane = 4 consecutive channels of one
// pixel), deterministic per-M-block batch-norm partial sums.  `red` = >= WGM*2*BN floats of LDS, free to overwrite
// (all waves are past their last LDS read).
template <typename ET, int BM, int BN, int WGM, int WGN, int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4_t (&acc)[MT][NT], int tid, int wm, int wn, int m0,
                                              int n0, int mblk, float* red) {
  constexpr int TM = BM / WGM, TN = BN / WGN;
  const int lane = tid & 63;
  // ---- epilogue ------------------------------------------------------------------------------
  if (a.flags & IMM_DBG_NO_EPILOGUE) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 123456.789f) ((float*)a.y)[0] = t;   // keeps the accumulators live
    return;
  }
  // lane holds D[n = 4*(lane>>4)+r][m = lane&15] of each 16x16 tile
  const bool f_bias = a.flags & IMM_CONV_BIAS, f_relu = a.flags & IMM_CONV_RELU;
  const bool f_stats = a.flags & IMM_CONV_STATS, f_mask = a.flags & IMM_CONV_MASK;
  const bool f_f32 = a.flags & IMM_CONV_OUT_F32;
  float s1[NT][4], s2[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[j][r] = 0.f; s2[j][r] = 0.f; }

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * TN + j * 16 + 4 * (lane >> 4);
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (f_bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n + r < a.co) bv[r] = a.bias[n + r];
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int mi = m0 + wm * TM + i * 16 + (lane & 15);    // pixel index of this launch (drives validity, stats)
      int64_t m = mi;                                         // pixel index in the output tensor
      if (a.oscale != 1 && mi < a.M) {
        const int hw = a.ho * a.wo;
        const int img = mi / hw, rem = mi - img * hw;
        const int oy = rem / a.wo, ox = rem - oy * a.wo;
        m = ((int64_t)img * a.ho * a.oscale + oy * a.oscale + a.ooff_y) * (a.wo * a.oscale) + ox * a.oscale + a.ooff_x;
      }
      const bool mok = mi < a.M;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = acc[i][j][r] + bv[r];
        if (f_relu) v[r] = fmaxf(v[r], 0.f);
      }
      // with IMM_CONV_STATS | IMM_CONV_MASK the second partial sum is sum(v * mask_ref) instead of sum(v^2): the
      // batch-norm backward sums (sum dz, sum dz * out) of the layer whose output gradient this data gradient is
      float mv[4] = {0.f, 0.f, 0.f, 0.f};
      if (f_mask && mok) {
        const uint16_t* mp = a.mask + (int64_t)m * a.ldmask + n;
        if (n + 3 < a.co && (a.ldmask & 3) == 0) {
          const uint2 mw = *(const uint2*)mp;
          const uint16_t mh[4] = {(uint16_t)(mw.x & 0xffffu), (uint16_t)(mw.x >> 16), (uint16_t)(mw.y & 0xffffu), (uint16_t)(mw.y >> 16)};
#pragma unroll
          for (int r = 0; r < 4; ++r) { mv[r] = ET::to_f32(mh[r]); if (!(mv[r] > 0.f)) v[r] = 0.f; }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.co) { mv[r] = ET::to_f32(mp[r]); if (!(mv[r] > 0.f)) v[r] = 0.f; }
        }
      }
      if (f_stats && mok) {
        if (f_mask) {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * mv[r]; }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[j][r] += v[r]; s2[j][r] += v[r] * v[r]; }
        }
      }
      if (mok) {
        if (f_f32) {
          float* yp = (float*)a.y + (int64_t)m * a.ldy + n;
          if (n + 3 < a.co) {
            *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.co) yp[r] = v[r];
          }
        } else {
          uint16_t* yp = (uint16_t*)a.y + (int64_t)m * a.ldy + n;
          if (n + 3 < a.co) {
            *(uint2*)yp = make_uint2(ET::pack2(v[0], v[1]), ET::pack2(v[2], v[3]));
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < a.co) yp[r] = ET::from_f32(v[r]);
          }
        }
      }
    }
  }

  if (f_stats) {
    // reduce over the 16 pixel-lanes of each channel quad, then over the WGM wave rows through LDS
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
    if (tid < BN && n0 + tid < a.co) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < WGM; ++w) { t1 += red[(w * 2 + 0) * BN + tid]; t2 += red[(w * 2 + 1) * BN + tid]; }
      a.stats[((int64_t)mblk * 2 + 0) * a.co + n0 + tid] = t1;
      a.stats[((int64_t)mblk * 2 + 1) * a.co + n0 + tid] = t2;
    }
  }
}
