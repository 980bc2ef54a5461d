/* imm_hip.h — C-ABI of libimm_hip.so: the MI355X (gfx950) kernels behind the IMM training step.
 *
 * The reference (tomasjakab/imm) has NO native layer and NO FFI: every number on its hot path comes
 * from stock TensorFlow-1.10 ops called from Python.  This header therefore declares one entry
 * point per TF op call site on the path (SURVEY.md §2a op table); each declaration cites the
 * reference call site it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; nothing allocates, nothing
 *     synchronises, nothing throws; every launch goes to the caller's hipStream_t (passed as void*).
 *   - return value: 0 = ok, <0 = error (IMM_E_*); imm_last_error() returns a thread-local message.
 *   - activations are NHWC with an explicit pixel stride `ld*` (elements), 16-bit storage
 *     (imm_dtype BF16 or F16) unless a parameter says f32; channel counts that feed a
 *     convolution are multiples of 8 and padding channels hold zeros.
 *   - statistics, parameters, gradients of parameters, optimizer state: f32.
 */
#ifndef IMM_HIP_H
#define IMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMM_ABI_VERSION 21   /* 21: imm_masked_sse_all.  20: imm_vgg_head_fwd / imm_vgg_head_supported / imm_vgg_head_scratch_bytes (conv1_1 + conv1_2 in one launch).  19: imm_set_cu_limit / imm_get_cu_limit, imm_masked_sse_pool with pool_a == NULL.  18: imm_copy_f32.  17: imm_cost_ema, imm_rms16 (summaries).  16: imm_conv2d_variant.  15: imm_conv2d_dgrad_s2, imm_conv2d_nol, imm_conv_first, imm_wgrad_job.x_scale/x_shift/x_relu; entry points removed
                                  since 14 (imm_bn_bwd_reduce_finalize, imm_conv2d_stats_workspace_bytes) finally counted */

/* IMM_F32 (round 6): f32 activation storage — the exact-arithmetic WITNESS of the wiring, a test instrument (the reference computes
 * in fp32, imm_model.py:97): accepted by the entry points that say so below; convolutions then run plain f32 FMA kernels
 * (csrc/conv_f32.hip: no tuning, ~100x slower than the 16-bit MFMA kernels). */
enum imm_dtype { IMM_BF16 = 0, IMM_F16 = 1, IMM_F32 = 2 };

enum imm_error {
  IMM_OK = 0,
  IMM_E_INVALID = -1,      /* bad argument (null pointer, misaligned stride, non-positive dim) */
  IMM_E_UNSUPPORTED = -2,  /* shape / mode not implemented by these kernels */
  IMM_E_HIP = -3           /* a HIP runtime call failed */
};

/* convolution epilogue / mode flags */
#define IMM_CONV_BIAS 1      /* add f32 bias[co]                                               */
#define IMM_CONV_RELU 2      /* max(.,0)                                                       */
#define IMM_CONV_STATS 4     /* write per-M-block partial sum / sum-of-squares for batch norm   */
#define IMM_CONV_MASK 8      /* multiply by (mask_ref[m][n] > 0): ReLU backward fused in dgrad  */
#define IMM_CONV_OUT_F32 16  /* y is f32 instead of 16-bit                                      */

/* Geometry of one implicit-GEMM convolution launch.  Forward (tf.nn.conv2d SAME,
 * imm/tf_utils/nn_utils.py:100; imm/models/selfsup/vgg16.py:182) uses stride>=1, updiv=1.
 * Data-gradient (tf.gradients of the same op) is expressed as a convolution of dy with the
 * flipped/transposed packed weights: stride=1, updiv=<forward stride>, pad = k-1-pad_fwd. */
typedef struct imm_conv_desc {
  int32_t batch, hi, wi, ci;   /* input: ci = channels entering the K loop (multiple of 8)        */
  int32_t ldx;                 /* input pixel stride (elements, multiple of 8)                    */
  int32_t ho, wo, co;          /* output: co = real output channels                               */
  int32_t ldy;                 /* output pixel stride (elements, multiple of 4)                   */
  int32_t kh, kw, stride, pad_t, pad_l;
  int32_t updiv;               /* 1, or 2 for the transposed (stride-2 dgrad) gather              */
  int32_t kpad;                /* packed-weight row length: round_up(kh*kw*ci, 32)                */
  int32_t flags;               /* IMM_CONV_*                                                      */
  int32_t ldmask;              /* pixel stride of mask_ref (IMM_CONV_MASK)                        */
  /* output scatter (0/0/0 = dense): output pixel (y,x) of this launch is written to pixel
   * (y*out_scale + out_off_y, x*out_scale + out_off_x) of a [batch, ho*out_scale, wo*out_scale] tensor.
   * Used by the parity-class decomposition of the stride-2 data gradient (4 launches, out_scale = 2). */
  int32_t out_scale, out_off_y, out_off_x;
} imm_conv_desc;

/* ---- runtime ------------------------------------------------------------------------------- */
int imm_abi_version(void);
const char* imm_last_error(void);
/* sha256 (hex) of the sources this library was built from (every .hip and .h file under csrc/ plus this header, by repo-relative
 * path + content): the host binding compares it with the checkout and refuses (or rebuilds) a stale binary */
const char* imm_source_digest(void);
/* device properties the host needs to size launches: [0]=CU count, [1]=gfx arch number (950) */
int imm_device_info(int32_t* out2_host);
/* Confine the launches THIS THREAD issues from now on to a share of the chip: the persistent / chip-sized convolution kernels
 * (conv_halo, conv_halo2, conv_hdeep, conv_hdeep6) size their grids — and choose their tiles — as if the device had `cus` compute
 * units (a multiple of 8: one share per XCD; 0 = the whole device, the default).  The other CUs stay free for the kernels of a
 * concurrent stream: the weight-independent ground-truth half of the frozen VGG16 forward (concat([gt, pred]),
 * imm/models/imm_model.py:126) runs beside the latency-bound encoder chains this way (imm_amd/engine.py).  Results do not depend
 * on the limit except through the tile choice (accumulation order).  Row-count queries (imm_conv_stats_blocks ...) follow the
 * limit in force when they are called: query and launch under the same one (launches with IMM_CONV_STATS are never confined by
 * the engine).  Thread-local, like imm_last_error. */
int imm_set_cu_limit(int cus);
int imm_get_cu_limit(void);
/* timeline probe: slots[index] = the device's constant-rate wall clock (100 MHz ticks) when this one-thread kernel runs on
 * `stream`.  Placed between the launches of a captured program it gives a profiler-free timeline of a HIP-graph replay
 * (IMM_DEBUG_STAMPS=1 in imm_amd/engine.py). */
int imm_debug_stamp(uint64_t* slots, int index, void* stream);
/* dst[0..n) = src[0..n) by a kernel on `stream`; either side may be device memory or PINNED host memory (4-byte aligned; 16-byte
 * aligned pairs are copied 16 bytes at a time).  Used by
 * the host-bounced gradient exchange of the multi-rank TESTS (torch.distributed backend "gloo": several ranks on one GPU) instead of
 * hipMemcpyAsync, whose copy-engine writes are not seen by lines still valid in the L2 (imm_amd/train/cnn_train_multi.py). */
int imm_copy_f32(float* dst, const float* src, int64_t n, void* stream);
/* HIP-graph capture of a launch sequence on `stream` (replaces TF's session.run of a static graph,
 * imm/train/cnn_train_multi.py:459). */
int imm_graph_begin(void* stream);
int imm_graph_end(void* stream, void** graph_exec_out_host);
int imm_graph_launch(void* graph_exec, void* stream);
int imm_graph_destroy(void* graph_exec);

/* ---- weights ------------------------------------------------------------------------------- */
/* f32 HWIO master [kh,kw,ci_real,co_real] -> 16-bit packed Wt[rows][kpad] (k contiguous).
 * mode 0 (forward): row n = co, k = (ky*kw+kx)*ci_pad + c.
 * mode 1 (dgrad):   row n = ci, k = (ky'*kw+kx')*co_pad + c', value W[kh-1-ky'][kw-1-kx'][n][c'].
 * mode 4+2*py+px (stride-2 dgrad, input-pixel parity class (py,px); kh,kw = size of the FULL filter):
 *                   sub-filter of ny x nx taps, ny = (kh-py+1)/2, nx = (kw-px+1)/2; row n = ci,
 *                   k = (jy*nx+jx)*co_pad + c', value W[py+2(ny-1-jy)][px+2(nx-1-jx)][n][c'].
 * rows >= real count and padded k are written as zeros; `rows` is the allocated row count. */
int imm_pack_weights(const float* w, void* wt, int dtype, int mode, int kh, int kw, int ci_real, int co_real,
                     int c_pad, int rows, int kpad, void* stream);

/* Table-driven variants: ONE launch re-packs / reduces every tensor of a step.  jobs: device int64[n_jobs][12];
 * pack job  = {w, wt, mode, kh, kw, ci_real, co_real, c_pad, rows, kpad, 0, 0}  (imm_pack_weights_multi_blocks workgroups:
 *             mode 0 = 32 x 64 tiles transposed through LDS, the other modes 256*8 consecutive elements per workgroup)
 * reduce job = {slab, dw, nsplit, kh*kw, ci_pad, ci_real, co, kpad, lanes, 0, 0, 0}   (1024/lanes outputs per workgroup; lanes =
 *             split lanes that share the slabs of an output, a power of two <= 16, 0 = 16: choose it about nsplit)
 * blk_first: device int32[n_jobs+1] prefix sums of workgroups per job; n_blocks = blk_first[n_jobs]. */
int imm_pack_weights_multi_blocks(int mode, int rows, int kpad);
int imm_pack_weights_multi(const int64_t* jobs, const int32_t* blk_first, int n_jobs, int n_blocks, int dtype,
                           void* stream);
int imm_wgrad_reduce_multi(const int64_t* jobs, const int32_t* blk_first, int n_jobs, int n_blocks, void* stream);

/* ---- convolution (implicit GEMM on MFMA) ----------------------------------------------------- */
/* n <= 4 convolutions that read the same x and write (disjoint pixels of) the same y, issued as ONE launch when every member
 * takes the deep-K 64x64-tile kernel, else one after the other (identical results): the four parity classes of a stride-2
 * data gradient (tf.nn.conv2d_backprop_input at nn_utils.py:100 call sites with stride 2).  Members carry no bias;
 * descs is an array of n descriptors, wts an array of n packed filter images.  With IMM_CONV_STATS | IMM_CONV_MASK on the
 * members, stats_partial holds imm_conv2d_group_stats_blocks(descs, n) rows of [2][co]: member i's rows follow member
 * i-1's (mask_ref has the geometry of y). */
int imm_conv2d_group(const imm_conv_desc* descs, int n, int dtype, const void* x, const void* const* wts, void* y,
                     float* stats_partial, const void* mask_ref, void* stream);
int imm_conv2d_group_stats_blocks(const imm_conv_desc* descs, int n);
/* Normalise on load: the convolution of a conv + batch-norm + ReLU block's output WITHOUT that output ever being stored
 * (tf.layers.batch_normalization + tf.nn.relu feeding the next tf.nn.conv2d, nn_utils.py:201-209 -> :100).  x_raw is the 16-bit
 * output y of the PRECEDING convolution; this convolution reads relu(x_scale[c] * y + x_shift[c]) (x_relu != 0; zero padding is of
 * the normalised tensor), the affine + ReLU being applied once per halo pixel in LDS after the tile's DMA.  x_scale / x_shift:
 * f32 [ci] as written by imm_bn_finalize.  Everything else as imm_conv2d (bias, IMM_CONV_STATS partial rows — their count is
 * imm_conv2d_nol_stats_blocks(desc), not imm_conv_stats_blocks).  Served by the LDS-halo kernel of conv_halo.hip: 3x3, ci in
 * {32, 64}, co <= 64, stride 1 at >= 64x64 maps, or the 32 -> 64-channel stride-2 form; imm_conv2d_nol_supported(desc) = 1 / 0.
 * Results equal imm_bn_apply_relu + imm_conv2d up to the convolution's accumulation order. */
int imm_conv2d_nol_supported(const imm_conv_desc* desc_host);
int imm_conv2d_nol_stats_blocks(const imm_conv_desc* desc_host);
int imm_conv2d_nol(const imm_conv_desc* desc_host, int dtype, const void* x_raw, const float* x_scale, const float* x_shift,
                   int x_relu, const void* wt, const float* bias, void* y, float* stats_partial, void* stream);
/* The data gradient of a 3x3 STRIDE-2 SAME convolution (encoder conv_3 / conv_5 / conv_7, imm_model.py:197,204,211:
 * tf.nn.conv2d_backprop_input at those nn_utils.py:100 call sites) as ONE launch over ONE halo of dy: input pixel (2i+py, 2j+px)
 * receives only the taps of its parity class (4 / 2 / 2 / 1 of the nine); a workgroup keeps four accumulator sets for an 8x16
 * patch of class pixels and scatters them to the 16x32 patch of dx.  Replaces the four imm_conv2d_group class launches (same
 * values up to accumulation order).  dy 16-bit [batch,h,w], pixel stride lddy = the channels entering the K loop (multiple of
 * 64; channels beyond the real count hold zeros); wt = the flipped image of imm_pack_weights mode 1 (kh = kw = 3, c_pad = lddy:
 * rows >= c_dx, row length kpad = 9 * lddy); dx 16-bit [batch,2h,2w], pixel stride lddx, channels [0, c_dx) written
 * (c_dx % 8 == 0).  h % 8 == 0, w % 16 == 0.  imm_conv2d_dgrad_s2_supported: 1 when the shape is served, else 0. */
int imm_conv2d_dgrad_s2_supported(int batch, int h, int w, int lddy, int c_dx, int lddx);
int imm_conv2d_dgrad_s2(const void* dy, int lddy, const void* wt, int kpad, void* dx, int lddx, int c_dx, int dtype, int batch,
                        int h, int w, void* stream);
/* y[m][n] = epilogue( sum_k gather(x)[m][k] * wt[n][k] ).  stats_partial: [n_mblocks][2][co] f32
 * with n_mblocks = imm_conv_stats_blocks(desc).  Replaces tf.nn.conv2d+bias_add (nn_utils.py:100,108),
 * vgg conv+bias+relu (vgg16.py:182-189,230) and their data gradients.
 * Epilogue order: + bias, ReLU, mask (IMM_CONV_MASK: v = 0 where mask_ref <= 0, mask_ref 16-bit [pixels][ldmask] with the
 * geometry of y), store, partial sums.  IMM_CONV_STATS alone: rows of (sum v, sum v^2) — the batch statistics of a forward
 * convolution.  IMM_CONV_STATS | IMM_CONV_MASK: rows of (sum v, sum v * mask_ref) — the batch-norm BACKWARD sums (sum dz,
 * sum dz * out) of the conv+BN+ReLU block whose output gradient this data gradient produces (mask_ref = that block's stored
 * activation `out`), which replaces a separate reduction pass over dz (tf.gradients of nn_utils.py:201-209). */
int imm_conv2d(const imm_conv_desc* desc_host, int dtype, const void* x, const void* wt, const float* bias,
               void* y, float* stats_partial, const void* mask_ref, void* stream);
int imm_conv_stats_blocks(const imm_conv_desc* desc_host);
/* Which kernel imm_conv2d dispatches `desc` to: family * 100000 + tile variant (>= 100000; negative = invalid descriptor).
 * Families: 1 conv_igemm (im2col, BK = 32), 2 conv_igemm64 (im2col, BK = 64, LDS-DMA ring), 3 conv_halo (LDS halo, filter in LDS),
 * 4 conv_halo2 (LDS halo, filter in registers), 5 conv_hdeep (LDS halo of a 64-channel slice, filter taps streamed), 6 conv_hdeep6
 * (the 16x16x128 tile on 32-channel slices, six k-steps per barrier), 7 conv_s2f (3x3 stride-2 forward, parity-de-interleaved LDS
 * halo).  The twin of imm_conv2d_wgrad_variant: tests assert that the
 * kernel a case is named after is the one that ran; tools/layer_table.py prints it per layer.  Host-only, no launch. */
int imm_conv2d_variant(const imm_conv_desc* desc_host, int dtype);
/* imm_conv2d of a data gradient (no bias / ReLU / statistics) that ENTERS a tapped activation of the frozen VGG16 (conv3_2, conv4_2 of
 * imm_model.py:124-147), with the perceptual tap folded into its epilogue:
 *     y = [a_pred > 0] * ( round16(conv) + coef[idx] * loss_mask[pixel] * (a_pred - a_gt) )        (sign(a_pred - a_gt) with l1)
 * = imm_conv2d followed by imm_tap_grad(has_in = 1, relu = 1), bit for bit, without the extra pass over y.  a_pred / a_gt: the two
 * halves of the tapped activation, geometry of y, pixel stride lda; loss_mask f32 [batch,S,S] or NULL.  Served by the LDS-halo
 * deep-K kernel only: imm_conv2d_tap_supported(desc) != 0, else IMM_E_UNSUPPORTED (issue the two calls instead). */
int imm_conv2d_tap_supported(const imm_conv_desc* desc_host);
int imm_conv2d_tap(const imm_conv_desc* desc_host, int dtype, const void* x, const void* wt, void* y, const void* a_pred,
                   const void* a_gt, int lda, const float* loss_mask, int S, const float* coef, int idx, int l1, void* stream);
/* Filter gradient: slab[split][kpad][co] f32 partials over `nsplit` pixel ranges, then
 * imm_conv2d_wgrad_reduce sums the slabs into the HWIO f32 gradient [kh,kw,ci_real,co]
 * (tf.gradients of nn_utils.py:100 w.r.t. `w`).  desc describes the FORWARD convolution. */
int imm_conv2d_wgrad(const imm_conv_desc* desc_host, int dtype, const void* x, const void* dy, int lddy,
                     float* slab, int nsplit, void* stream);
/* > 0: the split count at which the LDS-resident-tile variant (3x3 s1, ci in {32,64}, co<=64, >=64x64) runs --
 * allocate that many slabs and pass it as nsplit; 0: any nsplit >= 1 (general kernel). */
int imm_conv2d_wgrad_splits(const imm_conv_desc* desc_host, int lddy);
int imm_conv2d_wgrad_reduce(const float* slab, int nsplit, int kh, int kw, int ci_pad, int ci_real, int co,
                            int kpad, float* dw, void* stream);
/* The filter gradients of MANY layers in as few launches as there are kernel variants among them (tf.gradients w.r.t. every
 * `w` of the towers' graph, cnn_train_multi.py:157,231: TF schedules them freely too — nobody reads a filter gradient before
 * apply_gradients).  A job = the arguments of one imm_conv2d_wgrad call.  imm_conv2d_wgrad_multi_plan writes a table of
 * imm_conv2d_wgrad_multi_table_bytes(n) bytes into HOST memory (launch list + the members' argument blocks, longest
 * workgroups first); the caller copies it once to device memory (16-byte aligned) and passes both copies to every launch:
 * the host copy carries the launch geometry, the device copy is what the kernels read.  Results are identical to n single
 * calls with the same split counts.  n <= 64.
 * imm_conv2d_wgrad_variant: the kernel (family * 100000 + tile variant) a layer's filter gradient runs with, its workgroups
 * per pixel split, its length in units (32-pixel steps or 8x16-pixel patches) and how many workgroups of that kernel are
 * resident per CU — what a caller needs to choose split counts for jobs that share a launch (jobs with equal return values
 * do; one round of resident, equally long workgroups is the target). */
typedef struct imm_wgrad_job {
  imm_conv_desc desc;          /* the FORWARD convolution */
  const void* x; const void* dy; float* slab;
  int32_t lddy, nsplit;
  /* normalise on load (imm_conv2d_nol below): x is the RAW output of the conv + batch-norm + ReLU block in front of this
   * convolution and the filter gradient is taken against relu(x_scale[c] * x + x_shift[c]) (x_relu != 0), rebuilt in LDS.
   * NULL / NULL / 0 = x is the stored activation.  Only for jobs whose variant is an LDS-halo kernel
   * (imm_conv2d_wgrad_variant(...) / 100000 == 2); imm_conv2d_wgrad_multi_plan refuses others. */
  const float* x_scale; const float* x_shift;
  int32_t x_relu, reserved;
} imm_wgrad_job;
int64_t imm_conv2d_wgrad_multi_table_bytes(int n);
int imm_conv2d_wgrad_variant(const imm_conv_desc* desc_host, int lddy, int dtype, int* wg_per_split_host, int* units_host,
                             int* wg_per_cu_host);
int imm_conv2d_wgrad_multi_plan(const imm_wgrad_job* jobs_host, int n, int dtype, void* table_host);
int imm_conv2d_wgrad_multi(const void* table_host, const void* table_dev, void* stream);
/* db[n] = sum_m dy[m][n], n < c_out, for convolutions not followed by batch norm (bias_add gradient,
 * nn_utils.py:108); c = padded channel count summed per row (multiple of 8);
 * partial: [nblk][c] scratch with nblk = imm_colsum_blocks(npix, c). */
int imm_colsum(const void* dy, int dtype, int64_t npix, int c, int c_out, int ld, float* partial, float* out,
               void* stream);
int imm_colsum_blocks(int64_t npix, int c);

/* ---- batch norm (tf.layers.batch_normalization(fused=True) + relu, nn_utils.py:201-209) ------ */
/* partial [nblk][2][c] -> scale/shift for the apply pass, saved mean/rstd for backward, moving
 * averages updated with momentum 0.99 / unbiased variance when training != 0. */
int imm_bn_finalize(const float* partial, int nblk, int c, int64_t count, const float* gamma, const float* beta,
                    float eps, float momentum, int training, float* moving_mean, float* moving_var,
                    float* scale, float* shift, float* mean, float* rstd, void* stream);
int imm_bn_apply_relu(const void* y, int dtype, int64_t npix, int c, int ldy, const float* scale,
                      const float* shift, int relu, void* x_out, int ldx, void* stream);
/* imm_bn_finalize + imm_bn_apply_relu in ONE launch, for layers with few partial rows (nblk <= ~256; c % 32 == 0): every
 * workgroup owns a 32-channel slice x a pixel range and redoes the finalize of its slice from the rows (fixed order: the same
 * scale / shift in every workgroup), workgroup (0, slice) writes scale / shift / mean / rstd and the moving statistics.
 * up2x_out != NULL: the x2 bilinear up-sampling that follows the block in the renderer (imm_model.py:175) is written in the
 * same pass, [B, 2h, 2w] with pixel stride ldu, from the 16-bit values of x_out (bitwise = imm_upsample2x_fwd(x_out)); x_out may then
 * be NULL (only the up-sampled tensor is written: the renderer's next convolution and its filter gradient read that one). */
int imm_bn_apply_fused(const float* partial, int nblk, int c, int64_t count, const float* gamma, const float* beta, float eps,
                       float momentum, int training, float* moving_mean, float* moving_var, float* scale, float* shift,
                       float* mean, float* rstd, const void* y, int dtype, int ldy, int relu, void* x_out, int ldx,
                       void* up2x_out, int ldu, int h, int w, void* stream);
/* out[g][0..width) = sum of rows [g*group, (g+1)*group) of in[rows][width] (f64 accumulation, row order): brings the partial rows
 * of a large layer (conv statistics, imm_bn_bwd_reduce) down to ceil(rows/group) rows that the fused apply passes finish. */
int imm_rows_reduce(const float* in, int rows, int width, int group, float* out, void* stream);
/* backward: reduce -> finalize (writes dgamma, dbeta, coef[3][c]) -> apply (dy_conv = ...) */
int imm_bn_bwd_reduce(const void* dout, int lddo, const void* y, int ldy, int dtype, int64_t npix, int c,
                      const float* scale, const float* shift, const float* mean, const float* rstd, int relu,
                      float* partial, void* stream);
int imm_bn_bwd_blocks(int64_t npix, int c);
/* imm_upsample2x_bwd followed by imm_bn_bwd_reduce in ONE pass, for the renderer blocks whose output is up-sampled (conv_2/4/6,
 * imm_model.py:166-175): dy_up 16-bit [batch,2h,2w] (stride lddy) = gradient of the up-sampled tensor; its adjoint is written
 * 16-bit to dprev [batch,h,w] (stride lddp; bitwise = imm_upsample2x_bwd) for the apply pass and summed into `partial`
 * (imm_bn_bwd_blocks(batch*h*w, c) rows) as imm_bn_bwd_reduce(dprev, ...) would. */
int imm_bn_bwd_reduce_up(const void* dy_up, int lddy, void* dprev, int lddp, const void* y, int ldy, int dtype, int batch, int h,
                         int w, int c, const float* scale, const float* shift, const float* mean, const float* rstd, int relu,
                         float* partial, void* stream);
/* partial rows [2][ldp] (ldp >= c: rows written by a producer with more channels than this layer).  from_out = 0: rows of
 * (sum dz, sum dz*xhat) from imm_bn_bwd_reduce; from_out = 1: rows of (sum dz, sum dz*out) from a data-gradient epilogue
 * (IMM_CONV_STATS | IMM_CONV_MASK) or imm_upsample2x_bwd_bn: sum dz*xhat = (sum dz*out - beta * sum dz) / gamma. */
int imm_bn_bwd_finalize(const float* partial, int nblk, int c, int ldp, int64_t count, const float* gamma, const float* beta,
                        const float* rstd, int from_out, float* dgamma, float* dbeta, float* coef, void* stream);
int imm_bn_bwd_apply(const void* dout, int lddo, const void* y, int ldy, int dtype, int64_t npix, int c,
                     const float* scale, const float* shift, const float* mean, const float* rstd, int relu,
                     const float* coef, void* dy_out, int lddy, void* stream);
/* imm_bn_bwd_finalize (rows of imm_bn_bwd_reduce, [2][c]) + imm_bn_bwd_apply in ONE launch (same scheme as imm_bn_apply_fused) */
int imm_bn_bwd_apply_fused(const float* partial, int nblk, int c, int64_t count, const float* gamma, const void* dout, int lddo,
                           const void* y, int ldy, int dtype, const float* scale, const float* shift, const float* mean,
                           const float* rstd, int relu, float* dgamma, float* dbeta, void* dy_out, int lddy, void* stream);

/* ---- resampling / pooling -------------------------------------------------------------------- */
/* tf.image.resize_images x2, bilinear, legacy align_corners=False (imm_model.py:175) and its adjoint */
int imm_upsample2x_fwd(const void* x, void* y, int dtype, int batch, int h, int w, int c, int ldx, int ldy,
                       void* stream);
int imm_upsample2x_bwd(const void* dy, void* dx, int dtype, int batch, int h, int w, int c, int lddy, int lddx,
                       void* stream);
/* the adjoint fused with the ReLU backward of the block that produced the up-sampled tensor and with the batch-norm
 * backward sums: dx = adjoint(dy) * [out > 0]; partial [nblk][2][c] = per-workgroup (sum dx, sum dx*out),
 * nblk = imm_upsample2x_bwd_bn_blocks(batch, h, w, c); out 16-bit [batch,h,w] with pixel stride ldo. */
int imm_upsample2x_bwd_bn(const void* dy, void* dx, int dtype, int batch, int h, int w, int c, int lddy, int lddx,
                          const void* out, int ldo, float* partial, void* stream);
int imm_upsample2x_bwd_bn_blocks(int batch, int h, int w, int c);
/* tf.image.resize_bilinear(align_corners=True) (imm_model.py:334) and its adjoint (dx must be zeroed
 * by the kernel itself: it is, via a gather formulation) */
int imm_resize_ac_fwd(const void* x, void* y, int dtype, int batch, int hi, int wi, int ho, int wo, int c,
                      int ldx, int ldy, void* stream);
int imm_resize_ac_bwd(const void* dy, void* dx, int dtype, int batch, int hi, int wi, int ho, int wo, int c,
                      int lddy, int lddx, void* stream);
/* tf.nn.max_pool 2x2/2 (selfsup/ops.py:16-26).  bwd routes to the first maximum in scan order and,
 * with relu_mask != 0, zeroes gradients where x <= 0 (ReLU backward of the producing conv). */
int imm_maxpool2_fwd(const void* x, void* y, int dtype, int batch, int h, int w, int c, void* stream);
int imm_maxpool2_bwd(const void* x, const void* dy, void* dx, int dtype, int batch, int h, int w, int c,
                     int relu_mask, void* stream);

/* ---- inputs ---------------------------------------------------------------------------------- */
/* f32 NHWC [npix,3] in [0,255] -> 16-bit [npix,8] (channels 3..7 zero): first encoder conv input */
int imm_pack_image(const float* src, void* dst, int dtype, int64_t npix, void* stream);
/* same source, horizontal taps unrolled into channels: dst[b,y,x,kx*3+ch] = src[b,y,x+kx-pad_l,ch] (zero outside
 * the image; channels >= 3*kw zero; pixel stride ld).  Turns the 7x7x3 first encoder convolution
 * (imm_model.py:190) into a 7x1 convolution over 32 channels with identical arithmetic. */
int imm_pack_image_taps(const float* src, void* dst, int dtype, int batch, int h, int w, int kw, int pad_l, int ld,
                        void* stream);

/* The first encoder convolution (7x7, 3 -> co <= 32 channels, stride 1, SAME + bias: imm_model.py:190 through nn_utils.py:100,108)
 * straight from the f32 image [batch,s,s,3]: the tap-unrolled operand tile of the 7x1 form above is built in LDS per 8x16-pixel
 * patch instead of being written to HBM by imm_pack_image_taps and read back.  wt = the SAME packed filter image the 7x1 form uses
 * (imm_pack_weights mode 0, kh 7, kw 1, ci_real 21, c_pad 32: row length kpad >= 224), so the arithmetic (16-bit operands, f32
 * accumulation) is unchanged.  y 16-bit [batch,s,s] with pixel stride ldy; flags: IMM_CONV_BIAS, IMM_CONV_STATS (partial rows
 * [imm_conv_first_stats_blocks(batch, s)][2][co]).  s % 16 == 0, co % 4 == 0.  imm_conv_first_supported: 1 / 0. */
int imm_conv_first_supported(int batch, int s, int co, int ldy);
int imm_conv_first_stats_blocks(int batch, int s);
int imm_conv_first(const float* image, const void* wt, int kpad, const float* bias, void* y, int ldy, float* stats_partial, int dtype,
                   int batch, int s, int co, int flags, void* stream);

/* ---- landmark bottleneck (imm_model.py:252-264 soft-argmax, :34-78 get_gaussian_maps) ---------- */
/* gauss_mode (config key gauss_mode, imm_model.py:48-72): 'rot' exp(-((y-mu_y)^2+(x-mu_x)^2)*inv_std^2) (every shipped
 * config), 'flat' exp(-(dist + 1e-5)^(1/4)), 'ankush' exp(-sqrt(1e-4+|mu_y-y|*inv_std)) * exp(-sqrt(1e-4+|mu_x-x|*inv_std)) */
#define IMM_GAUSS_ROT 0
#define IMM_GAUSS_FLAT 1
#define IMM_GAUSS_ANKUSH 2
/* heat f32 [B,h,w,ldh] (k<K) -> mu [B,K,2] (y,x), py [B,h,K], px [B,w,K];
 * gauss: 16-bit written at gauss_out[((b*s+y)*s+x)*ldg + k] for k<K. */
int imm_softargmax_gauss_fwd(const float* heat, int ldh, int batch, int h, int w, int k, float inv_std, int s,
                             float* mu, float* py, float* px, void* gauss_out, int ldg, int dtype, int gauss_mode,
                             void* stream);
/* dgauss 16-bit (same addressing as gauss_out) -> dheat 16-bit [B,h,w,lddh], channels >= K zeroed */
int imm_softargmax_gauss_bwd(const void* dgauss, int ldg, int dtype, int batch, int h, int w, int k, float inv_std,
                             int s, const float* mu, const float* py, const float* px, void* dheat, int lddh,
                             int gauss_mode, void* stream);
/* The pose head as ONE launch each way (imm_model.py:247-264: 1x1 convolution 8f -> K with bias, no batch norm / activation,
 * then the soft-argmax and the Gaussian render above).
 * fwd: feat 16-bit [B,h,w,ldf] (c channels, c % 32 == 0), wt = the forward-packed filter (imm_pack_weights mode 0: rows >=
 *      round_up(K,16), row length kpad >= c), bias f32[K] -> heat f32 [B,h,w,ldh] (channels < K written) and everything
 *      imm_softargmax_gauss_fwd writes.  K <= 64, h*w % 16 == 0.
 * bwd: everything imm_softargmax_gauss_bwd does (dheat 16-bit [B,h,w,lddh], lddh in {32, 64}: the filter gradient of the head
 *      still reads it) plus the head's data gradient dfeat 16-bit [B,h,w,lddf] = dheat x W^T (wt_dgrad = imm_pack_weights mode 1
 *      image, row length kpad_d >= lddh) and bias_partial f32 [B][K] = per-sample column sums of the stored dheat, to be summed
 *      over B by imm_wgrad_reduce_multi (job {bias_partial, db, B, 1, 1, 1, K, 1}).  Replaces imm_softargmax_gauss_bwd +
 *      imm_colsum + imm_conv2d(data gradient) of the head. */
int imm_pose_head_fwd(const void* feat, int ldf, int c, const void* wt, int kpad, const float* bias, int dtype, int batch, int h,
                      int w, int k, float inv_std, int s, float* heat, int ldh, float* mu, float* py, float* px, void* gauss_out,
                      int ldg, int gauss_mode, void* stream);
int imm_pose_head_bwd(const void* dgauss, int ldg, int dtype, int batch, int h, int w, int k, float inv_std, int s, const float* mu,
                      const float* py, const float* px, void* dheat, int lddh, int gauss_mode, const void* wt_dgrad, int kpad_d,
                      int c, void* dfeat, int lddf, float* bias_partial, void* stream);
/* render only (pose_embedding summary maps at other sizes; f32 output [B,s,s,K]) */
int imm_gauss_render_f32(const float* mu, int batch, int k, float inv_std, int s, float* out, int gauss_mode, void* stream);

/* ---- frozen VGG16 first layer (build_vgg16.py:22-26 grayscale+normalise, vgg16.py:345 conv1_1) -- */
/* images: gt f32 [B,S,S,3] and pred f32 [B,S,S,ldp] (first 3 ch) -> out 16-bit [2B,S,S,64] = concat([gt, pred], 0)
 * (imm_model.py:126).  halves: 1 = write only the gt images (out[0:B]), 2 = only the pred images (out[B:2B]), 3 = both;
 * the gt half does not depend on the network, so a caller may run it ahead on another stream. */
int imm_vgg_conv1_1_fwd(const float* gt, const float* pred, int ldp, int batch, int s, const float* w9x64,
                        const float* b64, void* out, int dtype, int halves, void* stream);
/* The head of the frozen VGG16 in one launch (vgg_head.hip): gray+normalise (build_vgg16.py:22-26) -> conv1_1 -> conv1_2
 * (vgg16.py:345-346: 3x3 SAME + bias + ReLU each) on concat([gt, pred], 0) (imm_model.py:126).  The persistent conv1_2 workgroup
 * produces the conv1_1 halo of its patches on the matrix cores (gray and filter as hi + lo pairs of the activation type: the f32
 * product to 2^-16), so conv1_1's 2B x S x S x 64 activation never travels through HBM; only images >= store_from of it are
 * stored to a11 (training: store_from = B, the prediction half = the ReLU mask of conv1_2's data gradient; 0 = all, 2B = none).
 * wt12 = conv1_2's packed forward filter image (imm_pack_weights mode 0, kpad12 = 576), y12 16-bit [2B,S,S,64],
 * gray_scratch >= imm_vgg_head_scratch_bytes(batch, s).  imm_vgg_head_supported: 16-bit dtype, S % 16 == 0, S >= 32. */
int imm_vgg_head_supported(int batch, int s, int dtype);
int64_t imm_vgg_head_scratch_bytes(int batch, int s);
int imm_vgg_head_fwd(const float* gt, const float* pred, int ldp, int batch, int s, const float* w9x64, const float* b64,
                     const void* wt12, int kpad12, const float* b12, void* a11, int store_from, void* y12, void* gray_scratch,
                     int dtype, void* stream);
/* dz 16-bit [B,S,S,64] (pred half, already ReLU-masked) -> dpred 16-bit [B,S,S,lddp]:
 * ch<3 = dgray/(3*255) + coef[input_idx]*mask[p]*(pred-gt), ch>=3 = 0.  coef is a device scalar table; input_idx = the
 * position of 'input' in perceptual.comp, or -1 when the raw image is not a loss feature; l1 != 0: sign(pred-gt)
 * (perceptual.l2: False, imm_model.py:132). */
int imm_vgg_conv1_1_bwd(const void* dz, int dtype, int batch, int s, const float* w9x64, const float* gt,
                        const float* pred, int ldp, const float* mask, const float* coef, int input_idx, int l1,
                        void* dpred, int lddp, void* stream);
/* the image-space term alone (no VGG feature tapped: reconstruction_loss 'l2', imm_model.py:385-387, or
 * perceptual.comp == ['input']): dpred ch<3 = coef[idx]*mask[p]*(pred-gt) (sign with l1), ch>=3 = 0. */
int imm_image_loss_grad(const float* gt, const float* pred, int ldp, int batch, int s, const float* mask,
                        const float* coef, int idx, int l1, void* dpred, int lddp, int dtype, void* stream);

/* ---- perceptual loss (imm_model.py:111-151; base_model.py:39-50) ----------------------------- */
#define IMM_SSE_BLOCKS 512
/* partial[IMM_SSE_BLOCKS] of sum mask[pixel/(s*s)...]*(a-b)^2 (l1 != 0: |a-b|, perceptual.l2: False, imm_model.py:132);
 * mask is the full-res f32 [B,S,S] mask, sampled with stride S/s (legacy resize == strided pick, imm_model.py:408-410).
 * a/b 16-bit [B,s,s,c]. */
/* imm_masked_sse of n <= 8 features in ONE launch: a[i]/b[i] 16-bit [B,s[i],s[i],c[i]] -> partial[i][IMM_SSE_BLOCKS]; the pointer and
 * size arrays are HOST arrays (copied into the kernel arguments). */
int imm_masked_sse_multi(int n, const void* const* a, const void* const* b, const int32_t* s_host, const int32_t* c_host,
                         float* const* partial, int dtype, int batch, const float* mask, int S, int l1, void* stream);
/* imm_masked_sse_multi AND imm_masked_sse_f32 (the f32 image pair of side S: the 'input' feature, imm_model.py:126-131) as ONE launch:
 * the error sums that sit back to back in front of the loss; every partial sum is bit for bit that of the two launches. */
int imm_masked_sse_all(int n, const void* const* a, const void* const* b, const int32_t* s_host, const int32_t* c_host,
                       float* const* partial, int dtype, int batch, const float* mask, int S, int l1,
                       const float* img_a, int lda, const float* img_b, int ldb, int img_c, float* img_partial, void* stream);
int imm_masked_sse(const void* a, const void* b, int dtype, int batch, int s, int c, const float* mask, int S, int l1,
                   float* partial, void* stream);
/* imm_masked_sse fused with the 2x2/2 max-pool that follows the tapped VGG layer (conv1_2, conv2_2): reads the two feature
 * halves once, writes the SSE partials and both pooled halves [batch, s/2, s/2, c].  pool_a == NULL: the ground-truth half is
 * pooled elsewhere (its own lane pools it with imm_maxpool2_fwd, imm_amd/engine.py), only pool_b is written. */
int imm_masked_sse_pool(const void* a, const void* b, int dtype, int batch, int s, int c, const float* mask, int S,
                        float* partial, void* pool_a, void* pool_b, void* stream);
int imm_masked_sse_f32(const float* a, int lda, const float* b, int ldb, int batch, int s, int c, const float* mask,
                       int l1, float* partial, void* stream);
/* nfeat features: partial [nfeat][IMM_SSE_BLOCKS], nel[nfeat] element counts, agg[nfeat] running
 * normalisers (updated when training).  Writes out[0..nfeat) loss terms, [nfeat..2nfeat) masked means,
 * [2nfeat..3nfeat) gradient coefficients c_k (d total / d (a_pred) = c_k*mask*(a_pred-a_gt), or c_k*mask*sign(.) with l1),
 * out[3nfeat] = 1000*sum terms, out[3nfeat+1] = wd_loss, out[3nfeat+2] = total.
 * mode IMM_LOSS_L2 (reconstruction_loss 'l2', imm_model.py:376,385-387,399): nfeat = 1 = the image, no normaliser:
 * out[3] = 1000*mean(mask*(pred-gt)^2), total = out[3]/255 + wd_loss, c_0 = (1000/255)*2/nel. */
#define IMM_LOSS_PERCEPTUAL 0
#define IMM_LOSS_L2 1
/* loss_scale (device f32 scalar, or NULL = 1): the gradient coefficients c_k are written multiplied by it — the seeds of the
 * whole backward chain (imm_tap_grad, imm_unpool_tap_grad, imm_vgg_conv1_1_bwd, imm_image_loss_grad read c_k), so that 16-bit
 * gradient tensors stored as f16 stay inside f16's range; the loss values themselves are NOT scaled.  The reference computes
 * in fp32 (imm_model.py:97) and needs none; imm_clip_adam_step divides the scale out again. */
int imm_perceptual_finalize(const float* partial, int nfeat, const float* nel, float* agg, int training,
                            const float* wd_loss, int l1, int mode, const float* loss_scale, float* out, void* stream);
/* Summaries of the reference's train loop that live on the device (round 5):
 * imm_cost_ema — BaseModel._add_cost_summary (base_model.py:52-60; tf.train.ExponentialMovingAverage(0.99) of reconstruction_loss,
 *   weights_loss, loss_total, applied with every training step through avg_ops): cost3 = the three f32 scalars in that order (the
 *   tail out[3 nfeat .. 3 nfeat + 3) of imm_perceptual_finalize), state4 = {shadow[3], local_step}, zero at start; the summarised
 *   value is shadow[i] itself (tensorflow 1.10: ExponentialMovingAverage(zero_debias=False)).  One tiny launch, capturable.
 * imm_rms16 — selfsup/vgg16.py:232-234 'activation/<layer>' = sqrt(mean(z^2)) of a 16-bit tensor of n elements (n % 8 == 0, 16-byte
 *   aligned); partial: nblk f32 of scratch (nblk <= 4096 workgroups); out[0] = the scalar.  Called on summary steps only. */
int imm_cost_ema(const float* cost3, float* state4, float decay, void* stream);
int imm_rms16(const void* x, int64_t n, int dtype, float* partial, int nblk, float* out, void* stream);
/* imm_maxpool2_bwd (no ReLU mask) followed by imm_tap_grad (has_in, relu) in one pass, for tapped layers that are pooled
 * next (conv1_2, conv2_2); dpool [batch, s/2, s/2, c] is the gradient of the pooled tensor.  Bitwise equal to the sequence. */
int imm_unpool_tap_grad(void* da, const void* dpool, const void* a_pred, const void* a_gt, int dtype, int batch, int s, int c,
                        const float* mask, int S, const float* coef, int idx, void* stream);
/* da = (has_in ? da : 0) + coef[idx]*mask*(a_pred-a_gt) (sign(a_pred-a_gt) with l1), then *= (a_pred>0) if relu.
 * 16-bit [B,s,s,c] */
int imm_tap_grad(void* da, int has_in, const void* a_pred, const void* a_gt, int dtype, int batch, int s, int c,
                 const float* mask, int S, const float* coef, int idx, int relu, int l1, void* stream);

/* ---- optimizer (cnn_train_multi.py:86-98 mean+clip_by_norm; scripts/train.py:92-98 Adam) ------ */
/* segment table: nseg tensors, seg_off[nseg+1] element offsets into the flat f32 buffers,
 * seg_wd[nseg] = weight-decay coefficient of that tensor (1e-5 for conv kernels, else 0).
 * All tables are device int32/f32 arrays built once by the host. */
/* optimizer (scripts/train.py:97-104, config key training.optim): the update after the tower mean and the per-tensor clip
 *   IMM_OPT_ADAM      tf.train.AdamOptimizer(lr): beta1 .9, beta2 .999, eps 1e-8; slots m, v
 *   IMM_OPT_ADADELTA  tf.train.AdadeltaOptimizer(lr, rho=0.95, epsilon=1e-6): slot v = accum, slot m = accum_update (both 0):
 *                     accum = rho accum + (1-rho) g^2; u = sqrt(accum_update + eps) / sqrt(accum + eps) * g;
 *                     accum_update = rho accum_update + (1-rho) u^2; w -= lr u          (rho = beta1, eps = eps)
 *   IMM_OPT_ADAGRAD   tf.train.AdagradOptimizer(lr): slot v = accumulator (initial value 0.1, set by the caller):
 *                     accum += g^2; w -= lr g / sqrt(accum) */
#define IMM_OPT_ADAM 0
#define IMM_OPT_ADADELTA 1
#define IMM_OPT_ADAGRAD 2
typedef struct imm_opt_hparams {
  float lr_start, lr_decay; int32_t lr_step; float lr_multiple;   /* staircase exponential decay */
  float beta1, beta2, eps, clip;                                   /* clip <= 0 disables clipping  */
  float grad_scale;                                                /* 1 / number of towers         */
  int32_t optim;                                                   /* IMM_OPT_*                    */
  int32_t scale_growth_interval;  /* loss scaling: double S after this many clean steps in a row (0 = static S)      */
  float scale_max;                /* ... up to this value (<= 0: unbounded)                                          */
} imm_opt_hparams;
/* wd_loss = sum_seg wd/2 * sum w^2  -> *out  (base_model.py:33-37) */
int imm_weight_decay_loss(const float* params, const int32_t* blk_seg, const int32_t* blk_begin,
                          const int32_t* blk_end, int nblk, const float* seg_wd, float* blk_partial, float* out,
                          void* stream);
/* g <- g*grad_scale + wd*w (in place), norm2[seg] = sum g^2; then Adam with per-tensor clip.
 * step_count: device int32 = TF global_step (drives the staircase learning rate only; incremented by this call);
 * adam_t: device int32 = number of Adam updates applied to these m/v slots, i.e. TF's beta{1,2}_power accumulators
 * (tf.train.AdamOptimizer keeps them apart from global_step: a restore without the optimizer slots restarts the bias
 * correction at t = 1 while global_step carries on; cnn_train_multi.py:404-433); incremented by this call;
 * lr_state: device f32[2] = {lr_t, lr} written for inspection.
 * loss_scale_state: NULL, or device f32[4] = {S, clean steps in a row, steps skipped so far, this step overflowed} for gradients
 * that were computed with their seeds multiplied by S (imm_perceptual_finalize): g is divided by S first; if any tensor's
 * norm is then not finite (an f16 gradient overflowed) the whole update is SKIPPED — weights, slots, step_count and adam_t
 * stay as they are — and S is halved (not below 1); after hp.scale_growth_interval clean steps in a row S doubles (up to
 * hp.scale_max).  The reference asserts on a NaN loss instead (cnn_train_multi.py:463); it has no reduced-precision mode.
 * "Not finite" is decided per chunk of the block table: a chunk whose sum of g^2 is NaN, inf or > 1e30.
 * Two launches: the gradient pass (+ the learning rate), then the update, whose workgroups each rebuild their tensor's norm and
 * the skip decision from the chunk sums (blk_partial f32[nblk]); lr_state is written by the first launch. */
int imm_clip_adam_step(float* params, float* grads, float* m, float* v, const int32_t* blk_seg,
                       const int32_t* blk_begin, const int32_t* blk_end, int nblk, int nseg,
                       const int32_t* seg_first_blk, const float* seg_wd, float* blk_partial, float* seg_norm2,
                       int32_t* step_count, int32_t* adam_t, float* lr_state, const imm_opt_hparams* hp_host,
                       float* loss_scale_state, void* stream);

/* ---- data-parallel gradient exchange (cnn_train_multi.py:66-106 average_gradients) ------------------------------------ */
/* One process per GPU; ONE sum all-reduce of (a bucket of) the flat f32 gradient buffer over RCCL / xGMI on the caller's
 * stream (capturable into the step's HIP graph); the division by the number of towers is imm_clip_adam_step's grad_scale, so
 * the reference's order "mean, then per-tensor clip" holds.  RCCL is bound at run time (the librccl already in the process,
 * else the system one).  unique id: 128 bytes made by rank 0, handed to every rank by any host-side channel. */
int imm_rccl_unique_id(void* out128_host);
int imm_rccl_init(int rank, int world, const void* unique_id128_host, void** comm_out_host);
int imm_rccl_allreduce(void* comm, float* buf, int64_t count, void* stream);
int imm_rccl_destroy(void* comm);

/* ---- workspace sizes: every scratch buffer is caller-allocated; these return its size in BYTES (< 0: unsupported shape) -- */
int64_t imm_conv2d_workspace_bytes(const imm_conv_desc* desc_host);                /* stats_partial of imm_conv2d            */
int64_t imm_conv2d_group_workspace_bytes(const imm_conv_desc* descs, int n);      /* stats_partial of imm_conv2d_group      */
int64_t imm_conv2d_wgrad_workspace_bytes(const imm_conv_desc* desc_host, int lddy, int nsplit);  /* slab (nsplit <= 0: the
                                                                                      forced / minimal split count)        */
int64_t imm_colsum_workspace_bytes(int64_t npix, int c);                          /* partial of imm_colsum                  */
int64_t imm_bn_bwd_workspace_bytes(int64_t npix, int c);                          /* partial of imm_bn_bwd_reduce           */
int64_t imm_upsample2x_bwd_bn_workspace_bytes(int batch, int h, int w, int c);    /* partial of imm_upsample2x_bwd_bn       */
int64_t imm_masked_sse_workspace_bytes(int nfeat);                                /* partial of imm_masked_sse* (per loss)  */

/* ---- thin-plate-spline augmentation (imm/utils/tps_sampler.py:76-99,142-157; imm/datasets/tps_dataset.py:70-96) ---- */
/* dst[b][p] = bilinear(src[b], sum_j basis_t[j][p] * w_tps[b][j][0..1]) with F.grid_sample's align_corners=True mapping and
 * zero padding.  src f32 NHWC [B,h,w,ld_src] (first c <= 8 channels), basis_t f32 [m3][h*w] (TPSGridGen's L matrix
 * transposed, m3 = control points + 3), w_tps f32 [B][m3][2] ((x, y) columns).  Outputs (each may be NULL): dst = all c
 * channels [B,h,w,ld_dst]; dst_c0 = channel 0 only [B,h,w] (the mask plane); dst_rest = channels 1..c-1 [B,h,w,ld_rest]
 * (the image, e.g. straight into the training step's input buffer). */
int imm_tps_warp(const float* src, int ld_src, int batch, int h, int w, int c, const float* basis_t, int m3,
                 const float* w_tps, float* dst, int ld_dst, float* dst_c0, float* dst_rest, int ld_rest, void* stream);
/* TPSRandomSampler(pad=True) (tps_sampler.py:24-29,89-92): the source is addressed as if replicate-padded by pad_y rows /
 * pad_x columns on each side (coordinates are clamped; no padded copy), warped on a grid_h x grid_w sampling grid (basis_t
 * f32 [m3][grid_h*grid_w]; normalised coordinates refer to the PADDED source, zero outside it), and the window
 * [crop_y, crop_y+out_h) x [crop_x, crop_x+out_w) of the result is written to dst [B,out_h,out_w,ld_dst].  The reference's
 * own arithmetic is pad_y = w/2, pad_x = h/2, grid = (h + h/2) x (w + w/2), crop = (w/2, h/2), out = (h + h/2 - 2(w/2)) x
 * (w + w/2 - 2(h/2)) — F.pad's argument order makes it swap the two paddings, and the output is smaller than the input;
 * imm_amd.data.tps.TPSRandomSampler reproduces exactly that. */
int imm_tps_warp_pad(const float* src, int ld_src, int batch, int h, int w, int c, int pad_y, int pad_x, int grid_h, int grid_w,
                     int crop_y, int crop_x, int out_h, int out_w, const float* basis_t, int m3, const float* w_tps,
                     float* dst, int ld_dst, void* stream);

/* ---- decoded-image ingest (imm/datasets/celeba_dataset.py:136-174, aflw_dataset.py:81-114: to_float -> bilinear
 *      align_corners=True resize -> central crop) ---- */
/* src: `batch` u8 HWC images of different sizes packed in one device buffer, image b at src + offsets[b] with
 * hw[2b] rows x hw[2b+1] columns x c channels (1 <= c <= 4).  Each is resized (TF1 resize_bilinear, align_corners=True) to
 * resize_h x resize_w and the window [crop_y0, crop_y0+out_h) x [crop_x0, crop_x0+out_w) of that is written as float32
 * (values stay in [0, 255]) to dst[b][y][x][0..c-1], pixel stride ld_dst floats (>= c: dst may point at channel 1 of
 * the mask||image stack imm_tps_warp reads).  Float32 arithmetic in TF's order, unfused: bit-exact vs the host oracle. */
int imm_resize_crop_u8(const uint8_t* src, const int64_t* offsets, const int32_t* hw, int batch, int c, int resize_h,
                       int resize_w, int crop_y0, int crop_x0, int out_h, int out_w, float* dst, int ld_dst, void* stream);

/* ---- host utility: CRC-32C of TensorFlow checkpoint bundles (cnn_train_multi.py:404-439 tf.train.Saver files) ---- */
/* *crc_inout = crc32c(*crc_inout continued over data[0..n)); start with 0.  Host memory, no GPU work. */
int imm_crc32c(const void* data, uint64_t n, uint32_t* crc_inout);

#ifdef __cplusplus
}
#endif
#endif /* IMM_HIP_H */
