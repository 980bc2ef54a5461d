"""Thin torch-tensor -> C-ABI wrappers.  Every function launches on torch's CURRENT stream and returns
immediately; tensors are caller-allocated device buffers (PyTorch is only the allocator here).
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import ConvDesc, OptHParams, call  # noqa: F401


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def dtype_enum(dt):
    if dt == torch.bfloat16:
        return L.IMM_BF16
    if dt == torch.float16:
        return L.IMM_F16
    if dt == torch.float32:
        return L.IMM_F32          # the f32 witness engine (plain f32 kernels; imm_amd/csrc/conv_f32.hip)
    raise ValueError('activation dtype must be torch.bfloat16 / torch.float16 (or torch.float32: the witness engine), got %r' % (dt,))


def round_up(x, m):
    return (x + m - 1) // m * m


def same_pad_before(n_in, k, s):
    """TF SAME: pad_before = max((ceil(in/s)-1)*s + k - in, 0) // 2  (SURVEY.md §8a S1)."""
    n_out = -(-n_in // s)
    return max((n_out - 1) * s + k - n_in, 0) // 2, n_out


def fwd_desc(batch, hi, wi, ci, ldx, co, ldy, k, stride, flags, ldmask=0, kw=None):
    """k x k SAME convolution; kw (e.g. 1) gives a k x kw kernel (SAME along both axes)."""
    kw = k if kw is None else kw
    pt, ho = same_pad_before(hi, k, stride)
    pl, wo = same_pad_before(wi, kw, stride)
    return ConvDesc(batch=batch, hi=hi, wi=wi, ci=ci, ldx=ldx, ho=ho, wo=wo, co=co, ldy=ldy, kh=k, kw=kw,
                    stride=stride, pad_t=pt, pad_l=pl, updiv=1, kpad=round_up(k * kw * ci, 32), flags=flags,
                    ldmask=ldmask)


def dgrad_desc(batch, hi, wi, ci_out, lddx, co_pad, lddy, k, stride, flags, ldmask=0):
    """Data gradient of the forward conv (hi,wi,ci)->(ho,wo,co): a conv of dy [ho,wo,co_pad] producing
    dx [hi,wi,ci_out] with flipped weights, stride 1, updiv = forward stride, pad = k-1-pad_fwd."""
    pt, ho = same_pad_before(hi, k, stride)
    pl, wo = same_pad_before(wi, k, stride)
    return ConvDesc(batch=batch, hi=ho, wi=wo, ci=co_pad, ldx=lddy, ho=hi, wo=wi, co=ci_out, ldy=lddx, kh=k, kw=k,
                    stride=1, pad_t=k - 1 - pt, pad_l=k - 1 - pl, updiv=stride, kpad=round_up(k * k * co_pad, 32),
                    flags=flags, ldmask=ldmask)


def dgrad_s2_class_descs(batch, hi, wi, ci_out, lddx, co_pad, lddy, k, flags=0, ldmask=0):
    """Stride-2 data gradient as 4 dense stride-1 convolutions, one per input-pixel parity class (py,px):
        dx[2a+py][2b+px] = sum_{ty,tx} dy[a-ty][b-tx] * W[py+2ty][px+2tx]
    (TF SAME on an even side with k=3 pads 0 before, so ky = iy - 2*oy has the parity of iy).  Returns
    [(desc, pack_mode)], or None when the geometry does not fit (odd side / pad_before != 0)."""
    pt, ho = same_pad_before(hi, k, 2)
    pl, wo = same_pad_before(wi, k, 2)
    if pt != 0 or pl != 0 or hi != 2 * ho or wi != 2 * wo:
        return None
    out = []
    for py in (0, 1):
        for px in (0, 1):
            ny, nx = (k - py + 1) // 2, (k - px + 1) // 2
            d = ConvDesc(batch=batch, hi=ho, wi=wo, ci=co_pad, ldx=lddy, ho=ho, wo=wo, co=ci_out, ldy=lddx, kh=ny, kw=nx,
                         stride=1, pad_t=ny - 1, pad_l=nx - 1, updiv=1, kpad=round_up(ny * nx * co_pad, 32), flags=flags,
                         ldmask=ldmask, out_scale=2, out_off_y=py, out_off_x=px)
            out.append((d, 4 + 2 * py + px))
    return out


def device_info():
    out = (C.c_int32 * 2)()
    call('imm_device_info', C.cast(out, C.c_void_p))
    return int(out[0]), int(out[1])


def set_cu_limit(cus):
    """Size the persistent convolution launches this thread issues from now on for `cus` compute units (0 = the whole device):
    include/imm_hip.h imm_set_cu_limit."""
    call('imm_set_cu_limit', int(cus))


# ---- graph capture ---------------------------------------------------------------------------
class Graph:
    """A captured launch sequence (HIP graph) on torch's current stream."""

    def __init__(self):
        self._exec = C.c_void_p()

    def capture_begin(self):
        call('imm_graph_begin', _s())

    def capture_end(self):
        call('imm_graph_end', _s(), C.byref(self._exec))

    def launch(self):
        call('imm_graph_launch', self._exec, _s())

    def __del__(self):
        try:
            if self._exec:
                L.load().imm_graph_destroy(self._exec)
        except Exception:
            pass


# ---- weights / conv --------------------------------------------------------------------------
def pack_weights(w, wt, mode, kh, kw, ci_real, co_real, c_pad, rows, kpad):
    call('imm_pack_weights', _p(w), _p(wt), dtype_enum(wt.dtype), mode, kh, kw, ci_real, co_real, c_pad, rows, kpad, _s())


class JobTable:
    """Device job table for the table-driven launches (imm_pack_weights_multi / imm_wgrad_reduce_multi)."""

    def __init__(self, jobs, items_per_job, items_per_block, device, blocks_per_job=None):
        first = [0]
        if blocks_per_job is None:
            blocks_per_job = [max(1, -(-int(n) // items_per_block)) for n in items_per_job]
        for n in blocks_per_job:
            first.append(first[-1] + int(n))
        rows = [list(j) + [0] * (12 - len(j)) for j in jobs]
        self.jobs = torch.tensor(rows, dtype=torch.int64, device=device)
        self.blk_first = torch.tensor(first, dtype=torch.int32, device=device)
        self.n_jobs, self.n_blocks = len(jobs), first[-1]


def pack_table(jobs, device):
    """Job table of imm_pack_weights_multi: jobs = (w, wt, mode, kh, kw, ci_real, co_real, c_pad, rows, kpad) tuples."""
    blocks = []
    for j in jobs:
        n = L.load().imm_pack_weights_multi_blocks(int(j[2]), int(j[8]), int(j[9]))
        if n <= 0:
            raise L.ImmHipError('imm_pack_weights_multi_blocks(%d, %d, %d) failed' % (j[2], j[8], j[9]))
        blocks.append(n)
    return JobTable(jobs, None, None, device, blocks_per_job=blocks)


def pack_weights_multi(tab, dtype):
    call('imm_pack_weights_multi', _p(tab.jobs), _p(tab.blk_first), tab.n_jobs, tab.n_blocks, dtype_enum(dtype), _s())


def reduce_table(jobs, outputs_per_job, device):
    """Job table of imm_wgrad_reduce_multi: jobs = (slab, dw, nsplit, taps, ci_pad, ci_real, co, kpad) tuples; the split-lane
    count of a job (field 8) is the power of two next to its slab count, capped at 16."""
    rows, blocks = [], []
    for j, n in zip(jobs, outputs_per_job):
        lanes = 1
        while lanes < min(16, int(j[2])):
            lanes *= 2
        rows.append(tuple(j[:8]) + (lanes,))
        blocks.append(max(1, -(-int(n) // (1024 // lanes))))
    return JobTable(rows, None, None, device, blocks_per_job=blocks)


def wgrad_reduce_multi(tab):
    call('imm_wgrad_reduce_multi', _p(tab.jobs), _p(tab.blk_first), tab.n_jobs, tab.n_blocks, _s())


def conv2d(desc, x, wt, bias, y, stats=None, mask=None):
    call('imm_conv2d', C.byref(desc), dtype_enum(x.dtype), _p(x), _p(wt), _p(bias), _p(y), _p(stats), _p(mask), _s())


def conv2d_tap_supported(desc):
    return bool(L.load().imm_conv2d_tap_supported(C.byref(desc)))


def conv2d_tap(desc, x, wt, y, a_pred, a_gt, lda, loss_mask, S, coef, idx, l1=False):
    call('imm_conv2d_tap', C.byref(desc), dtype_enum(x.dtype), _p(x), _p(wt), _p(y), _p(a_pred), _p(a_gt), lda, _p(loss_mask), S,
         _p(coef), idx, int(l1), _s())


def conv_stats_blocks(desc):
    n = L.load().imm_conv_stats_blocks(C.byref(desc))
    if n <= 0:
        raise L.ImmHipError('imm_conv_stats_blocks: ' + L.load().imm_last_error().decode())
    return n


def conv2d_wgrad(desc, x, dy, lddy, slab, nsplit):
    call('imm_conv2d_wgrad', C.byref(desc), dtype_enum(x.dtype), _p(x), _p(dy), lddy, _p(slab), nsplit, _s())


CONV_FAMILIES = {1: 'igemm', 2: 'igemm64', 3: 'halo', 4: 'halo2', 5: 'hdeep', 6: 'hdeep6', 7: 's2f', 8: 'f32'}


def conv2d_variant(desc, dtype):
    """(family name, variant key) of the kernel imm_conv2d dispatches `desc` to (imm_conv2d_variant: family * 100000 + tile
    variant; families in CONV_FAMILIES) — what a test asserts to prove that the kernel a case is named after is the one that ran."""
    key = L.load().imm_conv2d_variant(C.byref(desc), dtype_enum(dtype))
    if key < 0:
        raise L.ImmHipError('imm_conv2d_variant: ' + L.load().imm_last_error().decode())
    return CONV_FAMILIES[key // 100000], key


def conv2d_wgrad_variant(desc, lddy, dtype):
    """(variant key, workgroups per pixel split, length in units, resident workgroups per CU) of a layer's filter-gradient
    kernel: jobs with the same key share a launch of imm_conv2d_wgrad_multi."""
    wps, units, pcu = C.c_int(0), C.c_int(0), C.c_int(0)
    key = L.load().imm_conv2d_wgrad_variant(C.byref(desc), lddy, dtype_enum(dtype), C.byref(wps), C.byref(units), C.byref(pcu))
    if key < 0:
        raise L.ImmHipError('imm_conv2d_wgrad_variant: ' + L.load().imm_last_error().decode())
    return key, wps.value, units.value, pcu.value


class WgradMulti(object):
    """Planned multi-problem filter-gradient launch: jobs = [(desc, x, dy, lddy, slab, nsplit[, (x_scale, x_shift, x_relu)])];
    the optional triple = normalise on load (x is the raw output of the conv + BN + ReLU block in front of the layer).  Keeps
    the host table, its device copy and the operand tensors alive."""

    def __init__(self, jobs, dtype):
        n = len(jobs)
        self.keep = [t for j in jobs for t in (j[1], j[2], j[4])] + [t for j in jobs if len(j) > 6 and j[6] for t in j[6][:2]]
        arr = (L.WgradJob * n)()
        for i, job in enumerate(jobs):
            desc, x, dy, lddy, slab, nsplit = job[:6]
            arr[i].desc = desc
            arr[i].x, arr[i].dy, arr[i].slab = x.data_ptr(), dy.data_ptr(), slab.data_ptr()
            arr[i].lddy, arr[i].nsplit = int(lddy), int(nsplit)
            if len(job) > 6 and job[6]:
                arr[i].x_scale, arr[i].x_shift, arr[i].x_relu = job[6][0].data_ptr(), job[6][1].data_ptr(), int(bool(job[6][2]))
        nbytes = L.load().imm_conv2d_wgrad_multi_table_bytes(n)
        if nbytes <= 0:
            raise L.ImmHipError('imm_conv2d_wgrad_multi_table_bytes(%d)' % n)
        self.host = (C.c_ubyte * nbytes)()
        call('imm_conv2d_wgrad_multi_plan', C.cast(arr, C.c_void_p), n, dtype_enum(dtype), C.cast(self.host, C.c_void_p))
        self.dev = torch.frombuffer(self.host, dtype=torch.uint8).clone().to(jobs[0][1].device)
        self.n = n


def conv2d_wgrad_multi(multi):
    call('imm_conv2d_wgrad_multi', C.cast(multi.host, C.c_void_p), _p(multi.dev), _s())


def conv2d_wgrad_splits(desc, lddy):
    n = L.load().imm_conv2d_wgrad_splits(C.byref(desc), lddy)
    if n < 0:
        raise L.ImmHipError('imm_conv2d_wgrad_splits failed')
    return n


def conv2d_wgrad_reduce(slab, nsplit, kh, kw, ci_pad, ci_real, co, kpad, dw):
    call('imm_conv2d_wgrad_reduce', _p(slab), nsplit, kh, kw, ci_pad, ci_real, co, kpad, _p(dw), _s())


def colsum_blocks(npix, c):
    n = L.load().imm_colsum_blocks(npix, c)
    if n <= 0:
        raise L.ImmHipError('imm_colsum_blocks(%d,%d) unsupported' % (npix, c))
    return n


def colsum(dy, npix, c, c_out, ld, partial, out):
    call('imm_colsum', _p(dy), dtype_enum(dy.dtype), npix, c, c_out, ld, _p(partial), _p(out), _s())


# ---- batch norm ------------------------------------------------------------------------------
def bn_finalize(partial, nblk, c, count, gamma, beta, eps, momentum, training, mm, mv, scale, shift, mean, rstd):
    call('imm_bn_finalize', _p(partial), nblk, c, count, _p(gamma), _p(beta), eps, momentum, int(training), _p(mm),
         _p(mv), _p(scale), _p(shift), _p(mean), _p(rstd), _s())


def bn_apply_relu(y, npix, c, ldy, scale, shift, relu, x_out, ldx):
    call('imm_bn_apply_relu', _p(y), dtype_enum(y.dtype), npix, c, ldy, _p(scale), _p(shift), int(relu), _p(x_out), ldx, _s())


def bn_apply_fused(partial, nblk, c, count, gamma, beta, eps, momentum, training, mm, mv, scale, shift, mean, rstd, y, ldy,
                   relu, x_out, ldx, up2x=None, ldu=0, h=0, w=0):
    call('imm_bn_apply_fused', _p(partial), nblk, c, count, _p(gamma), _p(beta), eps, momentum, int(training), _p(mm), _p(mv),
         _p(scale), _p(shift), _p(mean), _p(rstd), _p(y), dtype_enum(y.dtype), ldy, int(relu), _p(x_out), ldx,
         _p(up2x), ldu, h, w, _s())


def bn_bwd_apply_fused(partial, nblk, c, count, gamma, dout, lddo, y, ldy, scale, shift, mean, rstd, relu, dgamma, dbeta,
                       dy_out, lddy):
    call('imm_bn_bwd_apply_fused', _p(partial), nblk, c, count, _p(gamma), _p(dout), lddo, _p(y), ldy, dtype_enum(y.dtype),
         _p(scale), _p(shift), _p(mean), _p(rstd), int(relu), _p(dgamma), _p(dbeta), _p(dy_out), lddy, _s())


def bn_bwd_blocks(npix, c):
    n = L.load().imm_bn_bwd_blocks(npix, c)
    if n <= 0:
        raise L.ImmHipError('imm_bn_bwd_blocks(%d,%d) unsupported' % (npix, c))
    return n


def rows_reduce(src, rows, width, group, dst):
    """dst[g] = sum of rows [g*group, (g+1)*group) of src[rows][width] (include/imm_hip.h: imm_rows_reduce)."""
    call('imm_rows_reduce', _p(src), rows, width, group, _p(dst), _s())


def bn_bwd_reduce(dout, lddo, y, ldy, npix, c, scale, shift, mean, rstd, relu, partial):
    call('imm_bn_bwd_reduce', _p(dout), lddo, _p(y), ldy, dtype_enum(y.dtype), npix, c, _p(scale), _p(shift), _p(mean),
         _p(rstd), int(relu), _p(partial), _s())


def bn_bwd_reduce_up(dy_up, lddy, dprev, lddp, y, ldy, batch, h, w, c, scale, shift, mean, rstd, relu, partial):
    call('imm_bn_bwd_reduce_up', _p(dy_up), lddy, _p(dprev), lddp, _p(y), ldy, dtype_enum(y.dtype), batch, h, w, c, _p(scale),
         _p(shift), _p(mean), _p(rstd), int(relu), _p(partial), _s())


def bn_bwd_finalize(partial, nblk, c, count, gamma, beta, rstd, dgamma, dbeta, coef, from_out=False, ldp=None):
    call('imm_bn_bwd_finalize', _p(partial), nblk, c, c if ldp is None else ldp, count, _p(gamma), _p(beta), _p(rstd),
         int(from_out), _p(dgamma), _p(dbeta), _p(coef), _s())


def bn_bwd_apply(dout, lddo, y, ldy, npix, c, scale, shift, mean, rstd, relu, coef, dy_out, lddy):
    call('imm_bn_bwd_apply', _p(dout), lddo, _p(y), ldy, dtype_enum(y.dtype), npix, c, _p(scale), _p(shift), _p(mean),
         _p(rstd), int(relu), _p(coef), _p(dy_out), lddy, _s())


# ---- resampling / pooling ---------------------------------------------------------------------
def upsample2x_fwd(x, y, batch, h, w, c, ldx, ldy):
    call('imm_upsample2x_fwd', _p(x), _p(y), dtype_enum(x.dtype), batch, h, w, c, ldx, ldy, _s())


def upsample2x_bwd(dy, dx, batch, h, w, c, lddy, lddx):
    call('imm_upsample2x_bwd', _p(dy), _p(dx), dtype_enum(dy.dtype), batch, h, w, c, lddy, lddx, _s())


def upsample2x_bwd_bn_blocks(batch, h, w, c):
    n = L.load().imm_upsample2x_bwd_bn_blocks(batch, h, w, c)
    if n <= 0:
        raise L.ImmHipError('imm_upsample2x_bwd_bn_blocks(%d,%d,%d,%d) unsupported' % (batch, h, w, c))
    return n


def upsample2x_bwd_bn(dy, dx, batch, h, w, c, lddy, lddx, out, ldo, partial):
    call('imm_upsample2x_bwd_bn', _p(dy), _p(dx), dtype_enum(dy.dtype), batch, h, w, c, lddy, lddx, _p(out), ldo, _p(partial), _s())


def resize_ac_fwd(x, y, batch, hi, wi, ho, wo, c, ldx, ldy):
    call('imm_resize_ac_fwd', _p(x), _p(y), dtype_enum(x.dtype), batch, hi, wi, ho, wo, c, ldx, ldy, _s())


def resize_ac_bwd(dy, dx, batch, hi, wi, ho, wo, c, lddy, lddx):
    call('imm_resize_ac_bwd', _p(dy), _p(dx), dtype_enum(dy.dtype), batch, hi, wi, ho, wo, c, lddy, lddx, _s())


def maxpool2_fwd(x, y, batch, h, w, c):
    call('imm_maxpool2_fwd', _p(x), _p(y), dtype_enum(x.dtype), batch, h, w, c, _s())


def maxpool2_bwd(x, dy, dx, batch, h, w, c, relu_mask):
    call('imm_maxpool2_bwd', _p(x), _p(dy), _p(dx), dtype_enum(x.dtype), batch, h, w, c, int(relu_mask), _s())


def pack_image_taps(src, dst, batch, h, w, kw, pad_l, ld):
    call('imm_pack_image_taps', _p(src), _p(dst), dtype_enum(dst.dtype), batch, h, w, kw, pad_l, ld, _s())


def pack_image(src, dst, npix):
    call('imm_pack_image', _p(src), _p(dst), dtype_enum(dst.dtype), npix, _s())


# ---- bottleneck ------------------------------------------------------------------------------
def gauss_mode_enum(mode):
    if mode not in L.GAUSS_MODES:
        raise ValueError('Unknown mode: ' + str(mode))       # imm_model.py:75
    return L.GAUSS_MODES[mode]


def softargmax_gauss_fwd(heat, ldh, batch, h, w, k, inv_std, s, mu, py, px, gauss_out, ldg, dtype, mode='rot'):
    call('imm_softargmax_gauss_fwd', _p(heat), ldh, batch, h, w, k, float(inv_std), s, _p(mu), _p(py), _p(px),
         _p(gauss_out), ldg, dtype_enum(dtype), gauss_mode_enum(mode), _s())


def softargmax_gauss_bwd(dgauss, ldg, batch, h, w, k, inv_std, s, mu, py, px, dheat, lddh, mode='rot'):
    call('imm_softargmax_gauss_bwd', _p(dgauss), ldg, dtype_enum(dheat.dtype), batch, h, w, k, float(inv_std), s, _p(mu),
         _p(py), _p(px), _p(dheat), lddh, gauss_mode_enum(mode), _s())


def pose_head_fwd(feat, ldf, c, wt, bias, batch, h, w, k, inv_std, s, heat, ldh, mu, py, px, gauss_out, ldg, dtype, mode='rot'):
    call('imm_pose_head_fwd', _p(feat), ldf, c, _p(wt), wt.shape[1], _p(bias), dtype_enum(dtype), batch, h, w, k, float(inv_std), s,
         _p(heat), ldh, _p(mu), _p(py), _p(px), _p(gauss_out), ldg, gauss_mode_enum(mode), _s())


def pose_head_bwd(dgauss, ldg, batch, h, w, k, inv_std, s, mu, py, px, dheat, lddh, wt_dgrad, c, dfeat, lddf, bias_partial, mode='rot'):
    call('imm_pose_head_bwd', _p(dgauss), ldg, dtype_enum(dheat.dtype), batch, h, w, k, float(inv_std), s, _p(mu), _p(py), _p(px),
         _p(dheat), lddh, gauss_mode_enum(mode), _p(wt_dgrad), wt_dgrad.shape[1], c, _p(dfeat), lddf, _p(bias_partial), _s())


def gauss_render_f32(mu, batch, k, inv_std, s, out, mode='rot'):
    call('imm_gauss_render_f32', _p(mu), batch, k, float(inv_std), s, _p(out), gauss_mode_enum(mode), _s())


# ---- VGG head / loss --------------------------------------------------------------------------
def vgg_head_supported(batch, s, dtype):
    """conv1_1 + conv1_2 of the frozen VGG16 in one launch (vgg_head.hip) serves this shape / storage type."""
    if dtype not in (torch.bfloat16, torch.float16):
        return False
    return bool(L.load().imm_vgg_head_supported(batch, s, dtype_enum(dtype)))


def vgg_head_scratch_bytes(batch, s):
    return int(L.load().imm_vgg_head_scratch_bytes(batch, s))


def vgg_head_fwd(gt, pred, ldp, batch, s, w11, b11, wt12, b12, a11, store_from, y12, gray_scratch):
    """gt f32 [B,S,S,3], pred f32 [B,S,S,ldp] -> y12 16-bit [2B,S,S,64] = relu(conv1_2(relu(conv1_1(gray(concat([gt, pred])))))),
    a11[store_from:] = conv1_1's activation of those images (selfsup/vgg16.py:345-346, build_vgg16.py:22-26)."""
    call('imm_vgg_head_fwd', _p(gt), _p(pred), ldp, batch, s, _p(w11), _p(b11), _p(wt12), wt12.shape[1], _p(b12), _p(a11),
         store_from, _p(y12), _p(gray_scratch), dtype_enum(y12.dtype), _s())


def vgg_conv1_1_fwd(gt, pred, ldp, batch, s, w, b, out, halves=3):
    call('imm_vgg_conv1_1_fwd', _p(gt), _p(pred), ldp, batch, s, _p(w), _p(b), _p(out), dtype_enum(out.dtype), halves, _s())


def vgg_conv1_1_bwd(dz, batch, s, w, gt, pred, ldp, mask, coef, dpred, lddp, input_idx=0, l1=False):
    call('imm_vgg_conv1_1_bwd', _p(dz), dtype_enum(dz.dtype), batch, s, _p(w), _p(gt), _p(pred), ldp, _p(mask), _p(coef),
         int(input_idx), int(l1), _p(dpred), lddp, _s())


def image_loss_grad(gt, pred, ldp, batch, s, mask, coef, idx, dpred, lddp, l1=False):
    call('imm_image_loss_grad', _p(gt), _p(pred), ldp, batch, s, _p(mask), _p(coef), int(idx), int(l1), _p(dpred), lddp,
         dtype_enum(dpred.dtype), _s())


def masked_sse(a, b, batch, s, c, mask, S, partial, l1=False):
    call('imm_masked_sse', _p(a), _p(b), dtype_enum(a.dtype), batch, s, c, _p(mask), S, int(l1), _p(partial), _s())


class SseMulti:
    """Argument block of imm_masked_sse_multi: feats = [(a, b, s, c, partial_row)], built once, launched every step."""

    def __init__(self, feats):
        n = len(feats)
        self.n = n
        self.keep = feats
        self.a = (C.c_void_p * n)(*[f[0].data_ptr() for f in feats])
        self.b = (C.c_void_p * n)(*[f[1].data_ptr() for f in feats])
        self.s = (C.c_int32 * n)(*[int(f[2]) for f in feats])
        self.c = (C.c_int32 * n)(*[int(f[3]) for f in feats])
        self.p = (C.c_void_p * n)(*[f[4].data_ptr() for f in feats])
        self.dtype = feats[0][0].dtype


def masked_sse_multi(g, batch, mask, S, l1=False):
    call('imm_masked_sse_multi', g.n, C.cast(g.a, C.c_void_p), C.cast(g.b, C.c_void_p), C.cast(g.s, C.c_void_p), C.cast(g.c, C.c_void_p),
         C.cast(g.p, C.c_void_p), dtype_enum(g.dtype), batch, _p(mask), S, int(l1), _s())


def masked_sse_all(g, batch, mask, S, img_a, lda, img_b, ldb, img_c, img_partial, l1=False):
    """masked_sse_multi(g, ...) and masked_sse_f32(img_a, lda, img_b, ldb, batch, S, img_c, mask, img_partial) in one launch."""
    call('imm_masked_sse_all', g.n, C.cast(g.a, C.c_void_p), C.cast(g.b, C.c_void_p), C.cast(g.s, C.c_void_p), C.cast(g.c, C.c_void_p),
         C.cast(g.p, C.c_void_p), dtype_enum(g.dtype), batch, _p(mask), S, int(l1), _p(img_a), lda, _p(img_b), ldb, img_c,
         _p(img_partial), _s())


def masked_sse_f32(a, lda, b, ldb, batch, s, c, mask, partial, l1=False):
    call('imm_masked_sse_f32', _p(a), lda, _p(b), ldb, batch, s, c, _p(mask), int(l1), _p(partial), _s())


LOSS_PERCEPTUAL, LOSS_L2 = 0, 1


def perceptual_finalize(partial, nfeat, nel, agg, training, wd_loss, out, l1=False, mode=LOSS_PERCEPTUAL, loss_scale=None):
    call('imm_perceptual_finalize', _p(partial), nfeat, _p(nel), _p(agg), int(training), _p(wd_loss), int(l1), int(mode),
         _p(loss_scale), _p(out), _s())


def tap_grad(da, has_in, a_pred, a_gt, batch, s, c, mask, S, coef, idx, relu, l1=False):
    call('imm_tap_grad', _p(da), int(has_in), _p(a_pred), _p(a_gt), dtype_enum(da.dtype), batch, s, c, _p(mask), S,
         _p(coef), idx, int(relu), int(l1), _s())


def copy_f32(dst, src):
    """dst <- src (f32, same numel) by a KERNEL on torch's current stream; either may be a pinned host tensor."""
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.numel() == src.numel()
    assert dst.is_contiguous() and src.is_contiguous() and (dst.is_cuda or dst.is_pinned()) and (src.is_cuda or src.is_pinned())
    call('imm_copy_f32', _p(dst), _p(src), dst.numel(), _s())


def upload(dst, src, what='tensor'):
    """dst (device) <- src (host tensor / ndarray), staged through PINNED memory, stream-synchronised and read back: for the
    ONE-TIME transfers (initial parameters, frozen VGG weights, restored checkpoints).  Per-step inputs go through PinnedStager
    (no read-back, no full synchronisation).

    Why not `dst.copy_(cpu_tensor)`: with eight processes sharing one GPU, about once in 100 engine constructions ONE rank started
    from other initial weights in exactly the 2.4 MB / 2.9 MB filters — uploaded one by one from pageable temporaries that were freed
    (and refilled with the next tensor's random numbers) as soon as `copy_` returned (round 5; found by comparing the replicas'
    parameters before the first update — DESIGN.md §7; the runtime's pin-in-place path for pageable sources above ~1 MB is the
    inferred culprit).  Here the source is pinned (read in place by the DMA engine), the stream is synchronised before the staging
    buffer may go, and the result is read back and compared bitwise.  A mismatch RAISES (round 6: no silent repetition — with the
    cause removed a wrong read-back is a new bug and must be seen)."""
    if not dst.is_cuda:
        dst.copy_(torch.as_tensor(src).reshape(dst.shape))
        return
    src = torch.as_tensor(src)
    if src.is_cuda:
        dst.copy_(src.reshape(dst.shape))
        return
    stage = torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True)
    stage.copy_(src.reshape(dst.shape))                    # host side: layout / dtype conversion into the pinned buffer
    back = torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True)
    stream = torch.cuda.current_stream(dst.device)
    dst.copy_(stage, non_blocking=True)
    back.copy_(dst, non_blocking=True)
    stream.synchronize()
    if not torch.equal(back.reshape(-1).view(torch.uint8), stage.reshape(-1).view(torch.uint8)):
        nbad = int((back.reshape(-1).view(torch.uint8) != stage.reshape(-1).view(torch.uint8)).sum())
        raise RuntimeError('imm_amd: host-to-device upload of %s (%d bytes) does not read back as written (%d bytes differ)'
                           % (what, stage.numel() * stage.element_size(), nbad))


class PinnedStager:
    """Per-step host -> device copies (a training loop fed from host tensors: IMMEngine.set_inputs): every destination has TWO
    persistent pinned staging buffers, used alternately; a buffer is refilled only after the copy that last read it has finished
    (an event per buffer — two steps back, so in steady state the wait is free), the copy itself is asynchronous on the caller's
    stream: no allocation, no read-back, no stream-wide synchronisation per step, and the host side of step n + 1 overlaps the
    device side of step n (VERDICT r5 item 8 / ADVICE r5: the verified `upload` serialised a host-fed loop with the device)."""

    def __init__(self):
        self._slots = {}

    def copy(self, dst, src, key=None):
        src = torch.as_tensor(src)
        if not dst.is_cuda or src.is_cuda:
            dst.copy_(src.reshape(dst.shape))             # device -> device (the loader's / the bench's tensors): stream-ordered
            return
        key = dst.data_ptr() if key is None else key
        ent = self._slots.get(key)
        if ent is None or ent['bufs'][0].shape != dst.shape or ent['bufs'][0].dtype != dst.dtype:
            ent = {'bufs': [torch.empty(dst.shape, dtype=dst.dtype, pin_memory=True) for _ in range(2)], 'evs': [None, None], 'i': 0}
            self._slots[key] = ent
        i = ent['i']
        ent['i'] = 1 - i
        if ent['evs'][i] is not None:
            ent['evs'][i].synchronize()                   # the copy that read this buffer two calls ago
        ent['bufs'][i].copy_(src.reshape(dst.shape))      # host side: layout / dtype conversion into pinned memory
        dst.copy_(ent['bufs'][i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dst.device))
        ent['evs'][i] = ev


def to_device_pinned(src, device, dtype=None):
    """A NEW device tensor holding `src` (host tensor / ndarray), copied from a pinned staging buffer that stays alive until the
    copy has finished (the host waits for THAT copy's event, not for the stream's earlier work).  What scripts and datasets use
    instead of `torch.from_numpy(x).to(device)`: a pageable temporary of more than ~1 MB that is freed right after `.to()` returns
    is the pattern behind round 5's torn upload (`upload` above); tests/test_host_cpu.py greps for it."""
    src = torch.as_tensor(src)
    device = torch.device(device)
    dtype = src.dtype if dtype is None else dtype
    if device.type != 'cuda' or src.is_cuda:
        return src.to(device=device, dtype=dtype)
    stage = torch.empty(src.shape, dtype=dtype, pin_memory=True)
    stage.copy_(src)
    dst = torch.empty(src.shape, dtype=dtype, device=device)
    dst.copy_(stage, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    ev.synchronize()
    return dst


def download(src):
    """Host copy of a device tensor through pinned memory, stream-synchronised (the mirror of upload(): checkpoints are written from
    these); returns an ordinary (pageable) CPU tensor."""
    if not src.is_cuda:
        return src.detach().clone()
    back = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
    back.copy_(src.detach(), non_blocking=True)
    torch.cuda.current_stream(src.device).synchronize()
    return back.clone()


def cost_ema(cost3, state4, decay=0.99):
    """BaseModel._add_cost_summary's moving averages (base_model.py:52-60): state4 = {biased[3], local_step}."""
    call('imm_cost_ema', _p(cost3), _p(state4), float(decay), _s())


def rms16(x, partial, out):
    """sqrt(mean(x^2)) of a contiguous 16-bit tensor (selfsup/vgg16.py:232-234 'activation/<layer>'); partial: f32 scratch."""
    assert x.is_contiguous() and x.numel() % 8 == 0
    call('imm_rms16', _p(x), x.numel(), dtype_enum(x.dtype), _p(partial), min(int(partial.numel()), 4096), _p(out), _s())


# ---- optimizer --------------------------------------------------------------------------------
def weight_decay_loss(params, tab, blk_partial, out):
    call('imm_weight_decay_loss', _p(params), _p(tab.blk_seg), _p(tab.blk_begin), _p(tab.blk_end), tab.nblk, _p(tab.seg_wd),
         _p(blk_partial), _p(out), _s())


def clip_adam_step(params, grads, m, v, tab, blk_partial, seg_norm2, step_count, adam_t, lr_state, hp, loss_scale_state=None):
    call('imm_clip_adam_step', _p(params), _p(grads), _p(m), _p(v), _p(tab.blk_seg), _p(tab.blk_begin), _p(tab.blk_end),
         tab.nblk, tab.nseg, _p(tab.seg_first_blk), _p(tab.seg_wd), _p(blk_partial), _p(seg_norm2), _p(step_count),
         _p(adam_t), _p(lr_state), C.byref(hp), _p(loss_scale_state), _s())


class SegmentTable:
    """Device-side description of the flat parameter buffer: tensors (segments) and the chunk each
    optimizer workgroup owns."""
    CHUNK = 2048     # 8192: 37 us for the update launch, 4096: 34, 2048: 32 (more 16-byte requests in flight per CU)

    def __init__(self, sizes, wds, device):
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + int(n))
        blk_seg, blk_begin, blk_end, first = [], [], [], [0]
        for s, n in enumerate(sizes):
            for b0 in range(0, int(n), self.CHUNK):
                blk_seg.append(s)
                blk_begin.append(offs[s] + b0)
                blk_end.append(offs[s] + min(int(n), b0 + self.CHUNK))
            first.append(len(blk_seg))
        self.offsets = offs
        self.nseg, self.nblk, self.total = len(sizes), len(blk_seg), offs[-1]
        i32 = dict(dtype=torch.int32, device=device)
        self.blk_seg = torch.tensor(blk_seg, **i32)
        self.blk_begin = torch.tensor(blk_begin, **i32)
        self.blk_end = torch.tensor(blk_end, **i32)
        self.seg_first_blk = torch.tensor(first, **i32)
        self.seg_wd = torch.tensor([float(w) for w in wds], dtype=torch.float32, device=device)


def debug_stamp(slots, index):
    """slots: int64 device tensor; slots[index] = device wall clock (100 MHz ticks) at this point of the current stream."""
    call('imm_debug_stamp', _p(slots), int(index), _s())


def tps_warp(src, basis_t, w_tps, dst=None, dst_c0=None, dst_rest=None):
    """src [B,H,W,C] f32 NHWC; basis_t [M+3, H*W]; w_tps [B, M+3, 2]; outputs as in include/imm_hip.h (imm_tps_warp)."""
    b, h, w, c = src.shape
    call('imm_tps_warp', _p(src), src.stride(2), b, h, w, c, _p(basis_t), basis_t.shape[0], _p(w_tps), _p(dst),
         dst.stride(2) if dst is not None else 0, _p(dst_c0), _p(dst_rest), dst_rest.stride(2) if dst_rest is not None else 0, _s())


def tps_warp_pad(src, basis_t, w_tps, pad_yx, grid_hw, crop_yx, dst):
    """TPSRandomSampler(pad=True): src [B,H,W,C] f32 NHWC read as if replicate-padded by pad_yx, warped on a grid_hw grid
    (basis_t [M+3, gh*gw]), window crop_yx + dst.shape[1:3] written to dst [B,oh,ow,C] (include/imm_hip.h: imm_tps_warp_pad)."""
    b, h, w, c = src.shape
    call('imm_tps_warp_pad', _p(src), src.stride(2), b, h, w, c, pad_yx[0], pad_yx[1], grid_hw[0], grid_hw[1], crop_yx[0], crop_yx[1],
         dst.shape[1], dst.shape[2], _p(basis_t), basis_t.shape[0], _p(w_tps), _p(dst), dst.stride(2), _s())


def resize_crop_u8(src_u8, offsets, hw, c, resize_hw, crop_yx, out_hw, dst, ld_dst=None):
    """src_u8: flat u8 device buffer of packed HWC images; offsets i64 [B]; hw i32 [B,2]; dst f32 view whose element
    (b,y,x,0) is dst.data_ptr() + ((b*oh+y)*ow+x)*ld_dst floats (include/imm_hip.h: imm_resize_crop_u8)."""
    call('imm_resize_crop_u8', _p(src_u8), _p(offsets), _p(hw), hw.shape[0], c, resize_hw[0], resize_hw[1], crop_yx[0], crop_yx[1],
         out_hw[0], out_hw[1], _p(dst), dst.stride(2) if ld_dst is None else ld_dst, _s())


def masked_sse_pool(a, b, batch, s, c, mask, S, partial, pool_a, pool_b):
    call('imm_masked_sse_pool', _p(a), _p(b), dtype_enum(a.dtype), batch, s, c, _p(mask), S, _p(partial), _p(pool_a), _p(pool_b), _s())


def unpool_tap_grad(da, dpool, a_pred, a_gt, batch, s, c, mask, S, coef, idx):
    call('imm_unpool_tap_grad', _p(da), _p(dpool), _p(a_pred), _p(a_gt), dtype_enum(da.dtype), batch, s, c, _p(mask), S,
         _p(coef), idx, _s())


class ConvGroup(object):
    """Descriptor array + filter pointer array for imm_conv2d_group (kept alive by this object)."""

    def __init__(self, descs, wts):
        assert 1 <= len(descs) <= 4 and len(descs) == len(wts)
        self.n = len(descs)
        self.descs = (ConvDesc * self.n)(*descs)
        self.wts = wts
        self.ptrs = (C.c_void_p * self.n)(*[w.data_ptr() for w in wts])


def conv2d_group(group, x, y, stats=None, mask=None):
    call('imm_conv2d_group', C.cast(group.descs, C.c_void_p), group.n, dtype_enum(x.dtype), _p(x), C.cast(group.ptrs, C.c_void_p),
         _p(y), _p(stats), _p(mask), _s())


def conv_first_supported(batch, s, co, ldy):
    return bool(L.load().imm_conv_first_supported(batch, s, co, ldy))


def conv_first_stats_blocks(batch, s):
    n = L.load().imm_conv_first_stats_blocks(batch, s)
    if n <= 0:
        raise L.ImmHipError('imm_conv_first_stats_blocks(%d, %d)' % (batch, s))
    return n


def conv_first(image, wt, bias, y, ldy, stats, batch, s, co, flags):
    """First encoder convolution (7x7 over RGB) straight from the f32 image (imm_conv_first); wt = the 7x1 packed filter image."""
    call('imm_conv_first', _p(image), _p(wt), int(wt.shape[1]), _p(bias), _p(y), ldy, _p(stats), dtype_enum(y.dtype), batch, s, co,
         flags, _s())


def conv2d_nol_supported(desc):
    """imm_conv2d_nol (normalise on load) serves this forward convolution?"""
    return bool(L.load().imm_conv2d_nol_supported(C.byref(desc)))


def conv2d_nol_stats_blocks(desc):
    n = L.load().imm_conv2d_nol_stats_blocks(C.byref(desc))
    if n <= 0:
        raise L.ImmHipError('imm_conv2d_nol_stats_blocks: ' + L.load().imm_last_error().decode())
    return n


def conv2d_nol(desc, x_raw, x_scale, x_shift, x_relu, wt, bias, y, stats=None):
    """Convolution of relu(x_scale * x_raw + x_shift) without that tensor being stored (imm_conv2d_nol)."""
    call('imm_conv2d_nol', C.byref(desc), dtype_enum(x_raw.dtype), _p(x_raw), _p(x_scale), _p(x_shift), int(bool(x_relu)), _p(wt),
         _p(bias), _p(y), _p(stats), _s())


def conv2d_dgrad_s2_supported(batch, h, w, lddy, c_dx, lddx):
    """dy [batch,h,w] with pixel stride lddy -> dx [batch,2h,2w,c_dx]: served by the one-launch four-class kernel?"""
    return bool(L.load().imm_conv2d_dgrad_s2_supported(batch, h, w, lddy, c_dx, lddx))


def conv2d_dgrad_s2(dy, lddy, wt, dx, lddx, c_dx, batch, h, w):
    """Data gradient of a 3x3 stride-2 SAME convolution as ONE launch (imm_conv2d_dgrad_s2): wt = the mode-1 packed image
    [rows >= c_dx][9 * lddy]."""
    call('imm_conv2d_dgrad_s2', _p(dy), lddy, _p(wt), int(wt.shape[1]), _p(dx), lddx, c_dx, dtype_enum(dy.dtype), batch, h, w, _s())


def conv2d_group_stats_blocks(group):
    n = L.load().imm_conv2d_group_stats_blocks(C.cast(group.descs, C.c_void_p), group.n)
    if n <= 0:
        raise L.ImmHipError('imm_conv2d_group_stats_blocks: ' + L.load().imm_last_error().decode())
    return n


# ---- RCCL through the C-ABI (imm_rccl_*): the data-parallel gradient exchange without torch.distributed in the loop ----------
class RcclComm:
    """Communicator for the current device.  The 128-byte unique id is made by rank 0 and exchanged over the (already
    initialised) torch.distributed process group, used here as the host-side side channel only."""

    def __init__(self, rank, world, group=None):
        import torch.distributed as dist
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            call('imm_rccl_unique_id', C.cast(buf, C.c_void_p))
            uid = torch.tensor(list(buf), dtype=torch.uint8)
        if world > 1:
            obj = [uid]
            # `rank` is the rank inside `group`; broadcast's src is a GLOBAL rank
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(obj, src=src, group=group)
            uid = obj[0]
        self._uid = (C.c_ubyte * 128)(*[int(x) for x in uid])
        self._comm = C.c_void_p()
        call('imm_rccl_init', rank, world, C.cast(self._uid, C.c_void_p), C.byref(self._comm))
        self.world = world

    def all_reduce_sum(self, flat_f32):
        """In-place sum over ranks, enqueued on torch's current stream (capturable into a HIP graph)."""
        assert flat_f32.dtype == torch.float32 and flat_f32.is_contiguous()
        call('imm_rccl_allreduce', self._comm, _p(flat_f32), flat_f32.numel(), _s())

    def destroy(self):
        if self._comm:
            call('imm_rccl_destroy', self._comm)
            self._comm = C.c_void_p()
