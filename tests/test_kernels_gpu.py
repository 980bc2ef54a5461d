"""Per-kernel parity: every C-ABI entry point of libimm_hip.so against the CPU oracle, on a real MI355X.

Inputs are seeded; 16-bit kernels are fed the same bf16/f16-rounded values the oracle sees, so the
only differences are accumulation order and the final rounding of the stored result.  Tolerances are
written next to each comparison.
"""
import math

import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'
L_CONV_BIAS_F32 = 1 | 16       # IMM_CONV_BIAS | IMM_CONV_OUT_F32


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd import ops as _ops
    torch.cuda.set_device(0)
    return _ops


def rnd(shape, seed, scale=1.0, dt=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dt)


def close(got, ref, rtol, atol_frac, what):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = float(ref.abs().max()) + 1e-30
    err = (got - ref).abs()
    tol = atol_frac * scale + rtol * ref.abs()
    bad = err > tol
    assert not bool(bad.any()), '%s: %d/%d elements off, max err %.4g (ref max %.4g), first bad idx %s got %.6g ref %.6g' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), scale,
        tuple(int(i) for i in bad.nonzero()[0]), float(got[bad][0]), float(ref[bad][0]))


def padded(x, ld):
    """[..., c] -> [..., ld] zero padded, contiguous, on device."""
    out = torch.zeros(x.shape[:-1] + (ld,), dtype=x.dtype)
    out[..., :x.shape[-1]] = x
    return out.to(DEV).contiguous()


# ----------------------------------------------------------------------------------------------
# convolution forward
# ----------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, ci_real, ci_pad, co, k, stride, out_f32, tag
    (2, 16, 32, 32, 32, 3, 1, False, 'enc3x3'),
    (2, 16, 32, 32, 64, 3, 2, False, 'stride2'),
    (2, 32, 3, 8, 32, 7, 1, False, 'first7x7'),
    (2, 16, 256, 256, 10, 1, 1, True, 'pose1x1'),
    (1, 32, 32, 32, 9, 3, 1, True, 'final9'),
    (2, 16, 266, 288, 256, 3, 1, False, 'concat266'),
    (1, 10, 64, 64, 64, 3, 1, False, 'ragged_m100'),
    (4, 128, 32, 32, 128, 3, 1, False, 'tile128x128'),
    (4, 128, 32, 32, 64, 3, 1, False, 'tile128x64'),
    (2, 8, 512, 512, 512, 3, 1, False, 'vgg5'),
    (2, 16, 256, 256, 30, 1, 1, True, 'pose1x1_k30'),
    # 3x3 s1, ci in {32,64}, co<=64, >=64x64: served by the LDS-resident-filter halo kernel (conv_halo.hip)
    (2, 64, 64, 64, 64, 3, 1, False, 'halo_64_64'),
    (1, 128, 32, 32, 32, 3, 1, False, 'halo_32_32'),
    (3, 64, 32, 32, 9, 3, 1, True, 'halo_32_9_f32'),
    (2, 64, 64, 64, 32, 3, 1, False, 'halo_64_32'),
    (70, 64, 32, 32, 64, 3, 1, False, 'halo_32_64_many_patches'),
    # 3x3 STRIDE 2, 32 -> 64 channels, >= 32x32 outputs: parity-de-interleaved LDS-halo kernel (conv_halo.hip, encoder conv_3)
    (2, 128, 32, 32, 64, 3, 2, False, 'halo_s2_128px'),
    (33, 64, 32, 32, 64, 3, 2, False, 'halo_s2_64px_many_patches'),
    (1, 64, 32, 32, 48, 3, 2, False, 'halo_s2_co48'),
    # 3x3 s1, ci % 64 == 0, co % 64 == 0, maps % 16 == 0, >= 192 workgroups: LDS-halo deep-K kernel (conv_hdeep.hip)
    (16, 64, 64, 64, 128, 3, 1, False, 'hdeep_bn128_one_slice'),
    (13, 32, 128, 128, 256, 3, 1, False, 'hdeep_bn64_two_slices'),
    (64, 16, 128, 128, 512, 3, 1, False, 'hdeep_bn128_whole_image_patches'),
    (7, 64, 192, 192, 128, 3, 1, False, 'hdeep_three_slices'),
    # more tiles than CUs: one persistent workgroup per CU walks them (ragged: some take two tiles, some one; one / two slices)
    (20, 64, 128, 128, 128, 3, 1, False, 'hdeep_persistent_ragged'),
    (9, 64, 64, 64, 256, 3, 1, False, 'hdeep_persistent_one_slice_two_nblk'),
    (130, 32, 128, 128, 64, 3, 1, False, 'hdeep_persistent_bn64'),   # 520 16x16x64 tiles >= 2 x CUs: 8 waves, persistent
    # small grids: 8x16 patches, 4 waves, two workgroups per CU
    (32, 16, 128, 128, 256, 3, 1, False, 'hdeep_small_patch'),
    (5, 40, 64, 64, 64, 3, 1, False, 'hdeep_small_patch_h40'),
    # 8x8 maps (VGG conv5): two whole images per workgroup; odd batch = a half-empty last pair
    (64, 8, 512, 512, 512, 3, 1, False, 'hdeep_map8'),
    (33, 8, 128, 128, 256, 3, 1, False, 'hdeep_map8_odd_batch'),
]


# The kernel family imm_conv2d must dispatch each case to on a 256-CU part (imm_conv2d_variant): a case named after a kernel proves
# nothing unless that kernel is the one that ran.  Tags without an entry fall to the im2col kernels.
CONV_FAMILY = {
    'tile128x64': 'halo2', 'halo_64_64': 'halo2', 'halo_32_32': 'halo2', 'halo_64_32': 'halo2', 'halo_32_64_many_patches': 'halo2',
    'halo_32_9_f32': 'halo', 'halo_s2_128px': 'halo', 'halo_s2_64px_many_patches': 'halo', 'halo_s2_co48': 'halo',
    'hdeep_bn128_one_slice': 'hdeep6', 'hdeep_bn128_whole_image_patches': 'hdeep6', 'hdeep_persistent_ragged': 'hdeep6',
    'hdeep_persistent_one_slice_two_nblk': 'hdeep6',
    'hdeep_bn64_two_slices': 'hdeep', 'hdeep_three_slices': 'hdeep', 'hdeep_persistent_bn64': 'hdeep', 'hdeep_small_patch': 'hdeep',
    'hdeep_map8': 'hdeep',
    # below the deep-K LDS-halo kernel's minimum grid (100 workgroups): the BK = 64 im2col kernel
    'hdeep_small_patch_h40': 'igemm64', 'hdeep_map8_odd_batch': 'igemm64', 'ragged_m100': 'igemm64', 'vgg5': 'igemm64',
    'hdeep_256to128': 'hdeep', 'hdeep_512to256': 'hdeep', 'ci266': 'igemm64', 'k3s2_b': 'igemm64',
}
# variant digits of the hdeep family (imm_hdeep_variant): persistent 10000, row at a time 20000, 8x8 maps 40000
CONV_VARIANT = {'hdeep_persistent_bn64': 510648, 'hdeep_persistent_ragged': 600001, 'hdeep_bn128_one_slice': 600000,
                'hdeep_map8': 560644}


def check_family(ops, desc, dt, tag):
    fam, key = ops.conv2d_variant(desc, dt)
    assert fam == CONV_FAMILY.get(tag, 'igemm'), (tag, fam, key)
    if tag in CONV_VARIANT:
        assert key == CONV_VARIANT[tag], (tag, key)


def run_conv(ops, x16, w, bias, k, stride, co, ci_pad, out_f32, extra_flags=0, mask=None, ldy=None):
    from imm_amd import _lib as L
    B, H, W, _ = x16.shape
    dt = x16.dtype
    ci_real = w.shape[2]
    xd = padded(x16, ci_pad)
    flags = extra_flags | (L.CONV_BIAS if bias is not None else 0) | (L.CONV_OUT_F32 if out_f32 else 0)
    ldy = ldy or ops.round_up(co, 8 if not out_f32 else 4)
    desc = ops.fwd_desc(B, H, W, ci_pad, ci_pad, co, ldy, k, stride, flags, ldmask=(mask.shape[-1] if mask is not None else 0))
    rows = ops.round_up(co, 128)
    wt = torch.zeros(rows, desc.kpad, dtype=dt, device=DEV)
    wd = w.float().to(DEV).contiguous()
    ops.pack_weights(wd, wt, 0, k, k, ci_real, co, ci_pad, rows, desc.kpad)
    y = torch.full((B, desc.ho, desc.wo, ldy), float('nan'), dtype=torch.float32 if out_f32 else dt, device=DEV)
    stats = None
    if extra_flags & L.CONV_STATS:
        stats = torch.full((ops.conv_stats_blocks(desc), 2, co), float('nan'), dtype=torch.float32, device=DEV)
    bd = bias.float().to(DEV) if bias is not None else None
    ops.conv2d(desc, xd, wt, bd, y, stats, mask)
    torch.cuda.synchronize()
    return y, stats, desc


@pytest.mark.parametrize('case', CONV_CASES, ids=[c[-1] for c in CONV_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_forward(ops, case, dt):
    B, H, ci_real, ci_pad, co, k, stride, out_f32, tag = case
    x = rnd((B, H, H, ci_real), 1, 1.0, dt)
    w = rnd((k, k, ci_real, co), 2, 0.05, dt)
    b = rnd((co,), 3, 0.5, torch.float32)
    y, _, desc = run_conv(ops, x, w, b, k, stride, co, ci_pad, out_f32)
    check_family(ops, desc, dt, tag)
    ref = O.conv2d_same(x.float(), w.float(), b, stride)
    # f32 out: accumulation-order error only; 16-bit out: one rounding (2^-8 bf16, 2^-11 f16)
    rt = 2e-3 if out_f32 else (1e-2 if dt == torch.bfloat16 else 2e-3)
    close(y[..., :co], ref, rt, 2e-3 if not out_f32 else 2e-4, 'conv_fwd/' + tag)
    if y.shape[-1] > co and not out_f32:
        pass  # padding channels are not written by the kernel (caller owns them)


@pytest.mark.parametrize('B,H,ci,co,dt', [(2, 32, 32, 64, torch.bfloat16), (2, 64, 64, 64, torch.bfloat16),
                                          (2, 64, 32, 32, torch.bfloat16), (3, 64, 64, 32, torch.bfloat16),
                                          (3, 64, 32, 64, torch.bfloat16), (40, 64, 64, 64, torch.bfloat16),
                                          (3, 128, 64, 64, torch.float16), (36, 64, 32, 32, torch.float16),
                                          (14, 32, 128, 256, torch.bfloat16), (64, 16, 256, 256, torch.float16),
                                          (24, 16, 128, 256, torch.bfloat16), (51, 8, 128, 256, torch.bfloat16),
                                          (20, 64, 128, 128, torch.bfloat16), (24, 64, 128, 128, torch.float16)],
                         ids=['igemm', 'halo64', 'halo32', 'halo64_32', 'halo32_64', 'halo64_persistent', 'halo64_f16',
                              'halo32_f16_persistent', 'hdeep', 'hdeep_f16', 'hdeep_small_patch', 'hdeep_map8_odd',
                              'hdeep_persistent', 'hdeep_persistent_f16'])
def test_conv_relu_stats_mask(ops, B, H, ci, co, dt):
    """Epilogue variants (BN partial sums, ReLU, ReLU-backward mask).  The halo cases run conv_halo2.hip (filter in
    registers, deferred epilogue); the *_persistent cases give every workgroup several patches, i.e. exercise the halo /
    mask DMA rings and the counted waits in steady state."""
    from imm_amd import _lib as L
    x = rnd((B, H, H, ci), 4, 1.0, dt)
    w = rnd((3, 3, ci, co), 5, 0.1, dt)
    b = rnd((co,), 6, 0.5, torch.float32)
    y, stats, desc = run_conv(ops, x, w, b, 3, 1, co, ci, False, extra_flags=L.CONV_STATS)
    ref = O.conv2d_same(x.float(), w.float(), b, 1)
    close(y, ref, 1e-2, 2e-3, 'conv+stats/y')
    y1, _, desc1 = run_conv(ops, x, w, b, 3, 1, co, ci, False)
    if ops.conv2d_variant(desc1, dt) == ops.conv2d_variant(desc, dt):
        assert torch.equal(y1, y), 'the stats flag must not change the output'
    else:
        # the 16x16x128 tile without partial sums runs conv_hdeep6.hip (32-channel K slices: another accumulation order)
        assert ops.conv2d_variant(desc1, dt)[0] == 'hdeep6' and ops.conv2d_variant(desc, dt)[0] == 'hdeep'
        close(y1, y, 1e-2 if dt == torch.bfloat16 else 2e-3, 1e-3, 'conv+stats vs conv (other kernel)')
    s = stats.sum(dim=0).cpu()
    close(s[0], ref.sum(dim=(0, 1, 2)), 1e-3, 1e-3, 'conv+stats/sum')      # f32 sums of f32 accumulators
    close(s[1], (ref ** 2).sum(dim=(0, 1, 2)), 1e-3, 1e-3, 'conv+stats/sumsq')
    y2, _, _ = run_conv(ops, x, w, b, 3, 1, co, ci, False, extra_flags=L.CONV_RELU)
    close(y2, torch.relu(ref), 1e-2, 2e-3, 'conv+relu')
    mref = rnd((B, H, H, co), 7, 1.0, dt).to(DEV).contiguous()
    y3, _, _ = run_conv(ops, x, w, b, 3, 1, co, ci, False, extra_flags=L.CONV_MASK, mask=mref)
    close(y3, ref * (mref.float().cpu() > 0), 1e-2, 2e-3, 'conv+mask')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_stride2_halo_bias_and_bn_sums(ops, dt):
    """The stride-2 LDS-halo forward as the encoder uses it (conv_3: bias + batch-norm partial sums, one row per persistent
    workgroup) against the oracle, and against the im2col kernel it replaces (IMM_CONV_DISABLE is per process: the im2col
    result comes from a shape just below the halo kernel's threshold instead — same arithmetic, other tiling)."""
    from imm_amd import _lib as L
    B, H, ci, co = 3, 128, 32, 64
    x = rnd((B, H, H, ci), 181, 1.0, dt)
    w = rnd((3, 3, ci, co), 182, 0.05, dt)
    b = rnd((co,), 183, 0.5, torch.float32)
    y, stats, desc = run_conv(ops, x, w, b, 3, 2, co, ci, False, extra_flags=L.CONV_STATS)
    assert (desc.ho, desc.pad_t, desc.pad_l) == (64, 0, 0) and stats.shape[0] <= 256
    ref = O.conv2d_same(x.float(), w.float(), b, 2)
    close(y, ref, 1e-2 if dt == torch.bfloat16 else 2e-3, 2e-3, 'halo_s2+bias')
    st = stats.sum(dim=0).cpu()
    close(st[0], ref.sum(dim=(0, 1, 2)), 1e-3, 1e-3, 'halo_s2/sum')
    close(st[1], (ref ** 2).sum(dim=(0, 1, 2)), 1e-3, 1e-3, 'halo_s2/sumsq')


S2F_CASES = [
    # B, H, ci, co, expected channel block, tag        (3x3 stride 2, ci % 64 == 0, co % 64 == 0: conv_s2f.hip)
    (32, 64, 64, 128, 128, 'enc_conv5_64to128'),      # 256 patches x one 128-channel block: 8 waves
    (32, 32, 128, 256, 64, 'enc_conv7_128to256'),     # 64 patches: 128-wide blocks would leave half the chip idle -> 64 (4 waves)
    (3, 32, 64, 64, 64, 'one_nblk_two_slices'),
    (2, 32, 192, 192, 64, 'six_slices_three_nblk'),     # two patches per image (32x32 -> 16x16 outputs), 192 = six 32-channel slices
    (5, 48, 64, 128, 64, 'ragged_rows_h48'),           # 24x24 outputs: wo % 16 != 0 -> NOT served (falls to the im2col kernel)
]


@pytest.mark.parametrize('case', S2F_CASES, ids=[c[-1] for c in S2F_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_stride2_forward_lds_halo(ops, case, dt):
    """Encoder conv_5 / conv_7 (imm_model.py:204,211): the stride-2 forward convolution through the parity-de-interleaved LDS
    halo (conv_s2f.hip) against the oracle — bias + batch-norm partial sums as the encoder uses it, the zero padding of S1 (0 top /
    left, 1 bottom / right: the last output row / column reads one row / column of zeros), several slices and channel blocks."""
    from imm_amd import _lib as L
    B, H, ci, co, bn, tag = case
    x = rnd((B, H, H, ci), 191, 1.0, dt)
    w = rnd((3, 3, ci, co), 192, 0.05, dt)
    b = rnd((co,), 193, 0.5, torch.float32)
    y, stats, desc = run_conv(ops, x, w, b, 3, 2, co, ci, False, extra_flags=L.CONV_STATS)
    fam, key = ops.conv2d_variant(desc, dt)
    if tag.startswith('ragged'):
        assert fam == 'igemm64', (fam, key)
    else:
        assert (fam, key) == ('s2f', 700000 + bn), (fam, key)
        assert stats.shape[0] == B * (H // 16) * (H // 32)
    assert (desc.ho, desc.pad_t, desc.pad_l) == (H // 2, 0, 0)
    ref = O.conv2d_same(x.float(), w.float(), b, 2)
    close(y, ref, 1e-2 if dt == torch.bfloat16 else 2e-3, 2e-3, 's2f/' + tag)
    st = stats.sum(dim=0).cpu()
    close(st[0], ref.sum(dim=(0, 1, 2)), 1e-3, 1e-3, 's2f/sum')
    close(st[1], (ref ** 2).sum(dim=(0, 1, 2)), 1e-3, 1e-3, 's2f/sumsq')
    y1, _, _ = run_conv(ops, x, w, None, 3, 2, co, ci, False)               # no bias, no sums
    close(y1, O.conv2d_same(x.float(), w.float(), None, 2), 1e-2 if dt == torch.bfloat16 else 2e-3, 2e-3, 's2f/nobias/' + tag)


# ----------------------------------------------------------------------------------------------
# data gradient
# ----------------------------------------------------------------------------------------------
DGRAD_CASES = [(2, 16, 32, 32, 32, 32, 3, 1, 'k3s1'), (2, 16, 32, 32, 64, 64, 3, 2, 'k3s2'),
               (2, 16, 256, 256, 10, 16, 1, 1, 'k1_co10'), (1, 32, 32, 32, 9, 16, 3, 1, 'k3_co9'),
               (2, 16, 266, 288, 256, 256, 3, 1, 'ci266'), (2, 32, 64, 64, 128, 128, 3, 2, 'k3s2_b'),
               (16, 32, 128, 128, 256, 256, 3, 1, 'hdeep_256to128'), (32, 16, 256, 256, 512, 512, 3, 1, 'hdeep_512to256')]


@pytest.mark.parametrize('case', DGRAD_CASES, ids=[c[-1] for c in DGRAD_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_dgrad(ops, case, dt):
    B, H, ci_real, ci_pad, co, co_pad, k, stride, tag = case
    w = rnd((k, k, ci_real, co), 11, 0.05, dt)
    xr = torch.zeros(B, H, H, ci_real, requires_grad=True)
    yref = O.conv2d_same(xr, w.float(), None, stride)
    dy = rnd(tuple(yref.shape), 12, 1.0, dt)
    (gx,) = torch.autograd.grad(yref, xr, dy.float())
    desc = ops.dgrad_desc(B, H, H, ci_real, ci_pad, co_pad, co_pad, k, stride, 0)
    check_family(ops, desc, dt, tag)
    rows = ops.round_up(ci_real, 128)
    wt = torch.zeros(rows, desc.kpad, dtype=dt, device=DEV)
    ops.pack_weights(w.float().to(DEV).contiguous(), wt, 1, k, k, ci_real, co, co_pad, rows, desc.kpad)
    dx = torch.zeros(B, H, H, ci_pad, dtype=dt, device=DEV)
    ops.conv2d(desc, padded(dy, co_pad), wt, None, dx)
    torch.cuda.synchronize()
    close(dx[..., :ci_real], gx, 1e-2 if dt == torch.bfloat16 else 2e-3, 2e-3, 'dgrad/' + tag)


@pytest.mark.parametrize('H,ci,co', [(16, 32, 64), (32, 64, 128), (64, 128, 256), (64, 32, 64)])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_dgrad_stride2_parity_classes(ops, H, ci, co, dt):
    """Stride-2 data gradient as four dense sub-convolutions (one per input-pixel parity class) scattered into dx."""
    B, k = 2, 3
    w = rnd((k, k, ci, co), 13, 0.05, dt)
    xr = torch.zeros(B, H, H, ci, requires_grad=True)
    yref = O.conv2d_same(xr, w.float(), None, 2)
    dy = rnd(tuple(yref.shape), 14, 1.0, dt)
    (gx,) = torch.autograd.grad(yref, xr, dy.float())
    descs = ops.dgrad_s2_class_descs(B, H, H, ci, ci, co, co, k)
    assert descs is not None and len(descs) == 4 and sorted(d.kh * d.kw for d, _ in descs) == [1, 2, 2, 4]
    dx = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    dyd = dy.to(DEV).contiguous()
    rows = ops.round_up(ci, 128)
    for d, mode in descs:
        wt = torch.zeros(rows, d.kpad, dtype=dt, device=DEV)
        ops.pack_weights(w.float().to(DEV).contiguous(), wt, mode, k, k, ci, co, co, rows, d.kpad)
        ops.conv2d(d, dyd, wt, None, dx)
    torch.cuda.synchronize()
    close(dx, gx, 1e-2, 2e-3, 'dgrad_s2_classes')       # every pixel written exactly once (no NaN left)


NOL_CASES = [
    # B, H, ci, co, stride, out_f32, tag           (the shapes whose input is a batch-norm output: encoder conv_2/3/4, renderer conv_6/8)
    (2, 128, 32, 32, 1, False, 'enc_conv2_32_32'),
    (2, 128, 32, 64, 2, False, 'enc_conv3_stride2'),
    (2, 64, 64, 64, 1, False, 'enc_conv4_64_64'),
    (3, 64, 64, 32, 1, False, 'halo_64_32'),
    (1, 128, 32, 9, 1, True, 'ren_conv8_f32_head'),
    (70, 64, 32, 64, 1, False, 'many_patches_32_64'),
    (33, 64, 32, 64, 2, False, 'stride2_many_patches'),
]


def _nol_inputs(B, H, ci, dt, seed):
    """raw conv output y (16-bit), per-channel scale (both signs) / shift, and the normalised tensor the stand-alone apply pass
    stores: relu(scale * y + shift) rounded to 16 bits."""
    from imm_amd import ops as _ops
    y = rnd((B, H, H, ci), seed, 1.0, dt).to(DEV).contiguous()
    g = torch.Generator().manual_seed(seed + 1)
    scale = (torch.randn(ci, generator=g) * 0.8 + 0.3).to(DEV)
    shift = (torch.randn(ci, generator=g) * 0.7 + 0.4).to(DEV)      # mostly positive: relu(shift) != 0, so a normalised padding pixel shows
    out = torch.empty_like(y)
    _ops.bn_apply_relu(y, B * H * H, ci, ci, scale, shift, True, out, ci)
    torch.cuda.synchronize()
    return y, scale, shift, out


@pytest.mark.parametrize('case', NOL_CASES, ids=[c[-1] for c in NOL_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_norm_on_load(ops, case, dt):
    """imm_conv2d_nol: the convolution reads the RAW output of the conv + BN + ReLU block in front of it and applies the affine +
    ReLU in its LDS halo tile — against the two-launch path (imm_bn_apply_relu, then imm_conv2d on the stored tensor: same
    16-bit operands, accumulation order may differ) and against the oracle convolution of that tensor; the batch-norm partial
    sums of ITS output too."""
    from imm_amd import _lib as L
    B, H, ci, co, stride, out_f32, _tag = case
    y, scale, shift, out = _nol_inputs(B, H, ci, dt, 900)
    w = rnd((3, 3, ci, co), 902, 0.05, dt)
    bias = rnd((co,), 903, 0.1, torch.float32).float()
    flags = L.CONV_BIAS | L.CONV_STATS | (L.CONV_OUT_F32 if out_f32 else 0)
    ldy = ops.round_up(co, 4 if out_f32 else 8)
    desc = ops.fwd_desc(B, H, H, ci, ci, co, ldy, 3, stride, flags)
    assert ops.conv2d_nol_supported(desc)
    rows = ops.round_up(co, 128)
    wt = torch.zeros(rows, desc.kpad, dtype=dt, device=DEV)
    ops.pack_weights(w.float().to(DEV).contiguous(), wt, 0, 3, 3, ci, co, ci, rows, desc.kpad)
    bd = bias.to(DEV)
    odt = torch.float32 if out_f32 else dt
    z_nol = torch.full((B, desc.ho, desc.wo, ldy), float('nan'), dtype=odt, device=DEV)
    st_nol = torch.full((ops.conv2d_nol_stats_blocks(desc), 2, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv2d_nol(desc, y, scale, shift, True, wt, bd, z_nol, st_nol)
    z_ref = torch.full((B, desc.ho, desc.wo, ldy), float('nan'), dtype=odt, device=DEV)
    st_ref = torch.full((ops.conv_stats_blocks(desc), 2, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv2d(desc, out, wt, bd, z_ref, st_ref)
    torch.cuda.synchronize()
    zo = O.conv2d_same(out.float().cpu(), w.float(), bias, stride)
    close(z_nol[..., :co], zo, 1e-2 if not out_f32 else 2e-3, 2e-3, 'conv_nol vs oracle')
    close(z_nol[..., :co], z_ref[..., :co], (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) if not out_f32 else 1e-4, 1e-4,
          'conv_nol vs apply + conv')
    close(st_nol.sum(0), st_ref.sum(0), 2e-4, 2e-4, 'conv_nol batch-norm partial sums')
    # without the ReLU (a block built with relu = False)
    ops.conv2d_nol(desc, y, scale, shift, False, wt, bd, z_nol, st_nol)
    out2 = torch.empty_like(y)
    ops.bn_apply_relu(y, B * H * H, ci, ci, scale, shift, False, out2, ci)
    ops.conv2d(desc, out2, wt, bd, z_ref, st_ref)
    torch.cuda.synchronize()
    close(z_nol[..., :co], z_ref[..., :co], (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) if not out_f32 else 1e-4, 1e-4,
          'conv_nol (no relu) vs apply + conv')


def test_conv_norm_on_load_rejects_unserved_shapes(ops):
    from imm_amd import _lib as L
    assert not ops.conv2d_nol_supported(ops.fwd_desc(2, 32, 32, 128, 128, 128, 128, 3, 1, L.CONV_BIAS))     # deep layer: hdeep
    assert not ops.conv2d_nol_supported(ops.fwd_desc(2, 64, 64, 64, 64, 128, 128, 3, 2, L.CONV_BIAS))       # stride 2 from 64 channels
    assert not ops.conv2d_nol_supported(ops.fwd_desc(2, 128, 128, 32, 32, 32, 32, 3, 1, L.CONV_BIAS | L.CONV_RELU))
    d = ops.fwd_desc(2, 32, 32, 128, 128, 128, 128, 3, 1, L.CONV_BIAS)
    x = torch.zeros(2, 32, 32, 128, dtype=torch.bfloat16, device=DEV)
    v = torch.zeros(128, device=DEV)
    with pytest.raises(L.ImmHipError):
        ops.conv2d_nol(d, x, v, v, True, torch.zeros(128, d.kpad, dtype=torch.bfloat16, device=DEV), v, torch.empty_like(x))
    # a filter-gradient job with normalise-on-load must be an LDS-halo variant
    d8 = ops.fwd_desc(2, 8, 8, 128, 128, 128, 128, 3, 1, 0)            # 8x8 maps: the transpose-read kernel's job
    assert ops.conv2d_wgrad_variant(d8, 128, torch.bfloat16)[0] // 100000 != 2
    x8 = torch.zeros(2, 8, 8, 128, dtype=torch.bfloat16, device=DEV)
    slab = torch.zeros(2, d8.kpad, 128, device=DEV)
    with pytest.raises(L.ImmHipError):
        ops.WgradMulti([(d8, x8, x8, 128, slab, 2, (v, v, True))], torch.bfloat16)


NOL_WGRAD_CASES = [
    # B, H, ci, co, lddy, stride, nsplit, tag
    (2, 128, 32, 32, 32, 1, 5, 'enc_conv2'),
    (2, 128, 32, 64, 64, 2, 4, 'enc_conv3_stride2'),
    (3, 64, 64, 64, 64, 1, 6, 'enc_conv4'),
    (1, 128, 32, 9, 32, 1, 3, 'ren_conv8_head'),
]


@pytest.mark.parametrize('case', NOL_WGRAD_CASES, ids=[c[-1] for c in NOL_WGRAD_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_wgrad_norm_on_load(ops, case, dt):
    """imm_conv2d_wgrad_multi with imm_wgrad_job.x_scale / x_shift: the filter gradient against relu(scale * y + shift) rebuilt in
    the LDS halo from the raw tensor y == the same job on the stored normalised tensor (same 16-bit operands, same split count:
    the slabs must agree to rounding of identical sums), and == autograd of the oracle convolution."""
    B, H, ci, co, lddy, stride, nsplit, _tag = case
    y, scale, shift, out = _nol_inputs(B, H, ci, dt, 950)
    wr = torch.zeros(3, 3, ci, co, requires_grad=True)
    zref = O.conv2d_same(out.float().cpu(), wr, None, stride)
    dz = rnd(tuple(zref.shape), 952, 1.0, dt)
    (gw,) = torch.autograd.grad(zref, wr, dz.float())
    desc = ops.fwd_desc(B, H, H, ci, ci, co, lddy, 3, stride, 0)
    key, _wps, _units, _pcu = ops.conv2d_wgrad_variant(desc, lddy, dt)
    assert key // 100000 == 2, key
    dzd = padded(dz, lddy)
    res = []
    for nol in (True, False):
        slab = torch.full((nsplit, desc.kpad, co), float('nan'), dtype=torch.float32, device=DEV)
        job = (desc, y, dzd, lddy, slab, nsplit, (scale, shift, True)) if nol else (desc, out, dzd, lddy, slab, nsplit)
        ops.conv2d_wgrad_multi(ops.WgradMulti([job], dt))
        dw = torch.full((3, 3, ci, co), float('nan'), dtype=torch.float32, device=DEV)
        ops.conv2d_wgrad_reduce(slab, nsplit, 3, 3, ci, ci, co, desc.kpad, dw)
        torch.cuda.synchronize()
        res.append(dw)
    close(res[0], gw, 2e-3, 5e-4, 'wgrad_nol vs oracle')
    close(res[0], res[1], 1e-5, 1e-6, 'wgrad_nol vs wgrad on the stored tensor')
    print('WGRAD_NOL bitwise equal to the stored-tensor job:', bool(torch.equal(res[0], res[1])))


S2D_CASES = [
    # B, H (input side = 2 x dy side), ci (dx channels), co (dy channels), tag
    (32, 128, 32, 64, 'enc_conv3_full_size_1024_tiles'),       # more tiles than CUs: tap-at-a-time form; dx channels < the 64-wide block
    (32, 64, 64, 128, 'enc_conv5_full_size_256_tiles'),        # one round of tiles: row-at-a-time form, two 64-channel slices
    (32, 32, 128, 256, 'enc_conv7_full_size_two_nblocks'),     # four slices, two channel blocks
    (2, 32, 64, 128, 'four_tiles'),
    # dy 64 channels -> dx <= 32 channels: the whole flipped filter in LDS, persistent workgroups (conv_halo.hip S2D)
    (2, 64, 32, 64, 'halo_form_small'),
    (70, 64, 32, 64, 'halo_form_many_patches_per_workgroup'),
    (1, 64, 12, 64, 'halo_form_dx12'),
    (5, 128, 20, 64, 'halo_form_dx20'),
    (3, 64, 40, 64, 'odd_batch_dx40'),                         # c_dx = 40: a partly filled channel block
    (1, 32, 8, 192, 'dx8_three_slices'),
    (5, 64, 200, 64, 'dx200_four_nblocks_ragged'),
]


@pytest.mark.parametrize('case', S2D_CASES, ids=[c[-1] for c in S2D_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_dgrad_stride2_one_launch(ops, case, dt):
    """imm_conv2d_dgrad_s2: the data gradient of a 3x3 stride-2 SAME convolution (encoder conv_3/5/7) as one launch with four
    accumulator sets over one dy halo, against autograd of the oracle's convolution on the same 16-bit values and against the
    four class launches it replaces (same sums in another order)."""
    B, H, ci, co, _tag = case
    k = 3
    w = rnd((k, k, ci, co), 13, 0.05, dt)
    xr = torch.zeros(B, H, H, ci, requires_grad=True)
    yref = O.conv2d_same(xr, w.float(), None, 2)
    dy = rnd(tuple(yref.shape), 14, 1.0, dt)
    (gx,) = torch.autograd.grad(yref, xr, dy.float())
    h = H // 2
    assert ops.conv2d_dgrad_s2_supported(B, h, h, co, ci, ci)
    rows = ops.round_up(ci, 128)
    wd = w.float().to(DEV).contiguous()
    wt = torch.zeros(rows, 9 * co, dtype=dt, device=DEV)
    ops.pack_weights(wd, wt, 1, k, k, ci, co, co, rows, 9 * co)
    dx = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    dyd = dy.to(DEV).contiguous()
    ops.conv2d_dgrad_s2(dyd, co, wt, dx, ci, ci, B, h, h)
    torch.cuda.synchronize()
    close(dx, gx, 1e-2, 2e-3, 'dgrad_s2_one_launch')           # every pixel written exactly once (no NaN left)
    # the four class launches
    dx4 = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    for d, mode in ops.dgrad_s2_class_descs(B, H, H, ci, ci, co, co, k):
        wt_c = torch.zeros(rows, d.kpad, dtype=dt, device=DEV)
        ops.pack_weights(wd, wt_c, mode, k, k, ci, co, co, rows, d.kpad)
        ops.conv2d(d, dyd, wt_c, None, dx4)
    torch.cuda.synchronize()
    close(dx, dx4, 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10, 1e-4, 'dgrad_s2_one_launch vs class launches')
    # padded output stride: channels beyond c_dx are left alone
    if ci % 32:
        ld = ops.round_up(ci, 32)
        dxp = torch.full((B, H, H, ld), 7.0, dtype=dt, device=DEV)
        ops.conv2d_dgrad_s2(dyd, co, wt, dxp, ld, ci, B, h, h)
        torch.cuda.synchronize()
        assert torch.equal(dxp[..., :ci], dx) and bool((dxp[..., ci:] == 7.0).all())


def test_conv_dgrad_stride2_one_launch_rejects_unserved_shapes(ops):
    assert not ops.conv2d_dgrad_s2_supported(2, 8, 8, 64, 32, 32)        # dy columns % 16
    assert not ops.conv2d_dgrad_s2_supported(2, 12, 16, 64, 32, 32)      # dy rows % 8
    assert not ops.conv2d_dgrad_s2_supported(2, 16, 16, 32, 32, 32)      # dy channels % 64
    assert not ops.conv2d_dgrad_s2_supported(2, 16, 16, 64, 12, 16)      # dx channels % 8
    assert ops.conv2d_dgrad_s2_supported(2, 16, 16, 64, 32, 32)
    dy = torch.zeros(2, 8, 8, 64, dtype=torch.bfloat16, device=DEV)
    wt = torch.zeros(128, 576, dtype=torch.bfloat16, device=DEV)
    dx = torch.zeros(2, 16, 16, 32, dtype=torch.bfloat16, device=DEV)
    from imm_amd import _lib as L
    with pytest.raises(L.ImmHipError):
        ops.conv2d_dgrad_s2(dy, 64, wt, dx, 32, 32, 2, 8, 8)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('B,H,cin,cout,l1', [(32, 16, 512, 512, False), (8, 32, 256, 256, False), (4, 32, 256, 256, True),
                                              (33, 8, 512, 512, False)],
                         ids=['vgg4_3_into_4_2', 'vgg3_3_into_3_2', 'l1', 'map8_odd_batch'])
def test_conv_dgrad_with_tap_epilogue(ops, B, H, cin, cout, l1, dt):
    """imm_conv2d_tap == imm_conv2d (data gradient) followed by imm_tap_grad(has_in, relu), bit for bit: the perceptual tap of
    conv3_2 / conv4_2 (imm_model.py:142-147) in the epilogue of the data gradient that enters the tapped layer."""
    S = 128
    w = rnd((3, 3, cin, cout), 171, 0.02, dt)
    dz = rnd((B, H, H, cout), 172, 1e-3, dt).to(DEV).contiguous()
    act = rnd((2 * B, H, H, cin), 173, 1.0, dt).to(DEV).contiguous()          # [gt ; pred] halves of the tapped activation
    mask = torch.rand(B, S, S, device=DEV)
    coef = torch.tensor([0.0, 3.7e-4, 0.0], device=DEV)
    desc = ops.dgrad_desc(B, H, H, cin, cin, cout, cout, 3, 1, 0)
    assert ops.conv2d_tap_supported(desc)
    wt = torch.zeros(ops.round_up(cin, 128), desc.kpad, dtype=dt, device=DEV)
    ops.pack_weights(w.float().to(DEV).contiguous(), wt, 1, 3, 3, cin, cout, cout, wt.shape[0], desc.kpad)
    for mk in (mask, None):
        ref = torch.full((B, H, H, cin), float('nan'), dtype=dt, device=DEV)
        ops.conv2d(desc, dz, wt, None, ref)
        ops.tap_grad(ref, True, act[B:], act[:B], B, H, cin, mk, S, coef, 1, True, l1)
        got = torch.full_like(ref, float('nan'))
        ops.conv2d_tap(desc, dz, wt, got, act[B:], act[:B], cin, mk, S, coef, 1, l1)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())
        assert float(got.float().abs().max()) > 0
    # shapes the LDS-halo deep-K kernel does not take are refused (the caller issues the two launches)
    assert not ops.conv2d_tap_supported(ops.dgrad_desc(2, 16, 16, 32, 32, 32, 32, 3, 1, 0))


# ----------------------------------------------------------------------------------------------
# filter gradient
# ----------------------------------------------------------------------------------------------
WGRAD_CASES = [(2, 16, 32, 32, 32, 32, 3, 1, 1, 'k3s1_split1'), (2, 16, 32, 32, 32, 32, 3, 1, 5, 'k3s1_split5'),
               (2, 16, 32, 32, 64, 64, 3, 2, 2, 'k3s2'), (2, 32, 3, 8, 32, 32, 7, 1, 4, 'first7x7'),
               (2, 16, 256, 256, 10, 16, 1, 1, 2, 'pose1x1'), (1, 32, 32, 32, 9, 16, 3, 1, 3, 'final9'),
               (2, 16, 266, 288, 256, 256, 3, 1, 2, 'concat266'), (2, 16, 128, 128, 128, 128, 3, 1, 1, 'c128')]


@pytest.mark.parametrize('case', WGRAD_CASES, ids=[c[-1] for c in WGRAD_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_wgrad(ops, case, dt):
    B, H, ci_real, ci_pad, co, lddy, k, stride, nsplit, tag = case
    x = rnd((B, H, H, ci_real), 21, 1.0, dt)
    wr = torch.zeros(k, k, ci_real, co, requires_grad=True)
    yref = O.conv2d_same(x.float(), wr, None, stride)
    dy = rnd(tuple(yref.shape), 22, 1.0, dt)
    (gw,) = torch.autograd.grad(yref, wr, dy.float())
    desc = ops.fwd_desc(B, H, H, ci_pad, ci_pad, co, lddy, k, stride, 0)
    slab = torch.full((nsplit, desc.kpad, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad(desc, padded(x, ci_pad), padded(dy, lddy), lddy, slab, nsplit)
    dw = torch.full((k, k, ci_real, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad_reduce(slab, nsplit, k, k, ci_pad, ci_real, co, desc.kpad, dw)
    torch.cuda.synchronize()
    close(dw, gw, 2e-3, 5e-4, 'wgrad/' + tag)     # f32 accumulate of exact bf16 products


WGRAD_HALO_CASES = [(1, 128, 32, 32, 32, 'h32_32'), (2, 64, 64, 64, 64, 'h64_64'), (2, 64, 64, 32, 32, 'h64_32'),
                    (3, 64, 32, 9, 32, 'h32_9'), (40, 64, 32, 64, 64, 'h32_64_many'),
                    # deeper filters at >= 64x64: 32-channel slices (blockIdx.y / .z)
                    (5, 64, 128, 64, 64, 'sliced_128_64'), (2, 64, 160, 96, 96, 'sliced_160_96')]


@pytest.mark.parametrize('case', WGRAD_HALO_CASES, ids=[c[-1] for c in WGRAD_HALO_CASES])
@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_wgrad_halo(ops, case, dt):
    """3x3 s1 filter gradient through the LDS-resident / transpose-read kernel (conv_wgrad_halo.hip)."""
    B, H, ci, co, lddy, tag = case
    x = rnd((B, H, H, ci), 121, 1.0, dt)
    wr = torch.zeros(3, 3, ci, co, requires_grad=True)
    yref = O.conv2d_same(x.float(), wr, None, 1)
    dy = rnd(tuple(yref.shape), 122, 1.0, dt)
    (gw,) = torch.autograd.grad(yref, wr, dy.float())
    desc = ops.fwd_desc(B, H, H, ci, ci, co, lddy, 3, 1, 0)
    nsplit = ops.conv2d_wgrad_splits(desc, lddy)
    assert nsplit > 0, 'case should select the halo kernel'
    slab = torch.full((nsplit, desc.kpad, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad(desc, x.to(DEV), padded(dy, lddy), lddy, slab, nsplit)
    dw = torch.full((3, 3, ci, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad_reduce(slab, nsplit, 3, 3, ci, ci, co, desc.kpad, dw)
    torch.cuda.synchronize()
    close(dw, gw, 2e-3, 5e-4, 'wgrad_halo/' + tag)
    # the general kernel (any other split count) must agree
    slab2 = torch.empty(3, desc.kpad, co, dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad(desc, x.to(DEV), padded(dy, lddy), lddy, slab2, 3)
    dw2 = torch.empty(3, 3, ci, co, dtype=torch.float32, device=DEV)
    ops.conv2d_wgrad_reduce(slab2, 3, 3, 3, ci, ci, co, desc.kpad, dw2)
    torch.cuda.synchronize()
    close(dw, dw2, 1e-3, 2e-4, 'wgrad_halo_vs_general/' + tag)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_conv_wgrad_multi_equals_single_launches(ops, dt):
    """imm_conv2d_wgrad_multi: many layers' filter gradients in one launch per kernel variant (transpose-read tiles of four widths,
    LDS-halo slices of three shapes, one job of the generic kernel) == the same jobs through imm_conv2d_wgrad one by one with the
    same split counts, bit for bit; and == autograd of the oracle convolution for the reduced gradients."""
    # B, H, ci, co, lddy, k, stride, nsplit
    jobs_def = [(2, 16, 256, 256, 256, 3, 1, 2), (2, 16, 256, 256, 256, 3, 1, 3), (4, 32, 128, 128, 128, 3, 1, 4),
                (2, 32, 64, 128, 128, 3, 2, 2), (2, 64, 32, 64, 64, 3, 2, 5), (1, 128, 32, 32, 32, 3, 1, 7),
                (2, 64, 64, 64, 64, 3, 1, 6), (2, 64, 64, 32, 32, 3, 1, 3), (2, 16, 256, 10, 16, 1, 1, 2),
                (5, 64, 128, 64, 64, 3, 1, 9), (1, 10, 64, 64, 64, 3, 1, 2)]
    made, multi_jobs, keys = [], [], []
    for i, (B, H, ci, co, lddy, k, stride, nsplit) in enumerate(jobs_def):
        x = rnd((B, H, H, ci), 500 + i, 1.0, dt)
        wr = torch.zeros(k, k, ci, co, requires_grad=True)
        yref = O.conv2d_same(x.float(), wr, None, stride)
        dy = rnd(tuple(yref.shape), 600 + i, 1.0, dt)
        (gw,) = torch.autograd.grad(yref, wr, dy.float())
        desc = ops.fwd_desc(B, H, H, ci, ci, co, lddy, k, stride, 0)
        xd, dyd = x.to(DEV).contiguous(), padded(dy, lddy)
        slab_m = torch.full((nsplit, desc.kpad, co), float('nan'), dtype=torch.float32, device=DEV)
        made.append((desc, xd, dyd, lddy, nsplit, k, ci, co, gw, slab_m))
        multi_jobs.append((desc, xd, dyd, lddy, slab_m, nsplit))
        keys.append(ops.conv2d_wgrad_variant(desc, lddy, dt)[0])
    assert len(set(keys)) >= 6 and 0 in [k // 100000 for k in keys], keys      # several variants of both families + the generic kernel
    multi = ops.WgradMulti(multi_jobs, dt)
    ops.conv2d_wgrad_multi(multi)
    torch.cuda.synchronize()
    for i, (desc, xd, dyd, lddy, nsplit, k, ci, co, gw, slab_m) in enumerate(made):
        dw = torch.full((k, k, ci, co), float('nan'), dtype=torch.float32, device=DEV)
        ops.conv2d_wgrad_reduce(slab_m, nsplit, k, k, ci, ci, co, desc.kpad, dw)
        torch.cuda.synchronize()
        close(dw, gw, 2e-3, 5e-4, 'wgrad_multi/job%d' % i)
        if keys[i] // 100000 == 2 and nsplit != ops.conv2d_wgrad_splits(desc, lddy):
            continue     # the single entry point only takes the halo kernel at its own split count: nothing to compare bitwise
        slab_s = torch.full_like(slab_m, float('nan'))
        ops.conv2d_wgrad(desc, xd, dyd, lddy, slab_s, nsplit)
        torch.cuda.synchronize()
        assert torch.equal(slab_s, slab_m), 'job %d (variant %d)' % (i, keys[i])


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('B,S', [(2, 64), (5, 128)])
def test_conv_wgrad_halo_7x1_first_layer(ops, B, S, dt):
    """Filter gradient of the tap-unrolled first encoder convolution (7x1 over 32 channels, imm_model.py:190) through the
    LDS-halo kernel (14x16-pixel halo, seven vertical taps): against autograd of the oracle convolution, alone and as a member
    of a multi-problem launch at another split count, and against the transpose-read kernel."""
    ci, co = 32, 32
    x = rnd((B, S, S, ci), 131, 1.0, dt)
    x[..., 21:] = 0          # channels 21..31 of the unrolled image are padding
    wr = torch.zeros(7, 1, ci, co, requires_grad=True)
    yref = O.conv2d_same(x.float(), wr, None, 1)
    dy = rnd(tuple(yref.shape), 132, 1.0, dt)
    (gw,) = torch.autograd.grad(yref, wr, dy.float())
    desc = ops.fwd_desc(B, S, S, ci, ci, co, co, 7, 1, 0, kw=1)
    key, wps, units, pcu = ops.conv2d_wgrad_variant(desc, co, dt)
    assert key // 100000 == 2 and wps == 1 and pcu == 2, key                 # LDS-halo family, whole filter = one slice
    xd, dyd = x.to(DEV).contiguous(), dy.to(DEV).contiguous()
    nsplit = ops.conv2d_wgrad_splits(desc, co)
    assert nsplit > 0
    res = []
    for ns, multi in ((nsplit, False), (7, True), (3, False)):       # (3, False): any other split count = transpose-read kernel
        slab = torch.full((ns, desc.kpad, co), float('nan'), dtype=torch.float32, device=DEV)
        if multi:
            ops.conv2d_wgrad_multi(ops.WgradMulti([(desc, xd, dyd, co, slab, ns)], dt))
        else:
            ops.conv2d_wgrad(desc, xd, dyd, co, slab, ns)
        dw = torch.full((7, 1, ci, co), float('nan'), dtype=torch.float32, device=DEV)
        ops.conv2d_wgrad_reduce(slab, ns, 7, 1, ci, ci, co, desc.kpad, dw)
        torch.cuda.synchronize()
        close(dw, gw, 2e-3, 5e-4, 'wgrad_halo_7x1/%d%s' % (ns, 'm' if multi else ''))
        res.append(dw)
    close(res[0], res[2], 1e-3, 2e-4, 'wgrad_halo_7x1 vs transpose-read')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('B,S,co', [(2, 128, 64), (9, 64, 64), (1, 64, 48)])
def test_conv_wgrad_halo_stride2(ops, B, S, co, dt):
    """Filter gradient of the encoder's first stride-2 convolution (3x3, 32 -> 64 channels, imm_model.py:195) through the LDS-halo
    kernel with the parity-de-interleaved input halo: against autograd of the oracle convolution, alone and as a member of a
    multi-problem launch at another split count, and against the transpose-read kernel."""
    ci = 32
    x = rnd((B, S, S, ci), 141, 1.0, dt)
    wr = torch.zeros(3, 3, ci, co, requires_grad=True)
    yref = O.conv2d_same(x.float(), wr, None, 2)
    dy = rnd(tuple(yref.shape[:3]) + (64,), 142, 1.0, dt)
    dy[..., co:] = 0
    (gw,) = torch.autograd.grad(yref, wr, dy[..., :co].float())
    desc = ops.fwd_desc(B, S, S, ci, ci, co, 64, 3, 2, 0)
    key, wps, units, pcu = ops.conv2d_wgrad_variant(desc, 64, dt)
    assert key == 200000 + 20000 + 3264 and wps == 1 and pcu == 1, key
    xd, dyd = x.to(DEV).contiguous(), dy.to(DEV).contiguous()
    nsplit = ops.conv2d_wgrad_splits(desc, 64)
    assert nsplit > 0
    res = []
    for ns, multi in ((nsplit, False), (5, True), (3, False)):       # (3, False): any other split count = transpose-read kernel
        slab = torch.full((ns, desc.kpad, co), float('nan'), dtype=torch.float32, device=DEV)
        if multi:
            ops.conv2d_wgrad_multi(ops.WgradMulti([(desc, xd, dyd, 64, slab, ns)], dt))
        else:
            ops.conv2d_wgrad(desc, xd, dyd, 64, slab, ns)
        dw = torch.full((3, 3, ci, co), float('nan'), dtype=torch.float32, device=DEV)
        ops.conv2d_wgrad_reduce(slab, ns, 3, 3, ci, ci, co, desc.kpad, dw)
        torch.cuda.synchronize()
        close(dw, gw, 2e-3, 5e-4, 'wgrad_halo_s2/%d%s' % (ns, 'm' if multi else ''))
        res.append(dw)
    close(res[0], res[2], 1e-3, 2e-4, 'wgrad_halo_s2 vs transpose-read')


def test_table_driven_pack_and_reduce(ops):
    """imm_pack_weights_multi (bit for bit) / imm_wgrad_reduce_multi (to f32 rounding) vs their single-tensor counterparts."""
    dt = torch.bfloat16
    layers = [(3, 3, 8, 32, 0), (3, 32, 32, 64, 0), (3, 266, 288, 256, 0), (3, 32, 16, 9, 1), (1, 256, 16, 10, 1), (7, 3, 8, 32, 0),
              (1, 256, 256, 10, 0), (3, 64, 128, 128, 4), (3, 64, 128, 128, 7)]
    jobs, items, refs = [], [], []
    for k, ci_real, c_pad, co, mode in layers:
        w = rnd((k, k, ci_real, co), 100 + ci_real, 0.1, torch.float32).to(DEV).contiguous()
        if mode == 0:
            rows, kpad = ops.round_up(co, 128), ops.round_up(k * k * c_pad, 32)
        else:
            rows, kpad = ops.round_up(ci_real, 128), ops.round_up(k * k * c_pad, 32)
        wt_ref = torch.zeros(rows, kpad, dtype=dt, device=DEV)
        wt = torch.full((rows, kpad), float('nan'), dtype=dt, device=DEV)
        ops.pack_weights(w, wt_ref, mode, k, k, ci_real, co, c_pad, rows, kpad)
        jobs.append((w.data_ptr(), wt.data_ptr(), mode, k, k, ci_real, co, c_pad, rows, kpad)); items.append(rows * kpad)
        refs.append((w, wt, wt_ref))
    tab = ops.pack_table(jobs, DEV)
    ops.pack_weights_multi(tab, dt)
    torch.cuda.synchronize()
    for _w, wt, wt_ref in refs:
        assert torch.equal(wt.view(torch.int16), wt_ref.view(torch.int16))
    jobs, items, refs = [], [], []
    for k, ci_real, ci_pad, co, nsplit in [(3, 32, 32, 32, 5), (7, 3, 8, 32, 3), (3, 266, 288, 256, 2), (1, 256, 256, 10, 1),
                                           (3, 64, 64, 64, 1), (3, 32, 32, 9, 54), (3, 128, 128, 128, 16), (1, 1, 1, 10, 32)]:
        kpad = ops.round_up(k * k * ci_pad, 32)
        slab = rnd((nsplit, kpad, co), 200 + co, 1.0, torch.float32).to(DEV).contiguous()
        dw_ref = torch.empty(k, k, ci_real, co, device=DEV); dw = torch.full((k, k, ci_real, co), float('nan'), device=DEV)
        ops.conv2d_wgrad_reduce(slab, nsplit, k, k, ci_pad, ci_real, co, kpad, dw_ref)
        jobs.append((slab.data_ptr(), dw.data_ptr(), nsplit, k * k, ci_pad, ci_real, co, kpad)); items.append(k * k * ci_real * co)
        refs.append((slab, dw, dw_ref))
    # split lanes per job chosen about its slab count (1 .. 16), and the all-16 form of rounds 1-2
    for tab in (ops.reduce_table(jobs, items, DEV), ops.JobTable(jobs, items, 64, DEV)):
        for _s, dw, _r in refs:
            dw.fill_(float('nan'))
        ops.wgrad_reduce_multi(tab)
        torch.cuda.synchronize()
        for _s, dw, dw_ref in refs:
            close(dw, dw_ref, 1e-6, 1e-6, 'wgrad_reduce_multi')      # interleaved partial sums vs a serial sum


def test_colsum(ops):
    dy = rnd((3000, 10), 31)
    dd = padded(dy, 16)
    part = torch.empty(ops.colsum_blocks(3000, 16), 16, dtype=torch.float32, device=DEV)
    out = torch.full((16,), -7.0, dtype=torch.float32, device=DEV)
    ops.colsum(dd, 3000, 16, 10, 16, part, out)
    torch.cuda.synchronize()
    close(out[:10], dy.float().sum(0), 1e-4, 1e-5, 'colsum')
    assert float((out[10:] + 7.0).abs().max()) == 0.0      # entries >= c_out untouched


# ----------------------------------------------------------------------------------------------
# batch norm
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('c,npix', [(32, 4096), (256, 512), (64, 1000), (16, 2048)])
def test_batch_norm_fwd_bwd(ops, c, npix):
    dt = torch.bfloat16
    y = (rnd((npix, c), 41) * 2 + 0.5).to(dt)
    gamma = rnd((c,), 42, 0.5, torch.float32) + 1.0
    beta = rnd((c,), 43, 0.5, torch.float32)
    yf = y.float()
    partial = torch.stack([yf.sum(0), (yf * yf).sum(0)]).reshape(1, 2, c).to(DEV).contiguous()
    mm, mv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    scale, shift, mean, rstd = (torch.empty(c, device=DEV) for _ in range(4))
    ops.bn_finalize(partial, 1, c, npix, gamma.to(DEV), beta.to(DEV), 1e-3, 0.99, True, mm, mv, scale, shift, mean, rstd)
    yd = y.to(DEV)
    xo = torch.empty(npix, c, dtype=dt, device=DEV)
    ops.bn_apply_relu(yd, npix, c, c, scale, shift, True, xo, c)
    torch.cuda.synchronize()
    yr = yf.reshape(1, 1, npix, c).clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref, (rmm, rmv) = O.batch_norm(yr, g_, b_, torch.zeros(c), torch.ones(c), True)
    ref = torch.relu(ref)
    close(xo, ref.reshape(npix, c), 1e-2, 2e-3, 'bn_fwd')
    close(mm, rmm, 1e-4, 1e-5, 'bn_moving_mean')
    close(mv, rmv, 1e-4, 1e-5, 'bn_moving_var')
    # eval mode: uses the moving statistics, leaves them untouched
    mm2, mv2 = mm.clone(), mv.clone()
    ops.bn_finalize(None, 0, c, npix, gamma.to(DEV), beta.to(DEV), 1e-3, 0.99, False, mm2, mv2, scale.clone(), shift.clone(),
                    mean.clone(), rstd.clone())
    torch.cuda.synchronize()
    assert torch.equal(mm2, mm) and torch.equal(mv2, mv)
    # backward
    dout = rnd((npix, c), 44)
    gy, gg, gb = torch.autograd.grad(ref, [yr, g_, b_], dout.float().reshape(1, 1, npix, c))
    nblk = ops.bn_bwd_blocks(npix, c)
    part = torch.empty(nblk, 2, c, dtype=torch.float32, device=DEV)
    dod = dout.to(DEV)
    ops.bn_bwd_reduce(dod, c, yd, c, npix, c, scale, shift, mean, rstd, True, part)
    dg, db, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(3, c, device=DEV)
    ops.bn_bwd_finalize(part, nblk, c, npix, gamma.to(DEV), beta.to(DEV), rstd, dg, db, coef)
    dyo = torch.empty(npix, c, dtype=dt, device=DEV)
    ops.bn_bwd_apply(dod, c, yd, c, npix, c, scale, shift, mean, rstd, True, coef, dyo, c)
    torch.cuda.synchronize()
    close(dg, gg, 2e-3, 1e-3, 'bn_dgamma')
    close(db, gb, 2e-3, 1e-3, 'bn_dbeta')
    close(dyo, gy.reshape(npix, c), 2e-2, 4e-3, 'bn_dy')


def test_masked_sse_multi_equals_single_launches(ops):
    """imm_masked_sse_multi: several features in one launch, partial sums bit for bit those of imm_masked_sse."""
    from imm_amd import _lib as L
    B, S = 3, 64
    g = torch.Generator().manual_seed(5)
    mask = torch.rand(B, S, S, generator=g).to(DEV)
    feats, single = [], []
    for s_, c in ((32, 256), (16, 512), (8, 64)):
        y = torch.randn(2 * B, s_, s_, c, generator=g).to(torch.bfloat16).to(DEV)
        part = torch.full((L.SSE_BLOCKS,), float('nan'), device=DEV)
        ref = torch.empty(L.SSE_BLOCKS, device=DEV)
        ops.masked_sse(y[:B], y[B:], B, s_, c, mask, S, ref)
        feats.append((y[:B], y[B:], s_, c, part)); single.append(ref)
    for l1 in (False, True):
        if l1:
            for (a, b, s_, c, _p), ref in zip(feats, single):
                ops.masked_sse(a, b, B, s_, c, mask, S, ref, l1=True)
        ops.masked_sse_multi(ops.SseMulti(feats), B, mask, S, l1=l1)
        torch.cuda.synchronize()
        for (_a, _b, _s, _c, part), ref in zip(feats, single):
            assert torch.equal(part, ref)
        # imm_masked_sse_all: the same launch + the f32 image pair of imm_masked_sse_f32 (the 'input' feature), bit for bit
        ldp = 8
        gt = (torch.rand(B, S, S, 3, generator=g) * 255).to(DEV)
        pred = torch.zeros(B, S, S, ldp); pred[..., :3] = torch.rand(B, S, S, 3, generator=g) * 255
        pred = pred.to(DEV)
        img_ref = torch.empty(L.SSE_BLOCKS, device=DEV)
        ops.masked_sse_f32(gt, 3, pred, ldp, B, S, 3, mask, img_ref, l1=l1)
        img_part = torch.full((L.SSE_BLOCKS,), float('nan'), device=DEV)
        for (_a, _b, _s, _c, part) in feats:
            part.fill_(float('nan'))
        ops.masked_sse_all(ops.SseMulti(feats), B, mask, S, gt, 3, pred, ldp, 3, img_part, l1=l1)
        torch.cuda.synchronize()
        assert torch.equal(img_part, img_ref)
        for (_a, _b, _s, _c, part), ref in zip(feats, single):
            assert torch.equal(part, ref)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('B,h,c', [(32, 16, 256), (3, 32, 128), (5, 64, 64), (2, 8, 16)])
def test_bn_bwd_reduce_with_upsampling_adjoint(ops, B, h, c, dt):
    """imm_bn_bwd_reduce_up == imm_upsample2x_bwd followed by imm_bn_bwd_reduce (renderer conv_2/4/6: the block's output is
    up-sampled x2, imm_model.py:175): the adjoint written for the apply pass and the partial rows, bit for bit."""
    dy_up = rnd((B, 2 * h, 2 * h, c), 191, 1.0, dt).to(DEV).contiguous()
    y = (rnd((B, h, h, c), 192) * 2 + 0.3).to(dt).to(DEV).contiguous()
    scale = (rnd((c,), 193, 0.3, torch.float32) + 1.0).to(DEV); shift = rnd((c,), 194, 0.5, torch.float32).to(DEV)
    mean = rnd((c,), 195, 0.5, torch.float32).to(DEV); rstd = (rnd((c,), 196, 0.1, torch.float32).abs() + 0.5).to(DEV)
    npix = B * h * h
    nblk = ops.bn_bwd_blocks(npix, c)
    d_ref = torch.full((B, h, h, c), float('nan'), dtype=dt, device=DEV)
    p_ref = torch.full((nblk, 2, c), float('nan'), device=DEV)
    ops.upsample2x_bwd(dy_up, d_ref, B, h, h, c, c, c)
    ops.bn_bwd_reduce(d_ref, c, y, c, npix, c, scale, shift, mean, rstd, True, p_ref)
    d_got = torch.full_like(d_ref, float('nan')); p_got = torch.full_like(p_ref, float('nan'))
    ops.bn_bwd_reduce_up(dy_up, c, d_got, c, y, c, B, h, h, c, scale, shift, mean, rstd, True, p_got)
    torch.cuda.synchronize()
    assert torch.equal(d_got, d_ref)
    assert torch.equal(p_got, p_ref), float((p_got - p_ref).abs().max())


@pytest.mark.parametrize('rows,width,group', [(1024, 64, 32), (512, 128, 32), (700, 64, 32), (33, 512, 32), (2048, 64, 64)])
def test_rows_reduce(ops, rows, width, group):
    """imm_rows_reduce: group sums of partial rows (f64 accumulation); bitwise repeatable."""
    g = torch.Generator().manual_seed(rows + width)
    src = (torch.randn(rows, width, generator=g) * 100).to(DEV)
    n = -(-rows // group)
    outs = []
    for _ in range(2):
        dst = torch.full((n, width), float('nan'), device=DEV)
        ops.rows_reduce(src, rows, width, group, dst)
        torch.cuda.synchronize()
        outs.append(dst)
    assert torch.equal(outs[0], outs[1])
    pad = torch.zeros(n * group, width, dtype=torch.float64)
    pad[:rows] = src.cpu().double()
    ref = pad.reshape(n, group, width).sum(1)
    assert torch.allclose(outs[0].cpu().double(), ref, rtol=2e-7, atol=1e-4)


@pytest.mark.parametrize('c,npix,nrows', [(32, 4096, 7), (256, 8192, 64), (64, 1000, 256), (128, 32768, 33)])
def test_batch_norm_finalize_fused_into_apply(ops, c, npix, nrows):
    """imm_bn_apply_fused / imm_bn_bwd_apply_fused == the finalize + apply pairs (same arithmetic; the partial rows are
    summed in a different but fixed order: f64 sums, so the f32 results agree to the last bits), in training and eval mode."""
    dt = torch.bfloat16
    y = (rnd((npix, c), 241) * 2 + 0.5).to(dt).to(DEV)
    gamma = (rnd((c,), 242, 0.5, torch.float32) + 1.0).to(DEV)
    beta = rnd((c,), 243, 0.5, torch.float32).to(DEV)
    yf = y.float()
    # nrows partial rows: split the pixels into nrows ranges
    idx = torch.linspace(0, npix, nrows + 1).long()
    part = torch.stack([torch.stack([yf[idx[r]:idx[r + 1]].sum(0), (yf[idx[r]:idx[r + 1]] ** 2).sum(0)]) for r in range(nrows)]).contiguous()
    res = {}
    for fused in (False, True):
        mm, mv = torch.full((c,), 0.3, device=DEV), torch.full((c,), 1.7, device=DEV)
        scale, shift, mean, rstd = (torch.full((c,), float('nan'), device=DEV) for _ in range(4))
        xo = torch.full((npix, c), float('nan'), dtype=dt, device=DEV)
        if fused:
            ops.bn_apply_fused(part, nrows, c, npix, gamma, beta, 1e-3, 0.99, True, mm, mv, scale, shift, mean, rstd, y, c, True, xo, c)
        else:
            ops.bn_finalize(part, nrows, c, npix, gamma, beta, 1e-3, 0.99, True, mm, mv, scale, shift, mean, rstd)
            ops.bn_apply_relu(y, npix, c, c, scale, shift, True, xo, c)
        torch.cuda.synchronize()
        res[fused] = (xo, mm, mv, scale, shift, mean, rstd)
    for a, b, what in zip(res[True], res[False], ('out', 'moving_mean', 'moving_var', 'scale', 'shift', 'mean', 'rstd')):
        close(a, b, 1e-6 if what != 'out' else 8e-3, 1e-6 if what != 'out' else 1e-3, 'bn_apply_fused/' + what)
    assert float((res[True][0].float() != res[False][0].float()).float().mean()) < 1e-3       # 16-bit outputs: (almost) all identical
    # eval mode: moving statistics, untouched
    scale, shift, mean, rstd = res[False][3:]
    mm, mv = res[False][1].clone(), res[False][2].clone()
    xo_e = torch.empty(npix, c, dtype=dt, device=DEV); xo_r = torch.empty_like(xo_e)
    s2, h2, m2, r2 = (torch.empty(c, device=DEV) for _ in range(4))
    ops.bn_apply_fused(None, 0, c, npix, gamma, beta, 1e-3, 0.99, False, mm, mv, s2, h2, m2, r2, y, c, True, xo_e, c)
    ops.bn_finalize(None, 0, c, npix, gamma, beta, 1e-3, 0.99, False, mm, mv, scale, shift, mean, rstd)
    ops.bn_apply_relu(y, npix, c, c, scale, shift, True, xo_r, c)
    torch.cuda.synchronize()
    assert torch.equal(xo_e, xo_r) and torch.equal(mm, res[False][1]) and torch.equal(mv, res[False][2])
    # with the renderer's x2 up-sampling written in the same pass: bitwise the separate kernel applied to the 16-bit output
    if npix % 64 == 0:
        Bq, hq = (npix // 64, 8) if npix <= 8192 else (npix // 1024, 32)
        up_f = torch.full((Bq, 2 * hq, 2 * hq, c), float('nan'), dtype=dt, device=DEV)
        xo_u = torch.empty(npix, c, dtype=dt, device=DEV)
        ops.bn_apply_fused(None, 0, c, npix, gamma, beta, 1e-3, 0.99, False, mm, mv, s2, h2, m2, r2, y, c, True, xo_u, c, up_f, c, hq, hq)
        up_r = torch.empty_like(up_f)
        ops.upsample2x_fwd(xo_r.reshape(Bq, hq, hq, c), up_r, Bq, hq, hq, c, c, c)
        torch.cuda.synchronize()
        assert torch.equal(xo_u, xo_r) and torch.equal(up_f, up_r)
    # backward
    dout = rnd((npix, c), 244).to(DEV)
    nblk = ops.bn_bwd_blocks(npix, c)
    assert nblk <= 256
    bpart = torch.empty(nblk, 2, c, dtype=torch.float32, device=DEV)
    ops.bn_bwd_reduce(dout, c, y, c, npix, c, scale, shift, mean, rstd, True, bpart)
    dg, db, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(3, c, device=DEV)
    ops.bn_bwd_finalize(bpart, nblk, c, npix, gamma, beta, rstd, dg, db, coef)
    dy_ref = torch.empty(npix, c, dtype=dt, device=DEV)
    ops.bn_bwd_apply(dout, c, y, c, npix, c, scale, shift, mean, rstd, True, coef, dy_ref, c)
    dg2, db2 = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    dy_f = torch.full((npix, c), float('nan'), dtype=dt, device=DEV)
    ops.bn_bwd_apply_fused(bpart, nblk, c, npix, gamma, dout, c, y, c, scale, shift, mean, rstd, True, dg2, db2, dy_f, c)
    torch.cuda.synchronize()
    close(dg2, dg, 1e-6, 1e-6, 'bn_bwd_fused/dgamma'); close(db2, db, 1e-6, 1e-6, 'bn_bwd_fused/dbeta')
    close(dy_f, dy_ref, 8e-3, 1e-3, 'bn_bwd_fused/dy')
    assert float((dy_f.float() != dy_ref.float()).float().mean()) < 1e-3


@pytest.mark.parametrize('c,ldp,npix', [(32, 32, 4096), (256, 320, 512), (64, 64, 1000)])
def test_batch_norm_bwd_from_producer_sums(ops, c, ldp, npix):
    """The fused backward: the producer of dz hands over rows of (sum dz, sum dz*out) (out = the block's stored
    ReLU output, rows possibly wider than the layer: ldp > c) and dz already masked; imm_bn_bwd_finalize(from_out=1)
    recovers sum dz*xhat = (sum dz*out - beta sum dz)/gamma and imm_bn_bwd_apply runs without the ReLU mask.  Must equal
    autograd of relu(batch_norm(y))."""
    dt = torch.bfloat16
    y = (rnd((npix, c), 141) * 2 + 0.5).to(dt)
    gamma = rnd((c,), 142, 0.5, torch.float32) + 1.0
    beta = rnd((c,), 143, 0.5, torch.float32)
    yf = y.float()
    partial = torch.stack([yf.sum(0), (yf * yf).sum(0)]).reshape(1, 2, c).to(DEV).contiguous()
    mm, mv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    scale, shift, mean, rstd = (torch.empty(c, device=DEV) for _ in range(4))
    ops.bn_finalize(partial, 1, c, npix, gamma.to(DEV), beta.to(DEV), 1e-3, 0.99, True, mm, mv, scale, shift, mean, rstd)
    yd = y.to(DEV)
    out = torch.empty(npix, c, dtype=dt, device=DEV)
    ops.bn_apply_relu(yd, npix, c, c, scale, shift, True, out, c)
    torch.cuda.synchronize()
    yr = yf.reshape(1, 1, npix, c).clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref, _ = O.batch_norm(yr, g_, b_, torch.zeros(c), torch.ones(c), True)
    ref = torch.relu(ref)
    dout = rnd((npix, c), 144)
    gy, gg, gb = torch.autograd.grad(ref, [yr, g_, b_], dout.float().reshape(1, 1, npix, c))
    # what a producer epilogue leaves: dz = dout * [out > 0] (16-bit) and 3 rows of partial sums, ldp floats wide
    of = out.float().cpu()
    dz = dout.float() * (of > 0)
    rows = torch.zeros(3, 2, ldp)
    for r in range(3):
        sl = slice(r * npix // 3, (r + 1) * npix // 3)
        rows[r, 0, :c] = dz[sl].sum(0); rows[r, 1, :c] = (dz[sl] * of[sl]).sum(0)
    rows[:, :, c:] = 123.0                     # columns of the producer's other channels: must be ignored
    dg, db, coef = torch.empty(c, device=DEV), torch.empty(c, device=DEV), torch.empty(3, c, device=DEV)
    ops.bn_bwd_finalize(rows.to(DEV), 3, c, npix, gamma.to(DEV), beta.to(DEV), rstd, dg, db, coef, from_out=True, ldp=ldp)
    dyo = torch.empty(npix, c, dtype=dt, device=DEV)
    ops.bn_bwd_apply(dz.to(dt).to(DEV), c, yd, c, npix, c, scale, shift, mean, rstd, False, coef, dyo, c)
    torch.cuda.synchronize()
    # xhat recovered from the 16-bit `out` = one more rounding than from y: |err| ~ sqrt(N) |dz| |out| 2^-9 (measured 4e-3 of
    # the largest dgamma at N = 4096) — the reason the engine keeps the separate reduction pass by default
    close(dg, gg, 2e-2, 1e-2, 'bn_dgamma(from_out)')
    close(db, gb, 2e-3, 1e-3, 'bn_dbeta(from_out)')
    close(dyo, gy.reshape(npix, c), 2e-2, 4e-3, 'bn_dy(from_out)')


@pytest.mark.parametrize('B,H,ci,co,dt', [(2, 32, 32, 64, torch.bfloat16), (40, 64, 64, 64, torch.bfloat16),
                                          (36, 64, 32, 32, torch.float16), (9, 128, 32, 32, torch.bfloat16),
                                          (14, 32, 128, 256, torch.bfloat16), (24, 16, 128, 256, torch.bfloat16),
                                          (32, 16, 256, 320, torch.bfloat16), (51, 8, 128, 256, torch.bfloat16)],
                         ids=['igemm', 'halo64_persistent', 'halo32_f16_persistent', 'halo32_128px', 'hdeep', 'hdeep_small_patch',
                              'hdeep_co320', 'hdeep_map8_odd'])
def test_conv_mask_with_bn_backward_sums(ops, B, H, ci, co, dt):
    """IMM_CONV_STATS | IMM_CONV_MASK: same output as the mask alone, partial sums = (sum v, sum v*mask_ref) of the masked
    f32 values (every kernel family that produces the output gradient of a conv+BN+ReLU block)."""
    from imm_amd import _lib as L
    x = rnd((B, H, H, ci), 4, 1.0, dt)
    w = rnd((3, 3, ci, co), 5, 0.1, dt)
    mref = rnd((B, H, H, co), 7, 1.0, dt).to(DEV).contiguous()
    ref = O.conv2d_same(x.float(), w.float(), None, 1)
    y3, _, _ = run_conv(ops, x, w, None, 3, 1, co, ci, False, extra_flags=L.CONV_MASK, mask=mref)
    y4, stats, _ = run_conv(ops, x, w, None, 3, 1, co, ci, False, extra_flags=L.CONV_MASK | L.CONV_STATS, mask=mref)
    assert torch.equal(y3, y4), 'the stats flag must not change the output'
    mf = mref.float().cpu()
    v = ref * (mf > 0)
    close(y4, v, 1e-2, 2e-3, 'conv+mask+stats/y')
    s = stats.sum(dim=0).cpu()
    close(s[0], v.sum(dim=(0, 1, 2)), 2e-3, 2e-3, 'bn-bwd sums/sum dz')
    close(s[1], (v * mf).sum(dim=(0, 1, 2)), 2e-3, 2e-3, 'bn-bwd sums/sum dz*out')


@pytest.mark.parametrize('H,ci,co', [(32, 64, 128), (64, 128, 256), (16, 32, 64)], ids=['grouped_64x64', 'grouped_deep', 'grouped_128x32'])
def test_conv_group_mask_with_bn_backward_sums(ops, H, ci, co):
    """The stride-2 data gradient (four parity classes, scattered output) with the mask + sums epilogue: output = masked
    sequential result; the rows of all members together sum to (sum dz, sum dz*out)."""
    from imm_amd import _lib as L
    B, k, dt = 8, 3, torch.bfloat16
    w = rnd((k, k, ci, co), 401, 0.05)
    dy = rnd((B, H // 2, H // 2, co), 402).to(DEV).contiguous()
    mref = rnd((B, H, H, ci), 403).to(DEV).contiguous()
    plain = ops.dgrad_s2_class_descs(B, H, H, ci, ci, co, co, k)
    classes = ops.dgrad_s2_class_descs(B, H, H, ci, ci, co, co, k, flags=L.CONV_MASK | L.CONV_STATS, ldmask=ci)
    rows = ops.round_up(ci, 128)
    wts = []
    for d, mode in classes:
        wt = torch.zeros(rows, d.kpad, dtype=dt, device=DEV)
        ops.pack_weights(w.float().to(DEV).contiguous(), wt, mode, k, k, ci, co, co, rows, d.kpad)
        wts.append(wt)
    ref = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    for (d, _m), wt in zip(plain, wts):
        ops.conv2d(d, dy, wt, None, ref)
    grp = ops.ConvGroup([d for d, _m in classes], wts)
    nrows = ops.conv2d_group_stats_blocks(grp)
    stats = torch.full((nrows, 2, ci), float('nan'), dtype=torch.float32, device=DEV)
    got = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    ops.conv2d_group(grp, dy, got, stats, mref)
    torch.cuda.synchronize()
    mf = mref.float()
    want = ref.float() * (mf > 0)
    assert torch.equal(got.float(), want)                 # masking a value that was rounded to 16 bits first or after: same bits
    s = stats.sum(dim=0).cpu()
    close(s[0], want.sum(dim=(0, 1, 2)).cpu(), 5e-3, 5e-3, 'group sums/sum dz')          # sums of the f32 accumulators vs of
    close(s[1], (want * mf).sum(dim=(0, 1, 2)).cpu(), 5e-3, 5e-3, 'group sums/sum dz*out')   # the 16-bit rounded values


@pytest.mark.parametrize('B,h,c', [(2, 8, 16), (3, 16, 64), (32, 16, 256), (5, 64, 64)])
def test_upsample2x_bwd_with_bn_backward_sums(ops, B, h, c):
    dt = torch.bfloat16
    dy = rnd((B, 2 * h, 2 * h, c), 52).to(DEV)
    out = rnd((B, h, h, c), 55).to(DEV)
    plain = torch.empty(B, h, h, c, dtype=dt, device=DEV)
    ops.upsample2x_bwd(dy, plain, B, h, h, c, c, c)
    nblk = ops.upsample2x_bwd_bn_blocks(B, h, h, c)
    part = torch.full((nblk, 2, c), float('nan'), dtype=torch.float32, device=DEV)
    dx = torch.empty(B, h, h, c, dtype=dt, device=DEV)
    ops.upsample2x_bwd_bn(dy, dx, B, h, h, c, c, c, out, c, part)
    torch.cuda.synchronize()
    want = plain.float() * (out.float() > 0)
    assert torch.equal(dx.float(), want)
    s = part.sum(dim=0)
    close(s[0], want.sum(dim=(0, 1, 2)), 5e-3, 5e-3, 'upsample_bwd_bn/sum dz')
    close(s[1], (want * out.float()).sum(dim=(0, 1, 2)), 5e-3, 5e-3, 'upsample_bwd_bn/sum dz*out')


# ----------------------------------------------------------------------------------------------
# resampling / pooling / packing
# ----------------------------------------------------------------------------------------------
def test_upsample2x(ops):
    B, h, c = 2, 8, 16
    x = rnd((B, h, h, c), 51)
    xr = x.float().requires_grad_(True)
    ref = O.resize_bilinear(xr, 2 * h, 2 * h)
    y = torch.empty(B, 2 * h, 2 * h, c, dtype=x.dtype, device=DEV)
    ops.upsample2x_fwd(x.to(DEV), y, B, h, h, c, c, c)
    dy = rnd((B, 2 * h, 2 * h, c), 52)
    (gx,) = torch.autograd.grad(ref, xr, dy.float())
    dx = torch.empty(B, h, h, c, dtype=x.dtype, device=DEV)
    ops.upsample2x_bwd(dy.to(DEV), dx, B, h, h, c, c, c)
    torch.cuda.synchronize()
    close(y, ref, 8e-3, 1e-3, 'upsample_fwd')
    close(dx, gx, 8e-3, 2e-3, 'upsample_bwd')


def test_resize_align_corners(ops):
    B, hi, ho, c = 2, 32, 16, 16
    x = rnd((B, hi, hi, c), 53)
    xr = x.float().requires_grad_(True)
    ref = O.resize_bilinear(xr, ho, ho, align_corners=True)
    y = torch.empty(B, ho, ho, c, dtype=x.dtype, device=DEV)
    ops.resize_ac_fwd(x.to(DEV), y, B, hi, hi, ho, ho, c, c, c)
    dy = rnd((B, ho, ho, c), 54)
    (gx,) = torch.autograd.grad(ref, xr, dy.float())
    dx = torch.empty(B, hi, hi, c, dtype=x.dtype, device=DEV)
    ops.resize_ac_bwd(dy.to(DEV), dx, B, hi, hi, ho, ho, c, c, c)
    torch.cuda.synchronize()
    close(y, ref, 8e-3, 2e-3, 'resize_ac_fwd')
    close(dx, gx, 8e-3, 2e-3, 'resize_ac_bwd')


def test_maxpool(ops):
    B, h, c = 2, 8, 16
    x = torch.relu(rnd((B, h, h, c), 55))     # post-ReLU activations: many exact zeros / ties at 0
    xr = x.float().requires_grad_(True)
    ref = O.max_pool2(xr)
    y = torch.empty(B, h // 2, h // 2, c, dtype=x.dtype, device=DEV)
    ops.maxpool2_fwd(x.to(DEV), y, B, h, h, c)
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu(), ref.detach())      # exact
    dy = rnd((B, h // 2, h // 2, c), 56)
    (gx,) = torch.autograd.grad(ref, xr, dy.float())
    for relu_mask in (0, 1):
        dx = torch.empty(B, h, h, c, dtype=x.dtype, device=DEV)
        ops.maxpool2_bwd(x.to(DEV), dy.to(DEV), dx, B, h, h, c, relu_mask)
        torch.cuda.synchronize()
        if relu_mask:
            # gradient routed to a zero activation is killed by the ReLU mask of the producing conv
            assert torch.equal(dx.float().cpu(), gx * (x.float() > 0))
        else:
            # windows whose max is exactly 0 are 4-way ties: any routing is a valid subgradient; the
            # values only matter where x > 0 (ReLU mask downstream), so compare there + total mass
            assert torch.equal((dx.float().cpu() * (x.float() > 0)), gx * (x.float() > 0))
            np.testing.assert_allclose(float(dx.float().sum()), float(dy.float().sum()), rtol=1e-3)


@pytest.mark.parametrize('B,S', [(2, 32), (5, 128)], ids=['igemm_32px', 'halo2_7x1_128px'])
def test_first_conv_as_tap_unrolled_7x1(ops, B, S):
    """conv 7x7x3 == pack_image_taps (7 horizontal taps -> 21 channels) + 7x1 conv over 32 channels.  At >= 64x64 the
    7x1 instantiation of conv_halo2.hip serves it (filter in registers, 14-row halo tiles)."""
    from imm_amd import _lib as L
    co, dt = 32, torch.bfloat16
    src = torch.rand(B, S, S, 3) * 255
    w = rnd((7, 7, 3, co), 57, 0.01, torch.float32)
    xin = torch.full((B, S, S, 32), float('nan'), dtype=dt, device=DEV)
    ops.pack_image_taps(src.to(DEV), xin, B, S, S, 7, 3, 32)
    torch.cuda.synchronize()
    xp = torch.nn.functional.pad(src.to(dt), (0, 0, 3, 3))
    for kx in range(7):
        assert torch.equal(xin[..., kx * 3:kx * 3 + 3].cpu(), xp[:, :, kx:kx + S]), kx
    assert float(xin[..., 21:].float().abs().max()) == 0.0
    desc = ops.fwd_desc(B, S, S, 32, 32, co, co, 7, 1, 0, kw=1)
    assert (desc.kh, desc.kw, desc.pad_t, desc.pad_l, desc.kpad) == (7, 1, 3, 0, 224)
    wt = torch.zeros(128, desc.kpad, dtype=dt, device=DEV)
    ops.pack_weights(w.to(DEV).contiguous(), wt, 0, 7, 1, 21, co, 32, 128, desc.kpad)
    y = torch.empty(B, S, S, co, dtype=dt, device=DEV)
    ops.conv2d(desc, xin, wt, None, y)
    ref = O.conv2d_same(src.to(dt).float(), w.to(dt).float(), None, 1)
    dy = rnd((B, S, S, co), 58)
    slab = torch.empty(3, desc.kpad, co, device=DEV)
    ops.conv2d_wgrad(desc, xin, dy.to(DEV), co, slab, 3)
    dw = torch.empty(7, 7, 3, co, device=DEV)
    ops.conv2d_wgrad_reduce(slab, 3, 7, 1, 32, 21, co, desc.kpad, dw)
    torch.cuda.synchronize()
    close(y, ref, 1e-2, 2e-3, 'conv7x1')
    # epilogue variant used by the encoders: bias + BN partial sums
    bias = rnd((co,), 59, 0.5, torch.float32)
    d2 = ops.fwd_desc(B, S, S, 32, 32, co, co, 7, 1, L.CONV_BIAS | L.CONV_STATS, kw=1)
    stats = torch.full((ops.conv_stats_blocks(d2), 2, co), float('nan'), dtype=torch.float32, device=DEV)
    y2 = torch.empty(B, S, S, co, dtype=dt, device=DEV)
    ops.conv2d(d2, xin, wt, bias.to(DEV), y2, stats)
    torch.cuda.synchronize()
    close(y2, ref + bias, 1e-2, 2e-3, 'conv7x1+bias')
    st = stats.sum(dim=0).cpu()
    close(st[0], (ref + bias).sum(dim=(0, 1, 2)), 1e-3, 1e-3, 'conv7x1/sum')
    close(st[1], ((ref + bias) ** 2).sum(dim=(0, 1, 2)), 1e-3, 1e-3, 'conv7x1/sumsq')
    wr = torch.zeros(7, 7, 3, co, requires_grad=True)
    (gw,) = torch.autograd.grad(O.conv2d_same(src.to(dt).float(), wr, None, 1), wr, dy.float())
    close(dw, gw, 2e-3, 5e-4, 'wgrad7x1')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('B,S,co', [(2, 128, 32), (33, 64, 32), (1, 256, 32), (3, 32, 20)],
                         ids=['128px', 'many_patches_64px', '256px', 'co20_32px'])
def test_first_conv_from_the_f32_image(ops, B, S, co, dt):
    """imm_conv_first: the 7x7x3 first encoder convolution straight from the f32 image (tap-unrolled tile built in LDS) against
    the oracle convolution of the 16-bit-rounded image and against the two-launch form it replaces on the forward chain
    (imm_pack_image_taps + the 7x1 convolution): same 16-bit operands, so outputs and batch-norm sums agree to rounding."""
    from imm_amd import _lib as L
    g = torch.Generator().manual_seed(77)
    src = torch.rand(B, S, S, 3, generator=g) * 255
    w = rnd((7, 7, 3, co), 57, 0.01, torch.float32)
    bias = rnd((co,), 59, 0.5, torch.float32)
    srcd = src.to(DEV).contiguous()
    ldy = ops.round_up(co, 8)
    assert ops.conv_first_supported(B, S, co, ldy)
    wt = torch.zeros(128, 224, dtype=dt, device=DEV)
    ops.pack_weights(w.to(DEV).contiguous(), wt, 0, 7, 1, 21, co, 32, 128, 224)
    y = torch.full((B, S, S, ldy), float('nan'), dtype=dt, device=DEV)
    stats = torch.full((ops.conv_first_stats_blocks(B, S), 2, co), float('nan'), dtype=torch.float32, device=DEV)
    ops.conv_first(srcd, wt, bias.to(DEV), y, ldy, stats, B, S, co, L.CONV_BIAS | L.CONV_STATS)
    # the form it replaces
    xin = torch.empty(B, S, S, 32, dtype=dt, device=DEV)
    ops.pack_image_taps(srcd, xin, B, S, S, 7, 3, 32)
    d2 = ops.fwd_desc(B, S, S, 32, 32, co, ldy, 7, 1, L.CONV_BIAS | L.CONV_STATS, kw=1)
    stats2 = torch.full((ops.conv_stats_blocks(d2), 2, co), float('nan'), dtype=torch.float32, device=DEV)
    y2 = torch.full((B, S, S, ldy), float('nan'), dtype=dt, device=DEV)
    ops.conv2d(d2, xin, wt, bias.to(DEV), y2, stats2)
    torch.cuda.synchronize()
    ref = O.conv2d_same(src.to(dt).float(), w.to(dt).float(), bias, 1)
    close(y[..., :co], ref, 1e-2, 2e-3, 'conv_first vs oracle')
    close(y[..., :co], y2[..., :co], 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10, 1e-4, 'conv_first vs pack + 7x1 conv')
    close(stats.sum(0), stats2.sum(0), 2e-4, 2e-4, 'conv_first batch-norm partial sums')
    if ldy > co:
        assert bool(torch.isnan(y[..., co:].float()).all())          # padding channels are left alone
    # no bias, no sums (the inference form)
    y3 = torch.full((B, S, S, ldy), float('nan'), dtype=dt, device=DEV)
    ops.conv_first(srcd, wt, None, y3, ldy, None, B, S, co, 0)
    torch.cuda.synchronize()
    close(y3[..., :co], ref - bias, 1e-2, 2e-3, 'conv_first without bias')


def test_first_conv_from_the_f32_image_rejects_unserved_shapes(ops):
    from imm_amd import _lib as L
    assert not ops.conv_first_supported(2, 72, 32, 32)       # side % 16
    assert not ops.conv_first_supported(2, 64, 64, 64)       # more than 32 filters
    assert not ops.conv_first_supported(2, 64, 30, 32)       # co % 4
    img = torch.zeros(2, 72, 72, 3, device=DEV)
    wt = torch.zeros(128, 224, dtype=torch.bfloat16, device=DEV)
    y = torch.zeros(2, 72, 72, 32, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(L.ImmHipError):
        ops.conv_first(img, wt, None, y, 32, None, 2, 72, 32, 0)


def test_pack_image(ops):
    src = torch.rand(5, 7, 7, 3) * 255
    dst = torch.empty(5, 7, 7, 8, dtype=torch.bfloat16, device=DEV)
    ops.pack_image(src.to(DEV), dst, 5 * 49)
    torch.cuda.synchronize()
    assert torch.equal(dst[..., :3].cpu(), src.to(torch.bfloat16)) and float(dst[..., 3:].float().abs().max()) == 0.0


# ----------------------------------------------------------------------------------------------
# bottleneck
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize('mode', ['rot', 'flat', 'ankush'])
@pytest.mark.parametrize('h,K,s', [(16, 10, 16), (16, 50, 16), (32, 30, 16)])
def test_softargmax_gauss(ops, h, K, s, mode):
    """All three get_gaussian_maps modes (imm_model.py:48-72), forward and backward, against the oracle + autograd.  'flat'
    and 'ankush' have a kink at the landmark (|.|, ^1/4): the gradient tolerance is wider where a grid point sits within
    1e-3 of mu, which the seeded heat-maps never produce."""
    B, ldh, ldg = 3, ops.round_up(K, 4), ops.round_up(256 + K, 32)
    heat = rnd((B, h, h, K), 61, 2.0, torch.float32)
    hr = heat.clone().requires_grad_(True)
    mu_r, py_r, px_r = O.soft_argmax(hr)
    g_r = O.gaussian_maps(mu_r, [s, s], 10.0, mode)
    mu = torch.empty(B, K, 2, device=DEV); py = torch.empty(B, h, K, device=DEV); px = torch.empty(B, h, K, device=DEV)
    joint = torch.zeros(B, s, s, ldg, dtype=torch.bfloat16, device=DEV)
    ops.softargmax_gauss_fwd(padded(heat, ldh), ldh, B, h, h, K, 10.0, s, mu, py, px, joint[..., 256:], ldg, torch.bfloat16, mode)
    torch.cuda.synchronize()
    close(mu, mu_r, 1e-4, 1e-5, 'mu')                     # f32 math: landmarks to ~1e-6
    assert float((mu.cpu() - mu_r.detach()).abs().max()) < 1e-5
    close(py, py_r, 1e-4, 1e-5, 'py'); close(px, px_r, 1e-4, 1e-5, 'px')
    close(joint[..., 256:256 + K], g_r, 8e-3, 1e-3, 'gauss')
    assert float(joint[..., :256].float().abs().max()) == 0.0 and float(joint[..., 256 + K:].float().abs().max()) == 0.0
    dg = rnd((B, s, s, K), 62)
    (gh,) = torch.autograd.grad(g_r, hr, dg.float())
    dj = torch.zeros(B, s, s, ldg, dtype=torch.bfloat16, device=DEV)
    dj[..., 256:256 + K] = dg.to(DEV)
    dheat = torch.full((B, h, h, 64), float('nan'), dtype=torch.bfloat16, device=DEV)
    ops.softargmax_gauss_bwd(dj[..., 256:], ldg, B, h, h, K, 10.0, s, mu, py, px, dheat, 64, mode)
    torch.cuda.synchronize()
    close(dheat[..., :K], gh, 1e-2, 2e-3, 'dheat')
    assert float(dheat[..., K:].float().abs().max()) == 0.0
    out = torch.empty(B, 128, 128, K, device=DEV)
    ops.gauss_render_f32(mu, B, K, 10.0, 128, out, mode)
    torch.cuda.synchronize()
    close(out, O.gaussian_maps(mu_r.detach(), [128, 128], 10.0, mode), 1e-4, 1e-5, 'render128')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
@pytest.mark.parametrize('h,K,mode', [(16, 10, 'rot'), (16, 50, 'rot'), (32, 30, 'rot'), (16, 10, 'ankush')])
def test_pose_head_fused(ops, h, K, mode, dt):
    """imm_pose_head_fwd / _bwd — the pose head as one launch each way (imm_model.py:247-264) — against the oracle (1x1
    convolution + soft-argmax + Gaussian maps, autograd for the backward) and against the launches they replace
    (imm_conv2d, imm_softargmax_gauss_fwd/bwd, imm_colsum, the 1x1 data gradient)."""
    B, C, s = 3, 256, 16
    ldh, ldg, lddh = ops.round_up(K, 4), ops.round_up(C + K, 64), ops.round_up(K, 32)
    feat = rnd((B, h, h, C), 161, 1.0, dt)
    w = rnd((1, 1, C, K), 162, 0.05, dt)
    bias = rnd((K,), 163, 0.3, torch.float32)
    fr = feat.float().clone().requires_grad_(True)
    wr = w.float().clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    heat_r = O.conv2d_same(fr, wr, br, 1)
    mu_r, py_r, px_r = O.soft_argmax(heat_r)
    g_r = O.gaussian_maps(mu_r, [s, s], 10.0, mode)
    # ---- forward
    fd = ops.fwd_desc(B, h, h, C, C, K, ldh, 1, 1, 0)
    wt = torch.zeros(128, fd.kpad, dtype=dt, device=DEV)
    ops.pack_weights(w.float().to(DEV).contiguous(), wt, 0, 1, 1, C, K, C, 128, fd.kpad)
    featd = feat.to(DEV).contiguous()
    heat = torch.zeros(B, h, h, ldh, device=DEV)
    mu = torch.empty(B, K, 2, device=DEV); py = torch.empty(B, h, K, device=DEV); px = torch.empty(B, h, K, device=DEV)
    joint = torch.zeros(B, s, s, ldg, dtype=dt, device=DEV)
    ops.pose_head_fwd(featd, C, C, wt, bias.to(DEV), B, h, h, K, 10.0, s, heat, ldh, mu, py, px, joint[..., C:], ldg, dt, mode)
    torch.cuda.synchronize()
    close(heat[..., :K], heat_r, 2e-3, 2e-4, 'heat')                 # f32 accumulation of exact 16-bit products
    assert float((mu.cpu() - mu_r.detach()).abs().max()) < 2e-5
    close(py, py_r, 1e-3, 1e-5, 'py'); close(px, px_r, 1e-3, 1e-5, 'px')
    close(joint[..., C:C + K], g_r, 8e-3 if dt == torch.bfloat16 else 2e-3, 1e-3, 'gauss')
    assert float(joint[..., :C].float().abs().max()) == 0.0 and float(joint[..., C + K:].float().abs().max()) == 0.0
    # the two-launch path it replaces: same heat-map to accumulation order, same landmarks
    fd2 = ops.fwd_desc(B, h, h, C, C, K, ldh, 1, 1, L_CONV_BIAS_F32)
    heat2 = torch.zeros(B, h, h, ldh, device=DEV)
    ops.conv2d(fd2, featd, wt, bias.to(DEV), heat2)
    mu2 = torch.empty_like(mu); py2 = torch.empty_like(py); px2 = torch.empty_like(px)
    ops.softargmax_gauss_fwd(heat2, ldh, B, h, h, K, 10.0, s, mu2, py2, px2, None, ldg, dt, mode)
    torch.cuda.synchronize()
    close(heat[..., :K], heat2[..., :K], 1e-4, 1e-5, 'heat vs conv2d')
    assert float((mu - mu2).abs().max()) < 1e-5
    # ---- backward
    dg = rnd((B, s, s, K), 164, 1.0, dt)
    gf, gw, gb = torch.autograd.grad(g_r, (fr, wr, br), dg.float())
    dj = torch.zeros(B, s, s, ldg, dtype=dt, device=DEV)
    dj[..., C:C + K] = dg.to(DEV)
    wtd = torch.zeros(ops.round_up(C, 128), lddh, dtype=dt, device=DEV)
    ops.pack_weights(w.float().to(DEV).contiguous(), wtd, 1, 1, 1, C, K, lddh, wtd.shape[0], lddh)
    dheat = torch.full((B, h, h, lddh), float('nan'), dtype=dt, device=DEV)
    dfeat = torch.full((B, h, h, C), float('nan'), dtype=dt, device=DEV)
    bpart = torch.full((B, K), float('nan'), device=DEV)
    ops.pose_head_bwd(dj[..., C:], ldg, B, h, h, K, 10.0, s, mu, py, px, dheat, lddh, wtd, C, dfeat, C, bpart, mode)
    torch.cuda.synchronize()
    dheat2 = torch.full_like(dheat, float('nan'))
    ops.softargmax_gauss_bwd(dj[..., C:], ldg, B, h, h, K, 10.0, s, mu, py, px, dheat2, lddh, mode)
    torch.cuda.synchronize()
    assert torch.equal(dheat, dheat2)                                 # the shared part of the pass: bit for bit
    tol = 1e-2 if dt == torch.bfloat16 else 2e-3
    close(dfeat, gf, tol, 2e-3, 'dfeat')
    # the bias gradient is analytically ZERO (a constant added to a landmark's heat-map changes neither softmax): autograd
    # returns cancellation noise (|gb| ~ 1e-6), the kernel the sum of the 16-bit-rounded dheat — noise of the storage rounding
    assert float(gb.abs().max()) < 1e-4 and float(bpart.sum(0).abs().max()) <= 4e-3 * float(dheat[..., :K].float().abs().sum() / K) + 1e-6
    np.testing.assert_allclose(bpart.sum(0).cpu().numpy(), dheat[..., :K].float().sum((0, 1, 2)).cpu().numpy(), rtol=1e-5, atol=1e-6)
    # the head's filter gradient still comes from the stored dheat (a member of the multi-problem launch): consistent with gw
    close(torch.einsum('bhwc,bhwk->ck', feat.float(), dheat[..., :K].float().cpu()).reshape(1, 1, C, K), gw, tol, 5e-3, 'dW from dheat')


# ----------------------------------------------------------------------------------------------
# VGG head, loss, optimizer
# ----------------------------------------------------------------------------------------------
def test_vgg_conv1_1(ops):
    B, S, ldp = 2, 32, 16
    gt = torch.rand(B, S, S, 3) * 255
    pred = torch.zeros(B, S, S, ldp); pred[..., :3] = torch.rand(B, S, S, 3) * 255
    w = rnd((3, 3, 1, 64), 71, 0.4, torch.float32); b = rnd((64,), 72, 0.1, torch.float32)
    pr = pred[..., :3].clone().requires_grad_(True)
    ims = torch.cat([gt, pr], 0)
    gray = ims.mean(dim=3, keepdim=True) / 255.0 - O.VGG_GRAY_MEAN / 255.0
    ref = torch.relu(O.conv2d_same(gray, w, b))
    out = torch.empty(2 * B, S, S, 64, dtype=torch.bfloat16, device=DEV)
    ops.vgg_conv1_1_fwd(gt.to(DEV), pred.to(DEV), ldp, B, S, w.reshape(9, 64).to(DEV).contiguous(), b.to(DEV), out)
    torch.cuda.synchronize()
    close(out, ref, 8e-3, 1e-3, 'vgg1_1_fwd')
    # halves: 1 = gt images only, 2 = pred images only; together bitwise equal to the single launch
    out2 = torch.zeros_like(out)
    ops.vgg_conv1_1_fwd(gt.to(DEV), pred.to(DEV), ldp, B, S, w.reshape(9, 64).to(DEV).contiguous(), b.to(DEV), out2, 1)
    torch.cuda.synchronize()
    assert torch.equal(out2[:B], out[:B]) and float(out2[B:].float().abs().max()) == 0.0
    ops.vgg_conv1_1_fwd(gt.to(DEV), pred.to(DEV), ldp, B, S, w.reshape(9, 64).to(DEV).contiguous(), b.to(DEV), out2, 2)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)
    dz = rnd((B, S, S, 64), 73) * (ref[B:].detach() > 0)
    mask = torch.rand(B, S, S)
    coef = torch.tensor([0.37, 0, 0, 0, 0, 0])
    loss = (ref[B:] * dz.float()).sum() + 0.5 * 0.37 * (mask.unsqueeze(-1) * (pr - gt) ** 2).sum()
    (gp,) = torch.autograd.grad(loss, pr)
    dpred = torch.full((B, S, S, ldp), float('nan'), dtype=torch.bfloat16, device=DEV)
    ops.vgg_conv1_1_bwd(dz.to(DEV), B, S, w.reshape(9, 64).to(DEV).contiguous(), gt.to(DEV), pred.to(DEV), ldp, mask.to(DEV),
                        coef.to(DEV), dpred, ldp)
    torch.cuda.synchronize()
    close(dpred[..., :3], gp, 1e-2, 2e-3, 'vgg1_1_bwd')
    assert float(dpred[..., 3:].float().abs().max()) == 0.0


VGG_HEAD_CASES = [
    # B, S, dtype, store_from (None = B), tag
    (2, 128, torch.bfloat16, None, 'two_patches_per_workgroup'),
    (1, 32, torch.bfloat16, None, 'one_patch_each'),
    (1, 48, torch.bfloat16, 0, 'patch_grid_not_a_power_of_two_store_all'),
    (5, 64, torch.bfloat16, None, 'ragged_patch_counts'),
    (2, 64, torch.float16, None, 'f16'),
    (1, 64, torch.bfloat16, 2, 'store_none'),
]


@pytest.mark.parametrize('B,S,dt,store_from,tag', VGG_HEAD_CASES, ids=[c[-1] for c in VGG_HEAD_CASES])
def test_vgg_head_fused(ops, B, S, dt, store_from, tag):
    """imm_vgg_head_fwd (gray -> conv1_1 -> conv1_2 in one launch, conv1_1's halo produced on the matrix cores from hi + lo
    pairs) against (a) the oracle in fp32 and (b) the two launches it replaces (imm_vgg_conv1_1_fwd + imm_conv2d): conv1_1's
    stored activation within one rounding of the 16-bit type of the VALU kernel's, conv1_2's output within the tolerance of
    test_conv_forward; images below store_from are not written (selfsup/vgg16.py:345-346, build_vgg16.py:22-26)."""
    from imm_amd import _lib as L
    assert ops.vgg_head_supported(B, S, dt)
    ldp = 8
    g = torch.Generator().manual_seed(1234 + S)
    gt = torch.rand(B, S, S, 3, generator=g) * 255
    pred = torch.zeros(B, S, S, ldp); pred[..., :3] = torch.rand(B, S, S, 3, generator=g) * 255
    w11 = rnd((3, 3, 1, 64), 81, 0.4, torch.float32); b11 = rnd((64,), 82, 0.1, torch.float32)
    w12 = rnd((3, 3, 64, 64), 83, 0.05, torch.float32); b12 = rnd((64,), 84, 0.1, torch.float32)
    sf = B if store_from is None else store_from
    fd = ops.fwd_desc(2 * B, S, S, 64, 64, 64, 64, 3, 1, L.CONV_BIAS | L.CONV_RELU)
    wt = torch.zeros(128, fd.kpad, dtype=dt, device=DEV)
    ops.pack_weights(w12.to(DEV), wt, 0, 3, 3, 64, 64, 64, 128, fd.kpad)
    gtd, pd = gt.to(DEV), pred.to(DEV)
    w11d, b11d, b12d = w11.reshape(9, 64).to(DEV).contiguous(), b11.to(DEV), b12.to(DEV)
    # (b) the two-launch sequence
    a_ref = torch.empty(2 * B, S, S, 64, dtype=dt, device=DEV)
    y_ref = torch.empty(2 * B, S, S, 64, dtype=dt, device=DEV)
    ops.vgg_conv1_1_fwd(gtd, pd, ldp, B, S, w11d, b11d, a_ref)
    ops.conv2d(fd, a_ref, wt, b12d, y_ref)
    # the fused launch
    sentinel = 7.0
    a11 = torch.full((2 * B, S, S, 64), sentinel, dtype=dt, device=DEV)
    y12 = torch.full((2 * B, S, S, 64), float('nan'), dtype=dt, device=DEV)
    scratch = torch.empty(ops.vgg_head_scratch_bytes(B, S), dtype=torch.uint8, device=DEV)
    ops.vgg_head_fwd(gtd, pd, ldp, B, S, w11d, b11d, wt, b12d, a11, sf, y12, scratch)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(y12.float()).any()), 'every output pixel is written'
    if sf > 0:
        assert bool((a11[:sf].float() == sentinel).all()), 'images below store_from are not stored'
    if sf < 2 * B:
        # one unit in the last place of the storage type: bf16 2^-7, f16 2^-10 relative (+ a fraction of the tensor's scale near 0)
        ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
        close(a11[sf:], a_ref[sf:], ulp, ulp * 0.05, 'conv1_1 activation (fused) vs vgg_conv1_1_fwd')
        assert float((a11[sf:] != a_ref[sf:]).float().mean()) < 0.02, 'all but a few last-bit roundings are identical'
    close(y12, y_ref, 1.6e-2, 2e-3, 'conv1_2 output (fused) vs conv1_1 + conv2d launches')
    # (a) the oracle, fp32 from the images
    ims = torch.cat([gt, pred[..., :3]], 0)
    gray = ims.mean(dim=3, keepdim=True) / 255.0 - O.VGG_GRAY_MEAN / 255.0
    a_o = torch.relu(O.conv2d_same(gray, w11, b11))
    y_o = torch.relu(O.conv2d_same(a_o.to(dt).float(), w12.to(dt).float(), b12))
    close(y12, y_o, 1.6e-2, 2e-3, 'conv1_2 output (fused) vs oracle')
    if sf < 2 * B:
        close(a11[sf:], a_o[sf:], 8e-3, 1e-3, 'conv1_1 activation (fused) vs oracle')


def test_perceptual_loss_pieces(ops):
    from imm_amd import _lib as L
    B, S = 2, 32
    mask = O.smooth_mask(S, S, margin=2, step=6).reshape(1, S, S).repeat(B, 1, 1).contiguous()
    feats = [(32, 3), (32, 64), (16, 128), (8, 256)]
    nf = len(feats)
    partial = torch.zeros(nf, L.SSE_BLOCKS, device=DEV)
    agg0 = torch.tensor([100.0, 1.6, 2.3, 1.8])
    agg = agg0.clone().to(DEV)
    tens, ref_terms, ref_m, ref_grads = [], [], [], []
    for i, (s, c) in enumerate(feats):
        if i == 0:
            a = torch.rand(B, s, s, 3) * 255; b = torch.zeros(B, s, s, 16); b[..., :3] = torch.rand(B, s, s, 3) * 255
            ops.masked_sse_f32(a.to(DEV), 3, b.to(DEV), 16, B, s, 3, mask.to(DEV), partial[i])
            af, bf = a, b[..., :3].clone().requires_grad_(True)
        else:
            a = torch.relu(rnd((B, s, s, c), 80 + i)); b = torch.relu(rnd((B, s, s, c), 90 + i))
            ops.masked_sse(a.to(DEV), b.to(DEV), B, s, c, mask.to(DEV), S, partial[i])
            af, bf = a.float(), b.float().requires_grad_(True)
        mk = O.loss_mask_at(mask.unsqueeze(-1), s)
        m = ((af - bf) ** 2 * mk).mean()
        wl = O.exp_running_avg(m, agg0[i])
        term = m / wl
        (g,) = torch.autograd.grad(1000.0 * term, bf)
        tens.append((a, b)); ref_terms.append(float(term)); ref_m.append(float(m)); ref_grads.append(g)
    nel = torch.tensor([float(B * s * s * c) for s, c in feats], device=DEV)
    wd = torch.tensor([0.125], device=DEV)
    out = torch.zeros(3 * nf + 3, device=DEV)
    ops.perceptual_finalize(partial, nf, nel, agg, True, wd, out)
    torch.cuda.synchronize()
    o = out.cpu()
    np.testing.assert_allclose(o[:nf].numpy(), ref_terms, rtol=2e-4)
    np.testing.assert_allclose(o[nf:2 * nf].numpy(), ref_m, rtol=2e-4)
    np.testing.assert_allclose(float(o[3 * nf]), 1000 * sum(ref_terms), rtol=2e-4)
    np.testing.assert_allclose(float(o[3 * nf + 2]), 1000 * sum(ref_terms) + 0.125, rtol=2e-4)
    np.testing.assert_allclose(agg.cpu().numpy(), [a + 0.01 * (m - a) for a, m in zip(agg0.tolist(), ref_m)], rtol=1e-5)
    # gradient injection at a tap (feature 2), with an incoming gradient and ReLU mask
    s, c = feats[2]
    a, b = tens[2]
    din = rnd((B, s, s, c), 99)
    da = din.clone().to(DEV)
    ops.tap_grad(da, True, b.to(DEV), a.to(DEV), B, s, c, mask.to(DEV), S, out[2 * nf:], 2, True)
    torch.cuda.synchronize()
    ref = (din.float() + ref_grads[2]) * (b.float() > 0)
    close(da, ref, 1e-2, 2e-3, 'tap_grad')
    da2 = torch.full((B, s, s, c), float('nan'), dtype=torch.bfloat16, device=DEV)
    ops.tap_grad(da2, False, b.to(DEV), a.to(DEV), B, s, c, mask.to(DEV), S, out[2 * nf:], 2, False)
    torch.cuda.synchronize()
    close(da2, ref_grads[2], 1e-2, 2e-3, 'tap_grad_noin')


def test_clip_adam_and_weight_decay(ops):
    sizes = [9 * 32 * 32, 32, 7, 20000, 1]
    wds = [1e-5, 0.0, 0.0, 1e-5, 0.0]
    tab = ops.SegmentTable(sizes, wds, DEV)
    g = torch.Generator().manual_seed(5)
    P = {('t%d/w' % i if wds[i] else 't%d/b' % i): torch.randn(n, generator=g) * 0.3 for i, n in enumerate(sizes)}
    G = {k: torch.randn(v.shape, generator=g) * (3.0 if i % 2 == 0 else 1e-3) for i, (k, v) in enumerate(P.items())}
    flat = lambda d: torch.cat([v.reshape(-1) for v in d.values()]).to(DEV)
    params, grads = flat(P), flat(G) * 2.0          # grads hold the SUM over 2 towers
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    part = torch.empty(tab.nblk, device=DEV); norm2 = torch.empty(tab.nseg, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV); lrs = torch.zeros(2, device=DEV)
    adam_t = torch.zeros(1, dtype=torch.int32, device=DEV)
    wdl = torch.zeros(1, device=DEV)
    ops.weight_decay_loss(params, tab, part, wdl)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(wdl), float(O.weight_decay_loss(P)), rtol=1e-5)
    hp = ops.OptHParams(lr_start=1e-3, lr_decay=0.95, lr_step=100000, lr_multiple=1.0, beta1=0.9, beta2=0.999, eps=1e-8,
                        clip=1.0, grad_scale=0.5)
    opt = O.new_adam_state(P)
    Pref = P
    for it in range(3):
        # oracle: tower-mean gradient + weight decay gradient, clip, Adam
        gref = {k: G[k] + (1e-5 * Pref[k] if k.endswith('/w') else 0) for k in P}
        gref = {k: O.clip_by_norm(x, 1.0) for k, x in gref.items()}
        Pref = O.adam_apply(Pref, gref, opt, lr=O.learning_rate(it))
        grads = flat(G) * 2.0
        ops.clip_adam_step(params, grads, m, v, tab, part, norm2, step, adam_t, lrs, hp)
        torch.cuda.synchronize()
        close(params, flat(Pref), 1e-5, 1e-6, 'adam params step %d' % it)
    assert int(step) == 3 and int(adam_t) == 3
    # TF keeps Adam's bias correction in beta{1,2}_power, apart from global_step: a restored global_step (here 250000, i.e.
    # two staircase decays of the learning rate) with fresh m = v = 0 slots must restart the correction at t = 1
    step.fill_(250000); adam_t.zero_(); m.zero_(); v.zero_()
    p0 = params.clone()
    grads = flat(G) * 2.0
    ops.clip_adam_step(params, grads, m, v, tab, part, norm2, step, adam_t, lrs, hp)
    torch.cuda.synchronize()
    lr = 1e-3 * 0.95 ** 2
    np.testing.assert_allclose(float(lrs[1]), lr, rtol=1e-6)
    np.testing.assert_allclose(float(lrs[0]), lr * (1 - 0.999) ** 0.5 / (1 - 0.9), rtol=1e-5)
    assert int(step) == 250001 and int(adam_t) == 1
    big = grads.abs() > 1e-4          # first bias-corrected step: |dw| = lr * |g| / (|g| + eps') ~ lr
    np.testing.assert_allclose((params - p0).abs()[big].max().item(), lr, rtol=1e-3)


def test_clip_adam_loss_scaling(ops):
    """Loss scaling inside imm_clip_adam_step (f16 gradient storage; the reference is fp32, imm_model.py:97): gradients that
    carry the factor S give the SAME update as unscaled ones (S is a power of two: bitwise), S doubles after
    scale_growth_interval clean steps, and a non-finite gradient anywhere skips the whole update — weights, slots, both
    counters untouched — and halves S."""
    sizes = [9 * 32 * 32, 32, 7, 20000, 1]
    wds = [1e-5, 0.0, 0.0, 1e-5, 0.0]
    tab = ops.SegmentTable(sizes, wds, DEV)
    g = torch.Generator().manual_seed(9)
    params0 = (torch.randn(tab.total, generator=g) * 0.3).to(DEV)
    G = (torch.randn(tab.total, generator=g) * 0.7).to(DEV)
    kw = dict(lr_start=1e-3, lr_decay=0.95, lr_step=100000, lr_multiple=1.0, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0,
              grad_scale=0.5, scale_growth_interval=2, scale_max=4096.0)
    hp = ops.OptHParams(**kw)

    def fresh():
        return dict(params=params0.clone(), m=torch.zeros_like(params0), v=torch.zeros_like(params0),
                    part=torch.empty(tab.nblk, device=DEV), norm2=torch.empty(tab.nseg, device=DEV),
                    step=torch.zeros(1, dtype=torch.int32, device=DEV), adam_t=torch.zeros(1, dtype=torch.int32, device=DEV),
                    lrs=torch.zeros(2, device=DEV))

    def run(st, grads, ls):
        ops.clip_adam_step(st['params'], grads, st['m'], st['v'], tab, st['part'], st['norm2'], st['step'], st['adam_t'], st['lrs'],
                           hp, ls)
        torch.cuda.synchronize()
    a, b = fresh(), fresh()
    ls = torch.tensor([1024.0, 0.0, 0.0, 0.0], device=DEV)
    scales = []
    for it in range(3):
        S = float(ls[0]); scales.append(S)
        run(a, G.clone(), None)
        run(b, G * S, ls)
        assert torch.equal(a['params'], b['params']) and torch.equal(a['m'], b['m']) and torch.equal(a['v'], b['v']), it
    assert scales == [1024.0, 1024.0, 2048.0] and ls.tolist() == [2048.0, 1.0, 0.0, 0.0]      # doubled after 2 clean steps
    assert int(b['step']) == 3 and int(b['adam_t']) == 3
    # overflow: one inf in the smallest tensor -> no update at all, S halved, counters as before, `skipped` = 1
    keep = {k: v.clone() for k, v in b.items()}
    bad = G * float(ls[0]); bad[-1] = float('inf')
    run(b, bad, ls)
    for k in ('params', 'm', 'v', 'step', 'adam_t'):
        assert torch.equal(b[k], keep[k]), k
    assert ls.tolist() == [1024.0, 0.0, 1.0, 1.0]
    bad = G * float(ls[0]); bad[5] = float('nan')
    run(b, bad, ls)
    assert torch.equal(b['params'], keep['params']) and ls.tolist() == [512.0, 0.0, 2.0, 1.0]
    # and the next clean step applies again (a one step behind b's counters now)
    run(a, G.clone(), None); run(b, G * float(ls[0]), ls)
    assert torch.equal(a['params'], b['params']) and ls.tolist() == [512.0, 1.0, 2.0, 0.0] and int(b['step']) == 4
    # S never grows past scale_max, never drops below 1
    ls = torch.tensor([4096.0, 1.0, 0.0, 0.0], device=DEV)
    run(b, G * 4096.0, ls)
    assert float(ls[0]) == 4096.0
    ls = torch.tensor([1.0, 0.0, 0.0, 0.0], device=DEV)
    bad = G.clone(); bad[0] = float('inf')
    run(b, bad, ls)
    assert float(ls[0]) == 1.0 and float(ls[3]) == 1.0


def test_clip_adam_many_workgroups(ops):
    """The optimizer's two launches on a table of thousands of chunks (more workgroups than fit the chip at once; every one
    of them rebuilds its tensor's norm and the skip decision from the chunk sums): same update as a table with 4x larger chunks
    (only the summation order of the norms differs), counters advanced exactly once per step."""
    sizes = [3 * 3 * 266 * 256, 256, 3 * 3 * 256 * 256, 5, 3 * 3 * 128 * 256 + 3, 1, 3 * 3 * 512 * 256]
    wds = [1e-5, 0.0, 1e-5, 0.0, 1e-5, 0.0, 1e-5]
    tabs = []
    for chunk in (2048, 8192):
        old = ops.SegmentTable.CHUNK
        ops.SegmentTable.CHUNK = chunk
        try:
            tabs.append(ops.SegmentTable(sizes, wds, DEV))
        finally:
            ops.SegmentTable.CHUNK = old
    assert tabs[0].nblk > 4 * 256 and tabs[0].nblk > 3 * tabs[1].nblk
    g = torch.Generator().manual_seed(17)
    p0 = (torch.randn(tabs[0].total, generator=g) * 0.3).to(DEV)
    G = (torch.randn(tabs[0].total, generator=g) * 0.05).to(DEV)
    hp = ops.OptHParams(lr_start=1e-3, lr_decay=0.95, lr_step=2, lr_multiple=1.0, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0,
                        grad_scale=1.0, scale_growth_interval=0, scale_max=0.0)
    res = []
    for tab in tabs:
        params, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
        part = torch.empty(tab.nblk, device=DEV); norm2 = torch.empty(tab.nseg, device=DEV)
        step = torch.zeros(1, dtype=torch.int32, device=DEV); adam_t = torch.zeros(1, dtype=torch.int32, device=DEV)
        lrs = torch.zeros(2, device=DEV)
        ls = torch.tensor([8.0, 0.0, 0.0, 0.0], device=DEV)
        for it in range(4):
            ops.clip_adam_step(params, G * 8.0 * (1.0 + it), m, v, tab, part, norm2, step, adam_t, lrs, hp, ls)
            torch.cuda.synchronize()
            assert int(step) == it + 1 and int(adam_t) == it + 1, it
        assert ls.tolist() == [8.0, 4.0, 0.0, 0.0]
        np.testing.assert_allclose(float(lrs[1]), 1e-3 * 0.95 ** 1, rtol=1e-6)        # step 3 (global_step 3 before): 3 // 2 = 1
        res.append((params, m, v, norm2.clone()))
    for a, b, name in zip(res[0], res[1], ('params', 'm', 'v', 'norm2')):
        close(a, b, 1e-5, 1e-7, 'chunk 2048 vs 8192: ' + name)
    # the norms themselves against torch
    off = tabs[0].offsets
    gl = G * 4.0
    for i in range(len(sizes)):
        # last step: g = G * 4 + wd * w_before_last_step; with wd = 1e-5 and |w| ~ 0.3 the wd part is < 1e-4 of the norm
        ref = float((gl[off[i]:off[i + 1]].double() ** 2).sum())
        np.testing.assert_allclose(float(res[0][3][i]), ref, rtol=2e-3)


def test_perceptual_finalize_loss_scale(ops):
    """imm_perceptual_finalize with a loss scale: loss values unchanged, gradient coefficients x S exactly."""
    from imm_amd import _lib as L
    nfeat = 3
    part = torch.rand(nfeat, L.SSE_BLOCKS, device=DEV)
    nel = torch.tensor([1000.0, 5000.0, 64.0], device=DEV)
    wd = torch.tensor([0.25], device=DEV)
    outs = []
    for ls in (None, torch.tensor([256.0, 0, 0, 0], device=DEV)):
        agg = torch.tensor([100.0, 1.6, 2.3], device=DEV)
        out = torch.zeros(3 * nfeat + 3, device=DEV)
        ops.perceptual_finalize(part, nfeat, nel, agg, True, wd, out, False, ops.LOSS_PERCEPTUAL, ls)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    a, b = outs
    assert torch.equal(a[:2 * nfeat], b[:2 * nfeat]) and torch.equal(a[3 * nfeat:], b[3 * nfeat:])
    assert torch.equal(a[2 * nfeat:3 * nfeat] * 256.0, b[2 * nfeat:3 * nfeat])


@pytest.mark.parametrize('optim', ['adadelta', 'adagrad'])
def test_clip_adadelta_adagrad(ops, optim):
    """The other two optimizers scripts/train.py:99-102 offers, behind the same mean -> per-tensor clip front end."""
    from imm_amd import _lib as L
    sizes = [9 * 32 * 32, 32, 7, 20000]
    wds = [1e-5, 0.0, 0.0, 1e-5]
    tab = ops.SegmentTable(sizes, wds, DEV)
    g = torch.Generator().manual_seed(6)
    P = {('t%d/w' % i if wds[i] else 't%d/b' % i): torch.randn(n, generator=g) * 0.3 for i, n in enumerate(sizes)}
    G = {k: torch.randn(v.shape, generator=g) * (3.0 if i % 2 == 0 else 1e-3) for i, (k, v) in enumerate(P.items())}
    flat = lambda d: torch.cat([v.reshape(-1) for v in d.values()]).to(DEV)
    params = flat(P)
    m = torch.zeros_like(params)
    v = torch.full_like(params, 0.1) if optim == 'adagrad' else torch.zeros_like(params)
    part = torch.empty(tab.nblk, device=DEV); norm2 = torch.empty(tab.nseg, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV); adam_t = torch.zeros(1, dtype=torch.int32, device=DEV)
    lrs = torch.zeros(2, device=DEV)
    hp = ops.OptHParams(lr_start=1e-2, lr_decay=0.95, lr_step=100000, lr_multiple=1.0, beta1=0.95 if optim == 'adadelta' else 0.9,
                        beta2=0.999, eps=1e-6 if optim == 'adadelta' else 1e-8, clip=1.0, grad_scale=0.5, optim=L.OPTIMIZERS[optim])
    opt = {'m': {k: torch.zeros_like(x) for k, x in P.items()}, 'v': {k: torch.zeros_like(x) for k, x in P.items()}} \
        if optim == 'adadelta' else O.new_adagrad_state(P)
    Pref = P
    for it in range(3):
        gref = {k: G[k] + (1e-5 * Pref[k] if k.endswith('/w') else 0) for k in P}
        gref = {k: O.clip_by_norm(x, 1.0) for k, x in gref.items()}
        Pref = O.adadelta_apply(Pref, gref, opt, lr=1e-2) if optim == 'adadelta' else O.adagrad_apply(Pref, gref, opt, lr=1e-2)
        grads = flat(G) * 2.0
        ops.clip_adam_step(params, grads, m, v, tab, part, norm2, step, adam_t, lrs, hp)
        torch.cuda.synchronize()
        close(params, flat(Pref), 1e-5, 1e-6, '%s params step %d' % (optim, it))
    close(v, flat(opt['v']), 1e-5, 1e-6, optim + ' accumulator')


def test_graph_capture_replay(ops):
    x = torch.rand(4, 4, 4, 3, device=DEV) * 255
    dst = torch.zeros(4, 4, 4, 8, dtype=torch.bfloat16, device=DEV)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = ops.Graph()
        g.capture_begin()
        ops.pack_image(x, dst, 64)
        g.capture_end()
        assert float(dst.float().abs().max()) == 0.0       # capture does not execute
        g.launch()
        s.synchronize()
        assert torch.equal(dst[..., :3], x.to(torch.bfloat16))
        x.mul_(0.5)
        g.launch()
        s.synchronize()
        assert torch.equal(dst[..., :3], x.to(torch.bfloat16))


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_masked_sse_fused_with_maxpool(ops, dt):
    """imm_masked_sse_pool == imm_masked_sse + imm_maxpool2_fwd on both feature halves (pool bit for bit)."""
    from imm_amd import _lib as L
    B, s, c, S = 3, 16, 64, 64
    a = rnd((B, s, s, c), 301, 1.0, dt).to(DEV).contiguous()
    b = rnd((B, s, s, c), 302, 1.0, dt).to(DEV).contiguous()
    mask = torch.rand(B, S, S, device=DEV)
    for mk in (mask, None):
        p1 = torch.zeros(L.SSE_BLOCKS, device=DEV); p2 = torch.zeros(L.SSE_BLOCKS, device=DEV)
        pa = torch.empty(B, s // 2, s // 2, c, dtype=dt, device=DEV); pb = torch.empty_like(pa)
        ops.masked_sse_pool(a, b, B, s, c, mk, S, p1, pa, pb)
        ops.masked_sse(a, b, B, s, c, mk, S, p2)
        ra = torch.empty_like(pa); rb = torch.empty_like(pb)
        ops.maxpool2_fwd(a, ra, B, s, s, c); ops.maxpool2_fwd(b, rb, B, s, s, c)
        torch.cuda.synchronize()
        assert torch.equal(pa, ra) and torch.equal(pb, rb)
        m4 = mk[:, ::S // s, ::S // s].unsqueeze(-1) if mk is not None else 1.0
        ref = float((m4 * (a.float() - b.float()) ** 2).sum())
        assert abs(float(p1.double().sum()) - ref) <= 1e-5 * abs(ref) + 1e-6
        assert abs(float(p1.double().sum()) - float(p2.double().sum())) <= 1e-5 * abs(ref) + 1e-6


def test_unpool_fused_with_tap_grad(ops):
    """imm_unpool_tap_grad == imm_maxpool2_bwd(relu_mask=0) then imm_tap_grad(has_in, relu), bit for bit."""
    dt = torch.bfloat16
    B, s, c, S = 2, 16, 64, 64
    ap = rnd((B, s, s, c), 311, 1.0, dt).to(DEV).contiguous()
    ap[0, :2, :2, :8] = 0.5          # ties inside a window: the first maximum takes the gradient
    ag = rnd((B, s, s, c), 312, 1.0, dt).to(DEV).contiguous()
    dpool = rnd((B, s // 2, s // 2, c), 313, 1.0, dt).to(DEV).contiguous()
    mask = torch.rand(B, S, S, device=DEV)
    coef = torch.tensor([0.0, 0.37, 0.0, 0.0, 0.0, 0.0], device=DEV)
    for mk in (mask, None):
        ref = torch.full((B, s, s, c), float('nan'), dtype=dt, device=DEV)
        ops.maxpool2_bwd(ap, dpool, ref, B, s, s, c, 0)
        ops.tap_grad(ref, True, ap, ag, B, s, c, mk, S, coef, 1, True)
        got = torch.full((B, s, s, c), float('nan'), dtype=dt, device=DEV)
        ops.unpool_tap_grad(got, dpool, ap, ag, B, s, c, mk, S, coef, 1)
        torch.cuda.synchronize()
        assert torch.equal(got, ref)


@pytest.mark.parametrize('H,ci,co', [(32, 64, 128), (64, 128, 256), (16, 32, 64)], ids=['grouped_64x64', 'grouped_deep', 'grouped_128x32'])
def test_conv_group_equals_sequential(ops, H, ci, co):
    """imm_conv2d_group (stride-2 dgrad parity classes in one launch) == the four launches, bit for bit."""
    B, k, dt = 8, 3, torch.bfloat16
    w = rnd((k, k, ci, co), 401, 0.05)
    dy = rnd((B, H // 2, H // 2, co), 402).to(DEV).contiguous()
    classes = ops.dgrad_s2_class_descs(B, H, H, ci, ci, co, co, k)
    rows = ops.round_up(ci, 128)
    wts = []
    for d, mode in classes:
        wt = torch.zeros(rows, d.kpad, dtype=dt, device=DEV)
        ops.pack_weights(w.float().to(DEV).contiguous(), wt, mode, k, k, ci, co, co, rows, d.kpad)
        wts.append(wt)
    ref = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    for (d, _m), wt in zip(classes, wts):
        ops.conv2d(d, dy, wt, None, ref)
    got = torch.full((B, H, H, ci), float('nan'), dtype=dt, device=DEV)
    grp = ops.ConvGroup([d for d, _m in classes], wts)
    ops.conv2d_group(grp, dy, got)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(got.float()).any())
    assert torch.equal(got, ref)


def test_cost_ema_and_activation_rms(ops):
    """Round 5: the reference's device-side summaries.  imm_cost_ema against oracle.cost_ema_update (base_model.py:52-60, the biased TF-1.10
    moving averages of three costs over several steps); imm_rms16 against sqrt(mean(z^2)) (selfsup/vgg16.py:232-234), bf16 and f16."""
    st = torch.zeros(4, device=DEV)
    ref = [0.0, 0.0, 0.0, 0.0]
    g = torch.Generator().manual_seed(3)
    for it in range(6):
        c = torch.rand(3, generator=g) * 1000.0
        ops.cost_ema(c.to(DEV), st, 0.99)
        ref, avg = O.cost_ema_update(ref, [float(v) for v in c])
    torch.cuda.synchronize()
    got = st.cpu()
    assert float(got[3]) == 6.0
    np.testing.assert_allclose(got[:3].numpy(), ref[:3], rtol=1e-5)
    np.testing.assert_allclose(got[:3].numpy(), avg, rtol=1e-5)
    part, out = torch.zeros(1024, device=DEV), torch.zeros(1, device=DEV)
    for dt in (torch.bfloat16, torch.float16):
        for shape in ((2, 16, 16, 64), (64, 128, 128, 64), (3, 5, 7, 8)):
            x = rnd(shape, 11, 2.0, dt).to(DEV)
            ops.rms16(x, part, out)
            torch.cuda.synchronize()
            want = float(x.float().pow(2).mean().sqrt())
            assert abs(float(out) - want) <= 1e-5 * want, (shape, float(out), want)


def test_upload_and_download_through_pinned_memory(ops):
    """Round 5: host <-> device transfers of parameters, frozen weights, inputs and checkpoints go through pinned staging buffers,
    stream-synchronised, and an upload is read back (ops.upload / ops.download; DESIGN.md §7: a pageable 2.4 / 2.9 MB source freed
    right after `copy_` was what made one of eight co-resident ranks start from other weights about once in 100 constructions)."""
    x = np.random.default_rng(0).standard_normal((3, 3, 320, 256)).astype(np.float32)      # 2.9 MB: the size class that was torn
    d = torch.full(x.shape, float('nan'), device=DEV)
    ops.upload(d, x, 'x')
    del x
    ref = np.random.default_rng(0).standard_normal((3, 3, 320, 256)).astype(np.float32)
    back = ops.download(d)
    assert not back.is_cuda and not back.is_pinned() and torch.equal(back, torch.from_numpy(ref))
    # layout and dtype conversion happen on the host side of the staging buffer
    src = torch.arange(35.).reshape(7, 5).t()                                               # non-contiguous
    d16 = torch.empty(5, 7, dtype=torch.bfloat16, device=DEV)
    ops.upload(d16, src)
    assert torch.equal(d16.float().cpu(), src.contiguous())
    # a flat destination takes any source of the same size; a device source is a plain stream-ordered copy
    flat = torch.empty(35, device=DEV)
    ops.upload(flat, src)
    assert torch.equal(flat.cpu(), src.reshape(-1))
    ops.upload(flat, (flat * 2).reshape(5, 7))
    assert torch.equal(flat.cpu(), 2 * src.reshape(-1))
    with pytest.raises(RuntimeError):
        ops.upload(flat, torch.zeros(36))
    # host destination (CPU tools): a plain copy
    h = torch.empty(35)
    ops.upload(h, src)
    assert torch.equal(h, src.reshape(-1)) and torch.equal(ops.download(h), h)


def test_pinned_stager_and_to_device_pinned(ops):
    """Round 6: per-step host inputs go through ops.PinnedStager — two persistent pinned buffers per
    destination, alternated, asynchronous, no read-back — and scripts / datasets put host arrays on the device with
    ops.to_device_pinned.  A source overwritten right after the call (what a training loop's generator does) must not reach the
    device; many consecutive steps keep the right bytes; buffers are reused, not re-allocated."""
    st = ops.PinnedStager()
    dst = torch.empty(32, 128, 128, 3, device=DEV)                                          # 6.3 MB: a real input batch
    g = torch.Generator().manual_seed(5)
    for it in range(6):
        src = torch.rand(32, 128, 128, 3, generator=g)
        want = src.clone()
        st.copy(dst, src, 'image')
        src.fill_(-1.0)                                                                      # the caller's buffer is free at once
        if it == 2:
            bufs = [b.data_ptr() for b in st._slots['image']['bufs']]
        if it >= 2:
            assert [b.data_ptr() for b in st._slots['image']['bufs']] == bufs
        assert torch.equal(dst.cpu(), want), it
    # dtype / layout conversion on the host side, device sources as plain stream-ordered copies
    d16 = torch.empty(5, 7, dtype=torch.bfloat16, device=DEV)
    srcT = torch.arange(35.).reshape(7, 5).t()
    st.copy(d16, srcT, 'x')
    assert torch.equal(d16.float().cpu(), srcT.contiguous())
    st.copy(d16, (d16.float() * 2).to(torch.bfloat16), 'x')
    assert torch.equal(d16.float().cpu(), 2 * srcT.contiguous())
    a = np.random.default_rng(1).standard_normal((64, 128, 128)).astype(np.float64)       # 8 MB, converted to f32 on the way
    t = ops.to_device_pinned(a, DEV, torch.float32)
    a_ref = a.astype(np.float32)
    a[:] = 0
    assert t.is_cuda and t.dtype == torch.float32 and torch.equal(t.cpu(), torch.from_numpy(a_ref))
