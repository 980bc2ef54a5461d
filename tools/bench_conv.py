"""Micro-benchmark of the implicit-GEMM convolution kernels on representative IMM layers (MI355X).
Usage: python tools/bench_conv.py [--ablate]   (writes a table to stdout)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imm_amd import _lib as L   # noqa: E402
from imm_amd import ops        # noqa: E402

DEV = 'cuda:0'
LAYERS = [  # tag, images, H, ci, co, k, stride
    ('enc_conv2  128^2 32->32', 32, 128, 32, 32, 3, 1),
    ('enc_conv3  128^2 32->64 s2', 32, 128, 32, 64, 3, 2),
    ('vgg1_2     128^2 64->64', 64, 128, 64, 64, 3, 1),
    ('vgg1_2 dgrad+mask 64->64', 32, 128, 64, 64, 3, 1),
    ('enc_conv4   64^2 64->64', 32, 64, 64, 64, 3, 1),
    ('ren_conv7d 128^2 32->64', 32, 128, 32, 64, 3, 1),
    ('vgg2_2      64^2 128->128', 64, 64, 128, 128, 3, 1),
    ('vgg3_2      32^2 256->256', 64, 32, 256, 256, 3, 1),
    ('vgg4_2      16^2 512->512', 64, 16, 512, 512, 3, 1),
    ('vgg5_1       8^2 512->512', 64, 8, 512, 512, 3, 1),
    ('ren_conv1   16^2 288->256', 32, 16, 288, 256, 3, 1),
    ('ren_conv7  128^2 64->32', 32, 128, 64, 32, 3, 1),
    ('ren_conv5   64^2 128->64', 32, 64, 128, 64, 3, 1),
    ('ren_conv3   32^2 256->128', 32, 32, 256, 128, 3, 1),
    ('ren_conv2   16^2 256->256', 32, 16, 256, 256, 3, 1),
    ('enc_conv6   32^2 128->128', 32, 32, 128, 128, 3, 1),
]


def time_launch(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ablate', action='store_true')
    ap.add_argument('--wgrad', action='store_true')
    ap.add_argument('--bn', action='store_true', help='flags of a conv + BN block: bias + batch-norm partial sums, no ReLU')
    ap.add_argument('--layers', default='', help="probe shapes instead of the table: 'images,H,ci,co[,k[,stride]];...'")
    args = ap.parse_args()
    global LAYERS
    if args.layers:
        LAYERS = []
        for item in args.layers.split(';'):
            v = [int(t) for t in item.split(',')]
            n, H, ci, co = v[:4]
            LAYERS.append(('probe %d x %d^2 %d->%d' % (n, H, ci, co), n, H, ci, co, v[4] if len(v) > 4 else 3, v[5] if len(v) > 5 else 1))
    torch.cuda.set_device(0)
    dt = torch.bfloat16
    variants = [('full', 0)]
    if args.ablate:
        variants += [('no_gload', 0x100), ('no_lds_store', 0x200), ('no_mfma', 0x400), ('no_epilogue', 0x800),
                     ('only_mfma', 0x100 | 0x200 | 0x800), ('only_loads', 0x400 | 0x800)]
    print('%-30s %-13s %10s %9s %9s' % ('layer', 'variant', 'us', 'TFLOP/s', 'GB/s(min)'))
    for tag, n, H, ci, co, k, stride in LAYERS:
        x = (torch.randn(n, H, H, ci, device=DEV) * 0.5).to(dt)
        w = torch.randn(k, k, ci, co, device=DEV) * 0.05
        b = torch.zeros(co, device=DEV)
        use_mask = 'mask' in tag
        mref = (torch.randn(n, H, H, co, device=DEV)).to(dt) if use_mask else None
        for vname, vflag in variants:
            if use_mask:
                desc = ops.fwd_desc(n, H, H, ci, ci, co, co, k, stride, L.CONV_MASK | vflag, ldmask=co)
            else:
                desc = ops.fwd_desc(n, H, H, ci, ci, co, co, k, stride, (L.CONV_BIAS | L.CONV_STATS if args.bn else L.CONV_BIAS | L.CONV_RELU) | vflag)
            wt = torch.zeros(ops.round_up(co, 128), desc.kpad, dtype=dt, device=DEV)
            ops.pack_weights(w, wt, 0, k, k, ci, co, ci, wt.shape[0], desc.kpad)
            y = torch.empty(n, desc.ho, desc.wo, co, dtype=dt, device=DEV)
            st = torch.empty(ops.conv_stats_blocks(desc), 2, co, device=DEV) if args.bn else None
            us = time_launch(lambda: ops.conv2d(desc, x, wt, None if use_mask else b, y, st, mref))
            flops = 2.0 * n * desc.ho * desc.wo * k * k * ci * co
            nbytes = x.numel() * 2 + y.numel() * 2 + wt.numel() * 2 + (mref.numel() * 2 if use_mask else 0)
            print('%-30s %-13s %10.1f %9.1f %9.0f  %s' % (tag, vname, us, flops / us / 1e6, nbytes / us / 1e3, ops.conv2d_variant(desc, dt)[0]))
        if args.wgrad:
            desc = ops.fwd_desc(n, H, H, ci, ci, co, co, k, stride, 0)
            dy = (torch.randn(n, desc.ho, desc.wo, co, device=DEV)).to(dt)
            npix = n * desc.ho * desc.wo
            bn_w = 128 if co > 64 else 64 if co > 32 else 32 if co > 16 else 16
            tiles = -(-desc.kpad // 128) * -(-co // bn_w)
            nsplit = max(1, min(-(-512 // tiles), max(1, npix // 512)))
            forced = ops.conv2d_wgrad_splits(desc, co)
            if forced > 0:
                nsplit = forced
            slab = torch.empty(nsplit, desc.kpad, co, device=DEV)
            us = time_launch(lambda: ops.conv2d_wgrad(desc, x, dy, co, slab, nsplit))
            flops = 2.0 * npix * k * k * ci * co
            print('%-30s %-13s %10.1f %9.1f   nsplit=%d' % (tag, 'wgrad', us, flops / us / 1e6, nsplit))


if __name__ == '__main__':
    main()
