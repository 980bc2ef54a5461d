"""A/B of the fused VGG head (imm_vgg_head_fwd) against the two launches it replaces, batch 32 x 128^2 (MI355X)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imm_amd import _lib as L   # noqa: E402
from imm_amd import ops        # noqa: E402

DEV = 'cuda:0'


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    torch.cuda.set_device(0)
    B, S, ldp = int(os.environ.get('B', 32)), int(os.environ.get('S', 128)), 3
    for dt in (torch.bfloat16, torch.float16):
        gt = torch.rand(B, S, S, 3, device=DEV) * 255
        pred = torch.rand(B, S, S, ldp, device=DEV) * 255
        w11 = torch.randn(9, 64, device=DEV) * 0.4; b11 = torch.randn(64, device=DEV) * 0.1
        w12 = torch.randn(3, 3, 64, 64, device=DEV) * 0.05; b12 = torch.randn(64, device=DEV) * 0.1
        fd = ops.fwd_desc(2 * B, S, S, 64, 64, 64, 64, 3, 1, L.CONV_BIAS | L.CONV_RELU)
        wt = torch.zeros(128, fd.kpad, dtype=dt, device=DEV)
        ops.pack_weights(w12, wt, 0, 3, 3, 64, 64, 64, 128, fd.kpad)
        a = torch.empty(2 * B, S, S, 64, dtype=dt, device=DEV)
        y = torch.empty(2 * B, S, S, 64, dtype=dt, device=DEV)
        y2 = torch.empty_like(y); a2 = torch.zeros_like(a)
        scratch = torch.empty(ops.vgg_head_scratch_bytes(B, S), dtype=torch.uint8, device=DEV)
        t1 = timeit(lambda: ops.vgg_conv1_1_fwd(gt, pred, ldp, B, S, w11, b11, a))
        t2 = timeit(lambda: ops.conv2d(fd, a, wt, b12, y))
        t12 = timeit(lambda: (ops.vgg_conv1_1_fwd(gt, pred, ldp, B, S, w11, b11, a), ops.conv2d(fd, a, wt, b12, y)))
        tf = timeit(lambda: ops.vgg_head_fwd(gt, pred, ldp, B, S, w11, b11, wt, b12, a2, B, y2, scratch))
        tf0 = timeit(lambda: ops.vgg_head_fwd(gt, pred, ldp, B, S, w11, b11, wt, b12, a2, 2 * B, y2, scratch))
        d = (y2.float() - y.float()).abs().max().item()
        da = (a2[B:].float() - a[B:].float()).abs().max().item()
        print('%s  conv1_1 %.1f us + conv1_2 %.1f us = %.1f us back to back | fused (store pred half) %.1f us | fused (store none) %.1f us'
              ' | max |dy| %.4g (y max %.3g)  max |da11| %.4g' % (str(dt)[6:], t1, t2, t12, tf, tf0, d, y.float().abs().max().item(), da))


if __name__ == '__main__':
    main()
