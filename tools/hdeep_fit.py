"""Per-shape timing of the LDS-halo deep-K kernel (conv_hdeep.hip) — time is linear in the number of 64-channel slices;
run under IMM_HDEEP_NO_BIG / IMM_HDEEP_SMALL_BELOW / IMM_HDEEP_NO_SMALL to compare its tile plans.
Usage: python tools/hdeep_fit.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imm_amd import _lib as L   # noqa: E402
from imm_amd import ops   # noqa: E402
from tools.bench_conv import time_launch   # noqa: E402

DEV = 'cuda:0'


def main():
    dt = torch.bfloat16
    torch.cuda.set_device(0)
    shapes = [(64, 16, 512, 512), (64, 16, 256, 512), (64, 32, 256, 256), (64, 32, 128, 256), (64, 64, 128, 128), (64, 64, 64, 128),
              (32, 16, 512, 512), (32, 32, 256, 256), (32, 64, 128, 128), (32, 16, 256, 256), (32, 32, 128, 128), (32, 64, 128, 64)]
    for n, H, ci, co in shapes:
        x = (torch.randn(n, H, H, ci, device=DEV) * 0.5).to(dt)
        w = torch.randn(3, 3, ci, co, device=DEV) * 0.05
        b = torch.zeros(co, device=DEV)
        desc = ops.fwd_desc(n, H, H, ci, ci, co, co, 3, 1, L.CONV_BIAS | L.CONV_RELU)
        wt = torch.zeros(ops.round_up(co, 128), desc.kpad, dtype=dt, device=DEV)
        ops.pack_weights(w, wt, 0, 3, 3, ci, co, ci, wt.shape[0], desc.kpad)
        y = torch.empty(n, H, H, co, dtype=dt, device=DEV)
        us = time_launch(lambda: ops.conv2d(desc, x, wt, b, y), reps=30)
        print('n=%d H=%d ci=%4d co=%4d  %7.1f us  %7.1f TF' % (n, H, ci, co, us, 2.0 * n * H * H * 9 * ci * co / us / 1e6))


if __name__ == '__main__':
    main()
