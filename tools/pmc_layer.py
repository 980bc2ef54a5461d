"""Run ONE convolution layer of tools/bench_conv.py's table a few times (for `rocprofv3 --pmc ...` runs).
Usage: python tools/pmc_layer.py <substring of the layer tag> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imm_amd import _lib as L   # noqa: E402
from imm_amd import ops        # noqa: E402
from tools.bench_conv import LAYERS, DEV   # noqa: E402


def main():
    key = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    torch.cuda.set_device(0)
    dt = torch.bfloat16
    for tag, n, H, ci, co, k, stride in LAYERS:
        if key not in tag:
            continue
        x = (torch.randn(n, H, H, ci, device=DEV) * 0.5).to(dt)
        w = torch.randn(k, k, ci, co, device=DEV) * 0.05
        b = torch.zeros(co, device=DEV)
        use_mask = 'mask' in tag
        mref = (torch.randn(n, H, H, co, device=DEV)).to(dt) if use_mask else None
        flags = L.CONV_MASK if use_mask else (L.CONV_BIAS | L.CONV_RELU)
        desc = ops.fwd_desc(n, H, H, ci, ci, co, co, k, stride, flags, ldmask=co if use_mask else 0)
        wt = torch.zeros(ops.round_up(co, 128), desc.kpad, dtype=dt, device=DEV)
        ops.pack_weights(w, wt, 0, k, k, ci, co, ci, wt.shape[0], desc.kpad)
        y = torch.empty(n, desc.ho, desc.wo, co, dtype=dt, device=DEV)
        for _ in range(reps):
            ops.conv2d(desc, x, wt, None if use_mask else b, y, None, mref)
        torch.cuda.synchronize()
        print('ran', tag, reps)


if __name__ == '__main__':
    main()
