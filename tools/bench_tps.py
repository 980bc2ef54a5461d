"""Times the GPU TPS augmentation (two warps of a batch-32 128x128 mask||image tensor = one training batch).
Usage: python tools/bench_tps.py   (the numpy restatement is timed by tests/test_tps_gpu.py::test_cpu_restatement_rate)"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imm_amd.data import tps   # noqa: E402


def main():
    B, S = 32, 128
    dev = 'cuda:0'
    torch.cuda.set_device(0)
    aug = tps.TPSPairAugmenter((S, S), device=dev, rng=np.random.RandomState(0))
    image = torch.rand(B, S, S, 3, device=dev) * 255
    mask = torch.rand(B, S, S, 1, device=dev)
    wt, ws = aug.target.sample_params(B), aug.source.sample_params(B)
    outs = [torch.empty(B, S, S, 3, device=dev), torch.empty(B, S, S, 3, device=dev), torch.empty(B, S, S, device=dev)]
    for _ in range(3):
        aug(image, mask, *outs, w_target=wt, w_source=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x4 = aug._stack
    e0.record()
    for _ in range(50):
        aug.target.warp(x4, wt, dst=aug._future, dst_c0=outs[2], dst_rest=outs[1])
        aug.source.warp(aug._future, ws, dst_rest=outs[0])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nbytes = B * S * S * 4 * 4 * 2 + B * S * S * (4 + 3 + 1 + 3) * 4          # reads of the two sources + all outputs
    print('GPU: %.1f us per batch of %d pairs (2 warps) = %.0f pairs/s; algorithmic %.1f MB -> %.0f GB/s' % (
        us, B, B / us * 1e6, nbytes / 1e6, nbytes / us / 1e3))


if __name__ == '__main__':
    main()
