"""Times the GPU TPS augmentation (two warps of a batch-32 128x128 mask||image tensor = one training batch) and the
numpy oracle on a bounded sample.  Usage: python tools/bench_tps.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imm_amd.data import tps   # noqa: E402
from oracle import tps_oracle as T   # noqa: E402


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
    img = (np.random.rand(4, S, S, 3) * 255).astype(np.float32)
    msk = np.random.rand(4, S, S, 1).astype(np.float32)
    w1, w2 = wt[:4].cpu().numpy(), ws[:4].cpu().numpy()
    T.apply_pair(img, msk, w1, w2)
    t0 = time.time()
    for _ in range(3):
        T.apply_pair(img, msk, w1, w2)
    dt = (time.time() - t0) / 3
    print('CPU oracle (numpy, basis rebuilt per call like a cold cache): %.1f ms per 4 pairs = %.0f pairs/s' % (dt * 1e3, 4 / dt))


if __name__ == '__main__':
    main()
