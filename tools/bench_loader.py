"""Throughput of the CelebA input pipeline (imm_amd/datasets): JPEG decode on host threads + resize/crop/TPS on the GPU,
next to the training step's rate.  Writes a synthetic CelebA tree of aligned-size (218x178) JPEGs to a temp directory.
Usage: python tools/bench_loader.py [--n 512] [--batch 32] [--threads 12] [--batches 60]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from dataset_fixtures import make_celeba_tree            # noqa: E402
from imm_amd.datasets import CelebADataset                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=512)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--threads', type=int, default=12)
    ap.add_argument('--batches', type=int, default=60)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    with tempfile.TemporaryDirectory() as root:
        t0 = time.time()
        make_celeba_tree(root, n=args.n, fmt='jpeg')
        print('wrote %d JPEGs in %.1f s; host cores %d' % (args.n, time.time() - t0, os.cpu_count()))
        ds = CelebADataset(root, 'train', dataset='celeba')
        for threads in sorted(set([0, 1, 4, args.threads, 2 * args.threads, 4 * args.threads])):
            ld = ds.get_dataset(args.batch, repeat=True, num_preprocess_threads=max(threads, 4), device='cuda:0')
            ld.decode_processes = threads           # 0 = threads of this process only
            # host stage alone (sample stream + decode)
            hb = ld.host_batches()
            next(hb)
            t0 = time.time()
            for _ in range(args.batches):
                next(hb)
            host = args.batches * args.batch / (time.time() - t0)
            hb.close()
            # whole loader (decode ahead in the producer thread, GPU stage in the consumer)
            it = iter(ld)
            next(it)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(args.batches):
                b = next(it)
            torch.cuda.synchronize()
            full = args.batches * args.batch / (time.time() - t0)
            it.close()
            ld.close()
            print('decode processes %2d%s: decode only %8.0f img/s   full loader %8.0f img/s' % (
                threads, ' (4 threads in-process)' if threads == 0 else '', host, full))
        # device stage alone: pack + H2D + resize/crop + two TPS warps on already decoded images
        samples, decoded = next(ds.get_dataset(args.batch, device='cuda:0').host_batches())
        for _ in range(3):
            ds._device_batch(samples, decoded, 'cuda:0')
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(50):
            ds._device_batch(samples, decoded, 'cuda:0')
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        print('device stage (pack + H2D + imm_resize_crop_u8 + 2x imm_tps_warp): %.3f ms per batch of %d = %.0f img/s'
              % (dt * 1e3, args.batch, args.batch / dt))
        assert b['image'].shape == (args.batch, 128, 128, 3)


if __name__ == '__main__':
    main()
