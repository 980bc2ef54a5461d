"""End-to-end rate of training from image FILES (CelebA loader: decode processes -> GPU resize/crop/TPS -> training step)
next to the same step on a resident synthetic batch.  Usage: python tools/bench_train_from_files.py [--steps 150]"""
import argparse
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from dataset_fixtures import make_celeba_tree            # noqa: E402
from imm_amd.datasets import CelebADataset                # noqa: E402
from imm_amd.models.imm_model import IMMModel             # noqa: E402
from imm_amd.train.cnn_train_multi import TrainStep       # noqa: E402
from imm_amd.utils.config import load_configs             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=150)
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--n', type=int, default=512)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    cfg = load_configs([os.path.join(ROOT, 'tests', 'configs', 'paths.yaml'), os.path.join(ROOT, 'tests', 'configs', 'smoke-10pts.yaml')])
    model = IMMModel(cfg.model, device='cuda:0')
    ts = TrainStep(model, args.batch, 128, world_size=1, use_graph=True)
    with tempfile.TemporaryDirectory() as root:
        make_celeba_tree(root, n=args.n, fmt='jpeg')
        loader = CelebADataset(root, 'train', dataset='celeba').get_dataset(args.batch, repeat=True, device='cuda:0')
        it = iter(loader)
        first = next(it)
        for _ in range(10):
            ts.step(first)
        ts.synchronize()
        t0 = time.time()
        for _ in range(args.steps):
            ts.step(None)                                  # inputs resident: the bench.py condition
        ts.synchronize()
        resident = args.steps * args.batch / (time.time() - t0)
        t0 = time.time()
        for _ in range(args.steps):
            ts.step(next(it))
        ts.synchronize()
        files = args.steps * args.batch / (time.time() - t0)
        t0 = time.time()
        for _ in range(args.steps):
            next(it)
        torch.cuda.synchronize()
        only_loader = args.steps * args.batch / (time.time() - t0)
        it.close()
        loader.close()
    print('training step, resident batch : %8.0f images/s' % resident)
    print('loader alone                  : %8.0f images/s' % only_loader)
    print('training from JPEG files      : %8.0f images/s  (%.0f %% of the resident-batch rate)' % (files, 100 * files / resident))


if __name__ == '__main__':
    main()
