"""Landmark-regression evaluation (reference: scripts/test.py).  Same flags; because the CelebA/MAFL/AFLW loaders are out
of this build's scope the two splits come from .npz files with `image`, `future_image` (NHWC float32,
[0,255]) and `future_landmarks` ([N,L,2] pixels, first two points = the eyes):
    python scripts/test.py --experiment-name celeba-10pts --train-npz mafl_train.npz --test-npz mafl_test.npz \\
        --checkpoint logs/model.ckpt-100.pt"""
from __future__ import print_function

import argparse
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from imm_amd.eval import eval_imm                     # noqa: E402
from imm_amd.models.imm_model import IMMModel        # noqa: E402
from imm_amd.utils.config import load_configs        # noqa: E402


def npz_batches(path, batch_size, device):
    d = np.load(path)
    n = d['image'].shape[0]
    for i in range(0, n, batch_size):
        sl = slice(i, min(i + batch_size, n))
        yield {'image': torch.from_numpy(d['image'][sl]).float().to(device),
               'future_image': torch.from_numpy(d['future_image'][sl]).float().to(device),
               'future_landmarks': d['future_landmarks'][sl]}


def main(args):
    config = load_configs([args.paths_config, osp.join('configs', 'experiments', args.experiment_name + '.yaml')]
                          if args.configs is None else args.configs)
    torch.cuda.set_device(0)
    net = IMMModel(config.model, device='cuda:0')
    if args.checkpoint is None or not osp.isfile(args.checkpoint):
        raise ValueError('Checkpoint file %s not found.' % args.checkpoint)
    ck = torch.load(args.checkpoint, map_location='cpu')
    d0 = np.load(args.train_npz)
    net._get_engine(min(args.batch_size, d0['image'].shape[0]), args.im_size).load_parameters(ck['params'], ck.get('state'))
    err = eval_imm.evaluate_regression(net, npz_batches(args.train_npz, args.batch_size, 'cuda:0'),
                                       npz_batches(args.test_npz, args.batch_size, 'cuda:0'),
                                       [args.im_size, args.im_size], batch_size=args.batch_size, bias=args.bias)
    print('')
    print('========================= RESULTS =========================')
    print('checkpoint: %s' % args.checkpoint)
    print('error: %.5f (fraction of the inter-ocular distance)' % err)
    print('===========================================================')


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Test model on face datasets.')
    parser.add_argument('--experiment-name', type=str, required=False, default=None, help='Name of the experiment to evaluate.')
    parser.add_argument('--train-dataset', type=str, default='mafl', help='kept for flag compatibility')
    parser.add_argument('--test-dataset', type=str, default='mafl', help='kept for flag compatibility')
    parser.add_argument('--paths-config', type=str, default='configs/paths/default.yaml', required=False)
    parser.add_argument('--iteration', type=int, default=None)
    parser.add_argument('--test-split', type=str, default='test')
    parser.add_argument('--buffer-name', type=str, default=None)
    parser.add_argument('--im-size', type=int, default=128)
    parser.add_argument('--bias', action='store_true', required=False, help='Use bias in the regression.')
    parser.add_argument('--batch-size', type=int, default=100, required=False)
    # additions of this build
    parser.add_argument('--configs', nargs='+', default=None, help='explicit config files (instead of --experiment-name)')
    parser.add_argument('--checkpoint', type=str, default=None)
    parser.add_argument('--train-npz', type=str, required=True)
    parser.add_argument('--test-npz', type=str, required=True)
    main(parser.parse_args())
