"""Landmark-regression evaluation (reference: scripts/test.py, same flags).  The regressor's training split and the
test split come from the MAFL / AFLW loaders (imm_amd/datasets, tps=False, ordered stream) like the reference:
    python scripts/test.py --experiment-name celeba-10pts --train-dataset mafl --test-dataset mafl [--iteration N]
or, without dataset files, from .npz files with `image`, `future_image` (NHWC float32, [0,255]) and `future_landmarks`
([N,L,2] (y, x) pixels, first two points = the eyes):
    python scripts/test.py --configs a.yaml b.yaml --train-npz mafl_train.npz --test-npz mafl_test.npz --checkpoint x.pt"""
from __future__ import print_function

import argparse
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from imm_amd import ops                               # noqa: E402
from imm_amd.eval import eval_imm                     # noqa: E402
from imm_amd.models.imm_model import IMMModel        # noqa: E402
from imm_amd.utils.config import load_configs        # noqa: E402
from imm_amd.utils.dataset_import import import_dataset   # noqa: E402


def npz_batches(path, batch_size, device):
    d = np.load(path)
    n = d['image'].shape[0]
    for i in range(0, n, batch_size):
        sl = slice(i, min(i + batch_size, n))
        yield {'image': ops.to_device_pinned(d['image'][sl], device, torch.float32),
               'future_image': ops.to_device_pinned(d['future_image'][sl], device, torch.float32),
               'future_landmarks': d['future_landmarks'][sl]}


def landmark_dataset(name, datadir, subset, im_size, max_samples=None):
    """scripts/test.py:88-116 of the reference."""
    if name == 'mafl':
        return import_dataset('celeba')(datadir, dataset='mafl', subset=subset, order_stream=True, max_samples=max_samples,
                                        tps=False, image_size=[im_size, im_size])
    if name == 'aflw':
        return import_dataset('aflw')(datadir, subset=subset, order_stream=True, max_samples=max_samples, tps=False,
                                      image_size=[im_size, im_size])
    raise ValueError('Dataset %s not supported.' % name)


def main(args):
    config = load_configs([args.paths_config, osp.join('configs', 'experiments', args.experiment_name + '.yaml')]
                          if args.configs is None else args.configs)
    torch.cuda.set_device(0)
    dev = 'cuda:0'
    net = IMMModel(config.model, device=dev)
    ckpt = args.checkpoint
    if ckpt is None:
        ckpt = osp.join(config.training.logdir, 'model.ckpt' + ('-%d' % args.iteration if args.iteration is not None else ''))
        if not osp.isfile(ckpt + '.index'):
            ckpt += '.pt'
    is_tf = osp.isfile(ckpt + '.index')       # a TensorFlow bundle prefix (the authors' released checkpoints)
    if not is_tf and not osp.isfile(ckpt):
        raise ValueError('Checkpoint file %s not found.' % ckpt)
    if args.train_npz is not None:
        first = min(args.batch_size, np.load(args.train_npz)['image'].shape[0])
        train_it = npz_batches(args.train_npz, args.batch_size, dev)
        test_it = npz_batches(args.test_npz, args.batch_size, dev)
    else:
        train_dset = landmark_dataset(args.train_dataset, config.training.datadir, 'train', args.im_size)
        test_dset = landmark_dataset(args.test_dataset, config.training.datadir, args.test_split, args.im_size)
        first = min(args.batch_size, train_dset.num_samples())
        train_it = train_dset.get_dataset(args.batch_size, repeat=False, shuffle=False, device=dev)
        test_it = test_dset.get_dataset(args.batch_size, repeat=False, shuffle=False, device=dev)
    eng = net._get_engine(first, args.im_size)
    if is_tf:
        from imm_amd.utils.tf_checkpoint import load_tf_checkpoint
        load_tf_checkpoint(eng, ckpt)
    else:
        ck = torch.load(ckpt, map_location='cpu')
        eng.load_parameters(ck['params'], ck.get('state'))
    err = eval_imm.evaluate_regression(net, train_it, test_it, [args.im_size, args.im_size], batch_size=args.batch_size,
                                       bias=args.bias)
    model_dataset = config.training.train_dset_params.dataset if hasattr(config.training, 'train_dset_params') and \
        'dataset' in config.training.train_dset_params else getattr(config.training, 'dset', '?')
    print('')
    print('========================= RESULTS =========================')
    print('model trained in unsupervised way on %s dataset' % model_dataset)
    print('regressor trained on %s training set' % args.train_dataset)
    print('error on %s datset %s set: %.5f (%.3f percent)' % (args.test_dataset, args.test_split, err, err * 100.0))
    print('===========================================================')


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Test model on face datasets.')
    parser.add_argument('--experiment-name', type=str, required=False, default=None, help='Name of the experiment to evaluate.')
    parser.add_argument('--train-dataset', type=str, default='mafl', help='Training dataset for regressor (mafl|aflw).')
    parser.add_argument('--test-dataset', type=str, default='mafl', help='Testing dataset for regressed landmarks (mafl|aflw).')
    parser.add_argument('--paths-config', type=str, default='configs/paths/default.yaml', required=False)
    parser.add_argument('--iteration', type=int, default=None, help='Checkpoint iteration to evaluate.')
    parser.add_argument('--test-split', type=str, default='test', help='Test split (val|test).')
    parser.add_argument('--buffer-name', type=str, default=None)
    parser.add_argument('--im-size', type=int, default=128)
    parser.add_argument('--bias', action='store_true', required=False, help='Use bias in the regression.')
    parser.add_argument('--batch-size', type=int, default=100, required=False)
    # additions of this build
    parser.add_argument('--configs', nargs='+', default=None, help='explicit config files (instead of --experiment-name)')
    parser.add_argument('--checkpoint', type=str, default=None, help='explicit checkpoint file (default: <logdir>/model.ckpt[-N].pt)')
    parser.add_argument('--train-npz', type=str, default=None)
    parser.add_argument('--test-npz', type=str, default=None)
    main(parser.parse_args())
