"""Training entry point — counterpart of /root/reference/scripts/train.py (same flags, same YAML files).

    python scripts/train.py --configs configs/paths/default.yaml configs/experiments/celeba-10pts.yaml [--ngpus N]
    torchrun --nnodes=1 --nproc-per-node N scripts/train.py --configs ... --ngpus N      (one process per GPU)

The training step runs on the HIP engine (imm_amd/engine.py -> libimm_hip.so).  Data: by default the dataset the
config names (`training.dset` + `train_dset_params` / `test_dset_params` under `training.datadir`, like the
reference: imm_amd/datasets — JPEG decode on host threads, resize/crop/TPS pairs on the GPU); `--synthetic` uses random
batches instead (no files needed), `--data-npz` pre-rendered pairs {image, future_image, mask} (float32 NHWC).
"""
from __future__ import print_function

import argparse
import os
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from imm_amd import ops                                 # noqa: E402
from imm_amd.models.imm_model import IMMModel          # noqa: E402
from imm_amd.train import cnn_train_multi as tru        # noqa: E402
from imm_amd.utils.config import load_configs           # noqa: E402


class model_factory():
    """scripts/train.py:30-40."""

    def __init__(self, network, **kwargs):
        self.network = network
        self.net_args = kwargs

    def create(self):
        return self.network(**self.net_args)


def smooth_mask(h, w, margin=10, step=20):
    """imm/datasets/tps_dataset.py:47-67."""
    from imm_amd.datasets.tps_dataset import smooth_mask as _m
    return torch.from_numpy(_m(h, w, margin, step))


def dataset_loaders(train_config, batch_per_rank, size, device, rank, world):
    """scripts/train.py:115-148 of the reference: the training stream (repeat, no shuffle, 12 decode threads) and the
    test stream (one pass) from the dataset class the config names."""
    from imm_amd.utils.dataset_import import import_dataset
    dset_class = import_dataset(train_config.dset)
    train_params, test_params = {}, {}
    train_subset, test_subset = 'train', 'test'
    if hasattr(train_config, 'train_dset_params'):
        train_params.update(train_config.train_dset_params)
        train_subset = train_params.pop('subset', train_subset)
    if hasattr(train_config, 'test_dset_params'):
        test_params.update(train_config.test_dset_params)
        test_subset = test_params.pop('subset', test_subset)
    if hasattr(train_config, 'max_test_samples'):
        raise ValueError('max_test_samples attribute deprecated')
    for p in (train_params, test_params):
        p.setdefault('image_size', [size, size])
    train = dset_class(train_config.datadir, subset=train_subset, **train_params)
    test = dset_class(train_config.datadir, subset=test_subset, **test_params)
    return (train.get_dataset(batch_per_rank, repeat=True, shuffle=False, num_preprocess_threads=12, device=device,
                              rank=rank, world=world),
            test.get_dataset(batch_per_rank, repeat=False, shuffle=False, num_preprocess_threads=12, device=device))


def synthetic_iter(batch, size, device, seed):
    """Random batches, put on the device through pinned staging (ops.to_device_pinned — never a raw host-to-device `.to` of a pageable
    temporary: ADVICE r5); host tensors handed to TrainStep.step would be staged by IMMEngine.set_inputs (ops.PinnedStager)."""
    g = torch.Generator().manual_seed(seed)
    mask = ops.to_device_pinned(smooth_mask(size, size).reshape(1, size, size, 1).repeat(batch, 1, 1, 1), device)
    while True:
        yield {'image': ops.to_device_pinned(torch.rand(batch, size, size, 3, generator=g) * 255, device),
               'future_image': ops.to_device_pinned(torch.rand(batch, size, size, 3, generator=g) * 255, device), 'mask': mask}


def npz_iter(path, batch, device, rank, world):
    d = np.load(path)
    n = d['image'].shape[0]
    i = rank * batch
    while True:
        idx = [(i + j) % n for j in range(batch)]
        yield {k: ops.to_device_pinned(d[k][idx], device, torch.float32) for k in ('image', 'future_image', 'mask')}
        i += batch * world


def tps_pair_iter(source, size, device, cfg_dataset):
    """imm/datasets/tps_dataset.py: every batch of single images becomes (image, future_image, mask) by two random
    thin-plate-spline warps of mask||image — on the GPU (imm_amd/data/tps.py) instead of a CPU py_func.  The warp
    parameters of the `dataset` config block are honoured when present (rotsd, scalesd, transsd, warpsd, *_points)."""
    from imm_amd.data.tps import TPSPairAugmenter
    kw = {}
    for k in ('vertical_points', 'horizontal_points', 'rotsd', 'scalesd', 'transsd', 'warpsd'):
        if cfg_dataset is not None and k in cfg_dataset:
            kw[k] = cfg_dataset[k]
    aug = TPSPairAugmenter((size, size), device=device, **kw)
    base_mask = ops.to_device_pinned(smooth_mask(size, size).reshape(1, size, size, 1), device)
    for batch in source:
        img = batch['image']
        yield aug(img, base_mask.expand(img.shape[0], -1, -1, -1))


def main(args):
    config = load_configs(args.configs)
    train_config = config.training
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != max(args.ngpus, 1):
        raise SystemExit('--ngpus %d needs `torchrun --nproc-per-node %d` (one process per GPU)' % (args.ngpus, args.ngpus))
    if args.ngpus == 0:
        raise SystemExit('CPU training (train_single_cpu, cnn_train_multi.py:252-301) is not provided: the product path '
                         'has no CPU fallback; oracle/imm_oracle.py is the CPU restatement used for checking')
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.distributed.init_process_group('nccl', device_id=torch.device(dev))

    if train_config.optim.lower() not in ('adam', 'adadelta', 'adagrad'):        # scripts/train.py:97-104
        raise ValueError('Optimizer = %s not suppoerted' % train_config.optim)
    hparams = dict(lr_start=float(train_config.lr.start_val), lr_decay=float(train_config.lr.decay),
                   lr_step=int(train_config.lr.step), lr_multiple=float(args.lr_multiple), clip=float(train_config.gradclip),
                   optim=train_config.optim.lower())
    batch_size = int(train_config.batch)
    size = int(args.image_size)
    factory = model_factory(IMMModel, config=config.model, global_step=None, device=dev, hparams=hparams, world_size=world)
    opts = {'gpu_ids': list(range(args.ngpus)), 'batch_size': batch_size, 'image_size': size,
            'log_dir': train_config.logdir, 'n_checkpoint': int(train_config.ncheckpoint), 'n_summary': 10,
            'n_test': int(train_config.n_test) if hasattr(train_config, 'n_test') else 500}
    step = tru.setup_training(opts, factory, clip_value=train_config.gradclip)
    eng = step.engine
    if args.checkpoint is not None and osp.exists(args.checkpoint + '.index'):
        # a TensorFlow bundle (the authors' released checkpoints, or one written by --tf-checkpoints)
        from imm_amd.utils.tf_checkpoint import load_tf_checkpoint
        skipped = load_tf_checkpoint(eng, args.checkpoint, restore_optim=args.restore_optim,
                                     ignore_missing_vars=args.ignore_missing_vars)
        print('RESTORING MODEL from: %s (TensorFlow bundle%s)' % (args.checkpoint, '; %d variables not found' % len(skipped) if skipped else ''))
    elif args.checkpoint is not None and osp.exists(args.checkpoint):
        ck = torch.load(args.checkpoint, map_location='cpu')
        eng.load_parameters(ck['params'], ck.get('state'))
        if 'step' in ck:                      # global_step is a model variable upstream: restored in both modes
            eng.step_count.fill_(int(ck['step']))
        # Adam's slots AND its bias-correction counter (TF: beta1_power / beta2_power) travel together: without
        # --restore-optim both start afresh (m = v = 0, t = 0) whatever the restored global_step is
        if args.restore_optim and 'adam_m' in ck:
            eng.adam_m.copy_(ck['adam_m']); eng.adam_v.copy_(ck['adam_v'])
            eng.adam_t.fill_(int(ck.get('adam_t', ck.get('step', 0))))
            if eng.loss_scale_state is not None and ck.get('loss_scale_state') is not None:
                eng.loss_scale_state.copy_(ck['loss_scale_state'])      # dynamic loss scale S + its clean-step counter (f16 storage)
        else:
            eng.reset_optimizer_slots()
    elif args.checkpoint is not None:
        print('No checkpoint at %s. Initializing randomly.' % args.checkpoint)
    if args.reset_global_step >= 0:
        eng.step_count.fill_(args.reset_global_step)
    per_rank = batch_size // world
    test_data = None
    if args.data_npz:
        data = npz_iter(args.data_npz, per_rank, dev, rank, world)
    elif args.synthetic:
        data = synthetic_iter(per_rank, size, dev, rank)
    else:
        train_data, test_data = dataset_loaders(train_config, per_rank, size, dev, rank, world)
        data = iter(train_data)
    if args.tps:
        ds = getattr(train_config, 'dataset', None)
        data = tps_pair_iter(data, size, dev, ds if ds is not None and hasattr(ds, '__contains__') else None)

    def save(n):
        os.makedirs(train_config.logdir, exist_ok=True)
        path = osp.join(train_config.logdir, 'model.ckpt-%d.pt' % n)
        torch.save({'params': eng.named_parameters(), 'state': eng.named_state(), 'adam_m': eng.adam_m.cpu(),
                    'adam_v': eng.adam_v.cpu(), 'step': int(eng.step_count), 'adam_t': int(eng.adam_t),
                    'loss_scale_state': None if eng.loss_scale_state is None else eng.loss_scale_state.cpu(),
                    'vgg_weights': getattr(step.model, 'vgg_source', 'unknown')}, path)
        print('saved', path)
        if args.tf_checkpoints:
            from imm_amd.utils.tf_checkpoint import save_tf_checkpoint
            save_tf_checkpoint(eng, osp.join(train_config.logdir, 'model.ckpt-%d' % n))

    writer = tru.SummaryWriter(train_config.logdir) if rank == 0 else None
    tru.train_loop(opts, step, data, args.num_steps, log_every=10, checkpoint_fn=save, test_dataset=test_data,
                   model=step.model, summary_writer=writer)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Train Unsupervised Sequence Model')
    parser.add_argument('--configs', nargs='+', default=[], help='Paths to the config files.')
    parser.add_argument('--ngpus', type=int, default=1, required=False, help='Number of GPUs to use for training.')
    parser.add_argument('--lr-multiple', type=float, default=1, help='multiplier on the learning rate.')
    parser.add_argument('--checkpoint', type=str, default=None, help='checkpoint file-name of the *FULL* model to restore.')
    parser.add_argument('--restore-optim', action='store_true', help='Restore the optimizer variables.')
    parser.add_argument('--reset-global-step', type=int, default=-1, help='Force the value of global step.')
    parser.add_argument('--ignore-missing-vars', action='store_true', help='Skip re-storing vars not in the checkpoint file.')
    # additions of this build
    parser.add_argument('--num-steps', type=int, default=30000000)
    parser.add_argument('--image-size', type=int, default=128)
    parser.add_argument('--tf-checkpoints', action='store_true', help='also write TensorFlow bundles model.ckpt-N.{index,data-*} '
                        '(variable names of the reference graph, readable by tf.train.Saver)')
    parser.add_argument('--synthetic', action='store_true', help='random batches instead of the configured dataset')
    parser.add_argument('--data-npz', type=str, default=None, help='optional .npz with image/future_image/mask float32 NHWC arrays')
    parser.add_argument('--tps', action='store_true', help='build (image, future_image, mask) from each batch\'s `image` by two '
                        'random TPS warps on the GPU (imm/datasets/tps_dataset.py)')
    main(parser.parse_args())
