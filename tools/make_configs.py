"""Writes configs/ — the experiment files scripts/train.py and scripts/test.py take, with the reference's keys and the
paper's hyper-parameters (the values of /root/reference/configs/experiments/*.yaml; `${...}` references resolve against
configs/paths/default.yaml, imm_amd/utils/config.py).  Usage: python tools/make_configs.py"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PATHS = """\
# Where things live; every experiment file refers to these through ${...}.
logdir: data/logs                                   # training logs and checkpoints
celeba_data_dir: data/datasets/celeba               # Img/img_align_celeba_hq, Anno/, Eval/, MAFL/
aflw_data_dir: data/datasets/aflw_release-2         # aflw_{train,test}_images.txt, *_keypoints.mat, output/
vgg16_path: data/models/vgg16.caffemodel.h5         # perceptual-loss VGG16 (caffe blobs; .npz of the same blobs also accepted)
"""

TEMPLATE = """\
# {title}
name: {name}

training:
  dset: {dset}
  datadir: ${{{datadir}}}
  logdir: ${{logdir}}/${{name}}
  train_dset_params: {train_params}
  test_dset_params: {test_params}
  batch: 50
  optim: Adam
  lr: {{start_val: 0.001, step: 100000, decay: 0.95}}    # staircase exponential decay
  gradclip: 1.0                                         # per-tensor gradient-norm clip
  ncheckpoint: 2000                                     # steps between checkpoints
  n_test: 1000                                          # steps between passes over the test split
  allow_growth: True                                    # (TensorFlow allocator switch; ignored here)

model:
  n_maps: {n_maps}                 # number of landmarks
  gauss_std: 0.10
  gauss_mode: 'rot'
  n_filters: 32
  block_sizes: [1, 1, 1]
  n_filters_render: 32
  renderer_stride: 2
  min_res: 16
  same_n_filt: False
  reconstruction_loss: perceptual   # perceptual | l2
  perceptual:
    l2: True
    comp: ['input', 'conv1_2', 'conv2_2', 'conv3_2', 'conv4_2', 'conv5_2']
    net_file: ${{vgg16_path}}
  loss_mask: True
{extra}  channels_bug_fix: True
"""


def main():
    os.makedirs(os.path.join(ROOT, 'configs', 'paths'), exist_ok=True)
    os.makedirs(os.path.join(ROOT, 'configs', 'experiments'), exist_ok=True)
    with open(os.path.join(ROOT, 'configs', 'paths', 'default.yaml'), 'w') as f:
        f.write(PATHS)
    for k in (10, 30, 50):
        jobs = [dict(name='celeba-%dpts' % k, title='CelebA, %d unsupervised landmarks' % k, dset='celeba',
                     datadir='celeba_data_dir', train_params='{dataset: celeba, subset: train}',
                     test_params='{dataset: mafl, subset: test, order_stream: True, max_samples: 1000}',
                     extra='  confidence: False\n' if k == 10 else ''),
                dict(name='aflw-%dpts-finetune' % k, title='AFLW fine-tuning of the CelebA model, %d landmarks' % k, dset='aflw',
                     datadir='aflw_data_dir', train_params='{subset: train}',
                     test_params='{subset: test, order_stream: True, max_samples: 1000}', extra='')]
        for j in jobs:
            with open(os.path.join(ROOT, 'configs', 'experiments', j['name'] + '.yaml'), 'w') as f:
                f.write(TEMPLATE.format(n_maps=k, **j))


if __name__ == '__main__':
    main()
