"""Unsupervised landmarks of a folder of images, drawn onto them — the counterpart of examples/visualize.ipynb of the
reference (same steps: resize every image to 128x128 with PIL, feed it as both `image` and `future_image`, run the model
in inference mode, denormalise `gauss_yx` from [-1, 1] to pixels, plot with one style per landmark).
    python scripts/visualize.py --experiment-name aflw-10pts-finetune --images-dir my_faces --out landmarks.png
The checkpoint is <logdir>/model.ckpt (a TensorFlow bundle, e.g. the authors' release) or model.ckpt.pt, or --checkpoint."""
from __future__ import print_function

import argparse
import os
import os.path as osp
import sys

import numpy as np
import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from imm_amd.models.imm_model import IMMModel                 # noqa: E402
from imm_amd.utils.config import load_configs                 # noqa: E402
from imm_amd.utils.plot_landmarks import plot_landmarks      # noqa: E402


def load_images(images_dir, image_size):
    from PIL import Image
    files = sorted(f for f in os.listdir(images_dir) if f.lower().endswith(('.jpg', '.jpeg', '.png', '.bmp')))
    if not files:
        raise ValueError('no images in %s' % images_dir)
    out = []
    for f in files:
        with Image.open(osp.join(images_dir, f)) as im:
            out.append(np.array(im.convert('RGB').resize(image_size[::-1]), dtype=np.float32))
    return files, np.stack(out)


def restore(net, eng, checkpoint):
    if osp.isfile(checkpoint + '.index'):
        from imm_amd.utils.tf_checkpoint import load_tf_checkpoint
        skipped = load_tf_checkpoint(eng, checkpoint, ignore_missing_vars=True)
        if skipped:
            print('vars-IGNORED (not restoring)')
            print(', '.join(skipped))
    elif osp.isfile(checkpoint):
        ck = torch.load(checkpoint, map_location='cpu')
        eng.load_parameters(ck['params'], ck.get('state'))
    else:
        raise Exception('model file does not exist at: ' + checkpoint)


def main(args):
    image_size = [args.im_size, args.im_size]
    config = load_configs([args.paths_config, osp.join('configs', 'experiments', args.experiment_name + '.yaml')]
                          if args.configs is None else args.configs)
    checkpoint = args.checkpoint
    if checkpoint is None:
        checkpoint = osp.join(config.training.logdir, 'model.ckpt')
        if not osp.isfile(checkpoint + '.index') and osp.isfile(checkpoint + '.pt'):
            checkpoint += '.pt'
    files, images = load_images(args.images_dir, image_size)
    torch.cuda.set_device(0)
    net = IMMModel(config.model, device='cuda:0')
    restore(net, net._get_engine(len(images), image_size[0]), checkpoint)
    x = torch.from_numpy(images).to('cuda:0')
    _, _, _, tensors = net.build({'image': x, 'future_image': x}, training_pl=False, output_tensors=True, build_loss=False)
    landmarks = tensors['gauss_yx'].float().cpu().numpy()
    landmarks = ((landmarks + 1) / 2.0) * np.array(image_size)          # denormalize landmarks
    from PIL import Image
    cols = min(4, len(images))
    rows = int(np.ceil(len(images) / float(cols)))
    tiles = [plot_landmarks(images[i], landmarks[i], size=2.5, scale=args.scale) for i in range(len(images))]
    tw, th = tiles[0].size
    sheet = Image.new('RGB', (cols * tw, rows * th), (255, 255, 255))
    for i, t in enumerate(tiles):
        sheet.paste(t, ((i % cols) * tw, (i // cols) * th))
    sheet.save(args.out)
    if args.save_landmarks:
        np.save(args.save_landmarks, landmarks)
    for f, lm in zip(files, landmarks):
        print(f, ' '.join('(%.1f, %.1f)' % (p[0], p[1]) for p in lm))
    print('wrote', args.out)


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Plot unsupervised landmarks on images.')
    parser.add_argument('--experiment-name', type=str, default='aflw-10pts-finetune')
    parser.add_argument('--paths-config', type=str, default='configs/paths/default.yaml')
    parser.add_argument('--configs', nargs='+', default=None, help='explicit config files (instead of --experiment-name)')
    parser.add_argument('--images-dir', type=str, required=True)
    parser.add_argument('--checkpoint', type=str, default=None)
    parser.add_argument('--im-size', type=int, default=128)
    parser.add_argument('--scale', type=int, default=3, help='enlargement of the output tiles')
    parser.add_argument('--out', type=str, default='landmarks.png')
    parser.add_argument('--save-landmarks', type=str, default=None, help='optional .npy for the [N,K,2] (y, x) pixel coordinates')
    main(parser.parse_args())
