"""Synthetic CelebA / AFLW directory trees for the dataset tests (tiny JPEG/PNG files written with PIL)."""
import os

import numpy as np


def _smooth_image(rng, h, w):
    """Low-frequency content so that JPEG round trips stay close; u8 HWC."""
    base = rng.rand(6, 6, 3)
    ys = np.linspace(0, 5, h)
    xs = np.linspace(0, 5, w)
    y0 = np.clip(np.floor(ys).astype(int), 0, 4)
    x0 = np.clip(np.floor(xs).astype(int), 0, 4)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    img = (base[y0][:, x0] * (1 - fy) * (1 - fx) + base[y0 + 1][:, x0] * fy * (1 - fx)
           + base[y0][:, x0 + 1] * (1 - fy) * fx + base[y0 + 1][:, x0 + 1] * fy * fx)
    return (img * 255).astype(np.uint8)


def make_celeba_tree(root, n=40, seed=0, fmt='png'):
    """n images 000001.jpg.. (file NAMES end in .jpg like the dataset; the content is PNG by default so that decoding is
    lossless and tests can compare against the exact pixels).  Partition: first 60% train (0), next 20% val (1), last
    20% test (2).  MAFL training = images 21..30, MAFL testing = images 31..36 (1-based).  Returns {name: u8 array}."""
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'Img', 'img_align_celeba_hq'))
    for d in ('Anno', 'Eval', 'MAFL'):
        os.makedirs(os.path.join(root, d))
    names = ['%06d.jpg' % (i + 1) for i in range(n)]
    pixels = {}
    lm_rows = []
    for i, nm in enumerate(names):
        h, w = (218, 178) if i % 3 else (200 + i, 160 + i)
        img = _smooth_image(rng, h, w)
        Image.fromarray(img).save(os.path.join(root, 'Img', 'img_align_celeba_hq', nm), format='PNG' if fmt == 'png' else 'JPEG',
                                  **({} if fmt == 'png' else {'quality': 95}))
        pixels[nm] = img
        lm_rows.append(nm + ' ' + ' '.join(str(int(v)) for v in rng.randint(10, 150, size=10)))
    with open(os.path.join(root, 'Anno', 'list_landmarks_align_celeba.txt'), 'w') as f:
        f.write('%d\nlefteye_x lefteye_y righteye_x righteye_y nose_x nose_y leftmouth_x leftmouth_y rightmouth_x rightmouth_y\n' % n)
        f.write('\n'.join(lm_rows) + '\n')
    with open(os.path.join(root, 'Eval', 'list_eval_partition.txt'), 'w') as f:
        for i, nm in enumerate(names):
            f.write('%s %d\n' % (nm, 0 if i < 0.6 * n else 1 if i < 0.8 * n else 2))
    with open(os.path.join(root, 'MAFL', 'training.txt'), 'w') as f:
        f.write('\n'.join(names[20:30]) + '\n')
    with open(os.path.join(root, 'MAFL', 'testing.txt'), 'w') as f:
        f.write('\n'.join(names[30:36]) + '\n')
    return names, pixels


def make_aflw_tree(root, n_train=20, n_test=6, seed=1):
    from PIL import Image
    from scipy.io import savemat
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(root, 'output'))
    pixels = {}
    for split, n in (('train', n_train), ('test', n_test)):
        names = ['%s_%03d.png' % (split, i) for i in range(n)]
        hw = np.zeros((n, 2), np.int32)
        for i, nm in enumerate(names):
            h, w = 90 + 3 * i, 100 + 2 * i
            img = _smooth_image(rng, h, w)
            Image.fromarray(img).save(os.path.join(root, 'output', nm))
            pixels[nm] = img
            hw[i] = (h, w)
        gt = rng.rand(n, 5, 2).astype(np.float64) * 90
        savemat(os.path.join(root, 'aflw_%s_keypoints.mat' % split), {'gt': gt, 'hw': hw})
        with open(os.path.join(root, 'aflw_%s_images.txt' % split), 'w') as f:
            f.write('\n'.join(names) + '\n')
    return pixels
