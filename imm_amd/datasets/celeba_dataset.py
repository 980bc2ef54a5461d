"""CelebA / MAFL (imm/datasets/celeba_dataset.py): index + split logic on the host, pixels on the GPU.

Directory layout expected under data_dir (celeba_dataset.py:14-17,30,38,48): Img/img_align_celeba_hq/*.jpg,
Anno/list_landmarks_align_celeba.txt, Eval/list_eval_partition.txt, MAFL/training.txt, MAFL/testing.txt."""
import os

import numpy as np

from .tps_dataset import TPSDataset


def load_dataset(data_root, dataset, subset):
    """celeba_dataset.py:14-93.  Every aligned-CelebA image gets a set label: for dataset='celeba' the official
    partition + 1 (train 1 / val 2 / test 3), for 'mafl' 1 on MAFL's training list; then MAFL's test images -> 4 and
    the LAST 10% of MAFL's training images (in landmark-file order) -> 5, in both modes — so CelebA training never sees
    MAFL test/validation faces.  Returns (image_dir, file names [N], keypoints [N,5,2] as (x, y))."""
    image_dir = os.path.join(data_root, 'Img', 'img_align_celeba_hq')
    with open(os.path.join(data_root, 'Anno', 'list_landmarks_align_celeba.txt'), 'r') as f:
        rows = f.read().splitlines()[2:]                       # count line + column header
    names = [r.split()[0] for r in rows]
    keypoints = np.array([[int(v) for v in r.split()[1:]] for r in rows], dtype=np.float32)
    assert names[0] == '000001.jpg'

    def listed(fname):
        with open(os.path.join(data_root, 'MAFL', fname), 'r') as f:
            members = set(f.read().splitlines())
        return [i for i, n in enumerate(names) if n in members]

    mafl_train = listed('training.txt')
    label = np.zeros(len(names), dtype=np.int32)
    if dataset == 'celeba':
        with open(os.path.join(data_root, 'Eval', 'list_eval_partition.txt'), 'r') as f:
            label[:] = [int(line.split()[1]) for line in f.readlines()]
        label += 1
    elif dataset == 'mafl':
        label[mafl_train] = 1
    else:
        raise ValueError('Dataset = %s not recognized.' % dataset)
    label[listed('testing.txt')] = 4
    n_validation = int(round(0.1 * len(mafl_train)))
    label[mafl_train[-n_validation:]] = 5                      # NB: n_validation == 0 selects all (slice [-0:]), as upstream

    wanted = {'celeba': {'train': 1, 'val': 2}, 'mafl': {'train': 1, 'test': 4, 'train10': 5}}[dataset]
    if subset not in wanted:
        raise ValueError('subset = %s for %s dataset not recognized.' % (subset, dataset))
    keep = label == wanted[subset]
    return image_dir, np.array(names)[keep], np.reshape(keypoints[keep], [-1, 5, 2])


class CelebADataset(TPSDataset):
    """Same constructor as the reference (celeba_dataset.py:101-118): `dataset` is 'celeba' or 'mafl', the remaining
    arguments are TPSDataset's."""
    LANDMARK_LABELS = {'left_eye': 0, 'right_eye': 1}
    N_LANDMARKS = 5
    CROP_PERCENT = 0.8            # the central 80 % of the resized image is kept (celeba_dataset.py:148)

    def __init__(self, data_dir, subset, dataset=None, max_samples=None, image_size=[128, 128], order_stream=False,
                 landmarks=False, tps=True, vertical_points=10, horizontal_points=10, rotsd=[0.0, 5.0], scalesd=[0.0, 0.1],
                 transsd=[0.1, 0.1], warpsd=[0.001, 0.005, 0.001, 0.01], name='CelebADataset'):
        if dataset is None:
            raise AssertionError('CelebADataset needs dataset="celeba" or "mafl"')
        TPSDataset.__init__(self, data_dir, subset, max_samples, image_size, order_stream, landmarks, tps, vertical_points,
                            horizontal_points, rotsd, scalesd, transsd, warpsd, name)
        self._dataset = dataset
        self._image_dir, self._images, self._keypoints = load_dataset(data_dir, dataset, subset)
        side = int(self._image_size[0])
        resize = int(np.round(side / self.CROP_PERCENT))
        self._resize_margin = (resize, int(np.round((resize - side) / 2.0)))

    def num_samples(self):
        return len(self._images)

    def _geometry(self):
        """celeba_dataset.py:148-152: resize to round(size / 0.8), keep the central size^2 window."""
        return self._resize_margin

    def _proc_landmarks(self, sample, original_hw):
        """celeba_dataset.py:154-158: (y, x) landmarks follow the resize and the crop offset."""
        resize, margin = self._resize_margin
        return self._resize_points(np.asarray(sample['landmarks'], np.float32), original_hw, [resize, resize]) - np.float32(margin)
