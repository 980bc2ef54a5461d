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
    LANDMARK_LABELS = {'left_eye': 0, 'right_eye': 1}
    N_LANDMARKS = 5

    def __init__(self, data_dir, subset, dataset=None, max_samples=None, image_size=[128, 128], order_stream=False,
                 landmarks=False, tps=True, vertical_points=10, horizontal_points=10, rotsd=[0.0, 5.0], scalesd=[0.0, 0.1],
                 transsd=[0.1, 0.1], warpsd=[0.001, 0.005, 0.001, 0.01], name='CelebADataset'):
        super(CelebADataset, self).__init__(
            data_dir, subset, max_samples=max_samples, image_size=image_size, order_stream=order_stream,
            landmarks=landmarks, tps=tps, vertical_points=vertical_points, horizontal_points=horizontal_points,
            rotsd=rotsd, scalesd=scalesd, transsd=transsd, warpsd=warpsd, name=name)
        assert dataset is not None
        self._dataset = dataset
        self._image_dir, self._images, self._keypoints = load_dataset(self._data_dir, self._dataset, self._subset)

    def num_samples(self):
        return len(self._images)

    def _get_sample_dtype(self):
        d = {'image': 'string', 'landmarks': 'float32'}
        d.update({k: 'int32' for k in self.LANDMARK_LABELS.keys()})
        return d

    def _get_sample_shape(self):
        d = {'image': None, 'landmarks': [self.N_LANDMARKS, 2]}
        d.update({k: [] for k in self.LANDMARK_LABELS.keys()})
        return d

    def _geometry(self):
        """celeba_dataset.py:148-152: resize to round(size / 0.8), keep the central size^2 window."""
        crop_percent = 0.8
        final_sz = self._image_size[0]
        resize_sz = np.round(final_sz / crop_percent).astype(np.int32)
        margin = np.round((resize_sz - final_sz) / 2.0).astype(np.int32)
        return int(resize_sz), int(margin)

    def _proc_landmarks(self, sample, original_hw):
        """celeba_dataset.py:154-158: (y, x) landmarks follow the resize and the crop offset."""
        resize_sz, margin = self._geometry()
        lm = self._resize_points(np.asarray(sample['landmarks'], np.float32), original_hw, [resize_sz, resize_sz])
        return lm - np.float32(margin)
