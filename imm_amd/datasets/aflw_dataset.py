"""AFLW (imm/datasets/aflw_dataset.py): index on the host, pixels on the GPU.

Layout under data_dir (aflw_dataset.py:15-37): aflw_{train,test}_images.txt, aflw_{train,test}_keypoints.mat with
'gt' [N,5,2] and 'hw' [N,2], images under output/."""
import os

import numpy as np

from .tps_dataset import TPSDataset


def load_dataset(data_dir, subset):
    """aflw_dataset.py:15-38: 'train'/'val' share the training list, the last 10% of it is the validation split;
    keypoints are returned with the two coordinates of 'gt' swapped."""
    from scipy.io import loadmat
    load_subset = 'train' if subset in ['train', 'val'] else 'test'
    with open(os.path.join(data_dir, 'aflw_' + load_subset + '_images.txt'), 'r') as f:
        images = f.read().splitlines()
    mat = loadmat(os.path.join(data_dir, 'aflw_' + load_subset + '_keypoints.mat'))
    keypoints = mat['gt'][:, :, [1, 0]]
    sizes = mat['hw']
    if subset in ['train', 'val']:
        n_validation = int(round(0.1 * len(images)))
        sel = slice(None, -n_validation) if subset == 'train' else slice(-n_validation, None)
        images, keypoints, sizes = images[sel], keypoints[sel], sizes[sel]
    return os.path.join(data_dir, 'output'), images, keypoints, sizes


class AFLWDataset(TPSDataset):
    """Same constructor as the reference (aflw_dataset.py:47-62); every sample also carries the annotated image `size`."""
    LANDMARK_LABELS = {'left_eye': 0, 'right_eye': 1}
    N_LANDMARKS = 5
    EXTRA_FIELDS = {'size': ('int32', 2)}

    def __init__(self, data_dir, subset, max_samples=None, image_size=[128, 128], order_stream=False, landmarks=False,
                 tps=True, vertical_points=10, horizontal_points=10, rotsd=[0.0, 5.0], scalesd=[0.0, 0.1],
                 transsd=[0.1, 0.1], warpsd=[0.001, 0.005, 0.001, 0.01], name='CelebADataset'):
        TPSDataset.__init__(self, data_dir, subset, max_samples, image_size, order_stream, landmarks, tps, vertical_points,
                            horizontal_points, rotsd, scalesd, transsd, warpsd, name)
        self._image_dir, self._images, self._keypoints, self._sizes = load_dataset(data_dir, subset)

    def num_samples(self):
        return len(self._images)

    def _get_image(self, idx):
        """aflw_dataset.py:116-123: like TPSDataset._get_image plus the 'hw' row of the annotation file."""
        sample = TPSDataset._get_image(self, idx)
        sample['landmarks'] = np.asarray(sample['landmarks'], dtype=np.float32)
        sample['size'] = np.asarray(self._sizes[idx], dtype=np.int32)
        return sample

    def _proc_landmarks(self, sample, original_hw):
        """aflw_dataset.py:95-98: rescaled by the ANNOTATED size ('hw' of the .mat), not the decoded one."""
        side = self._image_size[0]
        return self._resize_points(np.asarray(sample['landmarks'], np.float32), sample['size'], [side, side])
