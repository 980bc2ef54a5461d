"""Host side of the input pipelines (imm_amd/datasets): index/split logic, sample streams, landmark
geometry, batching/shuffle/repeat/rank dealing and decode — everything before the GPU — plus the image oracle's
cross-check.  The expectations for the split logic are derived by hand from the rules of
imm/datasets/celeba_dataset.py:14-93 and aflw_dataset.py:15-38 on the synthetic trees of tests/dataset_fixtures.py."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dataset_fixtures import make_aflw_tree, make_celeba_tree     # noqa: E402
from imm_amd.datasets import AFLWDataset, CelebADataset           # noqa: E402
from imm_amd.datasets import celeba_dataset, aflw_dataset         # noqa: E402
from imm_amd.datasets.impair_dataset import ImagePairDataset, PairBatchLoader   # noqa: E402
from imm_amd.datasets.tps_dataset import smooth_mask              # noqa: E402
from oracle import image_oracle as IO                             # noqa: E402


@pytest.fixture(scope='module')
def celeba(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('celeba'))
    names, pixels = make_celeba_tree(root, n=40)
    return root, names, pixels


@pytest.fixture(scope='module')
def aflw(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('aflw'))
    return root, make_aflw_tree(root)


def test_celeba_splits(celeba):
    root, names, _ = celeba
    # partition: 1..24 train, 25..32 val, 33..40 test; MAFL train = 21..30 (last 10% = image 30 -> label 5), MAFL test = 31..36
    _, tr, kp = celeba_dataset.load_dataset(root, 'celeba', 'train')
    assert list(tr) == names[:24]                  # none of MAFL's test images are in 1..24; image 30 is not either
    assert kp.shape == (24, 5, 2) and kp.dtype == np.float32
    _, va, _ = celeba_dataset.load_dataset(root, 'celeba', 'val')
    assert list(va) == names[24:29]                # 25..32 minus image 30 (MAFL validation) and 31, 32 (MAFL test)
    _, mt, _ = celeba_dataset.load_dataset(root, 'mafl', 'train')
    assert list(mt) == names[20:29]
    _, m10, _ = celeba_dataset.load_dataset(root, 'mafl', 'train10')
    assert list(m10) == names[29:30]
    _, mte, kpt = celeba_dataset.load_dataset(root, 'mafl', 'test')
    assert list(mte) == names[30:36]
    # keypoints come back as (x, y) rows of the annotation file
    with open(os.path.join(root, 'Anno', 'list_landmarks_align_celeba.txt')) as f:
        row31 = [int(v) for v in f.read().splitlines()[2 + 30].split()[1:]]
    np.testing.assert_array_equal(kpt[0], np.array(row31, np.float32).reshape(5, 2))
    with pytest.raises(ValueError):
        celeba_dataset.load_dataset(root, 'celeba', 'test')
    with pytest.raises(ValueError):
        celeba_dataset.load_dataset(root, 'lsun', 'train')


def test_aflw_splits(aflw):
    root, _ = aflw
    d, tr, kp, hw = aflw_dataset.load_dataset(root, 'train')
    assert d.endswith('output') and len(tr) == 18 and tr[0] == 'train_000.png' and kp.shape == (18, 5, 2) and hw.shape == (18, 2)
    _, va, kpv, _ = aflw_dataset.load_dataset(root, 'val')
    assert list(va) == ['train_018.png', 'train_019.png']
    _, te, _, _ = aflw_dataset.load_dataset(root, 'test')
    assert len(te) == 6
    from scipy.io import loadmat
    gt = loadmat(os.path.join(root, 'aflw_train_keypoints.mat'))['gt']
    np.testing.assert_array_equal(kpv[0], gt[18][:, [1, 0]])


def test_sample_streams(celeba):
    root, names, _ = celeba
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, max_samples=4, tps=False)
    got = list(ds.sample_image_pair())
    assert [os.path.basename(s['image']) for s in got] == names[30:34]
    assert got[0]['left_eye'] == 0 and got[0]['right_eye'] == 1
    # landmarks are handed on as (y, x)
    np.testing.assert_array_equal(got[0]['landmarks'], ds._keypoints[0][:, [1, 0]])
    # ordered stream without max_samples ends with the index
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, tps=False)
    assert len(list(ds.sample_image_pair())) == 6 == ds.num_samples()
    # max_samples beyond the index: the ordered stream just ends
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, max_samples=100, tps=False)
    assert len(list(ds.sample_image_pair())) == 6
    # random stream: np.random draws, reproducible under a seed, bounded by max_samples
    ds = CelebADataset(root, 'train', dataset='celeba', max_samples=7, tps=False)
    np.random.seed(3)
    a = [s['image'] for s in ds.sample_image_pair()]
    np.random.seed(3)
    idx = [np.random.randint(24) for _ in range(7)]
    assert a == [os.path.join(root, 'Img', 'img_align_celeba_hq', names[i]) for i in idx]
    with pytest.raises(ValueError):
        CelebADataset(root, 'train', dataset='celeba', landmarks=True, tps=True)


def test_landmark_geometry(celeba, aflw):
    root, _, pixels = celeba
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, tps=False)
    assert ds._geometry() == (160, 16) == IO.celeba_geometry(128)
    s = next(iter(ds.sample_image_pair()))
    hw = pixels[os.path.basename(s['image'])].shape[:2]
    lm = ds._proc_landmarks(s, np.array(hw, np.int32))
    want = IO.resize_points(s['landmarks'], hw, [160, 160]) - 16
    np.testing.assert_array_equal(lm, want.astype(np.float32))
    assert lm.dtype == np.float32
    assert CelebADataset(root, 'test', dataset='mafl', image_size=[64, 64], tps=False)._geometry() == (80, 8)
    aroot, _ = aflw
    ads = AFLWDataset(aroot, 'test', order_stream=True, tps=False)
    assert ads._geometry() == (128, 0)
    s = next(iter(ads.sample_image_pair()))
    lm = ads._proc_landmarks(s, np.array([1, 1], np.int32))          # the decoded size is NOT what AFLW scales by
    np.testing.assert_array_equal(lm, IO.resize_points(s['landmarks'], s['size'], [128, 128]))
    # 'gt' columns are swapped twice on the way (load_dataset, _get_image): the sample carries gt's own order
    from scipy.io import loadmat
    gt = loadmat(os.path.join(aroot, 'aflw_test_keypoints.mat'))['gt']
    np.testing.assert_allclose(s['landmarks'], gt[0].astype(np.float32))


def test_smooth_mask():
    m = smooth_mask(128, 128, 10, 20)
    assert m.shape == (128, 128) and m.dtype == np.float32
    np.testing.assert_array_equal(m, IO.smooth_mask(128, 128))
    assert (m[:10] == 0).all() and (m[:, -10:] == 0).all() and (m[30:98, 30:98] == 1).all()
    strip = m[64]
    assert np.all(np.diff(strip[10:30]) > 0) and np.all(np.diff(strip[98:118]) < 0)
    np.testing.assert_allclose(strip[10], 0.5 + 0.5 * np.tanh(-1 / 0.4), rtol=1e-6)
    np.testing.assert_array_equal(strip, strip[::-1])


def test_loader_host_batches(celeba):
    root, names, pixels = celeba
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, tps=False)
    ld = PairBatchLoader(ds, 4, num_preprocess_threads=3)
    batches = list(ld.host_batches())
    assert [len(b[0]) for b in batches] == [4, 2]                    # the ragged remainder is kept
    for samples, decoded in batches:
        for s, d in zip(samples, decoded):
            np.testing.assert_array_equal(d, pixels[os.path.basename(s['image'])])      # PNG content: lossless decode
    # two ranks deal whole batches round-robin and together cover the stream once
    r0 = list(PairBatchLoader(ds, 2, rank=0, world=2).host_batches())
    r1 = list(PairBatchLoader(ds, 2, rank=1, world=2).host_batches())
    seen = [os.path.basename(s['image']) for b in (r0[0], r1[0], r0[1]) for s in b[0]]
    assert seen == names[30:36] and len(r0) == 2 and len(r1) == 1
    # repeat: the finite stream restarts
    it = PairBatchLoader(ds, 4, repeat=True).host_batches()
    first = [next(it) for _ in range(5)]
    assert [len(b[0]) for b in first] == [4, 4, 4, 4, 4]
    assert [os.path.basename(s['image']) for s in first[1][0]] == names[34:36] + names[30:32]
    # shuffle: a permutation of the stream, reproducible under the loader's rng
    sh = PairBatchLoader(ds, 6, shuffle=True, shuffle_buffer=3, rng=random.Random(5))
    order = [os.path.basename(s['image']) for s in next(sh.host_batches())[0]]
    assert sorted(order) == names[30:36] and order != names[30:36]
    sh2 = PairBatchLoader(ds, 6, shuffle=True, shuffle_buffer=3, rng=random.Random(5))
    assert order == [os.path.basename(s['image']) for s in next(sh2.host_batches())[0]]


def test_decode_jpeg_and_arrays(tmp_path):
    from PIL import Image
    from imm_amd.datasets.impair_dataset import decode_image
    rng = np.random.RandomState(0)
    from dataset_fixtures import _smooth_image
    img = _smooth_image(rng, 60, 50)
    p = str(tmp_path / 'a.jpg')
    Image.fromarray(img).save(p, format='JPEG', quality=95)
    d = decode_image(p)
    assert d.shape == (60, 50, 3) and d.dtype == np.uint8 and np.abs(d.astype(int) - img.astype(int)).mean() < 3
    g = str(tmp_path / 'g.png')
    Image.fromarray(img[..., 0]).save(g)
    assert decode_image(g).shape == (60, 50, 3)                     # grey files are expanded to 3 channels (channels=3)
    np.testing.assert_array_equal(decode_image(img), img)
    with pytest.raises(TypeError):
        decode_image(img.astype(np.float32))


def test_box_helpers():
    ds = ImagePairDataset('', 'train')
    np.testing.assert_array_equal(ds._find_common_box([5, 6, 20, 30], [2, 9, 25, 28]), [2, 6, 25, 30])
    # 20 x 40 box to a square target: height grows to 40 around the centre
    np.testing.assert_array_equal(ds._fit_bbox([10, 10, 30, 50], [128, 128]), [0, 10, 40, 50])
    np.testing.assert_array_equal(ds._fit_bbox([10, 10, 50, 30], [128, 128]), [10, 0, 50, 40])
    img = np.arange(5 * 6 * 1).reshape(5, 6, 1)
    c = ds._crop_to_box(img, [-2, -1, 3, 4])
    assert c.shape == (5, 5, 1) and (c[:2] == 0).all() and (c[:, 0] == 0).all() and c[2, 1, 0] == 0 and c[4, 4, 0] == img[2, 3, 0]
    c = ds._crop_to_box(img, [3, 4, 7, 8])
    assert c.shape == (4, 4, 1) and c[0, 0, 0] == img[3, 4, 0] and (c[2:] == 0).all() and (c[:, 2:] == 0).all()
    pts = np.array([[10, 20], [30, 40]], np.int32)
    np.testing.assert_array_equal(ds._resize_points(pts, [100, 200], [50, 50]), [[5, 5], [15, 10]])

    class R(object):
        def __init__(self, vals): self.vals = list(vals)
        def random(self): return self.vals.pop(0)
    im0, im1 = np.arange(12).reshape(2, 3, 2), -np.arange(12).reshape(2, 3, 2)
    p0, p1 = np.array([[0., 0.]], np.float32), np.array([[1., 2.]], np.float32)
    a, b, q0, q1 = ds._jitter_im_and_points(im0, im1, p0, p1, rng=R([0.9, 0.1]))     # flip, no swap
    np.testing.assert_array_equal(a, im0[:, ::-1]); np.testing.assert_array_equal(q1, [[1., 0.]]); np.testing.assert_array_equal(q0, [[0., 2.]])
    a, b, q0, q1 = ds._jitter_im_and_points(im0, im1, p0, p1, rng=R([0.1, 0.9]))     # no flip, swap
    np.testing.assert_array_equal(a, im1); np.testing.assert_array_equal(q0, p1)


def test_image_oracle_against_torch():
    """The restated TF1 bilinear/align_corners resize against torch's independent implementation of the same mapping."""
    rng = np.random.RandomState(0)
    for (h, w, oh, ow) in [(218, 178, 160, 160), (37, 91, 128, 128), (300, 200, 80, 80), (5, 5, 5, 5), (9, 7, 1, 1)]:
        img = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
        got = IO.resize_bilinear(img, oh, ow)
        t = torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None]
        if oh > 1:
            want = torch.nn.functional.interpolate(t, size=(oh, ow), mode='bilinear', align_corners=True)[0].permute(1, 2, 0).numpy()
            np.testing.assert_allclose(got, want, atol=2e-3)
        else:
            np.testing.assert_array_equal(got[0, 0], img[0, 0].astype(np.float32))
    np.testing.assert_array_equal(IO.resize_bilinear(img, 9, 7), img.astype(np.float32))      # identity size: exact
    c = IO.celeba_image(rng.randint(0, 256, size=(218, 178, 3)).astype(np.uint8), 128)
    assert c.shape == (128, 128, 3)


def test_iterating_without_gpu_fails_loudly(celeba):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    root, _, _ = celeba
    ds = CelebADataset(root, 'test', dataset='mafl', order_stream=True, tps=False)
    with pytest.raises(Exception):
        next(iter(ds.get_dataset(2, device='cpu', prefetch=False)))
