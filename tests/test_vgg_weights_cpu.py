"""Weight import of the frozen perceptual VGG16 (imm_amd/utils/vgg_weights.py) against the formulas of
imm/models/selfsup/vgg16.py:17-47,74-92 worked by hand / in float64."""
import numpy as np
import pytest

from imm_amd.utils import vgg_weights as V


def _blobs(rng, cin1=1):
    data = {}
    cin = cin1
    for name, cout in (('conv1_1', 64), ('conv1_2', 64), ('conv2_1', 128), ('conv2_2', 128), ('conv3_1', 256), ('conv3_2', 256),
                       ('conv3_3', 256), ('conv4_1', 512), ('conv4_2', 512), ('conv4_3', 512), ('conv5_1', 512), ('conv5_2', 512)):
        data[name] = {'0': rng.randn(cout, cin, 3, 3).astype(np.float32), '1': rng.randn(cout).astype(np.float32)}
        data['batch_' + name] = {'0': rng.randn(cout).astype(np.float32) * 7, '1': (rng.rand(cout) * 30 + 1).astype(np.float32),
                                 '2': np.array([9.5], np.float32)}
        cin = cout
    return data


def test_fold_known_answer():
    # one output channel: var sum 35, scale 7 -> sigma = sqrt(1e-5 + 5); mean sum 14 -> mu = 2
    w, b = V.fold_batch_norm(np.full((1, 1, 2, 1), 3.0, np.float32), np.array([4.0], np.float32), [14.0], [35.0], [7.0])
    s = np.sqrt(1e-5 + 5.0)
    np.testing.assert_allclose(w, 3.0 / s, rtol=1e-7)
    np.testing.assert_allclose(b, (4.0 - 2.0) / s, rtol=1e-7)


def test_from_caffe_blobs_layout_and_fold():
    rng = np.random.RandomState(0)
    data = _blobs(rng)
    out = V.from_caffe_blobs(data)
    w = out['vgg16/conv2_1/weights']
    assert w.shape == (3, 3, 64, 128) and out['vgg16/conv1_1/weights'].shape == (3, 3, 1, 64)
    bn = data['batch_conv2_1']
    sigma = np.sqrt(1e-5 + bn['1'].astype(np.float64) / 9.5)
    # element [ky, kx, i, o] comes from caffe [o, i, ky, kx], divided by sigma[o]
    np.testing.assert_allclose(w[1, 2, 5, 7], data['conv2_1']['0'][7, 5, 1, 2] / sigma[7], rtol=1e-6)
    np.testing.assert_allclose(out['vgg16/conv2_1/biases'], (data['conv2_1']['1'] - bn['0'] / 9.5) / sigma, rtol=1e-5)
    raw = V.from_caffe_blobs(data, fold_bn=False)
    np.testing.assert_array_equal(raw['vgg16/conv2_1/weights'][0, 0, 3, 4], data['conv2_1']['0'][4, 3, 0, 0])


def test_bgr_flip_only_for_three_channel_first_layer():
    rng = np.random.RandomState(1)
    data = {'conv1_1': {'0': rng.randn(4, 3, 3, 3).astype(np.float32), '1': np.zeros(4, np.float32)}}
    w = V.from_caffe_blobs(data)['vgg16/conv1_1/weights']
    np.testing.assert_array_equal(w[0, 0, 0], data['conv1_1']['0'][:, 2, 0, 0])     # R <- caffe B position


def test_load_npz_roundtrip_and_errors(tmp_path):
    rng = np.random.RandomState(2)
    data = _blobs(rng)
    flat = {'%s/%s' % (g, k): v for g, d in data.items() for k, v in d.items()}
    p = str(tmp_path / 'vgg_caffe.npz')
    np.savez(p, **flat)
    w = V.load_vgg16(p)
    ref = V.from_caffe_blobs(data)
    assert set(w) == set(ref)
    np.testing.assert_array_equal(w['vgg16/conv4_2/weights'].numpy(), ref['vgg16/conv4_2/weights'])
    # already-converted keys pass through
    p2 = str(tmp_path / 'vgg_final.npz')
    np.savez(p2, **ref)
    w2 = V.load_vgg16(p2)
    np.testing.assert_array_equal(w2['vgg16/conv1_1/biases'].numpy(), ref['vgg16/conv1_1/biases'])
    # the colour network is refused (the IMM loss feeds a grayscale image)
    p3 = str(tmp_path / 'vgg_rgb.npz')
    flat3 = dict(flat); flat3['conv1_1/0'] = rng.randn(64, 3, 3, 3).astype(np.float32)
    np.savez(p3, **flat3)
    with pytest.raises(ValueError):
        V.load_vgg16(p3)
    with pytest.raises(ValueError):
        V.load_vgg16(str(tmp_path / 'weights.bin'))
    with pytest.raises(FileNotFoundError):          # .h5 goes through the pure-Python reader (tests/test_hdf5_lite_cpu.py)
        V.load_vgg16(str(tmp_path / 'vgg16.caffemodel.h5'))
