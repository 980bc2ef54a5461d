"""imm_amd/utils/hdf5_lite.py against a file written by the REAL HDF5 library (tests/golden/caffe_vgg_tiny.h5, libhdf5
1.10.6 via tests/golden/make_h5_golden.c: the Caffe-snapshot layout of the reference's vgg16.caffemodel.h5 at toy sizes,
values = sin(0.37 e + 1.3 b + 0.11 l)), and the perceptual-network import on top of it."""
import os

import numpy as np
import pytest

from imm_amd.utils import hdf5_lite as H
from imm_amd.utils import vgg_weights as V

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden', 'caffe_vgg_tiny.h5')
LAYERS = ['conv1_1', 'conv1_2', 'conv2_1', 'conv2_2', 'conv3_1', 'conv3_2', 'conv3_3', 'conv4_1', 'conv4_2', 'conv4_3',
          'conv5_1', 'conv5_2', 'conv5_3']


def val(l, b, n):
    return np.sin(0.37 * np.arange(n) + 1.3 * b + 0.11 * l).astype(np.float32)


def test_group_tree_and_contiguous_datasets():
    tree = H.load(GOLDEN)
    assert list(tree) == ['data', 'extra']
    data = tree['data']
    assert sorted(data) == sorted(LAYERS + ['batch_' + n for n in LAYERS])          # 26 groups: several symbol-table nodes
    cin = 1
    for l, name in enumerate(LAYERS):
        cout = 4 + l % 3
        w, b = data[name]['0'], data[name]['1']
        assert w.shape == (cout, cin, 3, 3) and w.dtype == np.float32 and b.shape == (cout,)
        np.testing.assert_array_equal(w.reshape(-1), val(l, 0, w.size))
        np.testing.assert_array_equal(b, val(l, 1, cout))
        bn = data['batch_' + name]
        assert list(bn) == ['0', '1', '2']
        np.testing.assert_array_equal(bn['0'], val(l, 2, cout))
        np.testing.assert_array_equal(bn['1'], np.float32(2.0) + val(l, 3, cout))
        np.testing.assert_array_equal(bn['2'], [3.0 + l])
        cin = cout
    # a sub-path loads just that group
    sub = H.load(GOLDEN, '/data/conv3_2')
    np.testing.assert_array_equal(sub['1'], data['conv3_2']['1'])
    with pytest.raises(KeyError):
        H.load(GOLDEN, '/data/conv9_9')


def test_optional_layouts_filters_and_types():
    ex = H.load(GOLDEN, '/extra')
    x = ex['chunked_deflate']                       # 8x8 chunks of a 37x21 array, shuffle + deflate
    assert x.shape == (37, 21) and x.dtype == np.float32
    np.testing.assert_array_equal(x.reshape(-1), val(20, 0, 37 * 21))
    y = ex['chunked_f64']                           # 2x4x3 chunks of 5x6x7, unfiltered
    assert y.dtype == np.float64
    np.testing.assert_array_equal(y.reshape(-1), val(21, 0, 210).astype(np.float64))
    np.testing.assert_array_equal(ex['ints'], np.arange(9) ** 2 - 7)
    assert ex['ints'].dtype == np.int32
    assert ex['scalar'].shape == () and float(ex['scalar']) == 2.5
    np.testing.assert_array_equal(ex['compact'], val(22, 0, 6))
    np.testing.assert_array_equal(ex['big_endian'], val(22, 0, 6))
    assert ex['big_endian'].dtype == np.float32     # converted to native order


def test_rejects_what_it_does_not_parse(tmp_path):
    p = tmp_path / 'x.h5'
    p.write_bytes(b'not an hdf5 file' * 10)
    with pytest.raises(ValueError, match='not an HDF5 file'):
        H.load(str(p))
    raw = bytearray(open(GOLDEN, 'rb').read())
    raw[8] = 2                                      # pretend superblock version 2
    p.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError, match='superblock version 2'):
        H.load(str(p))


def test_vgg_import_from_the_h5_snapshot():
    """load_vgg16 on the .h5: same result as the known-answer path from_caffe_blobs on the parsed blobs."""
    w = V.load_vgg16(GOLDEN)
    data = H.load(GOLDEN, '/data')
    ref = V.from_caffe_blobs({g: {k: np.asarray(v) for k, v in blobs.items()} for g, blobs in data.items()})
    assert sorted(w) == sorted(ref)
    for k in ref:
        np.testing.assert_array_equal(w[k].numpy(), ref[k].astype(np.float32))
    # spot-check the folding by hand for one output channel of conv1_1 (HWIO, BN folded)
    l, o = 0, 2
    wc = val(0, 0, 4 * 1 * 3 * 3).reshape(4, 1, 3, 3)
    sigma = np.sqrt(1e-5 + (np.float32(2.0) + val(0, 3, 4)) / 3.0)
    mu = val(0, 2, 4) / 3.0
    np.testing.assert_allclose(w['vgg16/conv1_1/weights'].numpy()[:, :, 0, o], wc[o, 0] / sigma[o], rtol=1e-6)
    np.testing.assert_allclose(w['vgg16/conv1_1/biases'].numpy()[o], (val(0, 1, 4)[o] - mu[o]) / sigma[o], rtol=1e-5)
    assert w['vgg16/conv1_1/weights'].shape == (3, 3, 1, 4)
