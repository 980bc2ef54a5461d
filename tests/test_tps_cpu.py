"""TPS augmentation, CPU side: the numpy restatement (oracle/tps_oracle.py) against vectors produced by the
reference's own code (tests/golden/make_tps_golden.py), and the host-side sampler logic."""
import os

import numpy as np
import pytest

from oracle import tps_oracle as T

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tps_golden.npz'))


def test_basis_matches_reference():
    b = T.tps_basis(12, 20, 3, 4)
    assert b.dtype == np.float32 and b.shape == (240, 15)
    np.testing.assert_array_equal(b, G['basis_12x20_3x4'])        # float64 math, one rounding: bit exact
    np.testing.assert_array_equal(T.tps_basis(128, 128, 10, 10)[::997], G['basis_128_rows'])


def test_sample_w_draw_order():
    rng = np.random.RandomState(5)
    np.testing.assert_allclose(T.sample_tps_w(4, 3, (0.01, 0.02), 10.0, 0.2, 0.3, rng), G['w_seed5'], rtol=0, atol=0)


def test_grid_and_warp_match_reference():
    basis = T.tps_basis(128, 128, 10, 10)
    grid = T.tps_grid(basis, G['w_target'], 128, 128)
    np.testing.assert_allclose(grid[:, ::9, ::7], G['grid_target_sub'], rtol=0, atol=2e-6)   # matmul summation order
    rng = np.random.RandomState(int(G['img_seed'][0]))
    img = (rng.rand(3, 128, 128, 4) * 255).astype(np.float32)
    img[..., 0] = rng.rand(3, 128, 128)
    fut = T.warp(img, G['w_target'])
    np.testing.assert_allclose(fut[:, ::5, ::3], G['future_sub'], rtol=0, atol=2e-3)         # values up to 255
    src = T.warp(fut, G['w_source'])
    np.testing.assert_allclose(src[:, ::5, ::3], G['source_sub'], rtol=0, atol=4e-3)
    np.testing.assert_allclose(fut[:1, ::5, ::3], G['sampler_forward_py_sub'], rtol=0, atol=2e-3)
    out = T.apply_pair(img[..., 1:], img[..., :1], G['w_target'], G['w_source'])
    np.testing.assert_array_equal(out['future_image'], fut[..., 1:])
    np.testing.assert_array_equal(out['mask'], fut[..., :1])
    np.testing.assert_array_equal(out['image'], src[..., 1:])


def test_padded_warp_matches_reference():
    """TPSRandomSampler(pad=True).forward_py of the reference on a non-square input: swapped paddings, smaller output."""
    out = T.warp_pad(G['pad_img'], G['pad_w'].astype(np.float32), 4, 5)
    assert out.shape == G['pad_out'].shape == (2, 16, 6, 3)
    np.testing.assert_allclose(out, G['pad_out'], rtol=0, atol=2e-3)


def test_grid_sample_known_answers():
    img = np.arange(2 * 3 * 4, dtype=np.float32).reshape(1, 3, 4, 2)
    # identity grid reproduces the image (align_corners=True puts -1/+1 on the corner pixel centres)
    gx, gy = np.meshgrid(np.linspace(-1, 1, 4), np.linspace(-1, 1, 3))
    ident = np.stack([gx, gy], axis=-1)[None].astype(np.float32)
    np.testing.assert_allclose(T.grid_sample(img, ident), img, atol=1e-5)
    # half a pixel to the right: mean of horizontal neighbours, the last column fades towards the zero padding
    shifted = ident.copy(); shifted[..., 0] += 1.0 / 3.0
    out = T.grid_sample(img, shifted)
    np.testing.assert_allclose(out[0, :, 0], 0.5 * (img[0, :, 0] + img[0, :, 1]), atol=1e-4)
    np.testing.assert_allclose(out[0, :, 3], 0.5 * img[0, :, 3], atol=1e-4)
    # far outside: zeros
    assert float(np.abs(T.grid_sample(img, ident + 5.0)).max()) == 0.0
