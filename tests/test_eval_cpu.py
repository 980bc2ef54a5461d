"""Host arithmetic of the evaluation path (imm_amd/eval/eval_imm.py) on hand-made data."""
import numpy as np
import pytest

from imm_amd.eval import eval_imm as E


def test_convert_landmarks():
    t = {'gauss_yx': np.array([[[-1.0, 1.0], [0.0, 0.5]]], np.float32), 'future_landmarks': np.array([[[3, 4], [5, 6], [7, 8]]])}
    x, y = E.convert_landmarks(t, [128, 128])
    np.testing.assert_allclose(x, [[0.0, 128.0, 64.0, 96.0]])
    np.testing.assert_allclose(y, [[3, 4, 5, 6, 7, 8]])


def test_interocular_error_known_answer():
    gt = np.array([[[0, 0], [0, 10], [5, 5]]], np.float32)          # eyes 10 apart
    pred = gt + np.array([[[3, 4], [0, 0], [0, 5]]], np.float32)    # distances 5, 0, 5
    assert abs(E.interocular_error(gt, pred) - (5 + 0 + 5) / 3 / 10) < 1e-6


def test_ridge_recovers_a_linear_map():
    rng = np.random.RandomState(0)
    K, L, n = 10, 5, 200
    lm = rng.rand(n, K, 2).astype(np.float32) * 2 - 1
    a = rng.randn(2 * K, 2 * L)
    gt = (((lm + 1) / 2 * 128).reshape(n, -1) @ a).reshape(n, L, 2)
    tr = {'gauss_yx': lm[:150], 'future_landmarks': gt[:150]}
    te = {'gauss_yx': lm[150:], 'future_landmarks': gt[150:]}
    pred = E.regress_landmarks(tr, te, [128, 128], bias=False)
    # relative to the scale of the targets (float32 inputs)
    assert np.max(np.abs(pred - gt[150:])) / np.max(np.abs(gt)) < 1e-4
    pred_b = E.regress_landmarks(tr, te, [128, 128], bias=True)
    assert np.max(np.abs(pred_b - gt[150:])) / np.max(np.abs(gt)) < 1e-4


def test_plot_landmarks_styles_and_drawing():
    from imm_amd.utils import plot_landmarks as PL
    assert PL.get_marker_style(0) == (PL.COLORS[0], 'v') and PL.get_marker_style(9) == (PL.COLORS[1], 'o')
    assert PL.get_marker_style(55) == (PL.COLORS[7], '+')
    with pytest.raises(ValueError):
        PL.get_marker_style(56)
    img = np.zeros((64, 64, 3), np.float32)
    lm = np.array([[10.0, 20.0], [40.0, 50.0]] + [[5.0 + k, 5.0] for k in range(48)])      # every marker shape appears
    out = np.asarray(PL.plot_landmarks(img, lm, size=2.5, scale=3))
    assert out.shape == (192, 192, 3)
    assert tuple(out[30, 60]) == PL.COLORS[0]                 # (y, x) = (10, 20) scaled by 3: centre of the first marker
    assert tuple(out[120, 150]) == PL.COLORS[1]
    assert (out[170:185, 60:140] == 0).all()                   # untouched background
    red = PL.plot_landmarks(img, lm[:1], style_fn=PL.single_marker_style((255, 0, 0), 'o'), scale=1)
    assert tuple(np.asarray(red)[10, 20]) == (255, 0, 0)
