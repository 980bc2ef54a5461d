"""Oracle vs the committed golden vectors (tests/golden/imm_step_golden.npz, made by make_golden.py).
Thread count / BLAS blocking changes the summation order, so comparisons carry fp32 tolerances."""
import os

import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'imm_step_golden.npz'), allow_pickle=False)


@pytest.mark.parametrize('K,B', [(10, 2), (30, 1)])
def test_oracle_matches_golden(K, B):
    cfg = O.default_model_config(K)
    P, S = O.init_params(cfg, 128, seed=1, vgg_seed=2)
    inp = O.synthetic_inputs(B, 128, seed=0)
    opt = O.new_adam_state(P)
    newP, newS, info = O.train_step(P, S, opt, [inp], cfg, clip=1.0, lr=O.learning_rate(0))
    o = info['outs'][0]
    t = 'k%d_b%d' % (K, B)
    np.testing.assert_allclose(o['gauss_yx'].detach().numpy(), G[t + '/gauss_yx'], atol=2e-5)
    np.testing.assert_allclose(float(o['loss']), float(G[t + '/loss']), rtol=1e-5)
    np.testing.assert_allclose(float(o['weights_loss']), float(G[t + '/weights_loss']), rtol=1e-6)
    np.testing.assert_allclose([float(x) for x in o['loss_terms']], G[t + '/loss_terms'], rtol=1e-3)
    np.testing.assert_allclose(o['future_im_pred'].detach().numpy()[:, ::16, ::16, :], G[t + '/pred_sample'], atol=2e-3)
    names = list(P.keys())
    gn = np.array([float(info['grads'][k].double().norm()) for k in names])
    big = G[t + '/grad_norms'] > 1e-3       # bias-before-BN gradients are cancellation noise
    np.testing.assert_allclose(gn[big], G[t + '/grad_norms'][big], rtol=5e-2)   # conditioning: see DESIGN.md
    np.testing.assert_allclose([float(newS['loss/%s_agg' % n]) for n in cfg.perceptual.comp], G[t + '/agg_after_step'], rtol=1e-4)
    if K == 10:
        assert list(G['param_names']) == names
