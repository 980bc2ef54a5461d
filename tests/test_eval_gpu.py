"""Evaluation path on the GPU: eval_imm.evaluate over a small synthetic 'dataset' with a ragged last
batch, landmarks against the oracle in BN-eval mode, then the Ridge / inter-ocular pipeline end to end."""
import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_evaluate_matches_oracle_and_regression_runs():
    from imm_amd.eval import eval_imm as E
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = Box(dict(O.default_model_config()))
    S = 128
    model = IMMModel(cfg, device=DEV, seed=1)
    batches = []
    for i, b in enumerate((3, 3, 2)):                 # ragged tail: a second engine (batch 2) shares the variables
        inp = O.synthetic_inputs(b, S, seed=10 + i)
        rng = np.random.RandomState(i)
        inp = dict(inp)
        inp['future_landmarks'] = (rng.rand(b, 5, 2) * S).astype(np.float32)
        batches.append(inp)
    res = E.evaluate(iter(batches), model, batch_size=3, eval_tensors=['gauss_yx', 'future_landmarks'])
    assert [a.shape for a in res['gauss_yx']] == [(3, 10, 2), (3, 10, 2), (2, 10, 2)]
    assert all(np.array_equal(a, b['future_landmarks']) for a, b in zip(res['future_landmarks'], batches))
    # oracle, eval mode (moving statistics), same seeded parameters
    P, St = O.init_params(O.default_model_config(), S, seed=1)
    for got, inp in zip(res['gauss_yx'], batches):
        out = O.forward(P, St, {k: v for k, v in inp.items() if k != 'future_landmarks'}, O.default_model_config(),
                        training=False, build_loss=False)
        assert float(np.abs(got - out['gauss_yx'].numpy()).max()) < 1e-3        # north_star tolerance on mu in [-1, 1]
    allt = {k: np.concatenate(v) for k, v in res.items()}
    pred = E.regress_landmarks(allt, allt, [S, S], bias=True)
    err = E.interocular_error(allt['future_landmarks'], pred)
    assert np.isfinite(err) and err >= 0.0
