"""Shape sweep of the whole step on the GPU: odd batches, K = 10/30/50, both 16-bit dtypes, 256-pixel inputs, eager and
HIP-graph execution.  Every combination must train (finite, decreasing loss over 4 steps on a fixed batch), and the first
loss must agree with the oracle where the oracle is cheap enough to run."""
import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

CASES = [(1, 128, 10, torch.bfloat16, True), (3, 128, 10, torch.bfloat16, False), (7, 128, 30, torch.float16, True),
         (2, 256, 10, torch.bfloat16, True), (13, 128, 50, torch.bfloat16, True), (48, 128, 10, torch.bfloat16, True)]


@pytest.mark.parametrize('B,S,K,dt,graph', CASES, ids=['b1', 'b3_eager', 'b7_k30_f16', 'b2_256px', 'b13_k50', 'b48'])
def test_step_runs_and_learns(B, S, K, dt, graph):
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(K)
    model = IMMModel(Box(dict(cfg)), dtype=dt, device=DEV, seed=1)
    ts = TrainStep(model, B, S, world_size=1, use_graph=graph)
    inp = O.synthetic_inputs(B, S, seed=3)
    dev_inp = {k: v.to(DEV) for k, v in inp.items()}
    losses = []
    for _ in range(4):
        loss = ts.step(dev_inp)
        ts.synchronize()
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    if B <= 3 and S == 128:
        P, St = O.init_params(cfg, S, seed=1)
        ref = float(O.forward(P, St, inp, cfg, training=True)['loss'])
        assert abs(losses[0] - ref) / abs(ref) < 2e-2, (losses[0], ref)
    del ts, model
    torch.cuda.empty_cache()
