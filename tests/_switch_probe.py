"""Child process of tests/test_switches_gpu.py: two training steps of a batch-2 model under whatever IMM_* environment the
parent set; prints one JSON line {loss0, loss1, mu_abs_sum, params_abs_sum, kernels_seen}."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import imm_oracle as O                      # noqa: E402  (tests may use the oracle's synthetic inputs)
from imm_amd.models.imm_model import IMMModel            # noqa: E402
from imm_amd.train.cnn_train_multi import TrainStep      # noqa: E402
from imm_amd.utils.box import Box                        # noqa: E402

B = int(os.environ.get('PROBE_BATCH', '2'))
cfg = O.default_model_config(10)
inputs = O.synthetic_inputs(B, 128, seed=0)
model = IMMModel(Box(dict(cfg)), dtype=torch.bfloat16, device='cuda:0')
ts = TrainStep(model, B, 128, world_size=1, use_graph=os.environ.get('PROBE_GRAPH', '1') != '0')
l0 = float(ts.step(inputs).clone())
ts.synchronize()
l1 = float(ts.step(None).clone())
ts.synchronize()
eng = ts.engine
out = {'loss0': l0, 'loss1': l1, 'mu_abs_sum': float(eng.mu.double().abs().sum()), 'params_abs_sum': float(eng.params.double().abs().sum()),
       'step_count': int(eng.step_count), 'n_launches': sum(1 for p in (eng.prog_fwd, eng.prog_bwd, eng.prog_opt) for l in p if l.fn is not None)}
if eng._stamp_mode:
    rep = eng.stamp_report()
    out['stamps'] = len(rep)
    out['stamps_monotone_lane0'] = all(b[0] >= a[0] for a, b in zip([r for r in rep if r[1] == 0], [r for r in rep if r[1] == 0][1:]))
print('PROBE ' + json.dumps(out))
