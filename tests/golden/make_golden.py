"""Generates tests/golden/imm_step_b2_k10.npz from the CPU oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference itself cannot run (no TensorFlow), so the fixture pins
the ORACLE's numbers: any later edit of oracle/imm_oracle.py that changes them is caught by
tests/test_golden.py, and the GPU tests compare the HIP path against the same vectors without needing
the oracle's code path to be the only witness.  Inputs are regenerated from seeds, only outputs are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import imm_oracle as O   # noqa: E402


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    out = {}
    for K, B in ((10, 2), (30, 1)):
        cfg = O.default_model_config(K)
        P, S = O.init_params(cfg, 128, seed=1, vgg_seed=2)
        inp = O.synthetic_inputs(B, 128, seed=0)
        opt = O.new_adam_state(P)
        newP, newS, info = O.train_step(P, S, opt, [inp], cfg, clip=1.0, lr=O.learning_rate(0))
        o = info['outs'][0]
        tag = 'k%d_b%d' % (K, B)
        out[tag + '/gauss_yx'] = o['gauss_yx'].detach().numpy()
        out[tag + '/loss'] = np.float64(float(o['loss']))
        out[tag + '/reconstruction_loss'] = np.float64(float(o['reconstruction_loss']))
        out[tag + '/weights_loss'] = np.float64(float(o['weights_loss']))
        out[tag + '/loss_terms'] = np.array([float(t) for t in o['loss_terms']])
        out[tag + '/loss_means'] = np.array([float(t) for t in o['loss_means']])
        out[tag + '/pred_sample'] = o['future_im_pred'].detach().numpy()[:, ::16, ::16, :]
        names = list(P.keys())
        out[tag + '/grad_norms'] = np.array([float(info['grads'][k].double().norm()) for k in names])
        out[tag + '/param_checksum_after_step'] = np.array([float(newP[k].double().sum()) for k in names])
        out[tag + '/agg_after_step'] = np.array([float(newS['loss/%s_agg' % n]) for n in cfg.perceptual.comp])
        out[tag + '/bn_mm_checksum'] = np.float64(sum(float(v.double().sum()) for k, v in newS.items() if k.endswith('moving_mean')))
        if K == 10:
            out['param_names'] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'imm_step_golden.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
