"""Finite-difference directional check of the engine's backward pass against its OWN forward pass (MI355X):

    [L(theta + eps d) - L(theta - eps d)] / (2 eps)   vs   <g_engine, d>

for directions d = the oracle's gradient restricted to one tensor group and normalised.  Prints a table over step sizes
and storage types; tests/test_step_gpu.py::test_backward_is_the_derivative_of_the_forward holds the engine to it with the
step sizes this table shows to be in the linear, noise-free range.   python tools/diag/diag_fd_check.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import imm_oracle as O   # noqa: E402  (test infrastructure: lives under tests/, the oracle only supplies the directions)

GROUPS = {
    'image_encoder': lambda k: k.startswith('model/image_encoder/'),
    'pose_encoder': lambda k: k.startswith('model/pose_encoder/'),
    'renderer': lambda k: k.startswith('model/renderer/'),
    'img_enc conv_1/w': lambda k: k == 'model/image_encoder/encoder/conv_1/w',
    'img_enc conv_5/w': lambda k: k == 'model/image_encoder/encoder/conv_5/w',
    'pose_enc conv_8/gamma': lambda k: k == 'model/pose_encoder/encoder/conv_8/gamma',
    'pose head w': lambda k: k == 'model/pose_encoder/conv_1/w',
    'renderer conv_1/w': lambda k: k == 'model/renderer/conv_1/w',
    'renderer conv_4/beta': lambda k: k == 'model/renderer/conv_4/beta',
    'renderer conv_8/w': lambda k: k == 'model/renderer/conv_8/w',
}


def directional(eng, direction_flat, eps, agg0):
    """central difference of the engine's training-mode loss along direction_flat (unit norm)."""
    p0 = eng.params.clone()
    vals = []
    for sgn in (+1.0, -1.0):
        eng.params.copy_(p0 + sgn * eps * direction_flat)
        eng.run(eng.prog_pack)
        eng.loss_agg.copy_(agg0)
        eng.forward(True)
        torch.cuda.synchronize()
        vals.append(float(eng.loss.double()))
    eng.params.copy_(p0)
    eng.run(eng.prog_pack)
    eng.loss_agg.copy_(agg0)
    return (vals[0] - vals[1]) / (2.0 * eps)


def main():
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = O.default_model_config(10)
    inputs = O.synthetic_inputs(batch, 128, seed=0)
    P, St = O.init_params(cfg, 128)
    _out, g_ref = O.loss_and_grads(P, St, inputs, cfg)
    for dt in (torch.float16, torch.bfloat16):
        model = IMMModel(Box(dict(cfg)), dtype=dt, device='cuda:0')
        eng = model._get_engine(batch, 128)
        eng.set_inputs(inputs['image'].cuda(), inputs['future_image'].cuda(), inputs['mask'].cuda())
        agg0 = eng.loss_agg.clone()
        eng.forward(True); eng.backward()
        torch.cuda.synchronize()
        eng.loss_agg.copy_(agg0)
        g_eng = eng.grads.clone() / eng.loss_scale
        names = [n for n, _s, _w in eng.spec]
        for gname, pred in GROUPS.items():
            d = torch.zeros_like(g_eng)
            for i, k in enumerate(names):
                if pred(k) and not (k.endswith('/b') and (k[:-2] + '/gamma') in g_ref):
                    d[eng.tab.offsets[i]:eng.tab.offsets[i + 1]] = g_ref[k].reshape(-1).to(d.device)
            d /= d.norm()
            analytic = float((g_eng.double() * d.double()).sum())
            # weight-decay part of the total loss (not in eng.grads: it is added inside imm_clip_adam_step)
            wd = 0.0
            for i, (k, _s, w) in enumerate(eng.spec):
                if w:
                    sl = slice(eng.tab.offsets[i], eng.tab.offsets[i + 1])
                    wd += w * float((eng.params[sl].double() * d[sl].double()).sum())
            pnorm = float((eng.params * (d != 0)).norm())
            row = []
            pnorm = max(pnorm, 0.01 * float((d != 0).sum()) ** 0.5)      # beta = 0 at initialisation
            for rel in (3e-5, 1e-4, 3e-4, 1e-3, 3e-3):
                eps = rel * pnorm
                fd = directional(eng, d, eps, agg0) - wd
                row.append('%8.3g' % (fd / analytic))
            print('FD %-8s %-24s <g,d> %10.4g  |theta_grp| %8.3g  FD/analytic at eps/|theta| 3e-5,1e-4,3e-4,1e-3,3e-3: %s'
                  % (str(dt).split('.')[-1], gname, analytic, pnorm, ' '.join(row)))
        del model, eng


if __name__ == '__main__':
    main()
