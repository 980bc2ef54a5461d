"""How well conditioned is the step's gradient, and how close is the engine to the fp32 oracle where it is?  Prints, per
trainable tensor, relative L2 error and cosine of the engine's gradient (bf16 and f16 storage) against the fp32 oracle
 (a) at the seeded initialisation and (b) after N training steps on a smooth batch (weights off the 0.01-std start, BN
statistics away from their degenerate initial state).  The bounds of tests/test_step_gpu.py::
test_gradient_parity_on_a_trained_model come from this table.      python tools/diag/diag_grad_conditioning.py [steps] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import imm_oracle as O   # noqa: E402  (test infrastructure: lives under tests/)


def smooth_batch(B, S, seed=3):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, 6, 6, generator=g)
    img = torch.nn.functional.interpolate(low, size=(S, S), mode='bilinear', align_corners=True).permute(0, 2, 3, 1) * 255
    fut = torch.roll(img, shifts=(5, -7), dims=(1, 2))
    return {'image': img.contiguous(), 'future_image': fut.contiguous(), 'mask': O.smooth_mask(S, S).reshape(1, S, S, 1).repeat(B, 1, 1, 1).contiguous()}


def compare(tag, eng, g_ref):
    worst = 0.0
    for k, v in g_ref.items():
        if k.endswith('/b') and (k[:-2] + '/gamma') in g_ref:
            continue
        if float(v.norm()) < 1e-7:
            continue
        a, b = (eng.gview[k].detach() / eng.loss_scale).cpu().double().flatten(), v.detach().double().flatten()
        rel = float((a - b).norm() / b.norm())
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
        worst = max(worst, rel)
        print('COND %-14s %-46s rel %.4f  cos %.5f  |g| %.4g' % (tag, k, rel, cos, float(b.norm())))
    print('COND %-14s WORST rel %.4f' % (tag, worst))


def main():
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    S = 128
    cfg = O.default_model_config(10)
    inputs = smooth_batch(B, S)
    P0, St0 = O.init_params(cfg, S)

    def engine_grads(dt, P, St_named):
        model = IMMModel(Box(dict(cfg)), dtype=dt, device='cuda:0')
        eng = model._get_engine(B, S)
        if P is not None:
            eng.load_parameters(P, St_named)
        eng.set_inputs(inputs['image'].cuda(), inputs['future_image'].cuda(), inputs['mask'].cuda())
        eng.forward(True); eng.backward()
        torch.cuda.synchronize()
        return eng

    # (a) initialisation
    _o, g0 = O.loss_and_grads(P0, St0, inputs, cfg)
    for dt in (torch.bfloat16, torch.float16):
        compare('init/' + str(dt).split('.')[-1], engine_grads(dt, None, None), g0)

    # (b) after training
    model = IMMModel(Box(dict(cfg)), dtype=torch.bfloat16, device='cuda:0')
    ts = TrainStep(model, B, S, world_size=1, use_graph=True)
    for i in range(steps):
        ts.step(inputs if i == 0 else None)
    ts.synchronize()
    eng = ts.engine
    P1 = type(P0)((k, v.cpu()) for k, v in eng.named_parameters().items())
    named_state = {k: v.cpu() for k, v in eng.named_state().items()}
    St1 = type(St0)(St0)
    St1.update(named_state)
    out1, g1 = O.loss_and_grads(P1, St1, inputs, cfg)
    print('COND trained: oracle loss %.4f after %d steps' % (float(out1['loss']), steps))
    for dt in (torch.bfloat16, torch.float16):
        compare('trained/' + str(dt).split('.')[-1], engine_grads(dt, P1, named_state), g1)
    # the oracle against itself in float64: the conditioning floor
    P64, S64, in64 = O.to_dtype(P1, torch.float64), O.to_dtype(St1, torch.float64), O.to_dtype(inputs, torch.float64)
    _o64, g64 = O.loss_and_grads(P64, S64, in64, cfg)
    worst = 0.0
    for k, v in g1.items():
        if k.endswith('/b') and (k[:-2] + '/gamma') in g1 or float(v.norm()) < 1e-7:
            continue
        worst = max(worst, float((v.double() - g64[k]).norm() / g64[k].norm()))
    print('COND trained: fp32 oracle vs fp64 oracle WORST rel %.5f' % worst)


if __name__ == '__main__':
    main()
