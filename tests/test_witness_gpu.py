"""The f32-storage WITNESS engine (round 6): IMMModel(dtype=torch.float32) runs the SAME launch program
(imm_amd/engine.py: the same buffers, strides, fork / join structure, packed filter layouts, batch-norm partial rows, split-K
slabs, flat gradient layout) with f32 activation storage and plain f32 FMA convolutions (imm_amd/csrc/conv_f32.hip) — the
reference computes in fp32 (/root/reference/imm/models/imm_model.py:97).  What the 16-bit engines can only bound by their storage
emulation (encoder gradients at initialisation 0.38-0.42 relative) this run settles in exact arithmetic: landmarks,
loss terms, reconstruction and EVERY initial-state gradient tensor against the fp32 oracle.  A test instrument, not a product path
(~100x slower than the bf16 kernels); TF 1.10 itself cannot run here, so the oracle stays the restatement (parity unpinned)."""
import json

import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def witness(K=10, B=2, S=128, **cfg_over):
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(K)
    cfg.update(cfg_over)
    model = IMMModel(Box(dict(cfg)), dtype=torch.float32, device=DEV)
    inputs = O.synthetic_inputs(B, S, seed=0)
    return cfg, model, inputs


# measured on MI355X (B=2, K=10, 128x128, initial weights), f32 witness vs the fp32 oracle: see the bounds' comments
BOUNDS = dict(mu=1e-5, loss=1e-5, terms=1e-4, recon=1e-4, heat=1e-4, agg=1e-5)


@pytest.mark.timeout(900)
def test_f32_witness_forward_and_every_gradient_match_the_fp32_oracle():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    cfg, model, inputs = witness()
    P, St = O.init_params(cfg, 128)
    out, g = O.loss_and_grads(P, St, inputs, cfg)
    _, loss, _, t = model.build(inputs, True, output_tensors=True)
    eng = model.engine
    assert eng.f32 and eng.dt == torch.float32 and eng.loss_scale_state is None and not eng.fused_head
    # every convolution launch of the program is the plain f32 kernel
    variants = set(l.variant for p in (eng.prog_fwd, eng.prog_bwd) for l in p if l.tag in ('conv_fwd', 'conv_dgrad', 'vgg_fwd', 'vgg_dgrad') and l.variant)
    assert variants and all(v.startswith('f32:') or v in ('group', 's2d') for v in variants), variants
    eng.backward()
    torch.cuda.synchronize()
    got = dict(mu=float((t['gauss_yx'].cpu() - out['gauss_yx'].detach()).abs().max()),
               loss=abs(float(loss) - float(out['loss'])) / abs(float(out['loss'])),
               terms=max(abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(eng.loss_terms.cpu(), out['loss_terms'])),
               recon=rel(t['future_im_pred'], out['future_im_pred']),
               heat=rel(eng.heat[..., :10], out['heatmaps']),
               agg=max(abs(float(eng.loss_agg[i]) - float(out['new_state']['loss/%s_agg' % n])) / abs(float(out['new_state']['loss/%s_agg' % n]))
                       for i, n in enumerate(cfg.perceptual.comp)))
    gv = eng.named_gradients()
    worst, rows = 0.0, {}
    for k, v in g.items():
        if k.endswith('/b') and (k[:-2] + '/gamma') in g:
            assert float(gv[k].abs().max()) == 0.0        # analytically zero (the batch norm removes the mean): the oracle holds noise
            continue
        if float(v.norm()) < 1e-7:
            continue
        a, b = gv[k].detach().cpu().double().flatten(), v.detach().double().flatten()
        e = float((a - b).norm() / b.norm())
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        rows[k] = (e, cos)
        worst = max(worst, e)
    got['grad_worst'] = worst
    got['grad_median'] = float(np.median([e for e, _c in rows.values()]))
    got['cos_min'] = min(c for _e, c in rows.values())
    print('\nWITNESS_F32 ' + json.dumps(got))
    print('WITNESS_F32_GRADS ' + json.dumps({k: [round(e, 6), round(c, 8)] for k, (e, c) in rows.items()}))
    for k, lim in BOUNDS.items():
        assert got[k] <= lim, (k, got[k], lim, got)
    # THE point of the witness: every initial-state gradient tensor within 1e-2 of the fp32 oracle (whose own distance to an fp64
    # run of itself is 2e-3 .. 7e-3 there) — the 16-bit engines sit at 0.004 .. 0.42 for the same tensors
    assert worst <= 1e-2 and got['cos_min'] >= 0.9999, (worst, got['cos_min'], {k: v for k, v in rows.items() if v[0] > 5e-3})


@pytest.mark.timeout(900)
def test_f32_witness_training_steps_follow_the_oracle():
    """Three whole steps (forward, backward, per-tensor clip, Adam, re-pack) of the witness against oracle.train_step: with exact
    activations the trajectories stay together — parameters after 3 updates, the loss of every step and the landmark of the last
    forward pass — where the bf16 engine's first update direction is only cos >= 0.55 per tensor (tests/test_step_gpu.py)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    cfg, model, inputs = witness()
    ts = TrainStep(model, 2, 128, world_size=1, use_graph=True)
    P, St = O.init_params(cfg, 128)
    opt = O.new_adam_state(P)
    losses = []
    P0 = {k: v.clone() for k, v in P.items()}
    for it in range(3):
        loss = ts.step(inputs)
        ts.synchronize()
        P, St, info = O.train_step(P, St, opt, [inputs], cfg, clip=1.0, lr=O.learning_rate(it))
        losses.append((float(loss), float(info['outs'][0]['loss'])))
    eng = model.engine
    assert int(eng.step_count) == 3
    coss = {}
    for k, v in eng.named_parameters().items():
        # tensors whose gradient is ANALYTICALLY zero hold cancellation noise in the oracle (a bias in front of a batch norm; the pose
        # head's bias: a constant added to a heat-map leaves its softmax unchanged): Adam turns that noise into +-lr random walks
        #  — nothing to compare there
        if (k.endswith('/b') and (k[:-2] + '/gamma') in P) or k == 'model/pose_encoder/conv_1/b':
            continue
        ue, uo = (v.cpu().double() - P0[k].double()).flatten(), (P[k].double() - P0[k].double()).flatten()
        coss[k] = float((ue * uo).sum() / (ue.norm() * uo.norm() + 1e-300))
    worst = sorted(coss.items(), key=lambda kv: kv[1])[:5]
    med = float(np.median(list(coss.values())))
    print('\nWITNESS_F32_STEPS ' + json.dumps(dict(losses=losses, cos_min=worst[0][1], cos_median=med, worst=worst)))
    for a_, b_ in losses:
        assert abs(a_ - b_) <= 1e-4 * abs(b_), losses
    # exact activations keep the two Adam trajectories together: the accumulated update of EVERY tensor after three steps points
    # the oracle's way — measured min 0.944 / median 0.984 over the 72 tensors (Adam normalises element by element, so the elements
    # whose gradient is within the oracle's own f32 noise, 2e-3 .. 7e-3 of the tensor, move by +-lr at random: they cap the
    # cosine); the bf16 engine's FIRST update direction reaches cos >= 0.55 / median 0.75 (tests/test_step_gpu.py)
    assert worst[0][1] >= 0.90 and med >= 0.97, (worst, med)


@pytest.mark.timeout(900)
def test_f32_witness_variants_k30_and_l2():
    """The witness on the other configurations' wiring: K=30 (wider bottleneck, another concat padding) and the image-space 'l2'
    reconstruction loss (no VGG in the program)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    for kw in (dict(K=30, B=1), dict(K=10, B=2, reconstruction_loss='l2')):
        K, B = kw.pop('K'), kw.pop('B')
        cfg, model, inputs = witness(K=K, B=B, **kw)
        P, St = O.init_params(cfg, 128)
        out, g = O.loss_and_grads(P, St, inputs, cfg)
        _, loss, _, t = model.build(inputs, True, output_tensors=True)
        eng = model.engine
        eng.backward()
        torch.cuda.synchronize()
        assert float((t['gauss_yx'].cpu() - out['gauss_yx'].detach()).abs().max()) <= 1e-5
        assert abs(float(loss) - float(out['loss'])) <= 1e-5 * abs(float(out['loss']))
        gv = eng.named_gradients()
        worst = 0.0
        for k, v in g.items():
            if (k.endswith('/b') and (k[:-2] + '/gamma') in g) or float(v.norm()) < 1e-7:
                continue
            worst = max(worst, rel(gv[k], v))
        print('\nWITNESS_F32_VARIANT %r worst gradient %.3e' % (kw or {'K': K}, worst))
        assert worst <= 1e-2, (kw, worst)
