"""End-to-end parity of the MI355X training step against the CPU oracle (same seeds, same inputs).

Two oracles are used:
  * `emul`  — the oracle fed the same 16-bit-rounded weights and rounding its activations (and, through
              autograd, their gradients) to bf16 where the engine stores bf16: isolates kernel/wiring
              errors from storage precision;
  * `fp32`  — the plain fp32 restatement of the TF1 graph: the parity target of BASELINE.json
              ("landmarks within 1e-3 of the TF1 reference").
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)


def make(batch, K=10, S=128, seed_in=0, dp_buckets=None):
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(K)
    model = IMMModel(Box(dict(cfg)), dtype=torch.bfloat16, device=DEV, dp_buckets=dp_buckets)
    inputs = O.synthetic_inputs(batch, S, seed=seed_in)
    eng = model._get_engine(batch, S)
    P, St = O.init_params(cfg, S)
    # the engine draws its own parameters with the same seeded generator: they must be identical
    for k, v in eng.named_parameters().items():
        assert torch.equal(v.cpu(), P[k]), k
    for k in St:
        if k.startswith('vgg16/'):
            assert torch.equal(eng.vgg_w[k].cpu(), St[k]), k
    return cfg, model, eng, inputs, P, St


def emul_params(P, St):
    Pe = type(P)((k, bf(v) if k.endswith('/w') else v) for k, v in P.items())
    Se = type(St)((k, bf(v) if (k.startswith('vgg16/') and k.endswith('/weights') and 'conv1_1' not in k) else v)
                  for k, v in St.items())
    return Pe, Se


def rel(a, b):
    a, b = a.detach().float().cpu().double(), b.detach().float().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope='module')
def fwd2():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    cfg, model, eng, inputs, P, St = make(2)
    _, loss, avg_ops, tensors = model.build(inputs, True, output_tensors=True)
    eng.backward()
    torch.cuda.synchronize()
    Pe, Se = emul_params(P, St)
    out_e, g_e = O.loss_and_grads(Pe, Se, inputs, cfg, act_round=bf)
    out_f, g_f = O.loss_and_grads(P, St, inputs, cfg)
    return dict(cfg=cfg, model=model, eng=eng, inputs=inputs, P=P, St=St, loss=loss, tensors=tensors,
                out_e=out_e, g_e=g_e, out_f=out_f, g_f=g_f, avg_ops=avg_ops)


def test_forward_parity(fwd2):
    eng, t = fwd2['eng'], fwd2['tensors']
    rows = {}
    for tag, out in (('emul', fwd2['out_e']), ('fp32', fwd2['out_f'])):
        rows[tag] = dict(
            mu_maxabs=float((t['gauss_yx'].cpu() - out['gauss_yx'].detach()).abs().max()),
            pred_rel=rel(t['future_im_pred'], out['future_im_pred']),
            heat_rel=rel(t['heatmaps'], out['heatmaps']),
            loss_rel=abs(float(fwd2['loss']) - float(out['loss'])) / abs(float(out['loss'])),
            terms_rel=max(abs(float(a) - float(b)) / abs(float(b)) for a, b in zip(eng.loss_terms.cpu(), out['loss_terms'])))
    print('\nFORWARD_PARITY ' + json.dumps(rows))
    # landmarks: BASELINE.json tolerance 1e-3 on mu in [-1,1] against the fp32 restatement
    assert rows['fp32']['mu_maxabs'] < 1e-3, rows
    assert rows['emul']['mu_maxabs'] < 5e-4, rows
    # reconstruction: bf16 storage noise (~0.4 % relative L2 per layer, drifting smoothly through the
    # 16 conv+BN blocks in front of `pred`: see test_layerwise_forward_diagnostics) => 8 % measured
    assert rows['emul']['pred_rel'] < 0.12 and rows['fp32']['pred_rel'] < 0.12, rows
    assert rows['emul']['heat_rel'] < 0.05 and rows['fp32']['heat_rel'] < 0.05, rows
    # loss and its six terms (f32 reductions of bf16 features)
    assert rows['emul']['loss_rel'] < 1e-3 and rows['fp32']['loss_rel'] < 1e-3, rows
    assert rows['emul']['terms_rel'] < 1e-2 and rows['fp32']['terms_rel'] < 1e-2, rows
    assert fwd2['avg_ops'] == [] and t['pose_embedding'].shape == (2, 128, 128, 3)
    assert t['gauss_y_prob'].shape == (2, 16, 10) and t['heatmaps'].shape == (2, 16, 16, 10)


def test_layerwise_forward_diagnostics(fwd2):
    """Prints the relative error of every stored activation against both oracles (no hard limits
    beyond a coarse sanity bound): a wiring bug shows up as a jump at one layer, storage noise as a
    slow drift."""
    eng = fwd2['eng']
    B = eng.B
    table = []
    for tag, out in (('emul', fwd2['out_e']), ('fp32', fwd2['out_f'])):
        acts = out['acts']
        for lays in (eng.enc_im, eng.enc_pose, eng.ren):
            for lay in lays:
                ref = acts[lay.scope]
                if lay.bn and lay.out is None:
                    # the block's normalised output is never stored (normalise on load: its consumers rebuild it in LDS; an
                    # up-sampled renderer block: only the up-sampled tensor has readers); materialise it here with the
                    # stand-alone apply pass from the same scale / shift
                    from imm_amd import ops as _ops
                    tmp = torch.empty_like(lay.y)
                    _ops.bn_apply_relu(lay.y, lay.npix, lay.co, lay.ldy, lay.scale, lay.shift, lay.relu, tmp, lay.ldy)
                    torch.cuda.synchronize()
                    got = tmp[..., :lay.co]
                else:
                    got = lay.out[..., :lay.co] if lay.bn else lay.y[..., :lay.co]
                table.append((tag, lay.scope, rel(got, ref)))
                rc = acts[lay.scope + ':conv']
                table.append((tag, lay.scope + ':conv', rel(lay.y[..., :lay.co], rc)))
        for name, (y, H) in eng.vgg_activations().items():
            table.append((tag, 'vgg/' + name, rel(y, acts['vgg'][name])))
    for row in table:
        print('ACT_PARITY %-5s %-48s %.4g' % row)
    assert max(r[2] for r in table if not r[1].startswith('model/renderer/conv_8')) < 0.2


def test_gradient_parity(fwd2):
    """The step's gradient is ill-conditioned at initialisation: the fp32 oracle differs from an fp64
    run of itself by 2e-3..7e-3 and from its own bf16-storage emulation by up to 0.44 relative L2
    (encoders), 0.003 at the last conv (numbers in DESIGN.md).  So the bar for the bf16 engine is:
    (i) tight agreement where the problem is well conditioned (last renderer convs), and
    (ii) everywhere, no further from the fp32 oracle than the oracle's own bf16 emulation is."""
    eng = fwd2['eng']
    g_e, g_f = fwd2['g_e'], fwd2['g_f']
    fails = []
    for k, v in g_f.items():
        if k.endswith('/b') and (k[:-2] + '/gamma') in g_f:
            # conv bias in front of a batch norm: analytically zero gradient; the oracle (like TF)
            # produces rounding noise there, the engine writes exact zeros (DESIGN.md numerics)
            assert float(eng.gview[k].abs().max()) == 0.0
            if float(v.norm()) > 1e-3 * float(g_f[k[:-2] + '/gamma'].norm()) + 1e-6:
                fails.append(('bias-not-noise', k, float(v.norm())))
            continue
        if float(v.norm()) < 1e-6:
            continue     # pose 1x1 bias: |g| ~ 3e-8, pure cancellation noise in every implementation
        e_eng, e_emul = rel(eng.gview[k], v), rel(g_e[k], v)
        print('GRAD_PARITY %-48s engine-vs-fp32 %.4g   emul-vs-fp32 %.4g   |g|=%.4g' % (k, e_eng, e_emul, float(v.norm())))
        if e_eng > 1.5 * e_emul + 0.02:
            fails.append(('worse-than-bf16-emulation', k, e_eng, e_emul))
    for k, lim in (('model/renderer/conv_8/w', 1e-2), ('model/renderer/conv_8/b', 1e-3), ('model/renderer/conv_7/gamma', 2e-2),
                   ('model/renderer/conv_7/beta', 2e-2), ('model/renderer/conv_7/w', 8e-2)):
        if rel(eng.gview[k], g_f[k]) > lim:
            fails.append(('tight', k, rel(eng.gview[k], g_f[k]), lim))
    assert not fails, fails


def test_f16_storage_scales_the_error_down():
    """Same engine with f16 storage (3 more mantissa bits than bf16): errors against the fp32 oracle must
    shrink accordingly where f16's range suffices (renderer / image encoder), which separates storage
    noise from wiring mistakes in the backward pass."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(10)
    inputs = O.synthetic_inputs(2, 128)
    P, St = O.init_params(cfg, 128)
    out_f, g_f = O.loss_and_grads(P, St, inputs, cfg)
    res = {}
    for dt in (torch.float16, torch.bfloat16):
        model = IMMModel(Box(dict(cfg)), dtype=dt, device=DEV)
        _, loss, _, t = model.build(inputs, True, output_tensors=True)
        model.engine.backward()
        torch.cuda.synchronize()
        eng = model.engine
        assert (eng.loss_scale > 1.0) == (dt == torch.float16)       # f16 runs with loss scaling, bf16 without
        gv = eng.named_gradients()                                    # loss scale divided out
        res[dt] = dict(pred=rel(t['future_im_pred'], out_f['future_im_pred']),
                       mu=float((t['gauss_yx'].cpu() - out_f['gauss_yx'].detach()).abs().max()),
                       ren1=rel(gv['model/renderer/conv_1/w'], g_f['model/renderer/conv_1/w']),
                       ren5=rel(gv['model/renderer/conv_5/w'], g_f['model/renderer/conv_5/w']),
                       enc8=rel(gv['model/image_encoder/encoder/conv_8/w'], g_f['model/image_encoder/encoder/conv_8/w']),
                       enc1=rel(gv['model/image_encoder/encoder/conv_1/w'], g_f['model/image_encoder/encoder/conv_1/w']))
    print('\nF16_VS_BF16 ' + json.dumps({str(k): v for k, v in res.items()}))
    f, b = res[torch.float16], res[torch.bfloat16]
    assert f['pred'] < 0.4 * b['pred'] and f['pred'] < 0.03, res
    assert f['mu'] < 1e-3, res
    for k in ('ren1', 'ren5', 'enc8', 'enc1'):
        assert f[k] < 0.5 * b[k], (k, res)


def test_one_step_matches_oracle(fwd2):
    cfg, eng, inputs = fwd2['cfg'], fwd2['eng'], fwd2['inputs']
    eng.optimizer_step()
    torch.cuda.synchronize()
    Pe, Se = emul_params(fwd2['P'], fwd2['St'])
    opt = O.new_adam_state(Pe)
    newP, newS, info = O.train_step(Pe, Se, opt, [inputs], cfg, clip=1.0, lr=O.learning_rate(0), act_round=bf)
    got = eng.named_parameters()
    bad, coss = [], []
    for k, v in newP.items():
        if k.endswith('/b'):
            continue   # gradient is cancellation noise in the oracle (see test_gradient_parity)
        # the first Adam step moves every element by ~lr*sign(g): compare update DIRECTIONS.  With the
        # bf16-limited gradient agreement measured above the sign vectors agree on >= ~80 % of elements.
        du_ref = (v - Pe[k]).detach().flatten().double()
        du_got = (got[k].cpu() - fwd2['P'][k]).flatten().double()
        cos = float((du_ref * du_got).sum() / (du_ref.norm() * du_got.norm() + 1e-30))
        # round 5: the floor for the ill-conditioned tensors raised from 0.5 (measured bf16 minimum 0.60 against the fp32 oracle)
        lim = 0.97 if k in ('model/renderer/conv_8/w', 'model/renderer/conv_7/gamma', 'model/renderer/conv_7/beta') else 0.55
        print('STEP_PARITY %-48s cos(update) %.4f' % (k, cos))
        coss.append(cos)
        if cos < lim:
            bad.append((k, cos))
        mag = float(du_got.abs().max())
        if not (mag <= 1.05e-3):
            bad.append((k, 'update magnitude', mag))      # |lr_t * m/(sqrt(v)+eps)| <= lr at t=1
    assert not bad, bad
    assert float(np.median(np.array(coss))) >= 0.75, float(np.median(np.array(coss)))
    st = eng.named_state()
    for k, v in newS.items():
        if k.startswith('vgg16/'):
            continue
        tol = 2e-2 if 'moving' in k else 5e-3
        assert rel(st[k], v) < tol, (k, rel(st[k], v))
    assert int(eng.step_count) == 1


def test_one_step_of_the_f16_engine_matches_the_fp32_oracle():
    """The tight witness of the update direction (VERDICT r4 item 5 ii): the f16 engine (loss scale 4096) against the PLAIN fp32
    oracle — no storage emulation in between.  The first Adam step moves every element by lr * sign(g), so cos(update) is the
    fraction of elements whose gradient sign agrees: measured 0.82 (pose-encoder conv_4 gamma, the worst-conditioned tensor at
    the 0.01-std initialisation) .. 1.0, median 0.94 — bounds 0.75 / 0.97 for the three well-conditioned tensors / median 0.9
    (bf16 reaches 0.60 / median 0.80 on the same comparison)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(10)
    inputs = O.synthetic_inputs(2, 128)
    P, St = O.init_params(cfg, 128)
    newP, _newS, _info = O.train_step(P, St, O.new_adam_state(P), [inputs], cfg, clip=1.0, lr=O.learning_rate(0))
    model = IMMModel(Box(dict(cfg)), dtype=torch.float16, device=DEV)
    model.build(inputs, True)
    eng = model.engine
    eng.backward(); eng.optimizer_step()
    torch.cuda.synchronize()
    assert int(eng.step_count) == 1 and float(eng.loss_scale_state[0]) == 4096.0      # no overflow, update applied
    got = eng.named_parameters()
    bad, coss = [], []
    for k, v in newP.items():
        if k.endswith('/b'):
            continue
        du_ref = (v - P[k]).detach().flatten().double()
        du_got = (got[k].cpu() - P[k]).flatten().double()
        cos = float((du_ref * du_got).sum() / (du_ref.norm() * du_got.norm() + 1e-30))
        coss.append(cos)
        lim = 0.97 if k in ('model/renderer/conv_8/w', 'model/renderer/conv_7/gamma', 'model/renderer/conv_7/beta') else 0.75
        if cos < lim:
            bad.append((k, cos))
        assert float(du_got.abs().max()) <= 1.05e-3, k
    assert not bad, bad
    assert float(np.median(np.array(coss))) >= 0.9, float(np.median(np.array(coss)))


def test_eval_mode_and_model_only(fwd2):
    model, inputs, cfg = fwd2['model'], fwd2['inputs'], fwd2['cfg']
    eng = fwd2['eng']
    agg0, mm0 = eng.loss_agg.clone(), {k: v.clone() for k, v in eng.state.items()}
    _, loss, _, t = model.build(inputs, False, output_tensors=True)
    torch.cuda.synchronize()
    assert torch.equal(agg0, eng.loss_agg) and all(torch.equal(mm0[k], eng.state[k]) for k in mm0)   # S12
    P, St = eng.named_parameters(), eng.named_state()
    Sfull = dict(fwd2['St']); Sfull.update({k: v.cpu() for k, v in St.items()})
    Pe, Se = emul_params(type(fwd2['P'])((k, v.cpu()) for k, v in P.items()), type(fwd2['St'])(Sfull))
    out = O.forward(Pe, Se, inputs, cfg, training=False, act_round=bf)
    assert float((t['gauss_yx'].cpu() - out['gauss_yx']).abs().max()) < 1e-3
    assert abs(float(loss) - float(out['loss'])) / abs(float(out['loss'])) < 2e-2
    _, loss2, _ = model.build(inputs, False, build_loss=False)
    assert loss2 is None


def test_reference_error_types():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = Box(dict(O.default_model_config()))
    inp = O.synthetic_inputs(1, 128)
    del inp['mask']
    with pytest.raises(RuntimeError):
        IMMModel(cfg, device=DEV).build(inp, True)
    c2 = Box(dict(O.default_model_config())); c2.reconstruction_loss = 'huber'
    with pytest.raises(ValueError):
        IMMModel(c2, device=DEV).build(O.synthetic_inputs(1, 128), True)
    c3 = Box(dict(O.default_model_config())); c3.gauss_mode = 'nope'
    with pytest.raises(ValueError):
        IMMModel(c3, device=DEV).build(O.synthetic_inputs(1, 128), True)
    bad = O.synthetic_inputs(1, 128); bad['future_image'] = bad['future_image'][:, :, :64]; bad['image'] = bad['image'][:, :, :64]
    with pytest.raises(AssertionError):
        IMMModel(cfg, device=DEV).build(bad, True)


def test_graph_replay_equals_eager_and_is_deterministic():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    res = []
    for use_graph, split in ((False, False), (True, False), (True, False), (True, True)):
        cfg, model, eng, inputs, P, St = make(4)
        ts = TrainStep(model, 4, 128, world_size=1, use_graph=use_graph, split_graphs=split)
        for it in range(3):
            loss = ts.step(inputs)
        ts.synchronize()
        res.append((float(loss), eng.params.clone(), eng.loss_agg.clone(), int(eng.step_count)))
    assert res[1][0] == res[2][0] and torch.equal(res[1][1], res[2][1])          # replay is deterministic
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])          # graph == eager, bitwise
    assert res[0][3] == 3 and torch.equal(res[0][2], res[1][2])
    # the multi-GPU structure (graph(fwd+bwd) | all-reduce | graph(optimizer)) computes the same thing
    assert res[3][0] == res[1][0] and torch.equal(res[3][1], res[1][1])
    assert np.isfinite(res[0][0])


def test_full_size_step_properties():
    """BASELINE.json configs[1]: batch 32, 128x128, K=10 — size-independent properties."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    cfg, model, eng, inputs, P, St = make(32)
    ts = TrainStep(model, 32, 128, world_size=1, use_graph=True)
    losses = []
    for it in range(6):
        losses.append(ts.step(inputs).clone())
    ts.synchronize()
    losses = [float(l) for l in losses]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses                      # same batch: the loss must go down
    assert bool(torch.isfinite(eng.params).all()) and bool(torch.isfinite(eng.grads).all())
    mu = eng.mu.cpu()
    assert float(mu.abs().max()) <= 1.0                        # expectation of linspace(-1,1)
    # per-tensor clip: every clipped gradient has norm <= 1 (+ fp slack)
    n2 = eng.seg_norm2.cpu()
    assert bool((n2 >= 0).all())
    # the 6 bug-fix channels of the last conv get no loss gradient (S10): their bias gradient is exactly 0
    assert float(eng.gview['model/renderer/conv_8/b'][3:].abs().max()) == 0.0


@pytest.mark.parametrize('K,dt', [(10, torch.bfloat16), (50, torch.float16)], ids=['configs1_k10_bf16', 'configs4_k50_f16'])
def test_full_size_forward_matches_oracle(K, dt):
    """The benchmark's own size against the oracle (not only properties): batch 32 per GPU at 128x128 — BASELINE.json
    configs[1] (K=10, bf16) and the per-GPU shape of configs[4] (K=50, f16) — forward + perceptual loss vs oracle.forward on
    identical inputs: landmarks <= 1e-3 (the BASELINE target), total loss <= 1e-3 relative, each of the six terms <= 1e-2.
    At this batch every large-grid kernel variant the bench runs is selected (persistent conv_hdeep / conv_halo2 tiles, the
    grouped stride-2 data gradients, the pre-reduced batch-norm rows), which the batch-2 tests do not reach."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    B = 32
    cfg = O.default_model_config(K)
    inputs = O.synthetic_inputs(B, 128, seed=0)
    model = IMMModel(Box(dict(cfg)), dtype=dt, device=DEV)
    _, loss, _, t = model.build(inputs, True, output_tensors=True)
    torch.cuda.synchronize()
    eng = model.engine
    P, St = O.init_params(cfg, 128)
    with torch.no_grad():
        out = O.forward(P, St, inputs, cfg, training=True)
    mu_err = float((t['gauss_yx'].cpu() - out['gauss_yx']).abs().max())
    loss_rel = abs(float(loss) - float(out['loss'])) / abs(float(out['loss']))
    terms = eng.loss_terms.cpu()
    oterms = torch.stack([x.detach().float() for x in out['loss_terms']])
    term_rel = float(((terms - oterms).abs() / oterms.abs()).max())
    pred_rel = rel(t['future_im_pred'], out['future_im_pred'])
    print('\nFULLSIZE K=%d %s: mu_maxabs %.3g loss_rel %.3g terms_rel %.3g pred_rel %.3g' % (K, dt, mu_err, loss_rel, term_rel, pred_rel))
    assert t['gauss_yx'].shape == (B, K, 2)
    assert mu_err < 1e-3 and loss_rel < 1e-3 and term_rel < 1e-2
    assert pred_rel < (0.12 if dt == torch.bfloat16 else 0.03)
    # and one full step at this size runs clean (f16: with the loss scale; no overflow at the initial scale)
    eng.backward(); eng.optimizer_step()
    torch.cuda.synchronize()
    assert int(eng.step_count) == 1 and bool(torch.isfinite(eng.params).all())


def test_config4_256px_k30_forward_and_step():
    """BASELINE.json configs[3]: K=30 at 256x256 — exercises the align-corners 32->16 embedding resize
    (imm_model.py:324-335), the 32x32 heat-map bottleneck and the 10-conv renderer."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    cfg, model, eng, inputs, P, St = make(1, K=30, S=256)
    _, loss, _, t = model.build(inputs, True, output_tensors=True)
    torch.cuda.synchronize()
    out = O.forward(P, St, inputs, cfg, training=True)
    mu_err = float((t['gauss_yx'].cpu() - out['gauss_yx']).abs().max())
    loss_rel = abs(float(loss) - float(out['loss'])) / abs(float(out['loss']))
    print('\nCONFIG4 mu_maxabs %.3g loss_rel %.3g pred_rel %.3g' % (mu_err, loss_rel, rel(t['future_im_pred'], out['future_im_pred'])))
    assert t['future_im_pred'].shape == (1, 256, 256, 3) and t['heatmaps'].shape == (1, 32, 32, 30)
    assert mu_err < 1e-3 and loss_rel < 1e-3
    assert rel(t['future_im_pred'], out['future_im_pred']) < 0.15
    ts = TrainStep(model, 1, 256, world_size=1, use_graph=True)
    l0 = float(ts.step(inputs).clone()); ts.synchronize()
    for _ in range(4):
        l1 = ts.step(inputs)
    ts.synchronize()
    assert np.isfinite(float(l1)) and float(l1) < l0 and bool(torch.isfinite(eng.params).all())


def test_config5_k50_f16_step():
    """BASELINE.json configs[4] shape per GPU: K=50, f16 storage / f16 MFMA."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(50)
    model = IMMModel(Box(dict(cfg)), dtype=torch.float16, device=DEV)
    inputs = O.synthetic_inputs(2, 128)
    _, loss, _, t = model.build(inputs, True, output_tensors=True)
    torch.cuda.synchronize()
    P, St = O.init_params(cfg, 128)
    out = O.forward(P, St, inputs, cfg, training=True)
    mu_err = float((t['gauss_yx'].cpu() - out['gauss_yx']).abs().max())
    loss_rel = abs(float(loss) - float(out['loss'])) / abs(float(out['loss']))
    print('\nCONFIG5 mu_maxabs %.3g loss_rel %.3g pred_rel %.3g' % (mu_err, loss_rel, rel(t['future_im_pred'], out['future_im_pred'])))
    assert t['gauss_yx'].shape == (2, 50, 2) and mu_err < 1e-3 and loss_rel < 1e-3
    assert rel(t['future_im_pred'], out['future_im_pred']) < 0.03      # f16: 3 more mantissa bits than bf16
    ts = TrainStep(model, 2, 128, world_size=1, use_graph=True)
    losses = [float(ts.step(inputs).clone()) for _ in range(4)]
    ts.synchronize()
    ls = ts.engine.loss_scale_state.tolist()
    print('CONFIG5 loss scale state after 4 steps (S, clean, skipped, overflow): %r' % (ls,))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert ls[2] <= 1.0 and int(ts.engine.step_count) == 4 - int(ls[2])     # at most one step lost to finding the scale


def test_training_reduces_the_loss_on_a_fixed_batch():
    """Property of the whole step (forward, perceptual loss, backward, clip, Adam) beyond one-step parity: repeated steps
    on one batch of smooth images drive the loss down and keep every buffer finite; the six loss normalisers follow their
    0.99 moving average (base_model.py:40-48)."""
    import torch
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    B, S = 8, 128
    g = torch.Generator().manual_seed(3)
    low = torch.rand(B, 3, 6, 6, generator=g)
    img = torch.nn.functional.interpolate(low, size=(S, S), mode='bilinear', align_corners=True).permute(0, 2, 3, 1) * 255
    fut = torch.roll(img, shifts=(5, -7), dims=(1, 2))                 # a shifted copy: structure to reconstruct
    inputs = {'image': img.contiguous(), 'future_image': fut.contiguous(), 'mask': torch.ones(B, S, S, 1)}
    model = IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device='cuda:0')
    ts = TrainStep(model, B, S, world_size=1, use_graph=True)
    eng = model.engine
    agg0 = eng.loss_agg.clone()
    losses = []
    for i in range(120):
        loss = ts.step(inputs if i == 0 else None)
        if i % 10 == 0 or i == 119:
            ts.synchronize()
            losses.append(float(loss))
    ts.synchronize()
    assert all(np.isfinite(losses)), losses
    # the normalisers divide each term by its running mean, so the reported loss starts near 1000 * 6 terms-ish and
    # falls as the reconstruction improves faster than the averages follow
    assert losses[-1] < 0.8 * losses[0], losses
    assert min(losses[-3:]) < min(losses[:3]), losses
    assert bool(torch.isfinite(eng.params).all()) and bool(torch.isfinite(eng.adam_v).all())
    assert int(eng.step_count) == 120
    assert not torch.equal(eng.loss_agg, agg0)
    mu = eng.mu.float().cpu().numpy()
    assert np.abs(mu).max() <= 1.0 + 1e-3                              # landmarks stay inside the image frame


def test_bench_rccl_path_single_rank():
    """bench.py with --force-dist: RCCL process group of one rank, the split-graph step with the gradient all-reduce between
    the backward and the optimizer graphs — the code path every rank runs under `torch.distributed.run` at N > 1."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--force-dist', '--steps', '3', '--warmup', '1',
                          '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                                   # exactly one JSON line on stdout
    d = json.loads(lines[0])
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['roofline']['frac'] > 0
    # round 5: over RCCL the default is chosen by a start-up self-check — the one-graph step with two overlapped buckets against the
    # eager one-stream step — and recorded in the line
    c = d['config']['collective']
    assert c['selfcheck']['passed'] is True and c['selfcheck']['update_rel_diff'] < 1e-3 and c['selfcheck']['replicas_identical'] is True
    assert c['mode'] == 'graph' and c['buckets'] == 2 and c['graph_resident'] is True
    # an explicit choice skips the self-check
    out = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--force-dist', '--steps', '2', '--warmup', '1', '--windows', '1',
                          '--spin-seconds', '0', '--no-cpu-baseline', '--no-pmc', '--collective', 'pg'], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    c = json.loads([l for l in out.stdout.decode().splitlines() if l.strip()][0])['config']['collective']
    assert c['mode'] == 'pg' and c['buckets'] == 1 and c['selfcheck'] is None


def _smooth_batch(B, S, seed=3):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, 6, 6, generator=g)
    img = torch.nn.functional.interpolate(low, size=(S, S), mode='bilinear', align_corners=True).permute(0, 2, 3, 1) * 255
    fut = torch.roll(img, shifts=(5, -7), dims=(1, 2))
    mask = O.smooth_mask(S, S).reshape(1, S, S, 1).repeat(B, 1, 1, 1)
    return {'image': img.contiguous(), 'future_image': fut.contiguous(), 'mask': mask.contiguous()}


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16], ids=['bf16', 'f16'])
def test_gradient_parity_on_a_trained_model(dt):
    """Chain-level gradient parity where the problem is WELL CONDITIONED.  At the 0.01-std initialisation the gradient is
    ill conditioned (test_gradient_parity); after 60 Adam steps on a smooth batch the fp32 oracle agrees with an fp64 run of
    itself to 1.3e-4 on every tensor (tools/diag/diag_grad_conditioning.py), so a mis-scaled or mis-wired gradient anywhere in
    the backward chain would show.  Bound for the bf16 engine against the fp32 oracle at the SAME trained parameters and
    state: EVERY trainable tensor no further from the fp32 oracle than 1.25 x the oracle's own bf16-storage emulation + 0.03 (the
    emulation rounds weights and forward activations; the engine also stores every gradient tensor in bf16) AND within 0.30
    relative L2 / cosine >= 0.97 — the absolute pair being waived for a tensor only when the emulation itself is outside it in
    this trained state and the engine is at least as close to the fp32 oracle as the emulation (round 4: a rounding-order change
    in one data-gradient kernel landed the trajectory in a state whose first pose-encoder beta is at 0.387 for the emulation,
    0.378 for the engine); median over the tensors <= 0.10.  The 60-step trajectory is chaotic: numerically
    equivalent builds (another summation order of the batch-norm partial rows is enough) end in different trained states.
    Measured over such builds: worst 0.095 .. 0.23, median 0.04 .. 0.07, worst cosine 0.98 .. 0.9955; tensor by tensor the
    engine sits at 0.9 .. 2.2 x the emulation where the emulation itself is small (0.03), within +-10 % where it is large:
    what is left is bf16 storage, not wiring.
    f16 (BASELINE configs[4]; the reference is fp32, imm_model.py:97): the same test with HALF the bounds (0.15 / cosine 0.99 /
    median 0.05; still no further from the oracle than 1.25 x its bf16 emulation + 0.03).  Without loss scaling the stored dy tensors of the trained pose encoder underflow f16 (relative error up to
    3.6 on its first convolution, round 2); with the dynamic loss scale of imm_clip_adam_step the f16 engine has to be at
    least as close to the fp32 oracle as the bf16 engine is allowed to be, and no step of the 60 may be lost to an overflow
    after the first few (the scale search)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    B, S, steps = 4, 128, 60
    cfg = O.default_model_config(10)
    inputs = _smooth_batch(B, S)
    model = IMMModel(Box(dict(cfg)), dtype=dt, device=DEV)
    ts = TrainStep(model, B, S, world_size=1, use_graph=True)
    for i in range(steps):
        ts.step(inputs if i == 0 else None)
    ts.synchronize()
    eng = ts.engine
    if dt == torch.float16:
        ls = eng.loss_scale_state.tolist()
        print('TRAINED_GRAD f16 loss scale state (S, clean, skipped, overflow): %r, steps applied %d' % (ls, int(eng.step_count)))
        assert ls[0] >= 2.0 and ls[2] <= 4 and int(eng.step_count) == steps - int(ls[2])
    else:
        assert eng.loss_scale_state is None and int(eng.step_count) == steps
    P0, St0 = O.init_params(cfg, S)
    P1 = type(P0)((k, v.cpu()) for k, v in eng.named_parameters().items())
    St1 = type(St0)(St0)
    St1.update({k: v.cpu() for k, v in eng.named_state().items()})
    agg = eng.loss_agg.clone()
    eng.forward(True); eng.backward()
    torch.cuda.synchronize()
    eng.loss_agg.copy_(agg)
    out_f, g_f = O.loss_and_grads(P1, St1, inputs, cfg)
    Pe, Se = emul_params(P1, St1)
    _oe, g_e = O.loss_and_grads(Pe, Se, inputs, cfg, act_round=bf)
    assert abs(float(eng.loss) - float(out_f['loss'])) / abs(float(out_f['loss'])) < 2e-3
    bad, rels, worst, waivers = [], [], (0.0, 1.0), []
    # f16 storage rounds 8x finer than bf16: measured worst 0.035, median 0.008, cosine 0.9994 (round 3) — held to half the
    # bf16 bounds (which stay where the chaotic 60-step trajectory needs them, see above)
    lim_rel, lim_cos, lim_med = (0.30, 0.97, 0.10) if dt == torch.bfloat16 else (0.15, 0.99, 0.05)
    gv = eng.named_gradients()            # loss scale divided out (f16); == gview for bf16
    for k, v in g_f.items():
        if k.endswith('/b') and (k[:-2] + '/gamma') in g_f:
            assert float(gv[k].abs().max()) == 0.0       # analytically zero (BN removes the mean); oracle: noise
            continue
        if float(v.norm()) < 1e-7:
            continue
        a, b = gv[k].detach().cpu().double().flatten(), v.detach().double().flatten()
        e = float((a - b).norm() / b.norm())
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        e_emul = rel(g_e[k], v)
        c = g_e[k].detach().double().flatten()
        cos_emul = float((c * b).sum() / (c.norm() * b.norm()))
        rels.append(e)
        worst = (max(worst[0], e), min(worst[1], cos))
        print('TRAINED_GRAD %-48s rel %.4f cos %.5f  emul %.4f cos %.5f' % (k, e, cos, e_emul, cos_emul))
        # (i) tensor by tensor the engine stays with the oracle's own bf16-storage emulation: 1.25 x + 0.03 (measured
        #     ratios 0.9 .. 1.2 where the emulation is large, up to 2.2 where it is ~0.03); (ii) the absolute bounds — unless the emulation itself is outside them for this tensor of this
        #     trained state (the 60-step trajectory is chaotic: a rounding-order change in any kernel lands in another state, and
        #     the first pose-encoder layer of some states is ill conditioned: round 4 met beta at 0.378 with the emulation at
        #     0.387): then the engine must be at least as close to the fp32 oracle as the emulation is
        tight = e <= 1.25 * e_emul + 0.03
        waived = not (e <= lim_rel and cos >= lim_cos) and (e <= e_emul and cos >= cos_emul - 1e-3)
        absolute = (e <= lim_rel and cos >= lim_cos) or waived
        # (iii) ADVICE r4: the waiver has a hard outer cap of its own (a mis-scaled gradient in an ill-conditioned tensor must not
        #     pass just because the emulation is bad there too) and at most ONE tensor of a trained state may use it
        if waived:
            waivers.append((k, e, cos))
            absolute = e <= 0.45 and cos >= 0.90
        if not (tight and absolute):
            bad.append((k, e, cos, e_emul, cos_emul))
    med = float(np.median(np.array(rels)))
    print('TRAINED_GRAD worst rel %.4f, worst cos %.5f, median rel %.4f' % (worst + (med,)))
    assert not bad, bad
    assert len(waivers) <= 1, waivers
    if waivers:
        print('TRAINED_GRAD waiver used by', waivers)
    assert med <= lim_med, med


def test_backward_is_the_derivative_of_the_forward():
    """Finite-difference check of the engine's backward pass against its OWN forward pass (independent of any oracle and of
    the conditioning of the gradient): for a direction d confined to one tensor (or one sub-network),
        [L(theta + eps d) - L(theta - eps d)] / (2 eps)  ==  <g_engine, d>,     d = g_engine restricted / its norm.
    A mis-scaled gradient entering an encoder, a wrong wgrad/dgrad pairing or a wrong BN-backward coefficient changes the
    right-hand side only.  f16 storage (8x less rounding noise than bf16 in the forward), eps = 3e-4 |theta_d|: the step-size
    table of tools/diag/diag_fd_check.py shows the central difference within 0.2 % of the analytic value there for single tensors
    (the loss is strongly curved along the gradient: measured ratios 0.97 .. 1.00 at 3e-4, bound 5 %).  Sub-networks whose gradient is too small for the
    f32 resolution of the loss at that step (the pose encoder at initialisation: |g| ~ 1, loss ~ 5e4) are checked as one
    group with a looser bound."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.engine import WEIGHT_DECAY
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(10)
    inputs = O.synthetic_inputs(2, 128, seed=0)
    model = IMMModel(Box(dict(cfg)), dtype=torch.float16, device=DEV)
    eng = model._get_engine(2, 128)
    eng.set_inputs(inputs['image'].to(DEV), inputs['future_image'].to(DEV), inputs['mask'].to(DEV))
    agg0 = eng.loss_agg.clone()
    eng.forward(True); eng.backward()
    torch.cuda.synchronize()
    eng.loss_agg.copy_(agg0)
    g = eng.grads.clone() / eng.loss_scale          # f16 engine: gradients carry the loss scale until the optimizer step
    p0 = eng.params.clone()
    names = [n for n, _s, _w in eng.spec]

    def total_loss(theta):
        eng.params.copy_(theta)
        eng.run(eng.prog_pack)
        eng.loss_agg.copy_(agg0)
        eng.forward(True)
        torch.cuda.synchronize()
        return float(eng.loss.double())

    def check(sel, rel_eps, tol, what):
        d = torch.zeros_like(g)
        for i in sel:
            d[eng.tab.offsets[i]:eng.tab.offsets[i + 1]] = g[eng.tab.offsets[i]:eng.tab.offsets[i + 1]]
        analytic = float(d.double().norm())
        d /= d.norm()
        # the total loss also carries the weight decay, whose gradient is added inside imm_clip_adam_step, not in eng.grads
        wd = sum(WEIGHT_DECAY * float((p0[eng.tab.offsets[i]:eng.tab.offsets[i + 1]].double()
                                       * d[eng.tab.offsets[i]:eng.tab.offsets[i + 1]].double()).sum())
                 for i in sel if eng.spec[i][2])
        pn = max(float((p0 * (d != 0)).norm()), 0.01 * float((d != 0).sum()) ** 0.5)
        eps = rel_eps * pn
        fd = (total_loss(p0 + eps * d) - total_loss(p0 - eps * d)) / (2 * eps) - wd
        print('FD_CHECK %-44s <g,d> %10.4g  FD %10.4g  ratio %.4f' % (what, analytic, fd, fd / analytic))
        return None if abs(fd / analytic - 1.0) <= tol else (what, analytic, fd)

    bad = []
    try:
        for i, k in enumerate(names):
            part = k.split('/')[1]
            if part == 'pose_encoder' or (k.endswith('/b') and (k[:-2] + '/gamma') in names):
                continue
            gi, pi = g[eng.tab.offsets[i]:eng.tab.offsets[i + 1]], p0[eng.tab.offsets[i]:eng.tab.offsets[i + 1]]
            if float(gi.norm()) * 3e-4 * max(float(pi.norm()), 0.01 * gi.numel() ** 0.5) < 0.05:
                continue                 # loss change below ~50 x the f32 resolution of a loss of ~5e4 at this step size
            bad.append(check([i], 3e-4, 0.05, k))
        for part, tol in (('renderer', 0.03), ('pose_encoder', 0.12)):
            sel = [i for i, k in enumerate(names) if k.split('/')[1] == part and not (k.endswith('/b') and (k[:-2] + '/gamma') in names)]
            bad.append(check(sel, 3e-4, tol, part + ' (all tensors)'))
    finally:
        eng.params.copy_(p0)
        eng.run(eng.prog_pack)
    bad = [b for b in bad if b is not None]
    assert not bad, bad


def test_native_rccl_exchange_single_rank(monkeypatch):
    """collective='native' / 'graph': the gradient exchange through the C-ABI (imm_rccl_unique_id / init / allreduce / destroy) on
    the communication stream of TrainStep resp. as a node of the step's one graph — a one-rank RCCL communicator on the test box
    (sum over one rank = identity), so the step must equal the plain single-GPU step bit for bit, with one bucket and with two
    overlapped buckets.  The modes are chosen by ARGUMENT (IMMModel(dp_buckets=...), TrainStep(collective=...)); the last case
    goes through the environment variables, which only supply those arguments' defaults."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    res = {}
    for v in ('IMM_RCCL_NATIVE', 'IMM_DP_BUCKETS', 'IMM_RCCL_GRAPH'):
        monkeypatch.delenv(v, raising=False)
    # the engine is built per bucket mode (two buckets: the renderer's filter gradients are issued and reduced early: the same
    # sums in another order), so every mode is compared with the plain single-graph step of an engine built the same way
    for native, buckets, in_graph, via_env in ((False, 1, False, False), (True, 1, False, False), (True, 1, True, False),
                                               (False, 2, False, False), (True, 2, False, False), (True, 2, False, True)):
        if via_env:
            monkeypatch.setenv('IMM_RCCL_NATIVE', '1'); monkeypatch.setenv('IMM_DP_BUCKETS', str(buckets))
            cfg, model, eng, inputs, P, St = make(4)
            ts = TrainStep(model, 4, 128, world_size=1, use_graph=True, split_graphs=True)
        else:
            cfg, model, eng, inputs, P, St = make(4, dp_buckets=buckets)
            ts = TrainStep(model, 4, 128, world_size=1, use_graph=True, split_graphs=native,
                           collective='graph' if in_graph else 'native' if native else 'pg')
        assert (ts.native_comm is not None) == native and ts.graph_resident == in_graph and ts.buckets == buckets
        assert ts.collective == (None if not native else 'graph' if in_graph else 'native')
        for it in range(3):
            loss = ts.step(inputs)
        ts.synchronize()
        res[(native, buckets, in_graph, via_env)] = (float(loss), eng.params.clone())
        if native:
            ts.native_comm.destroy()
        if in_graph:
            assert len(ts._graphs) == 1
    for key, (loss, params) in res.items():
        ref = res[(False, key[1], False, False)]
        assert loss == ref[0] and torch.equal(params, ref[1]), key
    # and the two engine builds agree to rounding: the same filter-gradient sums in another order, carried through three updates whose
    # re-packed 16-bit filters flip by an ulp here and there (measured 0 ... 1.4e-5 of the third step's loss over the builds of round 6)
    assert abs(res[(False, 1, False, False)][0] - res[(False, 2, False, False)][0]) <= 5e-5 * abs(res[(False, 1, False, False)][0])


def test_exchange_ordering_is_enforced_poisoned_gradients_and_delayed_collective(monkeypatch):
    """the hand-offs between the step's stream, the communication stream and the optimizer have only ever met a
    ONE-rank communicator, where a mis-ordered all-reduce changes no value.  TrainStep(debug_poison=True) makes ordering visible on
    one GPU: the gradient buffer is NaN between steps (every element a backward pass rewrites) and every exchange leaves `*= 2`
    behind on the stream it was issued on.  Expected result = the eager ONE-stream sequence fwd, bwd, grads *= 2, clip + Adam.
      * all modes (C-ABI collective on its own stream / as nodes of the step's graph, one bucket / two overlapped buckets), with
        the collective DELAYED by a 3 ms spin on its stream: bit-identical to the expected result — events, not luck, order them;
      * negative control: the same delayed exchange with the join in front of the optimizer removed is caught (wrong or NaN)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    for v in ('IMM_RCCL_NATIVE', 'IMM_DP_BUCKETS', 'IMM_RCCL_GRAPH', 'IMM_DEBUG_POISON_GRADS'):
        monkeypatch.delenv(v, raising=False)

    def delayed(comm):
        real = comm.all_reduce_sum

        def f(flat):
            torch.cuda._sleep(6_000_000)              # ~3 ms on the collective's stream, in front of the collective
            real(flat)
        comm.all_reduce_sum = f
    for buckets in (1, 2):
        cfg, model, eng, inputs, P, St = make(4, dp_buckets=buckets)
        ref = TrainStep(model, 4, 128, world_size=1, use_graph=False)
        ref.after_backward = lambda e: e.grads.mul_(2.0)
        for it in range(3):
            loss = ref.step(inputs)
        ref.synchronize()
        want_loss, want = float(loss), eng.params.clone()
        assert bool(torch.isfinite(want).all())
        for mode in ('native', 'graph'):
            cfg, model, eng, inputs, P, St = make(4, dp_buckets=buckets)
            ts = TrainStep(model, 4, 128, world_size=1, use_graph=True, split_graphs=True, collective=mode, debug_poison=True)
            delayed(ts.native_comm)
            for it in range(3):
                loss = ts.step(inputs)
            ts.synchronize()
            assert float(loss) == want_loss and torch.equal(eng.params, want), (mode, buckets)
            assert bool(torch.isnan(eng.grads).any())                 # the poison really sits there between steps
            assert ts.graph_resident == (mode == 'graph') and ts.buckets == buckets
            ts.native_comm.destroy()
    # negative control: one bucket, the exchange on its own stream, delayed, and NOT joined in front of the optimizer
    cfg, model, eng, inputs, P, St = make(4, dp_buckets=1)
    ts = TrainStep(model, 4, 128, world_size=1, use_graph=True, split_graphs=True, collective='native', debug_poison=True)
    delayed(ts.native_comm)

    def broken_exchange(e):
        ev = torch.cuda.Event(); ev.record(ts.stream)
        ts.comm_stream.wait_event(ev)
        with torch.cuda.stream(ts.comm_stream):
            ts.native_comm.all_reduce_sum(e.grads); ts._probe(e.grads)
        ts._graphs[-1].launch()                                      # <- no wait for the communication stream
    ts._native_exchange = broken_exchange
    for it in range(3):
        ts.step(inputs)
    ts.synchronize(); torch.cuda.synchronize()
    assert not torch.equal(eng.params, want) or not bool(torch.isfinite(eng.params).all())
    ts.native_comm.destroy()


def test_step_phases_show_an_exposed_exchange_and_hide_an_overlapped_one(monkeypatch):
    """TrainStep.measure_phases (bench.py: step.phases_ms) reads forward / backward / exchange-exposed / optimizer
    off device-clock probes of a replayed step.  Verified where it can be on one GPU: the C-ABI collective on its own stream DELAYED
    by a ~3 ms spin shows up as ~3 ms of `exchange_exposed` with ONE bucket (nothing beside it) and with TWO buckets the spin in
    front of the renderer bucket hides behind the encoders' backward (only the second bucket's spin stays exposed); without the
    delay the exposed time is a few tens of microseconds.  The probed graphs are dropped again: the next step re-captures."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    for v in ('IMM_RCCL_NATIVE', 'IMM_DP_BUCKETS', 'IMM_RCCL_GRAPH', 'IMM_DEBUG_POISON_GRADS'):
        monkeypatch.delenv(v, raising=False)
    spins = []

    def delayed(comm):
        real = comm.all_reduce_sum

        def f(flat):
            if spins and spins[0]:
                torch.cuda._sleep(6_000_000)          # ~3 ms in front of the collective, on the collective's stream
            real(flat)
        comm.all_reduce_sum = f
    got = {}
    for buckets in (1, 2):
        cfg, model, eng, inputs, P, St = make(4, dp_buckets=buckets)
        ts = TrainStep(model, 4, 128, world_size=1, use_graph=True, split_graphs=True, collective='native')
        delayed(ts.native_comm)
        ts.step(inputs); ts.synchronize()
        spins[:] = [False]
        ts._graphs = None
        quiet = ts.measure_phases(3)
        spins[:] = [True]
        slow = ts.measure_phases(3)
        spins[:] = [False]
        got[buckets] = (quiet, slow)
        for ph in (quiet, slow):
            assert ph['steps'] == 3 and all(ph[k] >= 0.0 for k in ('forward', 'backward', 'exchange_exposed', 'optimizer')), ph
            assert 0.05 < ph['forward'] < 20.0 and 0.05 < ph['backward'] < 20.0 and ph['optimizer'] < 5.0, ph
        assert ts._graphs is None                       # the probed capture is gone
        loss = ts.step(inputs); ts.synchronize()
        assert np.isfinite(float(loss))
        ts.native_comm.destroy()
    q1, s1 = got[1]
    q2, s2 = got[2]
    spin_ms = s1['exchange_exposed'] - q1['exchange_exposed']
    assert q1['exchange_exposed'] < 0.5 and 1.0 < spin_ms < 12.0, (q1, s1)           # one bucket: the whole delay is exposed
    # two buckets: two spins are issued (one per bucket); the first one runs beside the encoders' backward, so less than the
    # two spins' total is exposed — at least the part the encoders' backward covers
    assert s2['exchange_exposed'] - q2['exchange_exposed'] < 2.0 * spin_ms - 0.2, (q2, s2, spin_ms)


def test_forward_only_iterations(capsys):
    """The reference's fwd_only switch (cnn_train_multi.py:378,447-449): TrainStep.forward_only = forward + perceptual loss as one
    graph replay — the loss of a training step's forward pass, no update, global_step untouched; train_loop(fwd_only=True)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train import cnn_train_multi as tru
    cfg, model, eng, inputs, P, St = make(2)
    ts = tru.TrainStep(model, 2, 128, world_size=1, use_graph=True)
    p0 = eng.params.clone()
    l_f = float(ts.forward_only(inputs)); ts.synchronize()
    assert int(eng.step_count) == 0 and torch.equal(eng.params, p0)
    out = O.forward(P, St, inputs, cfg, training=True)
    assert abs(l_f - float(out['loss'])) <= 1e-3 * abs(float(out['loss']))
    # eager form, on a fresh model (the loss normalisers of the first one have moved with its forward pass): the same loss
    cfg2, model2, eng2, _i, _P, _S = make(2)
    l_e = float(tru.TrainStep(model2, 2, 128, world_size=1, use_graph=False).forward_only(inputs))
    assert l_e == l_f and int(eng2.step_count) == 0

    def batches():
        while True:
            yield inputs
    tru.train_loop({'batch_size': 2}, ts, batches(), 4, log_every=2, fwd_only=True)
    txt = capsys.readouterr().out
    assert txt.count('[fwd_only]') == 2 and 'Avg. samples per second' in txt
    assert int(eng.step_count) == 0 and torch.equal(eng.params, p0)
    ts.step(inputs); ts.synchronize()                   # the training graph is a separate capture
    assert int(eng.step_count) == 1 and not torch.equal(eng.params, p0)


def test_train_loop_follows_the_device_step_when_updates_are_skipped(capsys):
    """f16 storage with a loss scale that starts far too high: the first updates overflow and are SKIPPED on the device (weights,
    slots and global_step untouched).  train_loop's host count follows the device's global_step, so the run ends with global_step ==
    num_steps (more iterations than steps), checkpoints carry the saved global_step, and the loss stays finite."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train import cnn_train_multi as tru
    from imm_amd.utils.box import Box
    model = IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.float16, device=DEV, hparams={'loss_scale': 2.0 ** 22})
    ts = tru.TrainStep(model, 2, 128, world_size=1, use_graph=True)
    inputs = O.synthetic_inputs(2, 128, seed=0)
    fed, saved = [0], []

    def batches():
        while True:
            fed[0] += 1
            yield inputs
    tru.train_loop({'batch_size': 2, 'n_summary': 0, 'n_test': 0, 'n_checkpoint': 2}, ts, batches(), 6, log_every=1,
                   checkpoint_fn=lambda step: saved.append((step, int(ts.engine.step_count))))
    eng = ts.engine
    skipped = int(eng.loss_scale_state[2])
    assert skipped >= 1 and int(eng.step_count) == 6 and int(eng.adam_t) == 6
    assert fed[0] == 6 + skipped                               # one extra iteration per skipped update
    assert all(s % 2 == 0 for s, _ in saved) and saved[-1][0] == 4
    # a checkpoint written "at step s" holds global_step s + 1 (the update of step s has been applied) or s (it was skipped)
    assert all(dev in (s, s + 1) for s, dev in saved), saved
    assert torch.isfinite(eng.params).all() and float(eng.loss) == float(eng.loss)


def test_filter_gradient_launches_are_chunked_by_the_table_caps(monkeypatch):
    """imm_conv2d_wgrad_multi takes at most 64 jobs of 16 kernel variants per table: the engine issues a deeper
    configuration's filter gradients as several multi-problem launches instead of failing at build.  Forced here with caps of 5
    jobs / 2 variants on the ordinary model: more launches, the same gradients (other split counts: f32 sums in another order)."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import imm_amd.engine as E
    cfg, model, eng, inputs, P, St = make(2)
    monkeypatch.setattr(E, 'WGRAD_MULTI_MAX_JOBS', 5)
    monkeypatch.setattr(E, 'WGRAD_MULTI_MAX_VARIANTS', 2)
    _c, _m, eng2, _i, _P, _S = make(2)
    n1 = sum(1 for l in eng.prog_bwd if l.tag == 'conv_wgrad')
    n2 = sum(1 for l in eng2.prog_bwd if l.tag == 'conv_wgrad')
    assert n1 == 1 and n2 >= 5, (n1, n2)
    for e in (eng, eng2):
        e.set_inputs(inputs['image'].to(DEV), inputs['future_image'].to(DEV), inputs['mask'].to(DEV))
        e.forward(True); e.backward()
    torch.cuda.synchronize()
    assert float(eng.loss) == float(eng2.loss)
    g1, g2 = eng.grads, eng2.grads
    assert float((g1 - g2).norm() / g1.norm()) < 1e-5
    for name in ('model/renderer/conv_8/w', 'model/image_encoder/encoder/conv_1/w', 'model/pose_encoder/encoder/conv_5/w'):
        a, b = eng.gview[name], eng2.gview[name]
        assert float((a - b).norm() / a.norm()) < 1e-4, name


def test_train_step_argument_checks():
    """TrainStep(collective=...): unknown names and 'graph' without graph capture are refused at construction."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from imm_amd.train.cnn_train_multi import TrainStep
    cfg, model, eng, inputs, P, St = make(2)
    with pytest.raises(ValueError):
        TrainStep(model, 2, 128, collective='ring')
    with pytest.raises(ValueError):
        TrainStep(model, 2, 128, use_graph=False, collective='graph')
    cfg2, model2, eng2, _i, _P, _S = make(2, dp_buckets=2)
    tg = TrainStep(model2, 2, 128, split_graphs=True, collective='graph')      # round 5: two overlapped buckets inside the one graph
    assert tg.graph_resident and tg.buckets == 2
    tg.native_comm.destroy()
    ts = TrainStep(model, 2, 128)                       # single rank, no split: no collective at all
    assert ts.collective is None and ts.buckets == 1 and ts.native_comm is None
