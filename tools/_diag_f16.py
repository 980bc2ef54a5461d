import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import imm_oracle as O
from imm_amd.models.imm_model import IMMModel
from imm_amd.utils.box import Box
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'imm_step_golden.npz'), allow_pickle=False)
for K, B in ((10, 2), (30, 1)):
    cfg = O.default_model_config(K)
    for dt in (torch.float16, torch.bfloat16):
        model = IMMModel(Box(dict(cfg)), dtype=dt, device='cuda:0')
        inp = O.synthetic_inputs(B, 128, seed=0)
        _, loss, _, tens = model.build(inp, True, output_tensors=True)
        eng = model.engine
        eng.backward(); torch.cuda.synchronize()
        t = 'k%d_b%d' % (K, B)
        got = tens['future_im_pred'].float().cpu().numpy()[:, ::16, ::16, :]; ref = G[t + '/pred_sample']
        names = [n for n, _s, _w in eng.spec]
        gv = eng.named_gradients()
        rec = {'mu': float(np.abs(tens['gauss_yx'].cpu().numpy() - G[t + '/gauss_yx']).max()),
               'loss_rel': abs(float(loss) - float(G[t + '/loss'])) / float(G[t + '/loss']),
               'terms_rel': float(np.max(np.abs(eng.loss_terms.cpu().numpy() - G[t + '/loss_terms']) / np.abs(G[t + '/loss_terms']))),
               'recon': float(np.linalg.norm(got - ref) / np.linalg.norm(ref)),
               'agg_rel': float(np.max(np.abs(eng.loss_agg.cpu().numpy() - G[t + '/agg_after_step']) / np.abs(G[t + '/agg_after_step'])))}
        if K == 10:
            gn = G[t + '/grad_norms']
            for k in ('model/renderer/conv_8/w', 'model/renderer/conv_8/b', 'model/renderer/conv_7/gamma', 'model/renderer/conv_7/w',
                      'model/renderer/conv_1/w', 'model/image_encoder/encoder/conv_8/w', 'model/pose_encoder/encoder/conv_8/w'):
                rec['gn ' + k] = abs(float(gv[k].double().norm()) / gn[names.index(k)] - 1.0)
        print('GOLDEN', t, str(dt), json.dumps(rec))
# one step: update cosines against the fp32 oracle (no storage emulation)
cfg = O.default_model_config(10)
inputs = O.synthetic_inputs(2, 128)
P, St = O.init_params(cfg, 128)
newP, newS, info = O.train_step(P, St, O.new_adam_state(P), [inputs], cfg, clip=1.0, lr=O.learning_rate(0))
for dt in (torch.float16, torch.bfloat16):
    model = IMMModel(Box(dict(cfg)), dtype=dt, device='cuda:0')
    model.build(inputs, True)
    eng = model.engine
    eng.backward(); eng.optimizer_step(); torch.cuda.synchronize()
    got = eng.named_parameters()
    cs = {}
    for k, v in newP.items():
        if k.endswith('/b'):
            continue
        du_ref = (v - P[k]).detach().flatten().double(); du_got = (got[k].cpu() - P[k]).flatten().double()
        cs[k] = float((du_ref * du_got).sum() / (du_ref.norm() * du_got.norm() + 1e-30))
    worst = sorted(cs.items(), key=lambda kv: kv[1])[:12]
    print('STEPCOS', str(dt), 'min %.4f median %.4f' % (min(cs.values()), float(np.median(list(cs.values())))), worst)
