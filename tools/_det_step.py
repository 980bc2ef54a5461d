import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import imm_oracle as O
from imm_amd.models.imm_model import IMMModel
from imm_amd.utils.box import Box
torch.cuda.set_device(0)
B = int(os.environ.get('DET_B', '2'))
model = IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device='cuda:0', world_size=2)
eng = model._get_engine(B, 128)
inp = O.synthetic_inputs(B, 128, seed=7)
eng.set_inputs(inp['image'].cuda(), inp['future_image'].cuda(), inp['mask'].cuda())
side = torch.cuda.Stream(); junk = torch.randn(6144, 6144, device='cuda:0')
def tensors():
    d = {'grads': eng.grads.clone(), 'loss': eng.loss_out.clone()}
    for nm, lays in (('enc_im', eng.enc_im), ('enc_pose', eng.enc_pose), ('ren', eng.ren)):
        for i, l in enumerate(lays):
            d['%s%d.y' % (nm, i + 1)] = l.y.clone()
            if getattr(l, 'stats', None) is not None: d['%s%d.stats' % (nm, i + 1)] = l.stats.clone()
    return d
ref = None; bad = {}
snap = eng.snapshot()
for r in range(int(os.environ.get('DET_REPS', '60'))):
    eng.restore(snap)
    if r % 2 == 1:
        with torch.cuda.stream(side):
            for _ in range(3): j2 = junk @ junk
    eng.forward(True); eng.backward(); torch.cuda.synchronize()
    t = tensors()
    if ref is None: ref = t
    else:
        for k, v in t.items():
            if not torch.equal(v.view(torch.uint8) if v.dtype != torch.float32 else v, ref[k].view(torch.uint8) if v.dtype != torch.float32 else ref[k]):
                bad[k] = bad.get(k, 0) + 1
print('DETSTEP B', B, 'mismatches', bad)
if 'grads' in bad:
    names = [n for n, _s, _w in eng.spec]
    eng.restore(snap); eng.forward(True); eng.backward(); torch.cuda.synchronize(); g0 = eng.grads.clone()
    diff = {}
    for r in range(40):
        eng.restore(snap)
        with torch.cuda.stream(side):
            for _ in range(3): j2 = junk @ junk
        eng.forward(True); eng.backward(); torch.cuda.synchronize()
        for i, n in enumerate(names):
            a, b = eng.grads[eng.tab.offsets[i]:eng.tab.offsets[i + 1]], g0[eng.tab.offsets[i]:eng.tab.offsets[i + 1]]
            if not torch.equal(a, b): diff[n] = diff.get(n, 0) + 1
    print('DETSTEP grads differing by tensor', diff)
