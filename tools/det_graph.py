"""Determinism of a graph-replayed forward + backward pass while ANOTHER process shares the GPU (round 5: how the missing wait
states behind conv_halo2.hip's inline-asm 16-byte stores were found — bit-identical alone, 0.3 % of the replays corrupted when
two processes ran side by side).  Run two copies at once on one GPU:

    python tools/det_graph.py A & python tools/det_graph.py B; wait

Each replays the step's first graph from the same state for DET_SECONDS (default 25) and compares every activation, the loss
and every gradient tensor bitwise with its own first replay: `mismatching 0` is the expected output.
DET_B = per-process batch (default 2), DET_K = landmarks (10), DET_DTYPE = bf16 | f16, DET_FULL=1 replays the optimizer graph as
well and compares the parameters, both Adam moments and the per-tensor clip norms (the eight-rank observation of DESIGN §7:
`for i in 0 1 2 3 4 5 6 7; do DET_B=4 DET_FULL=1 python tools/det_graph.py r$i & done; wait`)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from imm_amd.models.imm_model import IMMModel
from imm_amd.train.cnn_train_multi import TrainStep
torch.cuda.set_device(0)
tag = sys.argv[1] if len(sys.argv) > 1 else 'p'
B = int(os.environ.get('DET_B', '2'))
FULL = os.environ.get('DET_FULL', '0') != '0'
DT = torch.float16 if os.environ.get('DET_DTYPE', 'bf16') == 'f16' else torch.bfloat16
model = IMMModel(bench.model_config(int(os.environ.get('DET_K', '10'))), dtype=DT, device='cuda:0')
ts = TrainStep(model, B, 128, world_size=1, use_graph=True, split_graphs=True, collective='pg')
eng = ts.engine
inp = bench.synthetic_batch(B, 128, seed=7, device='cuda:0')
ts.step(inp); ts.synchronize()
snap = eng.snapshot()
torch.cuda.synchronize()              # (the snapshot's clones run on the default stream, the replays on ts.stream)
names = [n for n, _s, _w in eng.spec]
ref = None; bad = 0; diff = {}; adiff = {}
t_end = time.time() + float(os.environ.get('DET_SECONDS', '25'))
n = 0
while time.time() < t_end:
    with torch.cuda.stream(ts.stream):
        eng.restore(snap)
        ts._graphs[0].launch()
        if FULL: ts._graphs[1].launch()
    ts.synchronize(); n += 1
    g = eng.grads.clone(); l = eng.loss_out.clone()
    acts = {}
    for nm, lays in (('enc_im', eng.enc_im), ('enc_pose', eng.enc_pose), ('ren', eng.ren)):
        for i, ly in enumerate(lays):
            acts['%s%d.y' % (nm, i + 1)] = ly.y
            if getattr(ly, 'stats', None) is not None: acts['%s%d.stats' % (nm, i + 1)] = ly.stats
            if getattr(ly, 'out', None) is not None: acts['%s%d.out' % (nm, i + 1)] = ly.out
            if getattr(ly, 'scale', None) is not None: acts['%s%d.scale' % (nm, i + 1)] = ly.scale
    if FULL:
        acts['opt.params'] = eng.params; acts['opt.adam_m'] = eng.adam_m; acts['opt.adam_v'] = eng.adam_v
        acts['opt.seg_norm2'] = eng.seg_norm2; acts['opt.blk_partial'] = eng.opt_blk_partial
    acts['joint'] = eng.joint; acts['mu'] = eng.mu; acts['heat'] = eng.heat; acts['wd'] = eng.wd_loss
    for k_, (y_, _h) in eng.vgg_activations().items(): acts['vgg.' + k_] = y_
    if ref is None:
        ref = (g, l); aref = {k: v.clone() for k, v in acts.items()}
        torch.cuda.synchronize()      # the clones run on the default stream: they must not overlap the next replay on ts.stream
    elif not (torch.equal(g, ref[0]) and torch.equal(l, ref[1]) and
              (not FULL or all(torch.equal(acts[k], aref[k]) for k in ('opt.params', 'opt.adam_m', 'opt.adam_v')))):
        bad += 1
        for i, nm in enumerate(names):
            if not torch.equal(g[eng.tab.offsets[i]:eng.tab.offsets[i + 1]], ref[0][eng.tab.offsets[i]:eng.tab.offsets[i + 1]]):
                diff[nm] = diff.get(nm, 0) + 1
        if not torch.equal(l, ref[1]): diff['LOSS'] = diff.get('LOSS', 0) + 1
        if FULL:
            for i, nm in enumerate(names):
                sl = slice(eng.tab.offsets[i], eng.tab.offsets[i + 1])
                if not torch.equal(acts['opt.params'][sl], aref['opt.params'][sl]): diff['PARAM ' + nm] = diff.get('PARAM ' + nm, 0) + 1
        for k, v in acts.items():
            a_, b_ = (v, aref[k]) if v.dtype == torch.float32 else (v.view(torch.int16), aref[k].view(torch.int16))
            if not torch.equal(a_, b_): adiff[k] = adiff.get(k, 0) + 1
print('DETACTS', tag, adiff)
print('DETGRAPH', tag, 'runs', n, 'mismatching', bad, 'tensors', dict(sorted(diff.items(), key=lambda kv: -kv[1])[:12]), 'n_tensors_diff', len(diff))
