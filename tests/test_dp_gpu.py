"""Data-parallel step of the GPU ENGINE against the reference's two-tower semantics
(/root/reference/imm/train/cnn_train_multi.py:66-106 average_gradients = per-variable mean over towers, THEN per-tensor
clip, one Adam apply; :155 BN statistics per tower; :173 printed loss = tower mean) — on ONE MI355X:

  * two engines (towers) on the two halves of a batch -> flat gradient buffers summed -> grad_scale 1/2 folded into
    imm_clip_adam_step -> compared with oracle.train_step([towerA, towerB]);
  * the same step through the product's multi-rank path: two PROCESSES sharing the GPU, `TrainStep(world_size=2)` =
    graph(fwd + renderer bwd) | all-reduce(bucket 0) || graph(encoder bwd) | all-reduce(bucket 1) | graph(clip+Adam),
    with a real inter-process collective (gloo, host-bounced: RCCL refuses two ranks on one device).  A two-term sum is
    commutative bit for bit and the kernels are deterministic, so both ranks must end with parameters BITWISE equal to the
    single-process emulation, and to each other.

tests/test_dp_gloo.py covers the same host logic on CPU with the oracle's gradients."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import imm_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
B_GLOBAL, S_IMG, N_STEPS = 4, 128, 2


def _model(world, buckets=None):
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    return IMMModel(Box(dict(O.default_model_config(10))), dtype=torch.bfloat16, device=DEV, world_size=world, dp_buckets=buckets)


def _towers():
    full = O.synthetic_inputs(B_GLOBAL, S_IMG, seed=7)
    per = B_GLOBAL // 2
    return full, [{k: v[i * per:(i + 1) * per] for k, v in full.items()} for i in range(2)]


def emulate_two_towers(n_steps, buckets=1):
    """Single process: engines A and B are the two towers; A carries the variables (B is refreshed from A every step, like
    TF's reuse_variables), the flat gradient buffers are added on the device, A applies the update.
    `buckets`: the engines are built the way a rank with dp_buckets=<buckets> builds them (with two buckets the renderer's
    filter gradients are issued and reduced early, with other split counts: the same sums in another order)."""
    _full, towers = _towers()
    per = B_GLOBAL // 2
    mA, mB = _model(2, buckets), _model(2, buckets)
    A, Bn = mA._get_engine(per, S_IMG), mB._get_engine(per, S_IMG)
    losses = []
    for it in range(n_steps):
        if it > 0:       # tower B reads tower A's variables; its BN moving statistics / loss normalisers stay its own (:155)
            Bn.load_parameters(A.named_parameters())
        for eng, inp in ((A, towers[0]), (Bn, towers[1])):
            eng.set_inputs(inp['image'].to(DEV), inp['future_image'].to(DEV), inp['mask'].to(DEV))
            eng.forward(True)
            eng.backward()
        torch.cuda.synchronize()
        losses.append(0.5 * (float(A.loss) + float(Bn.loss)))
        if it == 0:
            gA0, gB0 = A.grads.clone(), Bn.grads.clone()
        A.grads.add_(Bn.grads)                      # what the sum all-reduce leaves on every rank
        A.optimizer_step()                          # grad_scale = 1/2 inside imm_clip_adam_step: mean, then clip, then Adam
        torch.cuda.synchronize()
    return A, Bn, losses, gA0, gB0


@pytest.fixture(scope='module')
def emu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return emulate_two_towers(N_STEPS)


def test_two_engines_equal_the_two_tower_reference(emu):
    A, Bn, losses, gA0, gB0 = emu
    cfg = O.default_model_config(10)
    _full, towers = _towers()
    P, St = O.init_params(cfg, S_IMG)
    Pe = type(P)((k, bf(v) if k.endswith('/w') else v) for k, v in P.items())
    Se = type(St)((k, bf(v) if (k.startswith('vgg16/') and k.endswith('/weights') and 'conv1_1' not in k) else v) for k, v in St.items())
    refP, _refS, info = O.train_step(Pe, Se, O.new_adam_state(Pe), towers, cfg, clip=1.0, lr=O.learning_rate(0), act_round=bf)
    # (i) the logged loss is the tower mean (:173)
    assert abs(losses[0] - float(info['loss'])) / abs(float(info['loss'])) < 1e-3, (losses[0], float(info['loss']))
    # (ii) the averaged gradient, tensor by tensor, where the problem is well conditioned (DESIGN.md numerics)
    gsum = 0.5 * (gA0 + gB0)
    names = [n for n, _s, _w in A.spec]
    for k, lim in (('model/renderer/conv_8/w', 2e-2), ('model/renderer/conv_8/b', 5e-3), ('model/renderer/conv_7/gamma', 5e-2),
                   ('model/renderer/conv_7/beta', 5e-2)):
        i = names.index(k)
        got = gsum[A.tab.offsets[i]:A.tab.offsets[i + 1]].cpu().reshape(info['grads'][k].shape)
        ref = info['grads'][k]
        e = float((got - ref).norm() / ref.norm())
        print('DP_GRAD %-32s rel %.3g' % (k, e))
        assert e < lim, (k, e)
    # (iii) after ONE update (a fresh one-step emulation) every tensor moves in the oracle's direction
    A1, _B1, _l, _a, _b = emulate_two_towers(1)
    got = A1.named_parameters()
    worst = []
    for k, v in refP.items():
        if k.endswith('/b') and (k[:-2] + '/gamma') in refP:
            continue                     # bias in front of a BN: noise gradient in the oracle, exact zero here
        du_ref = (v - Pe[k]).flatten().double()
        du_got = (got[k].cpu() - P[k]).flatten().double()
        if float(info['grads'][k].norm()) < 1e-6:
            continue                     # pose 1x1 bias: |g| ~ 3e-8, cancellation noise in every implementation
        # the first Adam update is lr * sign(g) element by element: elements whose gradient is rounding noise (the
        # 0.01-std initialisation is ill conditioned, DESIGN.md numerics) flip sign with any change of summation order,
        # so the agreement is weighted by the oracle's |g|
        wgt = info['grads'][k].flatten().double().abs()
        cos = float((wgt * du_ref * du_got).sum() / (((wgt * du_ref * du_ref).sum() * (wgt * du_got * du_got).sum()).sqrt() + 1e-30))
        lim = 0.97 if k in ('model/renderer/conv_8/w', 'model/renderer/conv_7/gamma', 'model/renderer/conv_7/beta') else 0.5
        if cos < lim:
            worst.append((k, cos))
        assert float(du_got.abs().max()) <= 1.05e-3, k       # |lr_t m/(sqrt(v)+eps)| <= lr at t = 1
    assert not worst, worst


def _rank_main(rank, world, port, ret, buckets):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from imm_amd.train.cnn_train_multi import TrainStep, mean_tower_loss, split_inputs
    full, _t = _towers()
    mine = split_inputs(full, world, rank)
    model = _model(world, buckets)
    ts = TrainStep(model, B_GLOBAL // world, S_IMG, world_size=world, use_graph=True)
    assert ts.split and ts.buckets == buckets, (ts.split, ts.buckets)     # default 1; 2 = overlapped buckets (opt-in)
    losses = []
    for it in range(N_STEPS):
        loss = ts.step(mine if it == 0 else None)
        ts.synchronize()
        losses.append(mean_tower_loss(loss, world))
    eng = ts.engine
    ret[rank] = {'params': eng.params.cpu(), 'losses': losses, 'step': int(eng.step_count), 'adam_t': int(eng.adam_t),
                 'mm': eng.state['model/renderer/conv_1/moving_mean'].cpu(), 'agg': eng.loss_agg.cpu()}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('buckets', [1, 2], ids=['one_bucket_default', 'two_buckets_optin'])
def test_two_ranks_on_one_gpu_run_the_split_graph_path_bitwise(emu, buckets):
    import torch.multiprocessing as mp
    A, Bn, losses, _gA0, _gB0 = emu if buckets == 1 else emulate_two_towers(N_STEPS, buckets)
    port = 29700 + (os.getpid() % 200) + 300 * buckets
    with mp.Manager() as mgr:          # shut the manager process down even when an assertion below fails
        ret = mgr.dict()
        mp.spawn(_rank_main, args=(2, port, ret, buckets), nprocs=2, join=True)
        r0, r1 = dict(ret[0]), dict(ret[1])
    assert r0['step'] == N_STEPS and r0['adam_t'] == N_STEPS
    assert torch.equal(r0['params'], r1['params'])                      # replicas stay identical
    assert torch.equal(r0['params'], A.params.cpu())                    # == the single-process two-tower emulation, bitwise
    assert r0['losses'] == r1['losses']                                 # every rank logs the tower mean
    np.testing.assert_allclose(r0['losses'], losses, rtol=1e-6)
    # BN moving statistics and the loss normalisers are rank-local (per tower, :155): they differ between the ranks and
    # equal the corresponding tower of the emulation
    assert not torch.equal(r0['mm'], r1['mm'])
    assert torch.equal(r0['mm'], A.state['model/renderer/conv_1/moving_mean'].cpu())
    assert torch.equal(r1['mm'], Bn.state['model/renderer/conv_1/moving_mean'].cpu())
    assert torch.equal(r0['agg'], A.loss_agg.cpu()) and torch.equal(r1['agg'], Bn.loss_agg.cpu())


def _rank_main_f16(rank, world, port, ret):
    """BASELINE configs[4] per-rank shape (K=50, f16 storage, dynamic loss scale) at world 2; the gradient of ONE rank overflows in
    the second step (an inf planted after its backward pass, before the exchange)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep, split_inputs
    from imm_amd.utils.box import Box
    full = O.synthetic_inputs(B_GLOBAL, S_IMG, seed=11)
    mine = split_inputs(full, world, rank)
    model = IMMModel(Box(dict(O.default_model_config(50))), dtype=torch.float16, device=DEV, world_size=world)
    ts = TrainStep(model, B_GLOBAL // world, S_IMG, world_size=world, use_graph=True)
    eng = ts.engine
    assert eng.loss_scale_state is not None and float(eng.loss_scale_state[0]) == 4096.0
    sums, scales, steps = [], [], []
    for it in range(4):
        ts.after_backward = (lambda e: e.gview['model/renderer/conv_3/w'].view(-1)[17:18].fill_(float('inf'))) if (it == 1 and rank == 1) else None
        ts.step(mine if it == 0 else None)
        ts.synchronize()
        bits = eng.params.view(torch.int32).to(torch.int64)
        sums.append(int(bits.sum()))
        scales.append(float(eng.loss_scale_state[0])); steps.append((int(eng.step_count), int(eng.adam_t)))
    ret[rank] = {'params': eng.params.cpu(), 'sums': sums, 'scales': scales, 'steps': steps, 'finite': bool(torch.isfinite(eng.params).all())}
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_f16_loss_scale_overflow_on_one_rank_skips_the_step_everywhere():
    """VERDICT r4 item 4c / BASELINE configs[4] (f16 + loss scaling at world > 1, never run together before): the skip decision is
    taken on the device from the norms of the ALL-REDUCED gradient (imm_clip_adam_step), so an overflow on one rank must make every
    rank skip the same update (weights, Adam slots, global_step, Adam's t untouched), halve the scale on every rank, and leave the
    replicas bit-identical (reference: cnn_train_multi.py:66-106 averages, then clips, then applies ONE update)."""
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    port = 29950 + (os.getpid() % 40)
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_main_f16, args=(2, port, ret), nprocs=2, join=True)
        r0, r1 = dict(ret[0]), dict(ret[1])
    assert r0['finite'] and r1['finite']
    assert torch.equal(r0['params'], r1['params']) and r0['sums'] == r1['sums']          # replicas identical after every step
    assert r0['steps'] == r1['steps'] == [(1, 1), (1, 1), (2, 2), (3, 3)]                # step 2 of 4 skipped on BOTH ranks
    assert r0['scales'] == r1['scales'] == [4096.0, 2048.0, 2048.0, 2048.0]              # ... and the scale halved on both
    assert r0['sums'][1] == r0['sums'][0] and r0['sums'][2] != r0['sums'][1]             # the skipped step moved nothing


@pytest.mark.timeout(900)
def test_bench_self_spawns_two_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher in front of it: bench.py re-executes itself under torch.distributed.run
    and prints ONE JSON line (here with the gloo stand-in for RCCL and both ranks on the one GPU of the test box)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--windows', '1',
                          '--spin-seconds', '0', '--backend', 'gloo', '--share-gpu', '--no-cpu-baseline', '--no-pmc'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=800)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 64 and d['config']['parallelism'] == 'dp2'
    assert d['config']['collective']['world_size'] == 2 and d['config']['collective']['buckets'] == 1
    assert d['config']['collective']['mode'] == 'pg' and d['step']['replicas_identical'] is True
    assert d['value'] > 0 and d['scaling'] == 'weak' and np.isfinite(d['step']['loss'])


@pytest.mark.timeout(1500)
def test_bench_self_spawns_eight_ranks(tmp_path):
    """The configuration the driver's scaling run ends with — `bench.py --gpus 8`: world 8, global batch 256 (BASELINE.json
    configs[2]), grad_scale 1/8, the self-spawn — executed for real, with gloo standing in for RCCL and the eight ranks sharing the
    one GPU of the test box (8 x 3.2 GB of engine buffers).  Nothing about speed is read from it; what is checked is that eight
    ranks rendezvous, run the split-graph path, and end with bit-identical parameters after the all-reduced updates."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    if torch.cuda.mem_get_info(0)[0] < 40 * 2 ** 30:
        pytest.skip('needs ~30 GB of free HBM for eight co-resident engines')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    d = _eight_rank_bench(env, [])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 256 and d['config']['parallelism'] == 'dp8'
    c = d['config']['collective']
    assert c['world_size'] == 8 and c['buckets'] == 1 and c['mode'] == 'pg' and c['backend'] == 'gloo'
    assert d['value'] > 0 and d['scaling'] == 'weak' and np.isfinite(d['step']['loss'])
    assert d['step']['replicas_identical'] is True, d['step'].get('replicas_diff')


def _eight_rank_bench(env, extra):
    """`bench.py --gpus 8 --backend gloo --share-gpu` (+ extra flags); returns the parsed line.
    EIGHT processes time-sharing one GPU is a situation only these tests create (the product runs one process per GPU).  Until the
    end of round 5 about one execution in ten of this file ended with ONE rank holding other bits in a few of the largest tensors:
    its INITIAL weights — per-tensor `copy_` from pageable temporaries, torn by the runtime's pin-in-place path for > 1 MB sources
    under eight-fold contention (found with a cross-rank checksum at the first exchange; fixed by ops.upload: pinned staging,
    stream-synchronised, read back).  Round 6: the repetition the harness kept is gone: the FIRST
    run whose replicas differ fails the calling test, and the line names the differing tensors / ranks (step.replicas_diff)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '1', '--windows', '1',
                          '--spin-seconds', '0', '--backend', 'gloo', '--share-gpu', '--no-cpu-baseline', '--no-pmc', '--collective', 'pg']
                         + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1400)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    if d['step']['replicas_identical'] is not True:
        print('the eight replicas sharing one GPU differ: step.replicas_diff = %r' % (d['step'].get('replicas_diff'),))
    return d


@pytest.mark.timeout(1500)
def test_bench_config4_eight_ranks_f16_loss_scale(tmp_path):
    """`bench.py --config 4 --gpus 8` — BASELINE configs[4] as stated (K=50, f16 storage + dynamic loss scale, 8 ranks), with gloo
    standing in for RCCL and the ranks sharing the one GPU: the f16 engine, its loss scale and the multi-rank path run TOGETHER,
    the line names the configuration, and the replicas end bit-identical."""
    import json
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    if torch.cuda.mem_get_info(0)[0] < 40 * 2 ** 30:
        pytest.skip('needs ~30 GB of free HBM for eight co-resident engines')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    d = _eight_rank_bench(env, ['--config', '4'])
    assert d['n_gpus'] == 8 and d['dtype'] == 'f16' and d['config']['n_maps'] == 50 and d['config']['baseline_config'] == 4
    assert d['config']['global_batch'] == 256 and 'configs[4]' in d['config']['workload'] and 'K=50' in d['metric']
    assert d['config']['loss_scale'] is not None and d['config']['loss_scale'][0] >= 1.0
    assert d['step']['replicas_identical'] is True and np.isfinite(d['step']['loss']), d['step'].get('replicas_diff')


@pytest.mark.timeout(600)
def test_graph_replay_is_deterministic_while_another_process_shares_the_gpu():
    """Round 5: two processes replaying the step on ONE GPU (what every two-rank test here does) used to corrupt ~0.3 % of the
    replays — 16-byte outputs of conv_halo2.hip's ci = 64 instantiations, whose inline-asm buffer_store_dwordx4 lacked the wait
    states a > 64-bit VMEM store needs before its data registers are rewritten (alone on the GPU the store always won the race).
    tools/det_graph.py compares every activation, the loss and every gradient of each replay with the process's first one."""
    import subprocess
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    env = dict(os.environ, DET_SECONDS='10')
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tools', 'det_graph.py'), t], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT) for t in ('A', 'B')]
    outs = [p.communicate(timeout=500)[0].decode() for p in ps]
    for p, o in zip(ps, outs):
        assert p.returncode == 0, o[-2000:]
        line = [l for l in o.splitlines() if l.startswith('DETGRAPH')][-1]
        runs, bad = int(line.split('runs')[1].split()[0]), int(line.split('mismatching')[1].split()[0])
        assert runs > 500 and bad == 0, line
