"""bench.py — IMM conditional-generation training step on MI355X: training images/sec at 128x128, K=10.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under
                                                            torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --config {1,3,4} [--gpus N]            the other BASELINE.json configurations, same one-line contract

One process per GPU; batch 32 per GPU (BASELINE.json configs[1]; N>1 = configs[2] weak scaling); synthetic
inputs resident in HBM; a step = forward (both encoders, landmark bottleneck, renderer, VGG16 perceptual
loss) + backward + RCCL gradient all-reduce + per-tensor clip + Adam.  Rank 0 prints ONE JSON line.
--config 3 = configs[3] (K=30 at 256x256, bf16, batch 16 per GPU), --config 4 = configs[4] (K=50 at 128x128, f16 storage with
the device-resident dynamic loss scale, batch 32 per GPU; --gpus 8 is the configuration as BASELINE.json states it).

Timing protocol: the step is first spun for a fixed wall time (default 1 s, untimed) so that the clocks have
settled, then W untimed warm-up steps, then `--windows` (default 3) windows of EXACTLY K steps, each bracketed by
barrier + torch.cuda.synchronize() on both sides and reduced with MAX over ranks; `ms_per_step` / `value` are the
MEDIAN window, all windows are listed under `step.windows_ms`.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 / f16 MFMA peak (MI355X_MICROARCH.md)
# BASELINE.json configurations this file can run (configs[0] is the CPU-only plumbing case, configs[2] = --config 1 --gpus 8)
WORKLOADS = {
    1: dict(batch=32, size=128, n_maps=10, dtype='bf16', name='BASELINE configs[1] (N>1: configs[2])',
            what='CelebA-shape 128x128 K=10 IMM training step'),
    3: dict(batch=16, size=256, n_maps=30, dtype='bf16', name='BASELINE configs[3]',
            what='K=30 at 256x256 IMM training step (align-corners resize path, wider bottleneck)'),
    4: dict(batch=32, size=128, n_maps=50, dtype='f16', name='BASELINE configs[4] (as stated at --gpus 8)',
            what='AFLW-finetune shape 128x128 K=50 IMM training step, f16 storage + dynamic loss scale'),
}
WL = dict(WORKLOADS[1])        # the selected workload (set by --config before anything runs)
BATCH_PER_GPU = 32
IMAGE_SIZE = 128
N_MAPS = 10


def select_workload(idx):
    global BATCH_PER_GPU, IMAGE_SIZE, N_MAPS
    WL.clear(); WL.update(WORKLOADS[idx]); WL['index'] = idx
    BATCH_PER_GPU, IMAGE_SIZE, N_MAPS = WL['batch'], WL['size'], WL['n_maps']


def torch_dtype(name=None):
    return torch.float16 if (name or WL['dtype']) == 'f16' else torch.bfloat16


IGEMM_TAGS = ('conv_fwd', 'vgg_fwd', 'conv_dgrad', 'vgg_dgrad')    # forward + data-gradient convolution launches
CONV_FAMILY = ('conv_igemm', 'conv_halo', 'conv_hdeep')            # kernel-name prefixes of those launches (not wgrad)


def model_config(n_maps):
    from imm_amd.utils.box import Box
    return Box(dict(gauss_std=0.10, gauss_mode='rot', n_maps=n_maps, n_filters=32, block_sizes=[1, 1, 1],
                    n_filters_render=32, renderer_stride=2, min_res=16, same_n_filt=False,
                    reconstruction_loss='perceptual',
                    perceptual=dict(l2=True, comp=['input', 'conv1_2', 'conv2_2', 'conv3_2', 'conv4_2', 'conv5_2'],
                                    net_file='synthetic'),
                    loss_mask=True, confidence=False, channels_bug_fix=True))


def smooth_mask(h, w, margin=10, step=20, b=0.4):
    """imm/datasets/tps_dataset.py:47-67."""
    def smooth_step(n, bb):
        return 0.5 + 0.5 * torch.tanh(torch.linspace(-1.0, 1.0, n) / bb)

    def strip(size):
        return torch.cat([torch.zeros(margin), smooth_step(step, b), torch.ones(size - 2 * margin - 2 * step),
                          smooth_step(step, -b), torch.zeros(margin)])
    return strip(h)[:, None] * strip(w)[None]


def synthetic_batch(batch, size, seed, device):
    g = torch.Generator().manual_seed(seed)
    im = torch.rand(batch, size, size, 3, generator=g) * 255.0
    fut = torch.rand(batch, size, size, 3, generator=g) * 255.0
    mask = smooth_mask(size, size).reshape(1, size, size, 1).repeat(batch, 1, 1, 1)
    def dev(t):
        # through pinned memory, stream-synchronised before the host tensor may go (imm_amd/ops.py upload: the runtime's path for
        # pageable sources of more than ~1 MB tore one upload in ~100 with eight processes on one GPU)
        if str(device) == 'cpu':
            return t.contiguous()
        d = t.contiguous().pin_memory().to(device, non_blocking=True)
        torch.cuda.current_stream(d.device).synchronize()
        return d
    return {'image': dev(im), 'future_image': dev(fut), 'mask': dev(mask)}


def forward_only_rates(dev, batches, steps=20):
    """images/s of TrainStep.forward_only (forward + perceptual loss as one graph replay; the reference's `fwd_only`,
    cnn_train_multi.py:447-449) at the given per-GPU batches: {'b<batch>': {'ms': .., 'images_per_s': ..}}."""
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    out = {}
    for b in batches:
        model = IMMModel(model_config(N_MAPS), dtype=torch_dtype(), device=dev)      # (its own model: the timed one is not touched)
        ts_f = TrainStep(model, b, IMAGE_SIZE, world_size=1, use_graph=True)
        inp = synthetic_batch(b, IMAGE_SIZE, seed=0, device=dev)
        ts_f.engine.set_inputs(inp['image'], inp['future_image'], inp['mask'])
        for _ in range(5):
            ts_f.forward_only(None)
        ts_f.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts_f.forward_only(None)
        ts_f.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out['b%d' % b] = {'ms': round(dt * 1e3, 4), 'images_per_s': round(b / dt, 1)}
    return out


# ---------------------------------------------------------------------------------------------------------
# HBM traffic of the dominant kernel family (rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE)
# ---------------------------------------------------------------------------------------------------------
def _pmc_rows(root, counter):
    import csv
    import glob
    for p in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        with open(p, newline='') as f:
            for r in csv.DictReader(f):
                if r.get('Counter_Name') == counter:
                    yield r['Kernel_Name'], float(r['Counter_Value'])


def pmc_traffic_live(timeout_s=240):
    """Runs this file's --pmc-pass (3 eager training steps, same workload) under `rocprofv3 --kernel-trace --pmc X`, once
    for FETCH_SIZE and once for WRITE_SIZE (the two do not fit one pass; no other tracing domain is enabled), and returns
    (MB per launch of the conv forward / data-gradient family, total GB per step, dispatch count) with the gfx950
    correction FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM section).  None when rocprofv3 is absent or a pass fails."""
    exe = shutil.which('rocprofv3')
    if exe is None:
        return None
    tmp = tempfile.mkdtemp(prefix='imm_pmc_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    steps = 3
    try:
        sums = {}
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(tmp, ctr)
            cmd = [exe, '--kernel-trace', '--output-format', 'csv', '--pmc', ctr, '-d', d, '--', sys.executable,
                   os.path.abspath(__file__), '--pmc-pass', '--steps', str(steps), '--config', str(WL.get('index', 1))]
            r = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                sys.stderr.write('bench: rocprofv3 %s pass failed (rc %d): %s\n' % (ctr, r.returncode, r.stderr.decode()[-400:]))
                return None
            fam_kb = tot_kb = 0.0
            fam_n = 0
            for name, val in _pmc_rows(d, ctr):
                if 'at::native' in name or 'rocclr' in name:
                    continue                         # torch's zero-fills / copies while the engine is being built: not the step
                tot_kb += val
                if any(k in name for k in CONV_FAMILY) and 'wgrad' not in name:
                    fam_kb += val; fam_n += 1
            if fam_n == 0:
                return None
            sums[ctr] = (fam_kb, fam_n, tot_kb)
        f, w = sums['FETCH_SIZE'], sums['WRITE_SIZE']
        mb_per_launch = (2.0 * f[0] / f[1] + w[0] / w[1]) / 1024.0
        gb_per_step = (2.0 * f[2] + w[2]) / 1024.0 / 1024.0 / steps
        return round(mb_per_launch, 2), round(gb_per_step, 3), f[1] // steps
    except Exception as e:      # a profiler hiccup must not lose the bench line
        sys.stderr.write('bench: live PMC traffic unavailable (%s)\n' % (e,))
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic_from_profiles():
    """Fallback: the newest committed PMC summary (profiles/r*_pmc_hbm_bytes.csv)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_bytes.csv')))
    if not files:
        return None, None
    n = mb = 0.0
    with open(files[-1]) as f:
        for row in csv.reader(f):
            if len(row) == 7 and any(k in row[0] for k in CONV_FAMILY) and 'wgrad' not in row[0]:
                n += float(row[1]); mb += float(row[1]) * (float(row[4]) + float(row[5]))
    return (round(mb / n, 2) if n else None), os.path.relpath(files[-1], ROOT)


def pmc_pass(steps):
    """Child of pmc_traffic_live: the same training step, eager (one dispatch per launch), no timing."""
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    torch.cuda.set_device(0)
    model = IMMModel(model_config(N_MAPS), dtype=torch_dtype(), device='cuda:0')
    ts = TrainStep(model, BATCH_PER_GPU, IMAGE_SIZE, world_size=1, use_graph=False)
    inputs = synthetic_batch(BATCH_PER_GPU, IMAGE_SIZE, seed=0, device='cuda:0')
    ts.engine.set_inputs(inputs['image'], inputs['future_image'], inputs['mask'])
    torch.cuda.synchronize()
    for _ in range(steps):
        ts.step(None)
    ts.synchronize()


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = "port"), bounded sample
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(budget_s=12.0):
    """CPU restatement of the TF1 graph (oracle/imm_oracle.py) timed on this node's host cores, bounded so
    that the default bench run stays within minutes, at the thread count the oracle runs FASTEST with (torch's CPU
    convolutions scale badly: "all cores" was the slowest point of the curve on the 128-core host of round 2):
      A  configs[0]: forward + perceptual loss, batch 4, swept over {1, 8, 16, 32, 64, all} threads
         (1 warm + 2 timed each, the faster one counts) -> best thread count
      B  configs[1]: training step (fwd + bwd + clip + Adam), batch 32, at that thread count: one untimed batch-4 step
         (autograd / allocator / thread-pool warm-up), then 3 timed steps, median            -> `value`
    `python bench.py --cpu-baseline-full` runs the unbounded protocol of BASELINE.md §3 (>= 3 warm + >= 10 timed per leg)."""
    from oracle import imm_oracle as O
    full = budget_s <= 0
    cfg = O.default_model_config(N_MAPS)
    P, S = O.init_params(cfg, IMAGE_SIZE)
    n_all = torch.get_num_threads()
    cores = os.cpu_count() or n_all

    def median(v):
        v = sorted(v)
        return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])

    def timed(fn, warm, n):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.time(); fn(); ts.append(time.time() - t0)
        return ts

    legs, sweep = {}, {}
    inp4 = O.synthetic_inputs(4, IMAGE_SIZE)
    counts = sorted(set(c for c in (1, 8, 16, 32, 64, n_all) if c <= n_all))
    try:
        for c in counts:
            torch.set_num_threads(c)
            with torch.no_grad():
                t = timed(lambda: O.forward(P, S, inp4, cfg, training=True), 3 if full else 1, 10 if full else 2)
            sweep[c] = round(4 / (median(t) if full else min(t)), 3)
        best = max(sweep, key=lambda c: sweep[c])
        legs['fwd_loss_b4_by_threads'] = {'images_per_s': sweep, 'best_threads': best}
        torch.set_num_threads(best)
        inp32 = O.synthetic_inputs(BATCH_PER_GPU, IMAGE_SIZE)
        opt = O.new_adam_state(P)
        O.train_step(P, S, O.new_adam_state(P), [inp4], cfg)            # untimed warm-up of the backward / optimizer code paths
        tB = timed(lambda: O.train_step(P, S, opt, [inp32], cfg), 3 if full else 0, 10 if full else (3 if IMAGE_SIZE <= 128 else 2))
        legs['train_step_b32'] = {'images_per_s': round(BATCH_PER_GPU / median(tB), 3), 'timed': len(tB), 'threads': best,
                                  'step_s': [round(x, 2) for x in tB]}
        if full and best != n_all:
            torch.set_num_threads(n_all)
            tD = timed(lambda: O.train_step(P, S, opt, [inp32], cfg), 1, 3)
            legs['train_step_b32_all_cores'] = {'images_per_s': round(BATCH_PER_GPU / median(tD), 3), 'timed': len(tD), 'threads': n_all}
    finally:
        torch.set_num_threads(n_all)
    return {'value': legs['train_step_b32']['images_per_s'], 'unit': 'images/s', 'cores': cores, 'threads': best, 'kind': 'port',
            'sample': '%d timed fp32 training steps (median) of batch %d at %dx%d K=%d on %d of %d host threads — the thread count '
                      'at which the batch-4 forward+loss leg ran fastest in a sweep over %s (torch-CPU restatement of the TF1 graph, '
                      'oracle/imm_oracle.py; TF 1.10 itself is not installable here)' % (len(tB), BATCH_PER_GPU, IMAGE_SIZE, IMAGE_SIZE,
                                                                                          N_MAPS, best, n_all, counts),
            'legs': legs}


def parity_vs_oracle(dev, batch=2):
    """The second half of BASELINE.json's metric ("landmark MSE vs TF1 ref"): one training-mode forward pass of the HIP path on a
    seeded batch of `batch` images of the workload's size and K against the CPU restatement of the TF1 graph (oracle/imm_oracle.py —
    TF 1.10 itself cannot run here) on the same inputs and the same seeded initial weights, for BOTH storage types:
    bf16 (the headline engine) and f16 (8x finer storage rounding: the tight witness — bounds recon 0.02, six terms 2e-3).  Part of
    the cpu_baseline leg (the oracle is the checker, never the thing measured); ~1 s of CPU work at 128x128."""
    from oracle import imm_oracle as O
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.utils.box import Box
    cfg = O.default_model_config(N_MAPS)
    inputs = O.synthetic_inputs(batch, IMAGE_SIZE, seed=0)
    P, S = O.init_params(cfg, IMAGE_SIZE)
    with torch.no_grad():
        ref = O.forward(P, S, inputs, cfg, training=True)
    rp = ref['future_im_pred']
    rt = torch.as_tensor([float(t) for t in ref['loss_terms']], dtype=torch.float64)

    def one(dt):
        model = IMMModel(Box(dict(cfg)), dtype=dt, device=dev)
        _, loss, _, tensors = model.build(inputs, True, output_tensors=True)
        torch.cuda.synchronize()
        d = tensors['gauss_yx'].detach().float().cpu() - ref['gauss_yx']
        pred = tensors['future_im_pred'].detach().float().cpu()
        terms = torch.as_tensor([float(t) for t in model.engine.loss_terms], dtype=torch.float64)
        return {'landmark_mse': float((d * d).mean()), 'mu_max_abs': float(d.abs().max()),
                'loss_rel': abs(float(loss) - float(ref['loss'])) / abs(float(ref['loss'])),
                'terms_max_rel': float(((terms - rt).abs() / rt.abs()).max()),
                'recon_rel_l2': float((pred - rp).norm() / rp.norm())}
    out = {'reference': 'oracle/imm_oracle.py (fp32 torch-CPU restatement of the TF1 graph; parity unpinned against TF itself)',
           'batch': batch, 'image_size': IMAGE_SIZE, 'n_maps': N_MAPS, 'dtype': 'bf16 storage vs fp32'}
    out.update(one(torch.bfloat16))
    # reconstruction: bf16 storage drift through 16 conv + BN blocks (18 at 256x256): measured 0.080 / 0.106
    out['bounds'] = {'mu_max_abs': 1e-3, 'loss_rel': 1e-3, 'terms_max_rel': 1e-2, 'recon_rel_l2': 0.10 if IMAGE_SIZE <= 128 else 0.13}
    out['f16'] = dict(one(torch.float16), dtype='f16 storage vs fp32',
                      bounds={'mu_max_abs': 1e-3, 'loss_rel': 1e-3, 'terms_max_rel': 2e-3, 'recon_rel_l2': 0.02})
    # the f32-storage WITNESS engine (round 6: the same launch program, plain f32 convolutions — a test instrument): what the
    # two 16-bit rows leave as storage noise it settles in exact arithmetic (every gradient tensor: tests/test_witness_gpu.py)
    if IMAGE_SIZE <= 128:
        out['f32_witness'] = dict(one(torch.float32), dtype='f32 storage (imm_amd/csrc/conv_f32.hip) vs fp32',
                                  bounds={'mu_max_abs': 1e-5, 'loss_rel': 1e-5, 'terms_max_rel': 1e-4, 'recon_rel_l2': 1e-4})
    out['within_bounds'] = all(out[k] <= v for k, v in out['bounds'].items()) and \
        all(out[row][k] <= v for row in ('f16', 'f32_witness') if row in out for k, v in out[row]['bounds'].items())
    return out


def collective_selfcheck(dev, world, inputs, buckets):
    """N > 1 default by construction: the candidate — the whole step as ONE HIP graph whose nodes include the
    all-reduce of `buckets` gradient buckets (the renderer's travelling on a forked stream while the encoders' backward nodes run) —
    is compared at start-up with the simplest correct sequence there is: eager, one stream, fwd, bwd, ONE torch.distributed
    all-reduce of the whole flat buffer, clip + Adam.  Same seeds, same inputs, two steps; the two paths use different
    communicators (the summation order over > 2 ranks may differ), so the UPDATES are compared to 1e-3 relative (a mis-ordered
    exchange — stale, partial or un-reduced gradients — moves an Adam update by O(1)); the candidate's replicas must be
    bit-identical across ranks.  Every rank must pass.  Runs in a CHILD process per rank (selfcheck_in_children): the candidate
    has only ever met a one-rank communicator on the build boxes, and a hang there must not take the measurement with it.
    Returns the record for the bench line."""
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    rec = {'candidate': 'graph, %d bucket(s)' % buckets, 'reference': 'eager one-stream step, one torch.distributed all-reduce',
           'steps': 2, 'bound_update_rel': 1e-3}
    try:
        mc = IMMModel(model_config(N_MAPS), dtype=torch_dtype(), device=dev, world_size=world, dp_buckets=buckets)
        tc = TrainStep(mc, BATCH_PER_GPU, IMAGE_SIZE, world_size=world, use_graph=True, split_graphs=True, collective='graph')
        mr = IMMModel(model_config(N_MAPS), dtype=torch_dtype(), device=dev, world_size=world, dp_buckets=buckets)
        tr = TrainStep(mr, BATCH_PER_GPU, IMAGE_SIZE, world_size=world, use_graph=False, split_graphs=True, collective='pg')
        p0 = tc.engine.params.clone()
        for t in (tc, tr):
            t.step(inputs); t.step(None); t.synchronize()
        uc, ur = tc.engine.params - p0, tr.engine.params - p0
        rel = float((uc - ur).norm() / ur.norm().clamp_min(1e-30))
        bits = tc.engine.params.view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=bits.device) % 8191 + 1)).sum()])
        got = [torch.zeros_like(chk) for _ in range(dist.get_world_size())]
        dist.all_gather(got, chk)
        same = all(bool(torch.equal(g, got[0])) for g in got)
        ok = bool(torch.isfinite(uc).all()) and rel < 1e-3 and same
        flag = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        rec.update(update_rel_diff=rel, replicas_identical=same, passed=bool(int(flag)))
    except Exception as e:          # a candidate that cannot even be built falls back like one that fails
        rec.update(passed=False, error='%s: %s' % (type(e).__name__, e))
    return rec


def selfcheck_child(args):
    """`bench.py --selfcheck-child`: one rank of the self-check's own process group (MASTER_PORT of the run + 7)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    dist.init_process_group('nccl', device_id=torch.device(dev))
    rec = collective_selfcheck(dev, world, synthetic_batch(BATCH_PER_GPU, IMAGE_SIZE, seed=rank, device=dev), args.buckets or 2)
    sys.stdout.write('SELFCHECK ' + json.dumps(rec) + '\n'); sys.stdout.flush()
    dist.barrier()
    dist.destroy_process_group()
    return 0 if rec.get('passed') else 3


def selfcheck_in_children(args, dev, world, ctrl, timeout_s=240.0):
    """Every rank runs the self-check in a child process (same GPU, a process group of the children's own) and waits for it with a
    time limit; a child that hangs, crashes or fails makes EVERY rank fall back (MIN over ranks on the parents' group)."""
    # the children rendezvous among themselves: their own TCP store on MASTER_PORT + 7, not the launcher's agent store
    env = {k: v for k, v in os.environ.items() if not k.startswith('TORCHELASTIC_')}
    env['MASTER_PORT'] = str(int(os.environ.get('MASTER_PORT', '29511')) + 7)
    cmd = [sys.executable, os.path.abspath(__file__), '--selfcheck-child', '--config', str(WL.get('index', 1)), '--gpus', str(world),
           '--buckets', str(args.buckets or 2)]
    rec = {'candidate': 'graph, %d bucket(s)' % (args.buckets or 2), 'passed': False, 'isolated': 'child process per rank, %.0f s limit' % timeout_s}
    t0 = time.time()
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        for line in r.stdout.decode().splitlines():
            if line.startswith('SELFCHECK '):
                rec.update(json.loads(line[10:]))
        if r.returncode not in (0, 3):
            rec.update(passed=False, error='child exit code %d: %s' % (r.returncode, r.stderr.decode()[-300:]))
    except subprocess.TimeoutExpired:
        rec.update(passed=False, error='child timed out after %.0f s' % timeout_s)
    rec['seconds'] = round(time.time() - t0, 1)
    flag = torch.tensor([1 if rec.get('passed') else 0], dtype=torch.int64, device=ctrl)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    rec['passed'] = bool(int(flag))
    return rec


# ---------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_spawn(args, argv):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU; the
    children's stdout is ours, so rank 0's JSON line is the only line printed."""
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.share_gpu:
        sys.stderr.write('bench.py: --gpus %d but only %d device(s) visible\n' % (args.gpus, n_dev))
        return 2
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--config', type=int, choices=sorted(WORKLOADS), default=1,
                    help='BASELINE.json configuration: 1 = configs[1] (the headline metric; --gpus 8 = configs[2]), 3 = configs[3] '
                         '(K=30 at 256x256), 4 = configs[4] (K=50, f16 + loss scale; --gpus 8 as stated)')
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--windows', type=int, default=3, help='timed windows of --steps steps each; the median is reported')
    ap.add_argument('--spin-seconds', type=float, default=1.0, help='untimed wall time of steps before the counted warm-up (clock settling)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-full', action='store_true', help='BASELINE.md §3 protocol (>=3 warm, >=10 timed): minutes of CPU time')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='do not collect live HBM traffic with rocprofv3 (falls back to profiles/)')
    ap.add_argument('--force-dist', action='store_true', help='initialise the process group and run the split-graph + all-reduce path even at 1 GPU')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: 'nccl' (= RCCL on ROCm) or 'gloo' (tests)")
    ap.add_argument('--share-gpu', action='store_true', help='tests: let several ranks share one device (needs --backend gloo)')
    ap.add_argument('--collective', choices=('auto', 'pg', 'native', 'graph'), default=None,
                    help="gradient exchange at N > 1: 'auto' (default over RCCL) = the step as ONE HIP graph with two overlapped "
                         "all-reduce buckets as its nodes IF a start-up self-check against the eager one-stream step passes on every "
                         "rank, else 'pg'/1 bucket — the line records which (config.collective.selfcheck); 'pg' torch.distributed "
                         "all-reduce between two graphs (default with gloo), 'native' imm_rccl_allreduce on its own stream, 'graph' "
                         "imm_rccl_allreduce captured into the step's single graph (RCCL only)")
    ap.add_argument('--buckets', type=int, choices=(1, 2), default=None,
                    help='all-reduce buckets (2: the renderer bucket travels while the encoders\' backward runs); default: 2 with '
                         "'auto', else IMM_DP_BUCKETS or 1")
    ap.add_argument('--pmc-pass', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--selfcheck-child', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()
    select_workload(args.config)

    if args.pmc_pass:
        pmc_pass(args.steps)
        return 0
    if args.selfcheck_child:
        return selfcheck_child(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return self_spawn(args, sys.argv[1:])

    # stdout carries exactly ONE JSON line: route everything else written to fd 1 (RCCL's version banner, library
    # chatter from any rank) to stderr and keep a private handle on the real stdout for the result
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    if args.share_gpu:
        local_rank %= max(n_dev, 1)
    elif local_rank >= n_dev:
        raise SystemExit('rank %d: local rank %d but only %d device(s) visible' % (rank, local_rank, n_dev))
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend)

    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep

    inputs = synthetic_batch(BATCH_PER_GPU, IMAGE_SIZE, seed=rank, device=dev)
    collective = args.collective
    if collective is None and use_dist and not any(os.environ.get(v, '0') != '0' for v in ('IMM_RCCL_GRAPH', 'IMM_RCCL_NATIVE')):
        collective = 'auto' if (args.backend == 'nccl' and not args.no_graph) else 'pg'
    selfcheck = None
    buckets = args.buckets
    if collective == 'auto':
        selfcheck = selfcheck_in_children(args, dev, world, dev if args.backend == 'nccl' else 'cpu')
        collective, buckets = ('graph', args.buckets or 2) if selfcheck['passed'] else ('pg', 1)
    model = IMMModel(model_config(N_MAPS), dtype=torch_dtype(), device=dev, world_size=world, dp_buckets=buckets)
    ts = TrainStep(model, BATCH_PER_GPU, IMAGE_SIZE, world_size=world, use_graph=not args.no_graph,
                   split_graphs=args.force_dist or collective == 'graph', collective=collective)
    eng = ts.engine
    eng.set_inputs(inputs['image'], inputs['future_image'], inputs['mask'])     # resident in HBM from here on
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        for _ in range(n):
            ts.step(None)
        ts.synchronize()

    # ---- clock settling: a fixed wall time of steps, the same count on every rank (the collectives must pair up) ----
    run_steps(2)                                  # graph capture, code-object loading
    barrier()
    t0 = time.perf_counter(); run_steps(10); t10 = (time.perf_counter() - t0) / 10
    ctrl = dev if args.backend == 'nccl' else 'cpu'      # small control collectives: device tensors for RCCL, host for gloo
    n_spin = torch.tensor([max(0, int(args.spin_seconds / max(t10, 1e-5)))], dtype=torch.int64, device=ctrl)
    if world > 1:
        dist.broadcast(n_spin, 0)
    run_steps(int(n_spin))
    run_steps(max(args.warmup, 1))
    windows = []
    for _ in range(max(args.windows, 1)):
        barrier()
        t0 = time.perf_counter()
        run_steps(args.steps)
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=ctrl)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t)
        windows.append(elapsed)
    elapsed = sorted(windows)[len(windows) // 2]
    loss = float(eng.loss)
    assert loss == loss, 'NaN loss'
    if world > 1 or args.force_dist:
        # the printed loss of the reference is the tower mean (cnn_train_multi.py:173)
        lt = eng.loss.detach().clone().reshape(1).to(ctrl)
        dist.all_reduce(lt)
        loss = float(lt) / world

    replicas_identical, replicas_diff = None, None
    if world > 1:
        # data-parallel invariant: after the same number of updates from all-reduced gradients every rank holds the SAME bits
        bits = eng.params.view(torch.int32).to(torch.int64)
        chk = torch.stack([bits.sum(), (bits * (torch.arange(bits.numel(), device=bits.device) % 8191 + 1)).sum()]).to(ctrl)
        got = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(got, chk)
        replicas_identical = all(bool(torch.equal(g, got[0])) for g in got)
        if not replicas_identical:
            # which tensors differ, and on which ranks (diagnosis: goes into the line as step.replicas_diff)
            names = [n for n, _s, _w in eng.spec]
            per = torch.stack([bits[eng.tab.offsets[i]:eng.tab.offsets[i + 1]].sum() for i in range(len(names))]).to(ctrl)
            allp = [torch.zeros_like(per) for _ in range(world)]
            dist.all_gather(allp, per)
            replicas_diff = {}
            for i, n in enumerate(names):
                odd = [r for r in range(world) if int(allp[r][i]) != int(allp[0][i])]
                if odd:
                    replicas_diff[n] = odd
            sys.stderr.write('bench: replicas differ in %d tensors: %r\n' % (len(replicas_diff), dict(list(replicas_diff.items())[:12])))

    # device-clock phases of the replayed step (every rank runs the probed steps: they contain the collectives)
    phases = None
    if not args.no_graph:
        try:
            phases = ts.measure_phases(5)
        except Exception as e:       # the probes must never lose the measurement
            phases = {'error': '%s: %s' % (type(e).__name__, e)}
        barrier()
    fwd_only = None
    if rank == 0 and not args.no_graph:
        # the reference's `fwd_only` timing switch (cnn_train_multi.py:378,447-449) on BASELINE.json configs[0]'s shape (batch 4:
        # forward + perceptual loss, no backward / update) and on this workload's batch
        try:
            fwd_only = forward_only_rates(dev, [4, BATCH_PER_GPU])
        except Exception as e:
            fwd_only = {'error': '%s: %s' % (type(e).__name__, e)}
    if world > 1:
        dist.barrier()

    if rank == 0:
        # per-kernel timing with HIP events on the launch stream (eager pass; graph replay hides launches)
        with torch.cuda.stream(ts.stream):
            eng._training = True
            rows = []
            for _ in range(3):
                rows = eng.run_timed(eng.prog_fwd) + eng.run_timed(eng.prog_bwd) + eng.run_timed(eng.prog_opt)
        by_tag = {}
        if os.environ.get('IMM_BENCH_DUMP'):
            for tag, ms, fl, nb, name in rows:
                sys.stderr.write('LAUNCH %-16s %-44s %9.1f us %8.1f TF\n' % (tag, name, ms * 1e3, fl / (ms * 1e-3) / 1e12 if fl else 0.0))
        for tag, ms, fl, nb, _name in rows:
            d = by_tag.setdefault(tag, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += ms; d[2] += fl; d[3] += nb
        ig = [by_tag[t] for t in IGEMM_TAGS if t in by_tag]
        n_l, ms_l, fl_l, nb_l = sum(d[0] for d in ig), sum(d[1] for d in ig), sum(d[2] for d in ig), sum(d[3] for d in ig)
        achieved = fl_l / (ms_l * 1e-3) / 1e12
        traffic, traffic_src, hbm_gb_step = None, None, None
        if world == 1 and not args.no_pmc:
            live = pmc_traffic_live()
            if live is not None:
                traffic, hbm_gb_step, _n = live
                traffic_src = 'live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) on this run\'s workload, 3 eager steps'
        if traffic is None:
            traffic, f = pmc_traffic_from_profiles()
            traffic_src = ('profiles: ' + f) if f else None
        roof = {'bound': 'mfma', 'kernel': 'conv_hdeep / conv_halo2 / conv_igemm64 / conv_igemm kernels (convolution fwd + data-gradient, %d launches/step)' % n_l,
                'achieved': round(achieved, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': traffic,
                'traffic_unit': 'MB HBM per launch (rocprofv3 PMC FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE)',
                'traffic_source': traffic_src,
                'algorithmic_mb_per_launch': round(nb_l / n_l / 1e6, 2),
                'avg_launch_us': round(ms_l * 1e3 / n_l, 2), 'gflop_per_launch': round(fl_l / n_l / 1e9, 3)}
        breakdown = {t: {'launches': d[0], 'ms': round(d[1], 3), 'tflops': (round(d[2] / (d[1] * 1e-3) / 1e12, 1) if d[2] else None)}
                     for t, d in sorted(by_tag.items(), key=lambda kv: -kv[1][1])}
        total_ms = sum(d[1] for d in by_tag.values())
        tr = [by_tag[t] for t in ('conv_fwd', 'conv_dgrad', 'conv_wgrad') if t in by_tag]
        step_flops = eng.step_flops()
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            'metric': 'training images/sec at %dx%d K=%d' % (IMAGE_SIZE, IMAGE_SIZE, N_MAPS),
            'value': round(world * BATCH_PER_GPU * args.steps / elapsed, 2),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': WL['dtype'], 'data': 'synthetic',
            'config': {'workload': '%s: %s (fwd + VGG16 perceptual loss + bwd + clip + Adam), batch %d per GPU'
                                   % (WL['name'], WL['what'], BATCH_PER_GPU),
                       'baseline_config': WL.get('index', 1),
                       'loss_scale': (None if eng.loss_scale_state is None else [float(v) for v in eng.loss_scale_state.tolist()]),
                       'global_batch': world * BATCH_PER_GPU, 'image_size': IMAGE_SIZE, 'n_maps': N_MAPS,
                       'parallelism': 'dp%d' % world, 'hip_graph': not args.no_graph,
                       'collective': ({'backend': dist.get_backend(), 'world_size': dist.get_world_size(),
                                       'mode': ts.collective, 'buckets': ts.buckets, 'graph_resident': bool(ts.graph_resident),
                                       'native_rccl': ts.native_comm is not None, 'selfcheck': selfcheck,
                                       'payload_mb': round(eng.tab.total * 4 / 1e6, 2),
                                       # one ring over ONE xGMI link direction (~153 GB/s) moves 2 (N-1)/N of the payload per GPU:
                                       # the per-link bound of a ring all-reduce; with all 7 links in rings it is 1/7 of that
                                       'expected_ring_ms': round(2.0 * (world - 1) / max(world, 1) * eng.tab.total * 4 / 153e9 * 1e3, 4),
                                       'expected_ring_ms_7_links': round(2.0 * (world - 1) / max(world, 1) * eng.tab.total * 4 / (7 * 153e9) * 1e3, 4)}
                                      if dist.is_initialized() else None),
                       'weights': 'seeded random init; synthetic VGG16 (vgg16.caffemodel.h5 unavailable offline)'},
            'roofline': roof,
            'step': {'windows_ms': [round(w / args.steps * 1e3, 4) for w in windows],
                     'min_ms': round(min(windows) / args.steps * 1e3, 4), 'max_ms': round(max(windows) / args.steps * 1e3, 4),
                     'spin_steps': int(n_spin),
                     'conv_gflop_per_image': round(step_flops / BATCH_PER_GPU / 1e9, 2),
                     'step_tflops': round(step_flops / (ms_per_step * 1e-3) / 1e12, 1),
                     'frac_of_peak': round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     'trainable_conv_tflops': round(sum(d[2] for d in tr) / (sum(d[1] for d in tr) * 1e-3) / 1e12, 1) if tr else None,
                     'sum_kernel_ms_eager': round(total_ms, 3), 'hbm_gb_per_step': hbm_gb_step, 'loss': round(loss, 3),
                     'phases_ms': phases, 'fwd_only': fwd_only,
                     'hbm_bytes_allocated': eng.memory_bytes(), 'replicas_identical': replicas_identical,
                     'replicas_diff': (None if not replicas_diff else {k: v for k, v in list(replicas_diff.items())[:12]})},
            'kernels': breakdown,
        }
        if (not args.no_cpu_baseline and world == 1) or args.cpu_baseline_full:
            out['cpu_baseline'] = cpu_baseline(0.0 if args.cpu_baseline_full else 12.0)
            try:        # the checker must never lose the measurement
                out['parity'] = parity_vs_oracle(dev)
            except Exception as e:
                out['parity'] = {'error': '%s: %s' % (type(e).__name__, e)}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
