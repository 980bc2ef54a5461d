"""bench.py — IMM conditional-generation training step on MI355X: training images/sec at 128x128, K=10.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU; batch 32 per GPU (BASELINE.json configs[1]; N>1 = configs[2] weak scaling); synthetic
inputs resident in HBM; a step = forward (both encoders, landmark bottleneck, renderer, VGG16 perceptual
loss) + backward + RCCL gradient all-reduce + per-tensor clip + Adam.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
BATCH_PER_GPU = 32
IMAGE_SIZE = 128
N_MAPS = 10
IGEMM_TAGS = ('conv_fwd', 'vgg_fwd', 'conv_dgrad', 'vgg_dgrad')    # all launches of conv_igemm_kernel


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
    return {'image': im.to(device), 'future_image': fut.to(device), 'mask': mask.to(device).contiguous()}


def pmc_traffic_per_launch():
    """HBM MB per launch of the conv_igemm kernels from the committed PMC profile (separate rocprofv3 --pmc passes,
    profiles/r*_pmc_hbm_bytes.csv: newest round wins); None when no profile is present."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hbm_bytes.csv')))
    if not files:
        return None
    n = mb = 0.0
    with open(files[-1]) as f:
        for row in csv.reader(f):
            if len(row) == 7 and ('conv_igemm' in row[0] or 'conv_halo' in row[0] or 'conv_hdeep' in row[0]):
                n += float(row[1]); mb += float(row[1]) * (float(row[4]) + float(row[5]))
    return round(mb / n, 2) if n else None


def cpu_baseline(sample_batch=8, steps=2):
    """CPU restatement of the TF1 graph (oracle/imm_oracle.py) timed on this node's host cores: the same
    algorithmic work per image as the GPU step (forward + backward + clip + Adam)."""
    from oracle import imm_oracle as O
    cfg = O.default_model_config(N_MAPS)
    P, S = O.init_params(cfg, IMAGE_SIZE)
    opt = O.new_adam_state(P)
    inp = O.synthetic_inputs(sample_batch, IMAGE_SIZE)
    P, S, _ = O.train_step(P, S, opt, [inp], cfg)          # warm-up (thread pools, allocator)
    t0 = time.time()
    for _ in range(steps):
        P, S, _ = O.train_step(P, S, opt, [inp], cfg)
    dt = (time.time() - t0) / steps
    return {'value': round(sample_batch / dt, 3), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': '%d fp32 training steps of batch %d at 128x128 K=10 (torch-CPU restatement of the TF1 graph; '
                      'TF 1.10 itself is not installable here)' % (steps, sample_batch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='initialise RCCL and run the split-graph + all-reduce path even at 1 GPU')
    args = ap.parse_args()

    # stdout carries exactly ONE JSON line: route everything else written to fd 1 (RCCL's version banner, library
    # chatter from any rank) to stderr and keep a private handle on the real stdout for the result
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
    torch.cuda.set_device(local_rank)
    dev = 'cuda:%d' % local_rank
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=torch.device(dev))

    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep

    model = IMMModel(model_config(N_MAPS), dtype=torch.bfloat16, device=dev, world_size=world)
    ts = TrainStep(model, BATCH_PER_GPU, IMAGE_SIZE, world_size=world, use_graph=not args.no_graph, split_graphs=args.force_dist)
    eng = ts.engine
    inputs = synthetic_batch(BATCH_PER_GPU, IMAGE_SIZE, seed=rank, device=dev)
    eng.set_inputs(inputs['image'], inputs['future_image'], inputs['mask'])     # resident in HBM from here on
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        ts.step(None)
    ts.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts.step(None)
    ts.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(eng.loss)
    assert loss == loss, 'NaN loss'

    if rank == 0:
        # per-kernel timing with HIP events on the launch stream (eager pass; graph replay hides launches)
        roof = None
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
        roof = {'bound': 'mfma', 'kernel': 'conv_hdeep / conv_halo2 / conv_igemm64 / conv_igemm kernels (convolution fwd + data-gradient, %d launches/step)' % n_l,
                'achieved': round(achieved, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(achieved / PEAK_BF16_TFLOPS, 4), 'traffic': pmc_traffic_per_launch(),
                'traffic_unit': 'MB HBM per launch (rocprofv3 PMC FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE, profiles/)',
                'algorithmic_mb_per_launch': round(nb_l / n_l / 1e6, 2),
                'avg_launch_us': round(ms_l * 1e3 / n_l, 2), 'gflop_per_launch': round(fl_l / n_l / 1e9, 3)}
        breakdown = {t: {'launches': d[0], 'ms': round(d[1], 3), 'tflops': (round(d[2] / (d[1] * 1e-3) / 1e12, 1) if d[2] else None)}
                     for t, d in sorted(by_tag.items(), key=lambda kv: -kv[1][1])}
        total_ms = sum(d[1] for d in by_tag.values())
        step_flops = eng.step_flops()
        ms_per_step = elapsed / args.steps * 1e3
        out = {
            'metric': 'training images/sec at 128x128 K=10', 'value': round(world * BATCH_PER_GPU * args.steps / elapsed, 2),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'CelebA-shape 128x128 K=10 IMM training step (fwd + VGG16 perceptual loss + bwd + '
                                   'clip + Adam), batch %d per GPU' % BATCH_PER_GPU,
                       'global_batch': world * BATCH_PER_GPU, 'image_size': IMAGE_SIZE, 'n_maps': N_MAPS,
                       'parallelism': 'dp%d' % world, 'hip_graph': not args.no_graph,
                       'weights': 'seeded random init; synthetic VGG16 (vgg16.caffemodel.h5 unavailable offline)'},
            'roofline': roof,
            'step': {'conv_gflop_per_image': round(step_flops / BATCH_PER_GPU / 1e9, 2),
                     'step_tflops': round(step_flops / (ms_per_step * 1e-3) / 1e12, 1),
                     'frac_of_peak': round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     'sum_kernel_ms_eager': round(total_ms, 3), 'loss': round(loss, 3),
                     'hbm_bytes_allocated': eng.memory_bytes()},
            'kernels': breakdown,
        }
        if not args.no_cpu_baseline and world == 1:
            out['cpu_baseline'] = cpu_baseline()
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + '\n').encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
