"""Step time and kernel-class breakdown of the non-headline BASELINE.json configurations (parity-test cases, not bench lines) on
one GPU: configs[3] S=256 / K=30 bf16, the per-GPU shape of configs[4] S=128 / K=50 f16, and S=128 / K=30 bf16.

    python tools/bench_configs.py [--json profiles/r04_configs.json]

Per configuration: HIP-graph replay time (median of three 40-step windows after 1 s of untimed steps, the protocol of bench.py),
and one eager pass with a HIP-event pair around every launch (engine.run_timed) -> time and TFLOP/s per launch class and the
roofline fraction of the convolution forward + data-gradient family against the 2.5 PFLOP/s dense 16-bit MFMA peak."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from imm_amd.models.imm_model import IMMModel   # noqa: E402
from imm_amd.train.cnn_train_multi import TrainStep   # noqa: E402


def run(B, S, K, dt, label, steps=40, windows=3):
    dev = 'cuda:0'
    model = IMMModel(bench.model_config(K), dtype=dt, device=dev)
    ts = TrainStep(model, B, S, world_size=1, use_graph=True)
    eng = ts.engine
    inputs = bench.synthetic_batch(B, S, seed=0, device=dev)
    ts.step(inputs)
    ts.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:            # clocks settle
        for _ in range(10):
            ts.step(None)
        ts.synchronize()
    win = []
    for _ in range(windows):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step(None)
        ts.synchronize()
        win.append((time.perf_counter() - t0) / steps * 1e3)
    ms = sorted(win)[len(win) // 2]
    with torch.cuda.stream(ts.stream):
        eng._training = True
        rows = []
        for _ in range(3):
            rows = eng.run_timed(eng.prog_fwd) + eng.run_timed(eng.prog_bwd) + eng.run_timed(eng.prog_opt)
    by = {}
    for tag, t, fl, _nb, _name in rows:
        d = by.setdefault(tag, [0, 0.0, 0.0])
        d[0] += 1; d[1] += t; d[2] += fl
    fam = [by[t] for t in bench.IGEMM_TAGS if t in by]
    fam_ms, fam_fl, fam_n = sum(d[1] for d in fam), sum(d[2] for d in fam), sum(d[0] for d in fam)
    tr = [by[t] for t in ('conv_fwd', 'conv_dgrad', 'conv_wgrad') if t in by]
    flops = eng.step_flops()
    rec = {'config': label, 'batch': B, 'image_size': S, 'n_maps': K, 'dtype': str(dt).split('.')[-1],
           'ms_per_step': round(ms, 4), 'windows_ms': [round(w, 4) for w in win], 'images_per_s': round(B / ms * 1e3, 1),
           'conv_gflop_per_image': round(flops / B / 1e9, 2), 'step_tflops': round(flops / ms / 1e9, 1),
           'frac_of_peak': round(flops / ms / 1e9 / bench.PEAK_BF16_TFLOPS, 4),
           'roofline': {'bound': 'mfma', 'kernel': 'convolution fwd + data-gradient family (%d launches/step)' % fam_n,
                        'achieved': round(fam_fl / fam_ms / 1e9, 1), 'peak': bench.PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(fam_fl / fam_ms / 1e9 / bench.PEAK_BF16_TFLOPS, 4)},
           'trainable_conv_tflops': round(sum(d[2] for d in tr) / sum(d[1] for d in tr) / 1e9, 1),
           'launches': sum(d[0] for d in by.values()),
           'kernels': {t: {'launches': d[0], 'ms': round(d[1], 3), 'tflops': round(d[2] / d[1] / 1e9, 1) if d[2] else None}
                       for t, d in sorted(by.items(), key=lambda kv: -kv[1][1])},
           'loss': round(float(eng.loss), 3), 'hbm_gb_allocated': round(eng.memory_bytes() / 1e9, 2),
           'loss_scale_state': None if eng.loss_scale_state is None else [float(v) for v in eng.loss_scale_state.tolist()]}
    print('%s: B=%d S=%d K=%d %s: %.3f ms/step = %.1f images/s; %.1f GFLOP/image -> %.0f TFLOP/s (%.3f of peak); conv fwd+dgrad %.0f TFLOP/s'
          % (label, B, S, K, rec['dtype'], ms, rec['images_per_s'], rec['conv_gflop_per_image'], rec['step_tflops'], rec['frac_of_peak'],
             rec['roofline']['achieved']), file=sys.stderr)
    del ts, model, eng
    torch.cuda.empty_cache()
    return rec


if __name__ == '__main__':
    torch.cuda.set_device(0)
    out = [run(16, 256, 30, torch.bfloat16, 'BASELINE configs[3]: K=30 at 256x256, 1 GPU'),
           run(32, 128, 50, torch.float16, 'BASELINE configs[4] per-GPU shape: K=50, f16 storage + loss scaling'),
           run(32, 128, 30, torch.bfloat16, 'K=30 at 128x128')]
    doc = {'what': 'non-headline BASELINE.json configurations on ONE MI355X (tools/bench_configs.py): graph-replay step time, eager per-class '
                   'launch timings (HIP events), roofline fraction of the convolution fwd + data-gradient family',
           'device': torch.cuda.get_device_name(0), 'configs': out}
    text = json.dumps(doc, indent=1)
    if '--json' in sys.argv:
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as f:
            f.write(text + '\n')
    print(text)
