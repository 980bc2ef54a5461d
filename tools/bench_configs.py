"""Step time of the non-headline BASELINE.json configurations (parity-test cases, not bench lines): S=256/K=30 bf16 and
S=128/K=50 f16 on one GPU.  Usage: python tools/bench_configs.py [B S K dtype]..."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402
from imm_amd.models.imm_model import IMMModel   # noqa: E402
from imm_amd.train.cnn_train_multi import TrainStep   # noqa: E402


def run(B, S, K, dt, steps=20, warmup=5):
    dev = 'cuda:0'
    model = IMMModel(bench.model_config(K), dtype=dt, device=dev)
    ts = TrainStep(model, B, S, world_size=1, use_graph=True)
    inputs = bench.synthetic_batch(B, S, seed=0, device=dev)
    for _ in range(warmup):
        ts.step(inputs)
    ts.synchronize()
    t0 = time.time()
    for _ in range(steps):
        ts.step(inputs)
    ts.synchronize()
    dt_s = (time.time() - t0) / steps
    eng = ts.engine
    print('B=%d S=%d K=%d %s: %.3f ms/step = %.1f images/s; %.1f GFLOP/image -> %.0f TFLOP/s; loss %.3f; %.2f GB allocated' % (
        B, S, K, str(dt).split('.')[-1], dt_s * 1e3, B / dt_s, eng.step_flops() / B / 1e9, eng.step_flops() / dt_s / 1e12,
        float(eng.loss), eng.memory_bytes() / 1e9))
    del ts, model
    torch.cuda.empty_cache()


if __name__ == '__main__':
    torch.cuda.set_device(0)
    run(16, 256, 30, torch.bfloat16)
    run(32, 128, 50, torch.float16)
    run(32, 128, 30, torch.bfloat16)
