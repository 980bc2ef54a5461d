"""Profiler-free look at one HIP-graph replay of the training step: (1) host time of each graph launch call against the
device time per step (is the host ahead of the GPU?), (2) the device wall-clock probes of IMM_DEBUG_STAMPS (lane
boundaries, or every launch with IMM_DEBUG_STAMPS=all).  Usage: IMM_DEBUG_STAMPS=marks python tools/graph_timeline.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from imm_amd.models.imm_model import IMMModel
    from imm_amd.train.cnn_train_multi import TrainStep
    from imm_amd.utils.box import Box
    dev = 'cuda:0'
    B, S = 32, 128
    model = IMMModel(Box(bench.model_config(10)), dtype=torch.bfloat16, device=dev)
    ts = TrainStep(model, B, S, world_size=1, use_graph=True)
    inputs = bench.synthetic_batch(B, S, 0, dev)
    ts.step(inputs)
    for _ in range(300):
        ts.step(None)
    ts.synchronize()
    n = 40
    host = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t_all = time.perf_counter()
    for _ in range(n):
        t = time.perf_counter()
        ts.step(None)
        host.append((time.perf_counter() - t) * 1e3)
    t_issue = (time.perf_counter() - t_all) * 1e3
    e1.record()
    ts.synchronize()
    torch.cuda.synchronize()      # e0 / e1 sit on the caller's stream, which waits for the step's stream
    dev_ms = e0.elapsed_time(e1) / n
    print('device %.3f ms/step; host issue of %d steps %.2f ms total (%.3f ms/step); per-call host ms: first 8 %s ... last 4 %s' % (
        dev_ms, n, t_issue, t_issue / n, ['%.2f' % h for h in host[:8]], ['%.2f' % h for h in host[-4:]]))
    eng = ts.engine
    if eng._stamp_mode:
        rep = eng.stamp_report()
        prev = {}
        for t, lane, label in sorted(rep):
            d = t - prev.get(lane, t)
            prev[lane] = t
            print('%9.1f us  (+%7.1f on lane)  lane %d  %s' % (t, d, lane, label))


if __name__ == '__main__':
    main()
