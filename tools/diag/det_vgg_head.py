"""Determinism of imm_vgg_head_fwd: N launches on the same inputs, every output compared bitwise with the first (optionally while a
second stream keeps the chip busy with stores, which is what perturbs the order in which VMEM operations complete)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imm_amd import _lib as L   # noqa: E402
from imm_amd import ops        # noqa: E402

DEV = 'cuda:0'
torch.cuda.set_device(0)
B, S, ldp, dt = 32, 128, 3, torch.bfloat16
N = int(os.environ.get('N', 300))
gt = torch.rand(B, S, S, 3, device=DEV) * 255
pred = torch.rand(B, S, S, ldp, device=DEV) * 255
w11 = torch.randn(9, 64, device=DEV) * 0.4; b11 = torch.randn(64, device=DEV) * 0.1
w12 = torch.randn(3, 3, 64, 64, device=DEV) * 0.05; b12 = torch.randn(64, device=DEV) * 0.1
fd = ops.fwd_desc(2 * B, S, S, 64, 64, 64, 64, 3, 1, L.CONV_BIAS | L.CONV_RELU)
wt = torch.zeros(128, fd.kpad, dtype=dt, device=DEV)
ops.pack_weights(w12, wt, 0, 3, 3, 64, 64, 64, 128, fd.kpad)
scratch = torch.empty(ops.vgg_head_scratch_bytes(B, S), dtype=torch.uint8, device=DEV)
side = torch.cuda.Stream()
noise = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
bad_y = bad_a = 0
ref_y = ref_a = None
for i in range(N):
    a = torch.zeros(2 * B, S, S, 64, dtype=dt, device=DEV)
    y = torch.empty(2 * B, S, S, 64, dtype=dt, device=DEV)
    if i % 2:
        with torch.cuda.stream(side):
            noise.add_(1.0)
    ops.vgg_head_fwd(gt, pred, ldp, B, S, w11, b11, wt, b12, a, B, y, scratch)
    torch.cuda.synchronize()
    if ref_y is None:
        ref_y, ref_a = y, a
    else:
        bad_y += int(not torch.equal(y, ref_y)); bad_a += int(not torch.equal(a, ref_a))
print('vgg_head determinism: %d launches, %d with a different conv1_2 output, %d with a different stored conv1_1 half' % (N, bad_y, bad_a))
sys.exit(1 if (bad_y or bad_a) else 0)
