import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from imm_amd import ops
import test_kernels_gpu as T
torch.cuda.set_device(0)
for c in T.CONV_CASES:
    B, H, ci_real, ci_pad, co, k, stride, out_f32, tag = c
    flags = 1 | (16 if out_f32 else 0)
    ldy = ops.round_up(co, 8 if not out_f32 else 4)
    d = ops.fwd_desc(B, H, H, ci_pad, ci_pad, co, ldy, k, stride, flags)
    print('CONV', tag, ops.conv2d_variant(d, torch.bfloat16))
for c in T.DGRAD_CASES:
    B, H, ci_real, ci_pad, co, co_pad, k, stride, tag = c
    d = ops.dgrad_desc(B, H, H, ci_real, ci_pad, co_pad, co_pad, k, stride, 0)
    print('DGRAD', tag, ops.conv2d_variant(d, torch.bfloat16))
