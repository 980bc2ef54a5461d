"""CPU-only dry run of the engine's program construction (no launches): catches Python-level errors in the launch-program
builder before a GPU call is spent.  Prints the launch counts by tag and the filter-gradient split plan."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from imm_amd import engine as E, ops          # noqa: E402
from imm_amd.utils.box import Box             # noqa: E402
from oracle import imm_oracle as O            # noqa: E402

ops.device_info = lambda: (256, 950)
E.IMMEngine.run = lambda self, prog: None
E.IMMEngine._pack_vgg = lambda self: None
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dt = torch.float16 if 'f16' in sys.argv else torch.bfloat16
eng = E.IMMEngine(Box(dict(O.default_model_config(K))), B, S, device='cpu', act_dtype=dt)
for name, prog in (('fwd', eng.prog_fwd), ('bwd', eng.prog_bwd), ('opt', eng.prog_opt)):
    c = collections.Counter(l.tag for l in prog if l.fn is not None)
    print(name, sum(c.values()), dict(c))
tot = 0
for lay in eng.enc_im + eng.enc_pose + [eng.pose_head] + eng.ren:
    nb = lay.slab.numel() * 4
    tot += nb
    print('%-44s nsplit %4d  slab %7.2f MB' % (lay.scope, lay.nsplit, nb / 1e6))
print('slabs total %.1f MB' % (tot / 1e6))
