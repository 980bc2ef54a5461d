"""Split-K slab sizes per trainable convolution (MI355X; needs a GPU for the engine build).  Usage: python tools/slab_report.py"""
import torch, sys
sys.path.insert(0, '/root/repo')
from imm_amd.engine import IMMEngine
import bench
cfg = bench.model_config(10)
e = IMMEngine(cfg, 32, 128)
tot = 0
for lay in e.enc_im + e.enc_pose + [e.pose_head] + e.ren:
    b = lay.slab.numel() * 4
    tot += b
    print('%-40s nsplit=%4d slab=%7.2f MB' % (lay.scope if hasattr(lay, 'scope') else '?', lay.nsplit, b / 1e6))
print('total slab MB', tot / 1e6)
