import os, sys, torch
sys.path.insert(0, '/root/repo')
from imm_amd import _lib as L
from imm_amd import ops
from tools.bench_conv import time_launch
DEV='cuda:0'; dt=torch.bfloat16
torch.cuda.set_device(0)
for (n,H,co) in [(64,16,512),(64,32,256),(64,64,128)]:
    for ci in (64,128,256,512):
        x=(torch.randn(n,H,H,ci,device=DEV)*0.5).to(dt); w=torch.randn(3,3,ci,co,device=DEV)*0.05; b=torch.zeros(co,device=DEV)
        desc=ops.fwd_desc(n,H,H,ci,ci,co,co,3,1,L.CONV_BIAS|L.CONV_RELU)
        wt=torch.zeros(ops.round_up(co,128),desc.kpad,dtype=dt,device=DEV); ops.pack_weights(w,wt,0,3,3,ci,co,ci,wt.shape[0],desc.kpad)
        y=torch.empty(n,H,H,co,dtype=dt,device=DEV)
        us=time_launch(lambda: ops.conv2d(desc,x,wt,b,y), reps=30)
        print('n=%d H=%d co=%d ci=%4d  %7.1f us  %7.1f TF' % (n,H,co,ci,us,2.0*n*H*H*9*ci*co/us/1e6))
