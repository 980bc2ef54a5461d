import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imm_amd import ops, _lib as L
torch.cuda.set_device(0)
DEV='cuda:0'
def probe(B,H,ci,co,dt=torch.bfloat16, reps=300):
    g=torch.Generator().manual_seed(1)
    x=(torch.randn(B,H,H,ci,generator=g)).to(dt).to(DEV)
    w=(torch.randn(3,3,ci,co,generator=g)*0.05).to(DEV)
    b=torch.randn(co,generator=g).to(DEV)
    desc=ops.fwd_desc(B,H,H,ci,ci,co,co,3,2,L.CONV_BIAS|L.CONV_STATS)
    wt=torch.zeros(ops.round_up(co,128),desc.kpad,dtype=dt,device=DEV)
    ops.pack_weights(w,wt,0,3,3,ci,co,ci,wt.shape[0],desc.kpad)
    nb=ops.conv_stats_blocks(desc)
    side=torch.cuda.Stream()
    junk=torch.randn(4096,4096,device=DEV)
    ref=None; bad=0
    for r in range(reps):
        y=torch.full((B,desc.ho,desc.wo,co),float('nan'),dtype=dt,device=DEV)
        st=torch.full((nb,2,co),float('nan'),device=DEV)
        if r%3==1:
            with torch.cuda.stream(side):
                junk2=junk@junk
        ops.conv2d(desc,x,wt,b,y,st)
        torch.cuda.synchronize()
        if ref is None: ref=(y.clone(),st.clone())
        elif not (torch.equal(ref[0].view(torch.int16),y.view(torch.int16)) and torch.equal(ref[1],st)): bad+=1
    print('DET', ops.conv2d_variant(desc,dt), (B,H,ci,co), 'mismatching runs', bad, 'of', reps, 'nan in y', bool(torch.isnan(ref[0].float()).any()))
probe(2,64,64,128); probe(2,32,128,256); probe(32,64,64,128); probe(32,32,128,256)
