import sys, numpy as np, torch
sys.path.insert(0,'dba-fusion_amd')
from dbaf_amd import synthetic as syn
from dbaf_amd.corr import CorrBlock
from dbaf_amd import projective_ops as pops
W = syn.window_25_96(0)
dev='cuda'
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
fm = t(syn.make_fmaps(W.B, 128, 64, 64, 1000)); ii, jj = t(W.ii), t(W.jj)
corr=None
for c0 in range(0, 96, 32):
    cb = CorrBlock(fm[ii[c0:c0+32]][None], fm[jj[c0:c0+32]][None]); corr = cb if corr is None else corr.cat(cb)
K = t(W.intrinsics)[None,None].expand(1,W.B,4).contiguous()
c_bench,_ = pops.projective_transform(t(W.poses)[None], t(W.disps)[None], K, ii, jj)
c_gt,_ = pops.projective_transform(t(W.poses_gt)[None], t(W.disps_gt)[None], K, ii, jj)
y,x = np.meshgrid(np.arange(64,dtype=np.float32), np.arange(64,dtype=np.float32), indexing='ij')
sm = np.zeros((96,64,64,2),np.float32); rng=np.random.default_rng(0)
for e in range(96):
    fx,fy = rng.uniform(-10,10,2); jit = 0.3*np.sin(0.2*x+0.13*y+e)
    sm[e,...,0]=x+fx+jit+0.02*x; sm[e,...,1]=y+fy+0.5*jit+0.01*y
c_sm = t(sm)[None]
def run(name,c):
    for _ in range(3): corr(c)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): corr(c)
    e1.record(); torch.cuda.synchronize()
    cc=c[0].cpu().numpy()
    print('%-28s %.1f us'%(name, e0.elapsed_time(e1)/20*1e3), ' frac OOB-ish', float(((cc[...,0]<-4)|(cc[...,0]>67)|(cc[...,1]<-4)|(cc[...,1]>67)).mean()))
run('bench state coords', c_bench); run('gt coords', c_gt); run('smooth synthetic', c_sm)
# variance of flow within rows
d = (c_bench[0,...,0].cpu().numpy() - x); print('row-wise spread of flow x (p50,p90,max):', np.percentile(d.max(-1)-d.min(-1),[50,90,100]))
