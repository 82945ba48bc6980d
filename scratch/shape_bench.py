import sys, numpy as np, torch
sys.path.insert(0,'dba-fusion_amd')
from dbaf_amd.corr import CorrBlock
dev='cuda'
rng=np.random.default_rng(0)
for (h,w) in [(64,64),(55,55),(28,107),(48,64),(56,56)]:
    n=24
    f1=torch.from_numpy(rng.standard_normal((1,n,128,h,w)).astype(np.float16)).to(dev)
    f2=torch.from_numpy(rng.standard_normal((1,n,128,h,w)).astype(np.float16)).to(dev)
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    cb=CorrBlock(f1,f2,num_levels=4,radius=3); torch.cuda.synchronize()
    e0.record(); cb=CorrBlock(f1,f2,num_levels=4,radius=3); e1.record(); torch.cuda.synchronize()
    tb=e0.elapsed_time(e1)*1e3/n
    y,x=np.meshgrid(np.arange(h,dtype=np.float32),np.arange(w,dtype=np.float32),indexing='ij')
    c=np.zeros((n,h,w,2),np.float32)
    for e in range(n):
        fx,fy=rng.uniform(-6,6,2); jit=0.3*np.sin(0.2*x+0.13*y+e)
        c[e,...,0]=x+fx+jit+0.02*x; c[e,...,1]=y+fy+0.5*jit+0.01*y
    ct=torch.from_numpy(c)[None].to(dev)
    for _ in range(3): cb(ct)
    e0.record()
    for _ in range(10): cb(ct)
    e1.record(); torch.cuda.synchronize()
    print("%3dx%-3d layout=%-9s build %.1f us/edge  lookup %.2f us/edge" % (h,w,cb.layout,tb,e0.elapsed_time(e1)*1e3/10/n))
