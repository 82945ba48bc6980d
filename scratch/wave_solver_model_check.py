import sys; sys.path.insert(0,'/root/repo/scratch')
import numpy as np
from wave_solver_model import WaveSolver, band_ok
def system(P, w, seed, extra=None):
    rng=np.random.default_rng(seed); n=6*P
    H=np.zeros((n,n)); fpose=list(range(P))
    pairs=[(p,q) for p in range(P) for q in range(max(0,p-w),p+1)]
    if extra: pairs+=extra
    for p,q in pairs:
        B=rng.standard_normal((6,6))*0.3
        if p==q: B=B+B.T
        H[6*p:6*p+6,6*q:6*q+6]+=B
        if p!=q: H[6*q:6*q+6,6*p:6*p+6]+=B.T
        fpose[p]=min(fpose[p],q)
    H+=np.eye(n)*(np.abs(H).sum(1).max()+1.0)
    return H, rng.standard_normal(n), fpose
for P,w,seed in [(24,4,0),(24,3,1),(25,4,2),(8,4,3),(3,2,4),(29,4,5),(63,4,6),(24,1,7),(16,0,8),(2,1,9)]:
    H,b,fp=system(P,w,seed)
    ok=band_ok(fp,6*P)
    lm,ep=1e-4,0.1
    Hd=H+np.diag(ep+lm*np.diag(H))
    xr=np.linalg.solve(Hd,b)
    ws=WaveSolver(np.tril(H)+np.tril(H,-1)*0,b,lm,ep)   # lower triangle only is read
    x,f=ws.solve()
    print(P,w,'band_ok',ok,'failed',f,'err',np.abs(x-xr).max()/np.abs(xr).max())
# band too wide must be rejected
H,b,fp=system(24,6,11); print('w=6 band_ok',band_ok(fp,144))
H,b,fp=system(24,5,12); print('w=5 band_ok',band_ok(fp,144))
H,b,fp=system(24,2,13,extra=[(20,3)]); print('arrow band_ok',band_ok(fp,144))
# non-SPD
H,b,fp=system(24,4,14); H[70,70]=-5
ws=WaveSolver(np.tril(H),b,1e-4,0.1); x,f=ws.solve(); print('nonspd failed',f, np.abs(x).max())
