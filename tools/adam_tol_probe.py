import sys, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import popvae_oracle as O
from tests.test_gpu_parity import _problem, _mk, _relerr
for (L,W,D) in [(1000,128,3),(3887,128,6),(900,256,3),(700,512,2),(600,128,20),(500,256,20),(1100,96,3),(2000,32,2)]:
    N,K=64,2
    dc,w=_problem(N,L,W,D,K,seed=11)
    e=_mk(dc,w,W,D,K)
    wo=[x.astype(np.float32).astype(np.float64) for x in w]
    opt=O.KerasAdam(wo)
    g=np.random.default_rng(2)
    for step in range(5):
        n=32 if step!=3 else 13
        rows=g.choice(N,n,replace=False).astype(np.int32)
        eps=g.normal(size=(n,K)).astype(np.float32)
        O.train_step(wo,opt,dc[rows],eps.astype(np.float64),D,K)
        e.train_step(rows,eps)
    em=max(_relerr(a,b) for a,b in zip(e.get_moments(0),opt.m))
    ev=max(_relerr(a,b) for a,b in zip(e.get_moments(1),opt.v))
    ew=max(np.abs(a.astype(np.float64)-b).max() for a,b in zip(e.get_weights(),wo))
    print(L,W,D,"m",em,"v",ev,"w",ew,flush=True)
    e.close()
