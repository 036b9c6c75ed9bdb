import ctypes as C, numpy as np, sys, os
sys.path.insert(0,'/root/repo')
from popvae_amd import engine as E
N,L,W,D=200,100000,128,6
g=np.random.default_rng(0)
packed=(g.integers(0,256,size=(N,(L+31)//32*8),dtype=np.uint8)&0x55)
e=E.Engine(n_snps=L,width=W,depth=D,latent_dim=2,batch_size=32,seed=1)
e.load_genotypes(packed); e.init_weights(1); e.lr=1e-6
rows=np.arange(32,dtype=np.int32)
for _ in range(5): e.train_step(rows)
buf=(C.c_ulonglong*2048)()
E.lib().pvae_debug_trace(buf,2048)
a=np.array(buf[:],dtype=np.int64).reshape(-1,2)
t0=a[0]
for i in range(14):
    r=a[i*8:(i+1)*8]
    nxt=a[(i+1)*8]
    w=[(x[0]-r[0][0])*10 for x in r]
    print(i, "waited_w %5d | prefetch_issued %5d | dense_start %5d | op_total %5d ns"%(w[1], w[5], w[2], (nxt[0]-r[0][0])*10))
r=a[60*8:61*8]
print("last dense op: mma_done->after_barrier %d ns; after barrier to op end:"%((r[4][0]-r[3][0])*10), (a[14*8][0]-r[4][0])*10 if a[14*8][0] else None, " mma_done rel last op start", (r[3][0]-a[13*8][0])*10)
