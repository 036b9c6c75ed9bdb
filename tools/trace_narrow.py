"""Per-op timeline of pvk_narrow_fwd (wave 0, thread 0; 100 MHz wall clock): build the engine with -DPVN_TRACE into
another file and point PVAE_LIB at it:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DPVN_TRACE -o /tmp/libpvae_trace.so popvae_amd/csrc/pvae_engine.hip
  PVAE_LIB=/tmp/libpvae_trace.so python tools/trace_narrow.py"""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E
N, L, W, D = 200, int(os.environ.get('TRACE_L', '100000')), 128, 6
g = np.random.default_rng(0)
packed = (g.integers(0, 256, size=(N, (L + 31) // 32 * 8), dtype=np.uint8) & 0x55)
e = E.Engine(n_snps=L, width=W, depth=D, latent_dim=2, batch_size=32, seed=1)
e.load_genotypes(packed); e.init_weights(1); e.lr = 1e-6
rows = np.arange(32, dtype=np.int32)
for _ in range(5):
    e.train_step(rows)
buf = (C.c_ulonglong * 2048)()
E.lib().pvae_debug_trace(buf, 2048)
a = np.array(buf[:], dtype=np.int64).reshape(-1, 2)
for i in range(2 * D + 1):
    r = a[i * 8:(i + 1) * 8]
    nxt = a[(i + 1) * 8]
    w = [(x[0] - r[0][0]) * 10 for x in r]
    cyc = [int(x[1] - r[0][1]) for x in r]
    print("   cycles:", cyc[1:5], "op cycles", int(nxt[1] - r[0][1]), " MHz ~", (nxt[1] - r[0][1]) / max(1, (nxt[0] - r[0][0])) * 100)
    print(i, "prefetch_issued %5d | mma_done %5d | epilogue_done %5d | after_barrier %5d | op_total %5d ns" % (
        w[1], w[2], w[3], w[4], (nxt[0] - r[0][0]) * 10))
print("backward chain (mma_issue_start, mma_done, drained, epilogue_done, after_barrier):")
for i in range(2 * D + 1):
    r = a[(16 + i) * 8:(17 + i) * 8]
    nxt = a[(17 + i) * 8]
    w = [(x[0] - r[0][0]) * 10 for x in r]
    print(i, "mma_start %5d | mma_done %5d | drained %5d | epilogue_done %5d | after_barrier %5d | op_total %5d ns" % (
        w[1], w[5], w[2], w[3], w[4], (nxt[0] - r[0][0]) * 10))
