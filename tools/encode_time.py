import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from popvae_amd import engine as E
N, L = 2504, 1_000_000
g = np.random.default_rng(0)
packed = (g.integers(0, 256, size=(N, L // 4), dtype=np.uint8) & 0x55)
e = E.Engine(n_snps=L, width=128, depth=6, latent_dim=2, batch_size=32, seed=1)
e.load_genotypes(packed); e.init_weights(1)
rows = np.arange(N, dtype=np.int32)
e.encode(rows)
t = time.perf_counter(); mu, lv, z = e.encode(rows); dt = time.perf_counter() - t
print("encode all 2504 rows at 1M SNPs: %.2f ms" % (dt * 1e3), mu.shape, float(np.abs(mu).max()))
