"""decoder.predict at 1M SNPs (pvae_decode -> pvk_dec_probs): wall time per call including the copy of the probabilities to the
host; run under `rocprofv3 --kernel-trace --stats` for the kernel's own time."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
e = E.Engine(n_snps=L, width=128, depth=6, latent_dim=2, batch_size=32, seed=1)
e.init_weights(1)
g = np.random.default_rng(0)
for n in (32, 128):
    z = g.normal(size=(n, 2)).astype(np.float32)
    e.decode(z)
    t = time.perf_counter()
    for _ in range(5):
        p = e.decode(z)
    dt = (time.perf_counter() - t) / 5
    print(f"decode {n} latent points at {L} SNPs: {dt * 1e3:.2f} ms per call ({n * L * 4 / dt / 1e9:.1f} GB/s of probabilities to the host), "
          f"mean p {float(p.mean()):.4f}")
