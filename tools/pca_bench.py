#!/usr/bin/env python3
"""PCA (pvae_pca_gram + host eigh) timing at a BASELINE shape: python tools/pca_bench.py [C4|C2|C1] [n_components]
The Gram kernel is MFMA-bound (float64): FLOP = 2 * L * (128 * 128 * tile pairs)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E, pca as P  # noqa: E402

SH = {"C4": (2504, 1_000_000), "C2": (1000, 100_000), "C1": (95, 3887), "C3": (929, 500_000)}
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
ncomp = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, L = SH[name]
g = np.random.default_rng(0)
lut = np.arange(256, dtype=np.uint8)
for sh in (0, 2, 4, 6):
    bad = ((lut >> sh) & 3) == 3
    lut[bad] &= ~np.uint8(2 << sh) & 0xFF
rb = (L + 31) // 32 * 8
packed = lut[g.integers(0, 256, size=(N, rb), dtype=np.uint8)]
if L % 32:
    packed[:, L // 4 + 1:] = 0
    packed[:, L // 4] &= (1 << (2 * (L % 4))) - 1
e = E.Engine(n_snps=L, width=32, depth=1, latent_dim=2, batch_size=32, seed=1)
e.load_genotypes(packed)
e.pca_gram("patterson")  # warm-up (module load, allocations)
t0 = time.perf_counter()
G = e.pca_gram("patterson")
t1 = time.perf_counter()
coords, evr = P.coords_from_gram(G, ncomp)
t2 = time.perf_counter()
nt = (N + 127) // 128
flop = 2.0 * L * 128 * 128 * (nt * (nt + 1) // 2)
print(f"{name}: N={N} L={L}  gram call {1e3 * (t1 - t0):.1f} ms (incl. {N * N * 8 / 1e6:.0f} MB D2H)  "
      f"= {flop / (t1 - t0) / 1e12:.1f} TFLOP/s f64 on {flop / 1e12:.2f} TFLOP;  host eigh {1e3 * (t2 - t1):.0f} ms")
print("  explained variance ratio:", np.array2string(evr[:5], precision=5))
