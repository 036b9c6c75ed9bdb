#!/usr/bin/env python3
"""A/B of an engine switch on identical inputs: a few train steps at a small learning rate, per-tensor difference of the Adam
moments between the default engine and one created with the given environment.
python tools/ab_steps.py L W D steps KEY=VALUE [...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E
L, W, D, steps = (int(x) for x in sys.argv[1:5])
env = dict(a.split("=", 1) for a in sys.argv[5:])
N, B, K = 64, 32, 2
g = np.random.default_rng(0)
lut = np.arange(256, dtype=np.uint8)
for sh in (0, 2, 4, 6):
    bad = ((lut >> sh) & 3) == 3
    lut[bad] &= ~np.uint8(2 << sh) & 0xFF
packed = lut[g.integers(0, 256, size=(N, (L + 31) // 32 * 8), dtype=np.uint8)]
if L % 32:
    packed[:, L // 4 + 1:] = 0
    packed[:, L // 4] &= (1 << (2 * (L % 4))) - 1
batches = [(g.choice(N, B if i % 2 == 0 else 19, replace=False).astype(np.int32),) for i in range(steps)]
batches = [(r[0], g.normal(size=(len(r[0]), K)).astype(np.float32)) for r in batches]
outs = []
for use in (False, True):
    for k in env:
        os.environ.pop(k, None)
    if use:
        os.environ.update(env)
    e = E.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=B, seed=1, lr=1e-5)
    e.load_genotypes(packed); e.init_weights(1)
    losses = [e.train_step(r, ep)[0] for r, ep in batches]
    outs.append((losses, e.get_moments(0), e.get_moments(1)))
    e.close()
(la, ma, va), (lb, mb, vb) = outs
print("losses default", la)
print("losses", env, lb)
print("m", " ".join(f"{np.abs(a - b).max() / max(np.abs(a).max(), 1e-30):.1e}" for a, b in zip(ma, mb)))
print("v", " ".join(f"{np.abs(a - b).max() / max(np.abs(a).max(), 1e-30):.1e}" for a, b in zip(va, vb)))
