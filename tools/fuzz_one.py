#!/usr/bin/env python3
"""One case of tests/test_gpu_random_shapes.py::test_multi_tile_equals_tile_path, per-tensor errors of the moments, with the kernel
switches given on the command line (KEY=VALUE ...) applied to the multi-tile run:  python tools/fuzz_one.py 17 [PVAE_NO_DEC3B=1]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine, prep
seed = int(sys.argv[1])
extra = dict(a.split("=", 1) for a in sys.argv[2:])
g = np.random.default_rng(1000 + seed)
L = int(g.choice([33, 64, 95, 200, 777, 1500, 4000, 9000])); W = int(g.choice([32, 64, 96, 128])); D = int(g.integers(1, 5))
K = int(g.choice([2, 3, 8])); B = int(g.choice([33, 48, 64, 65, 100, 128, 129, 200, 300])); n = int(g.integers(max(1, B // 3), B + 1))
N = B + 10
f = np.clip(g.beta(.5, .5, L), .02, .98)
dc = g.binomial(2, f, size=(N, L)) * 0.5
packed = prep.pack_dc(dc)
steps = [(g.choice(N, m, replace=False).astype(np.int32), g.normal(size=(m, K)).astype(np.float32)) for m in (n, B, max(1, n // 2))]
print("case", seed, dict(L=L, W=W, D=D, K=K, B=B, n=n), extra)
outs = []
for tiles in (False, True):
    for k in list(extra) + ["PVAE_NO_MULTITILE"]:
        os.environ.pop(k, None)
    if tiles:
        os.environ["PVAE_NO_MULTITILE"] = "1"
    else:
        os.environ.update(extra)
    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=B, seed=3)
    e.load_genotypes(packed); e.init_weights(11)
    if os.environ.get('LR'):
        e.lr = float(os.environ['LR'])
    res = [e.train_step(r, ep)[0] for r, ep in steps[:1]]
    m1 = e.get_moments(0)
    res += [e.train_step(r, ep)[0] for r, ep in steps[1:2]]
    outs.append((res, m1, e.get_moments(0), e.get_moments(1)))
    e.close()
(ra, m1a, ma, va), (rb, m1b, mb, vb) = outs
print("losses", ra, rb)
for name, xa, xb in (("m after step 1", m1a, m1b), ("m after step 2", ma, mb), ("v after step 2", va, vb)):
    print(name, " ".join(f"{np.abs(a - b).max() / max(np.abs(b).max(), 1e-30):.2e}" for a, b in zip(xa, xb)))
