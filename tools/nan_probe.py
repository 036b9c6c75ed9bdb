#!/usr/bin/env python3
"""Per-step losses at C4 with Keras' default lr = 1e-3 on uniformly random genotypes (what tools/kbench.py steps on), batch B,
through the one-weight-pass kernels and through 32-row tiles: where does the loss stop being finite, and do the two paths agree
up to there?   python tools/nan_probe.py [B] [steps] [L]"""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
if os.environ.get("_CHILD"):
    from popvae_amd import engine as E
    N = 2504
    g = np.random.default_rng(0)
    lut = np.arange(256, dtype=np.uint8)
    for sh in (0, 2, 4, 6):
        bad = ((lut >> sh) & 3) == 3
        lut[bad] &= ~np.uint8(2 << sh) & 0xFF
    packed = lut[g.integers(0, 256, size=(N, (L + 31) // 32 * 8), dtype=np.uint8)]
    e = E.Engine(n_snps=L, width=128, depth=6, latent_dim=2, batch_size=B, seed=1)
    e.load_genotypes(packed)
    e.init_weights(1)
    rows = np.concatenate([g.permutation(N) for _ in range(steps * B // N + 2)])[:steps * B].astype(np.int32).reshape(steps, B)
    out = []
    for r in rows:
        eps = np.random.default_rng(int(r[0])).standard_normal((B, 2)).astype(np.float32)
        out.append([float(x) for x in e.train_step(r, eps)])
    print(json.dumps(out))
    sys.exit(0)
res = {}
for name, env in (("one pass", {}), ("tiles", {"PVAE_NO_MULTITILE": "1"})):
    p = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=dict(os.environ, _CHILD="1", **env), capture_output=True, text=True, timeout=300)
    if p.returncode:
        print(p.stderr[-2000:]); sys.exit(1)
    res[name] = json.loads(p.stdout.strip().splitlines()[-1])
for s in range(steps):
    a, b = res["one pass"][s], res["tiles"][s]
    print(f"step {s}: one pass loss {a[0]:.6g} recon {a[1]:.6g} kl {a[2]:.6g} | tiles loss {b[0]:.6g} recon {b[1]:.6g} kl {b[2]:.6g}")
