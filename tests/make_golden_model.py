#!/usr/bin/env python3
"""Writes tests/golden/model_steps.npz: a tiny VAE (L=96 SNPs, width 32, depth 3, latent 2, 11 samples) stepped
three times by the float64 oracle (oracle/popvae_oracle.py), for both BCE forms (K5).  The reference has no
golden vectors of its own (no tests, TF/Keras not installable), so this fixture pins the restatement against
silent drift; tests/test_oracle_math.py checks the oracle against it and tests/test_gpu_parity.py the HIP engine."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import popvae_oracle as O  # noqa: E402

L, W, D, K, N = 96, 32, 3, 2, 11
g = np.random.default_rng(20240607)
codes = g.binomial(2, np.clip(g.beta(.5, .5, L), .05, .95), size=(N, L)).astype(np.uint8)
w0 = [x.astype(np.float32) for x in O.init_weights(L, W, D, K, seed=5, dtype=np.float64)]
for t in w0:
    if t.ndim == 1:
        t += g.normal(0, 0.05, t.shape).astype(np.float32)
rows = [g.choice(N, n, replace=False).astype(np.int32) for n in (11, 7, 11)]
eps = [g.normal(size=(len(r), K)).astype(np.float32) for r in rows]
out = {"codes": codes, "dims": np.array([L, W, D, K, N])}
for i, t in enumerate(w0):
    out[f"w0_{i}"] = t
for s in range(3):
    out[f"rows_{s}"], out[f"eps_{s}"] = rows[s], eps[s]
for mode in ("logits", "probs"):
    w = [x.astype(np.float64) for x in w0]
    opt = O.KerasAdam(w)
    dc = codes * 0.5
    for s in range(3):
        c = O.forward(w, dc[rows[s]], eps[s].astype(np.float64), D, K, bce_mode=mode)
        gr = O.backward(w, c, D, K, bce_mode=mode)
        out[f"{mode}_loss_{s}"] = np.array([c["loss"], c["recon"].mean(), c["kl"].mean()])
        if s == 0:
            for i, t in enumerate(gr):
                out[f"{mode}_grad0_{i}"] = t
        opt.step(w, gr)
    for i, t in enumerate(w):
        out[f"{mode}_w3_{i}"], out[f"{mode}_m3_{i}"], out[f"{mode}_v3_{i}"] = t, opt.m[i], opt.v[i]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "model_steps.npz"), **out)
print("written", len(out), "arrays")
