#!/usr/bin/env python3
"""Kernel-level timing at a BASELINE shape without torch: python tools/kbench.py [C4|C2] [steps] [batch_size]
Prints per-kernel-category HIP-event averages and the implied GB/s."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E  # noqa: E402

SH = {"C4": (2504, 1_000_000, 128, 6), "C2": (1000, 100_000, 128, 6), "C1": (95, 3887, 128, 6),
      "C3": (929, 500_000, 512, 6), "C5a": (1000, 200_000, 256, 5), "C5b": (1000, 200_000, 32, 3)}
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
N, L, W, D = SH[name]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
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
e = E.Engine(n_snps=L, width=W, depth=D, latent_dim=2, batch_size=B, seed=1)
e.load_genotypes(packed)
e.init_weights(1)
if os.environ.get("LR"):
    e.lr = float(os.environ["LR"])
rows = np.concatenate([g.permutation(N) for _ in range(steps * B // N + 2)])[:steps * B].astype(np.int32)
e.train_epoch(rows[:5 * B], np.zeros(0, np.int32))
e.set_timing(not os.environ.get('NOTIMING'))  # NOTIMING=1: no events (they cost stream time) -- for rocprofv3 timelines
t0 = time.perf_counter()
loss, _ = e.train_epoch(rows, np.zeros(0, np.int32))
dt = time.perf_counter() - t0
Wp = (W + 31) // 32 * 32 if W <= 128 else (256 if W <= 256 else 512)
algo = {"dec": 24 * L * Wp + 32 * L, "enc_fwd": 4 * L * Wp + 8 * L, "enc_bwd": 24 * L * Wp + 8 * L}
print(f"{name} B={B}: {dt / steps * 1e3:.3f} ms/step  {steps * B / dt:.0f} samples/s  loss {loss:.3f}")
tot = 0
for k in ("gather", "enc_fwd", "narrow_fwd", "dec", "dec_upd", "reduce", "narrow_bwd", "enc_bwd"):
    ms, n = e.get_timing(k)
    if n:
        a = ms / n
        tot += a * n / steps
        extra = f"  {algo[k] / a / 1e6:8.0f} GB/s" if k in algo else ""
        print(f"  {k:11s} {a * 1e3:9.1f} us x {n / steps:.0f}{extra}")
print(f"  sum/step    {tot * 1e3:9.1f} us")
vr = np.resize(rows, 8 * B).reshape(8, B)  # validation batches have the batch size too
e.set_timing(False)
e.synchronize()
t0 = time.perf_counter()
for r in vr:
    e.eval_step(r, None, want_loss=False)
e.synchronize()
print(f"  eval step   {(time.perf_counter() - t0) / 8 * 1e3:.3f} ms ({B} rows)")
