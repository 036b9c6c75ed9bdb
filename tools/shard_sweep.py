#!/usr/bin/env python3
"""Shard-size sweep: one GPU runs the train step of an SNP shard of BASELINE's C4 (1M SNPs, width 128) and C3 (500 k SNPs,
width 512) at 1/1, 1/2, 1/4 and 1/8 of the SNP axis -- what ONE rank of a 1 / 2 / 4 / 8-GPU run computes per step -- and
reports the step time, the per-kernel times and the compute-only speed-up t(L) / t(L / G) these imply.
A PROJECTION, NOT A MEASUREMENT of multi-GPU scaling: the two all-reduces of a step (16 KB / 64 KB, latency-bound) and any
skew between ranks are not in it.   python tools/shard_sweep.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import engine as E  # noqa: E402

STEPS = 71


def run(L, W, D, timing):
    N, B = 600, 32
    g = np.random.default_rng(0)
    rb = (L + 31) // 32 * 8
    packed = g.integers(0, 256, size=(N, rb), dtype=np.uint8) & 0x55  # codes 0 / 1
    e = E.Engine(n_snps=L, width=W, depth=D, latent_dim=2, batch_size=B, seed=1)
    e.load_genotypes(packed)
    e.init_weights(1)
    e.lr = 1e-4
    rows = np.concatenate([g.permutation(N) for _ in range(STEPS * B // N + 2)])[:STEPS * B].astype(np.int32)
    e.train_epoch(rows[:8 * B], np.zeros(0, np.int32))
    e.set_timing(timing)
    best = None
    for _ in range(3):
        e.synchronize()
        t0 = time.perf_counter()
        e.train_epoch(rows, np.zeros(0, np.int32))
        dt = (time.perf_counter() - t0) / STEPS
        best = dt if best is None else min(best, dt)
    kern = {}
    if timing:
        for k in ("gather", "enc_fwd", "narrow_fwd", "dec", "reduce", "narrow_bwd", "enc_bwd"):
            ms, n = e.get_timing(k)
            if n:
                kern[k] = round(ms / n * 1e3, 2)
    e.close()
    return best * 1e3, kern


out = {"_about": __doc__.strip(), "steps_per_timing": STEPS, "shapes": []}
for name, L0, W, D in (("C4", 1_000_000, 128, 6), ("C3", 500_000, 512, 6)):
    base = None
    for G in (1, 2, 4, 8):
        L = L0 // G
        ms, _ = run(L, W, D, False)          # step time without events (an event pair costs the stream 5-10 us)
        _, kern = run(L, W, D, True)         # per-kernel HIP-event times, a second run
        Wp = (W + 31) // 32 * 32 if W <= 128 else (256 if W <= 256 else 512)
        wide = 48 * L * Wp + 24 * L + 3 * 32 * L // 4
        base = ms if base is None else base
        row = {"workload": name, "gpus_projected": G, "n_snps_per_rank": L, "width": W, "ms_per_step": round(ms, 4),
               "kernel_us": kern, "wide_algorithmic_bytes": wide, "wide_hbm_frac_step": round(wide / (ms * 1e-3) / 8e12, 3),
               "compute_only_speedup_vs_1": round(base / ms, 2)}
        out["shapes"].append(row)
        print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
