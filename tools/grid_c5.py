"""BASELINE config 5 as a RUN: `popvae.py --search_network_sizes --width_range 32,256,512 --depth_range 3,5,8` (popvae.py:275-380)
on 1000 samples x 200 k SNPs of synthetic three-population genotypes handed over as a .popvae.hdf5 cache, one GPU, the nine
networks one after the other.  Prints the wall time per cell and the _param_grid.csv.
python tools/grid_c5.py [n_samples] [n_snps] [extra popvae.py flags ...]"""
import os, sys, time, tempfile, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import cli
from popvae_amd.weights_io import write_popvae_cache

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
extra = sys.argv[3:]
g = np.random.default_rng(20240001)
pop = np.arange(N) % 3
f0 = np.clip(g.beta(.5, .5, L), .02, .98)
f = np.clip(f0[None] + 0.1 * g.normal(size=(3, L)), .02, .98)
dc = (g.binomial(2, f[pop]) * 0.5).astype(np.float64)
d = tempfile.mkdtemp()
infile = os.path.join(d, "synth.popvae.hdf5")
write_popvae_cache(infile, dc, np.array([f"s{i:04d}" for i in range(N)], dtype=object))
del dc
out = os.path.join(d, "grid")
t0 = time.perf_counter()
rc = cli.main(["--infile", infile, "--out", out, "--seed", "42", "--search_network_sizes", "--width_range", "32,256,512",
               "--depth_range", "3,5,8"] + extra)
wall = time.perf_counter() - t0
grid = open(out + "_param_grid.csv").read()
cells = [{"width": w, "depth": dd, "seconds": round(s, 3), "epochs": ep, "ms_per_epoch": round(1e3 * s / ep, 2)}
         for w, dd, s, ep in cli.GRID_TIMINGS]
print(json.dumps({"what": f"popvae.py --search_network_sizes --width_range 32,256,512 --depth_range 3,5,8 on {N} x {L} synthetic "
                          "genotypes (.popvae.hdf5 input), one MI355X, cells run one after the other (grid: max_epochs / 4 epochs, "
                          "patience / 4, popvae.py:277,354); then the final fit at the chosen size",
                  "rc": rc, "wall_seconds_whole_command": round(wall, 2), "cells": cells,
                  "param_grid_csv": grid.splitlines()}, indent=1))
