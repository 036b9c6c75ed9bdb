"""The drop-in command line end to end at BASELINE config C2's size (1000 samples x 100 k SNPs, every default of popvae.py) on
synthetic three-population genotypes handed over as a .popvae.hdf5 cache (popvae.py:205-209): wall time of the whole run and
whether the latent means separate the populations.  python tools/cli_e2e.py [n_samples] [n_snps] [extra popvae.py flags ...]"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from popvae_amd import cli
from popvae_amd.weights_io import write_popvae_cache

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
extra = sys.argv[3:]
g = np.random.default_rng(20240001)
pop = np.arange(N) % 3
f0 = np.clip(g.beta(.5, .5, L), .02, .98)
f = np.clip(f0[None] + 0.1 * g.normal(size=(3, L)), .02, .98)
dc = (g.binomial(2, f[pop]) * 0.5).astype(np.float64)
d = tempfile.mkdtemp()
infile = os.path.join(d, "synth.popvae.hdf5")
t0 = time.perf_counter()
write_popvae_cache(infile, dc, np.array([f"s{i:04d}" for i in range(N)], dtype=object))
print(f"cache written in {time.perf_counter() - t0:.1f} s ({os.path.getsize(infile) / 1e6:.0f} MB)", flush=True)
out = os.path.join(d, "run")
t0 = time.perf_counter()
if os.environ.get("PROFILE"):  # where the host side of the run spends its time
    import cProfile, pstats
    pr = cProfile.Profile()
    rc = pr.runcall(cli.main, ["--infile", infile, "--out", out, "--seed", "42"] + extra)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(int(os.environ["PROFILE"]))
else:
    rc = cli.main(["--infile", infile, "--out", out, "--seed", "42"] + extra)
wall = time.perf_counter() - t0
hist = np.loadtxt(out + "_history.txt", skiprows=1)
lat = [r.split("\t") for r in open(out + "_latent_coords.txt").read().splitlines()[1:]]
xy = np.array([[float(r[0]), float(r[1])] for r in lat])
ids = np.array([int(r[4][1:]) for r in lat])
cent = np.array([xy[(ids % 3) == k].mean(0) for k in range(3)])
assign = np.argmin(((xy[:, None, :] - cent[None]) ** 2).sum(-1), axis=1)
acc = float((assign == ids % 3).mean())
print(f"rc {rc}; {len(hist)} epochs in {wall:.1f} s wall ({wall / len(hist) * 1e3:.1f} ms per epoch, everything included); "
      f"loss {hist[0, 1]:.1f} -> {hist[-1, 1]:.1f}, val_loss {hist[0, 2]:.1f} -> {hist[:, 2].min():.1f}; "
      f"nearest-centroid population accuracy of the latent means {acc:.3f}")
