"""fit loop on the GPU engine: trajectory parity with the oracle's ``fit`` under injected shuffle + noise,
and the command line end to end on a synthetic two-population VCF (the reference's pabu VCF cannot travel
to the GPU box)."""
import os

import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import cli, engine, prep
from popvae_amd import fit as F

pytestmark = pytest.mark.gpu


class Stepwise:
    """Engine adapter that runs an epoch as individual C-ABI train/eval steps with injected noise, the way
    the oracle's fit does (the C-side epoch loop draws its own Philox noise instead)."""

    def __init__(self, eng, eps_fn, B):
        self.e, self.eps_fn, self.B, self.epoch = eng, eps_fn, B, 0

    lr = property(lambda s: s.e.lr, lambda s, v: setattr(s.e, "lr", v))

    def train_epoch(self, rows, val_rows):
        K = self.e.latent_dim
        tl = [self.e.train_step(rows[i:i + self.B], self.eps_fn("train", self.epoch, st, (len(rows[i:i + self.B]), K)))[0]
              for st, i in enumerate(range(0, len(rows), self.B))]
        vl = [self.e.eval_step(val_rows[i:i + self.B], self.eps_fn("val", self.epoch, st, (len(val_rows[i:i + self.B]), K)))[0]
              for st, i in enumerate(range(0, len(val_rows), self.B))]
        self.epoch += 1
        return float(np.mean(tl)), float(np.mean(vl))

    def snapshot_best(self):
        self.e.snapshot_best()

    def encode(self, rows, eps=None):
        return self.e.encode(rows, np.zeros((len(rows), self.e.latent_dim), np.float32))


def _two_pop(N, L, seed):
    g = np.random.default_rng(seed)
    pop = np.arange(N) % 2
    f = np.clip(g.beta(.5, .5, L)[None] + 0.3 * (pop[:, None] - .5) * g.normal(size=L)[None], .02, .98)
    return g.binomial(2, f) * 0.5, pop


@pytest.mark.parametrize("N,B,ntrain", [(75, 32, 67),       # the reference's default batch size
                                         (150, 48, 131),     # --batch_size 48: two row tiles in one weight pass, ragged last batch
                                         (300, 128, 270)])   # --batch_size 128: four row tiles; 270 = 128 + 128 + 14
def test_fit_trajectory_matches_oracle(N, B, ntrain):
    L, W, D, K = 1500, 64, 3, 2
    dc, _ = _two_pop(N, L, 0)
    w = O.init_weights(L, W, D, K, seed=4, dtype=np.float32)
    train = np.random.default_rng(1).permutation(N)[:ntrain]
    test = np.setdiff1d(np.arange(N), train)

    def eps_fn(kind, epoch, step, shape):
        return np.random.default_rng([epoch, step, kind == "val"]).normal(size=shape).astype(np.float32)

    def perm_fn(epoch, n):
        return np.random.default_rng(100 + epoch).permutation(n)

    epochs, patience = 10, 4
    wo = [x.astype(np.float64) for x in w]
    ho, best_o, preds_o = O.fit(wo, dc, train, test, D, K, B, epochs, patience, perm_fn, eps_fn, prediction_freq=5)

    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=B, seed=0)
    e.load_genotypes(prep.pack_dc(dc))
    e.set_weights(w)
    hg, cbs = F.fit(Stepwise(e, eps_fn, B), train, test, epochs, patience, prediction_freq=5, n_samples=N,
                    samples=np.arange(N), out=None, perm_fn=perm_fn, verbose=0)
    assert len(hg["loss"]) == len(ho["loss"])
    np.testing.assert_allclose(hg["loss"], ho["loss"], rtol=2e-4)       # fp32 engine vs float64 oracle trajectory
    np.testing.assert_allclose(hg["val_loss"], ho["val_loss"], rtol=2e-4)
    np.testing.assert_array_equal(np.array(hg["lr"], np.float32), np.array(ho["lr"], np.float32))
    assert [ep for ep, _ in cbs["preds"].records] == [ep for ep, _ in preds_o]
    for (_, a), (_, b) in zip(cbs["preds"].records, preds_o):
        assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(b).max())
    # ModelCheckpoint + load_weights: restored weights = oracle's best-epoch weights
    e.restore_best()
    for a, b in zip(e.get_weights(), best_o):
        assert np.abs(a - b).max() <= 5e-4
    e.close()


def _write_vcf(path, gen, samples):
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
        for l in range(gen.shape[0]):
            calls = ["/".join("." if a < 0 else str(a) for a in gen[l, i]) for i in range(gen.shape[1])]
            fh.write(f"1\t{l + 1}\t.\tA\tC\t.\tPASS\t.\tGT\t" + "\t".join(calls) + "\n")


def test_cli_end_to_end(tmp_path, monkeypatch):
    N, L0 = 60, 2500
    g = np.random.default_rng(5)
    pop = np.arange(N) % 2
    f = np.clip(g.beta(.6, .6, L0)[None] + 0.45 * (pop[:, None] - .5) * g.normal(size=L0)[None], .01, .99)
    gen = (g.random((N, L0, 2)) < f[:, :, None]).astype(np.int8).transpose(1, 0, 2).copy()
    gen[g.random((L0, N)) < 0.05] = -1
    samples = [f"ind{i:02d}" for i in range(N)]
    vcf = str(tmp_path / "toy.vcf")
    _write_vcf(vcf, gen, samples)
    out = str(tmp_path / "run")
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = cli.main(["--infile", vcf, "--out", out, "--seed", "42", "--max_epochs", "40", "--patience", "10",
                   "--width", "64", "--depth", "3", "--save_weights", "--PCA", "--n_pc_axes", "5"])
    assert rc == 0
    # the integer path is bit-exact: same kept sites / imputed matrix / split as numpy's legacy stream gives
    rs = np.random.RandomState(42)
    dc_o, keep = O.filter_and_impute(gen, rs)
    train_o, test_o = O.train_test_split(N, 0.9, rs)
    hist = open(out + "_history.txt").read().splitlines()
    assert hist[0] == "\tloss\tval_loss\tlr" and len(hist) >= 12
    loss = [float(x.split("\t")[1]) for x in hist[1:]]
    assert loss[-1] < 0.97 * loss[0]
    lat = [x.split("\t") for x in open(out + "_latent_coords.txt").read().splitlines()]
    assert lat[0] == ["mean1", "mean2", "sd1", "sd2", "sampleID"] and [r[4] for r in lat[1:]] == samples
    xy = np.array([[float(r[0]), float(r[1])] for r in lat[1:]])
    # two populations separate linearly in the latent means (README.md:103 is the qualitative criterion)
    mu0, mu1 = xy[pop == 0].mean(0), xy[pop == 1].mean(0)
    proj = (xy - (mu0 + mu1) / 2) @ (mu1 - mu0)
    assert ((proj > 0) == (pop == 1)).mean() > 0.9
    preds = open(out + "_training_preds.txt").read().splitlines()
    assert preds[0] == "0\t1\tsampleID\tepoch" and (len(preds) - 1) % N == 0
    assert os.path.exists(out + "_weights.hdf5") or os.path.exists(out + "_weights.npz")
    if os.path.exists(out + "_weights.npz"):
        wz = np.load(out + "_weights.npz")
        assert wz["enc_dense0__kernel"].shape == (dc_o.shape[1], 64)
    # --PCA (popvae.py:528-540, 561-563): same coordinates as the oracle's allel.pca restatement on the imputed matrix
    pc = [x.split("\t") for x in open(out + "_pca.txt").read().splitlines()]
    assert pc[0] == ["PC1", "PC2", "PC3", "PC4", "PC5", "sampleID"] and [r[5] for r in pc[1:]] == samples
    got = np.array([[float(v) for v in r[:5]] for r in pc[1:]])
    want = O.pca(dc_o, "Patterson", 5)[0]
    sgn = np.sign(np.sum(got * want, axis=0))
    assert np.abs(got * sgn - want).max() <= 1e-8 * np.abs(want).max()
    rt = np.loadtxt(out + "_runtimes.txt")
    assert rt.shape == (2,) and (rt > 0).all()


def test_cli_grid_search_small(tmp_path, monkeypatch):
    N, L0 = 40, 600
    g = np.random.default_rng(6)
    f = np.clip(g.beta(.6, .6, L0), .05, .95)
    gen = (g.random((L0, N, 2)) < f[:, None, None]).astype(np.int8)
    vcf = str(tmp_path / "toy.vcf")
    _write_vcf(vcf, gen, [f"i{i}" for i in range(N)])
    out = str(tmp_path / "grid")
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = cli.main(["--infile", vcf, "--out", out, "--seed", "1", "--max_epochs", "8", "--patience", "8",
                   "--search_network_sizes", "--width_range", "32,64", "--depth_range", "2,3", "--prediction_freq", "2"])
    assert rc == 0
    rows = open(out + "_param_grid.csv").read().splitlines()
    assert rows[0] == "width,depth,val_loss" and [r.split(",")[:2] for r in rows[1:]] == \
        [["32", "2"], ["32", "3"], ["64", "2"], ["64", "3"]]
