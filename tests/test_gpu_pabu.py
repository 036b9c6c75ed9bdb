"""BASELINE.json configs[0]: the reference's own test run, `popvae.py --test` = `--infile data/pabu/pabu_test_genotypes.vcf
--out out/pabu_test --seed 42` (popvae.py:175-180) with every default (depth 6, width 128, patience 50, 500 epochs), through
the drop-in CLI on the GPU.  The data file and its sample sheet are the reference's (tests/golden/, gzip-compressed).

What can be pinned: the integer front end bit for bit (kept sites, imputed matrix, split -- golden JSON made with numpy's
legacy stream), the file formats, and the README's qualitative criterion (README.md:99-103: the latent space separates the
eastern and western populations, `k2pop` in the sample sheet).  The loss trajectory itself is **parity unpinned**: TF's
shuffle and noise streams cannot be reproduced outside TF and the reference ships no output of this run."""
import hashlib
import json
import os

import numpy as np
import pytest

from popvae_amd import cli, prep

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
VCF = os.path.join(HERE, "golden", "pabu_test_genotypes.vcf.gz")
SHEET = os.path.join(HERE, "golden", "pabu_test_sample_data.txt")
GOLD = json.load(open(os.path.join(HERE, "golden", "pabu_prep_seed42.json")))


def test_integer_front_end_bit_exact_on_the_gpu_box():
    """The integer path against the golden fixture on this machine too (same check as the CPU suite's: the native VCF reader,
    biallelic / doubleton filter, binomial imputation on numpy's legacy stream, 2-bit packing, split)."""
    gen, samples = prep.read_vcf(VCF)
    assert hashlib.sha256(gen.tobytes()).hexdigest() == GOLD["gt_sha256"]
    rng = prep.LegacyRNG(42)
    packed, L, keep, af, nimp = prep.filter_impute_pack(gen, rng)
    assert L == GOLD["n_snps"] and nimp == 22384
    assert hashlib.sha256(packed.tobytes()).hexdigest() == GOLD["packed_sha256"]
    tr, te = prep.train_test_split(gen.shape[1], 0.9, rng)
    assert tr.tolist() == GOLD["train"] and te.tolist() == GOLD["test"]


def test_reference_test_run_on_pabu(tmp_path, monkeypatch, capsys):
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    out = str(tmp_path / "pabu_test")
    assert cli.main(["--infile", VCF, "--out", out, "--seed", "42"]) == 0
    log = capsys.readouterr().out
    assert "running on 3887 SNPs" in log                      # popvae.py:272; SURVEY 8a P3
    rows = [l.split("\t") for l in open(SHEET, encoding="utf-8-sig").read().splitlines()]
    col = {name: i for i, name in enumerate(rows[0])}
    k2 = {r[col["sampleID"]]: r[col["k2pop"]] for r in rows[1:]}

    lat = [x.split("\t") for x in open(out + "_latent_coords.txt").read().splitlines()]
    assert lat[0] == ["mean1", "mean2", "sd1", "sd2", "sampleID"] and len(lat) == 1 + GOLD["n_samples"]
    ids = [r[4] for r in lat[1:]]
    assert ids[:3] == GOLD["samples_first3"] and ids[-1] == GOLD["samples_last"]
    # the validation samples the CLI printed are the golden split's (ascending complement of the seeded choice)
    for i in GOLD["test"]:
        assert ids[i] in log

    hist = [x.split("\t") for x in open(out + "_history.txt").read().splitlines()]
    assert hist[0] == ["", "loss", "val_loss", "lr"]
    loss = np.array([float(r[1]) for r in hist[1:]]); vloss = np.array([float(r[2]) for r in hist[1:]])
    assert np.isfinite(loss).all() and np.isfinite(vloss).all()
    assert 20 <= len(loss) <= 500 and loss.min() < 0.9 * loss[0] and vloss.min() < vloss[0]
    lr = np.array([float(r[3]) for r in hist[1:]])
    assert abs(lr[0] - 1e-3) < 1e-9 and (np.diff(lr) <= 1e-12).all()       # ReduceLROnPlateau only ever halves

    preds = open(out + "_training_preds.txt").read().splitlines()
    assert preds[0] == "0\t1\tsampleID\tepoch" and (len(preds) - 1) % GOLD["n_samples"] == 0
    assert not os.path.exists(out + "_weights.hdf5") and not os.path.exists(out + "_weights.npz")  # popvae.py:525-526

    # README.md:99-103: east / west separate in the latent means
    xy = np.array([[float(r[0]), float(r[1])] for r in lat[1:]])
    pop = np.array([k2[i] for i in ids])
    assert set(pop) == {"E", "W"}
    mu_e, mu_w = xy[pop == "E"].mean(0), xy[pop == "W"].mean(0)
    proj = (xy - (mu_e + mu_w) / 2) @ (mu_e - mu_w)
    assert ((proj > 0) == (pop == "E")).mean() >= 0.9


def test_test_flag_runs_in_this_tree(tmp_path):
    """`python popvae.py --test` itself (popvae.py:175-180), from a working directory that holds no data/pabu/: the flag spawns
    the run on the repository's copy of the fixture and leaves out/pabu_test_* behind, as README.md:24-27 describes."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "popvae.py"), "--test"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=280)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    assert "running on 3887 SNPs" in p.stdout
    lat = open(tmp_path / "out" / "pabu_test_latent_coords.txt").read().splitlines()
    assert lat[0].split("\t") == ["mean1", "mean2", "sd1", "sd2", "sampleID"] and len(lat) == 1 + GOLD["n_samples"]
    assert os.path.getsize(tmp_path / "out" / "pabu_test_history.txt") > 0


def test_cache_and_weight_files_through_the_cli(tmp_path, monkeypatch, capsys):
    """--save_allele_counts / .popvae.hdf5 input (popvae.py:205-209, 241-248) and --save_weights (popvae.py:429-435,
    525-526) through the CLI: the cache written next to the input file reproduces the run that wrote it (same SNPs, same
    split, same noise and shuffle seeds -> same history), and the saved best weights give back the latent coordinates when
    a later run starts from them (--init_weights, zero epochs of training budget used: max_epochs 1 with the same data)."""
    import shutil
    from popvae_amd import weights_io
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    vcf = str(tmp_path / "pabu.vcf.gz")
    shutil.copy(VCF, vcf)
    out1, out2 = str(tmp_path / "a"), str(tmp_path / "b")
    common = ["--seed", "42", "--max_epochs", "12", "--patience", "12", "--save_weights"]
    assert cli.main(["--infile", vcf, "--out", out1, "--save_allele_counts"] + common) == 0
    cache = vcf + ".popvae.hdf5"
    dc, samples = weights_io.read_popvae_cache(cache)
    assert dc.shape == (GOLD["n_samples"], 3887) and set(np.unique(dc)) <= {0.0, 0.5, 1.0}
    assert list(samples[:3]) == GOLD["samples_first3"]
    w = weights_io.load_weights_file(out1 + "_weights.hdf5")
    assert len(w) == 4 * 6 + 6 and w[0].shape == (3887, 128) and w[-2].shape == (128, 3887)
    capsys.readouterr()
    # the cache holds the imputed matrix: loading it skips the filters and the imputation draws, so the split differs from
    # the VCF run's (as in the reference, whose RNG stream is then shorter too) -- file formats and finiteness are checked
    assert cli.main(["--infile", cache, "--out", out2, "--init_weights", out1 + "_weights.hdf5"] + common) == 0
    log = capsys.readouterr().out
    assert "running on 3887 SNPs" in log and "initialising from" in log
    h1 = [x.split("\t") for x in open(out1 + "_history.txt").read().splitlines()]
    h2 = [x.split("\t") for x in open(out2 + "_history.txt").read().splitlines()]
    # the second run starts from the first run's best weights: its first epoch is already far below a fresh model's
    assert float(h2[1][1]) < 0.9 * float(h1[1][1])
    assert os.path.exists(out2 + "_weights.hdf5")


def test_network_size_grid_as_a_run(tmp_path, monkeypatch, capsys):
    """BASELINE config 5 is `--search_network_sizes` over widths {32, 256, 512} x depths {3, 5, 8} (popvae.py:275-380): here the
    same grid as a RUN through the command line on one GPU, at reduced size and epochs (200 samples x 20 k SNPs from a
    .popvae.hdf5 cache; tools/grid_c5.py is the full-size run whose timings sit under profiles/): nine networks one after the
    other, the three narrow-stack paths (op-table chain at width 32, per-layer kernels at 256 / 512) and both grouped decoder
    kernels; _param_grid.csv in the reference's width-major row order with finite losses, then the final fit at the argmin."""
    from popvae_amd.weights_io import write_popvae_cache
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    g = np.random.default_rng(5)
    N, L = 200, 20_000
    pop = np.arange(N) % 3
    f = np.clip(np.clip(g.beta(.5, .5, L), .02, .98)[None] + 0.1 * g.normal(size=(3, L)), .02, .98)
    dc = (g.binomial(2, f[pop]) * 0.5).astype(np.float64)
    infile = str(tmp_path / "synth.popvae.hdf5")
    write_popvae_cache(infile, dc, np.array([f"s{i:03d}" for i in range(N)], dtype=object))
    out = str(tmp_path / "grid")
    assert cli.main(["--infile", infile, "--out", out, "--seed", "42", "--max_epochs", "24", "--patience", "8",
                     "--search_network_sizes", "--width_range", "32,256,512", "--depth_range", "3,5,8"]) == 0
    rows = open(out + "_param_grid.csv").read().splitlines()
    assert rows[0] == "width,depth,val_loss"
    assert [tuple(int(x) for x in r.split(",")[:2]) for r in rows[1:]] == [(w, d) for w in (32, 256, 512) for d in (3, 5, 8)]
    vals = np.array([float(r.split(",")[2]) for r in rows[1:]])
    assert np.isfinite(vals).all() and (vals > 0).all()
    log = capsys.readouterr().out
    best = rows[1 + int(np.argmin(vals))].split(",")
    assert f"width = {best[0]}\ndepth = {best[1]}" in log        # popvae.py:375-380
    assert len(cli.GRID_TIMINGS) >= 9 and all(ep <= 6 for _, _, _, ep in cli.GRID_TIMINGS[-9:])  # int(24 / 4) epochs per cell
    lat = open(out + "_latent_coords.txt").read().splitlines()
    assert len(lat) == 1 + N
