"""BASELINE.json configs[0]: the reference's own test run, `popvae.py --test` = `--infile data/pabu/pabu_test_genotypes.vcf
--out out/pabu_test --seed 42` (popvae.py:175-180) with every default (depth 6, width 128, patience 50, 500 epochs), through
the drop-in CLI on the GPU.  The data file and its sample sheet are the reference's (tests/golden/, gzip-compressed).

What can be pinned: the integer front end bit for bit (kept sites, imputed matrix, split -- golden JSON made with numpy's
legacy stream), the file formats, and the README's qualitative criterion (README.md:99-103: the latent space separates the
eastern and western populations, `k2pop` in the sample sheet).  The loss trajectory itself is **parity unpinned**: TF's
shuffle and noise streams cannot be reproduced outside TF and the reference ships no output of this run."""
import json
import os

import numpy as np
import pytest

from popvae_amd import cli

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
VCF = os.path.join(HERE, "golden", "pabu_test_genotypes.vcf.gz")
SHEET = os.path.join(HERE, "golden", "pabu_test_sample_data.txt")
GOLD = json.load(open(os.path.join(HERE, "golden", "pabu_prep_seed42.json")))


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
