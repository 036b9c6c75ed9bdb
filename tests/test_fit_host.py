"""Host side of the fit loop (popvae_amd.fit) against the oracle's restatement of the Keras callbacks
(K10-K12) and against hand-derived expectations; file formats of the reference's writers.  CPU only."""
import io
import os

import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import fit as F


class FakeEngine:
    """Scripted engine: train_epoch returns a prepared (loss, val_loss) sequence."""

    def __init__(self, val_losses, lr=1e-3, K=2, N=7):
        self.val = list(val_losses)
        self._lr = np.float32(lr)
        self.i = 0
        self.snapshots = []
        self.perms = []
        self.K, self.N = K, N
        self.restored = False

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, v):
        self._lr = np.float32(v)

    def train_epoch(self, perm_rows, val_rows):
        self.perms.append(np.array(perm_rows))
        v = self.val[self.i]
        self.i += 1
        return 2.0 * v, v

    def snapshot_best(self):
        self.snapshots.append(self.i - 1)

    def restore_best(self):
        self.restored = True

    def encode(self, rows, eps=None):
        mu = np.arange(len(rows) * self.K, dtype=np.float32).reshape(len(rows), self.K) + self.i
        return mu, mu * 2, mu * 3

    def close(self):
        pass


def _oracle_trace(val, patience, lr_patience, lr0=1e-3, use_ck=True):
    ck, es, rl = O.ModelCheckpointBest(), O.EarlyStopping(patience), O.ReduceLROnPlateau(lr_patience)
    lr = np.float32(lr0)
    lrs, saves, stop_at = [], [], None
    for ep, v in enumerate(val):
        if use_ck and ck.on_epoch_end(ep, v):
            saves.append(ep)
        stop = es.on_epoch_end(ep, v)
        lrs.append(np.float32(lr))
        lr = rl.on_epoch_end(ep, v, lr)
        if stop:
            stop_at = ep
            break
    return lrs, saves, stop_at


SEQS = [
    ("improving", list(np.linspace(10, 1, 30)), 8),
    ("plateau", [5, 4, 3] + [3.5] * 40, 8),
    ("noisy", list(5 + np.sin(np.arange(60)) * 0.5 + np.arange(60) * 0.01), 12),
    ("nan", [5, 4, float("nan"), 3.9, 3.8] + [4.0] * 30, 8),
    ("equal-is-not-better", [3.0] * 20, 4),
    ("float-patience-grid", [5, 4, 3] + [3.5] * 40, 12.5),   # popvae.py:277: patience / 4 = 12.5
    ("patience-0", [3, 2, 4, 5], 0),
]


@pytest.mark.parametrize("name,val,patience", SEQS)
def test_callback_state_machines_match_oracle(name, val, patience):
    lr_pat = int(patience / 4)
    eng = FakeEngine(val)
    hist, cbs = F.fit(eng, np.arange(5), np.arange(5, 7), max_epochs=len(val), patience=patience,
                      lr_patience=lr_pat, verbose=0)
    lrs, saves, stop_at = _oracle_trace(val, patience, lr_pat)
    n = len(hist["loss"])
    assert n == (stop_at + 1 if stop_at is not None else len(val))
    assert [float(x) for x in hist["lr"]] == [float(x) for x in lrs[:n]]
    assert eng.snapshots == saves
    assert cbs["checkpoint"].best_epoch == (saves[-1] if saves else -1)
    np.testing.assert_array_equal(np.array(hist["val_loss"]), np.array(val[:n], dtype=np.float32).astype(float))


def test_known_answers_k10_k11_k12():
    # K11: stops at wait == patience, never at epoch 0; grid patience 12.5 stops at wait == 13
    val = [1.0] + [2.0] * 40
    h, cbs = F.fit(FakeEngine(val), np.arange(4), np.arange(2), len(val), patience=12.5, lr_patience=3, verbose=0)
    assert len(h["loss"]) == 14 and cbs["earlystop"].stopped_epoch == 13
    # K12: lr halves after `patience` non-improving epochs, logged lr is the pre-reduction one
    assert [float(x) for x in h["lr"][:8]] == pytest.approx([1e-3] * 4 + [5e-4] * 3 + [2.5e-4], rel=1e-6)
    # K10: strict '<'
    eng = FakeEngine([2.0, 2.0, 1.0, 1.0])
    F.fit(eng, np.arange(4), np.arange(2), 4, patience=50, verbose=0)
    assert eng.snapshots == [0, 2]


def test_fit_uses_permutation_of_train_rows_and_prediction_freq(tmp_path):
    train = np.array([9, 3, 5, 1, 7])
    eng = FakeEngine([3, 2, 1, 0.5, 0.4, 0.3, 0.2], N=10)
    out = str(tmp_path / "run")
    hist, cbs = F.fit(eng, train, np.array([0, 2]), 7, patience=50, prediction_freq=3, n_samples=10,
                      samples=np.array([f"s{i}" for i in range(10)], dtype=object), out=out,
                      perm_fn=lambda ep, n: np.arange(n)[::-1], verbose=0)
    for p in eng.perms:
        assert list(p) == list(train[::-1])
    assert [e for e, _ in cbs["preds"].records] == [0, 3, 6]
    lines = open(out + "_training_preds.txt").read().splitlines()
    assert lines[0] == "0\t1\tsampleID\tepoch"          # popvae.py:456-466: int column labels, then the two names
    assert len(lines) == 1 + 3 * 10
    assert lines[1].split("\t")[2:] == ["s0", "0"] and lines[-1].split("\t")[2:] == ["s9", "6"]


def test_history_and_latent_writers(tmp_path):
    hist = {"loss": [3.5, 2.25], "val_loss": [4.0, 3.0], "lr": [np.float32(1e-3), np.float32(5e-4)]}
    p = str(tmp_path / "h.txt")
    F.write_history(p, hist)
    lines = open(p).read().splitlines()
    assert lines[0] == "\tloss\tval_loss\tlr"             # unnamed index column (popvae.py:504-505)
    assert lines[1].split("\t")[:3] == ["0", "3.5", "4.0"] and lines[1].split("\t")[3] == "0.001"
    assert lines[2].split("\t")[3] == "0.0005"
    mu = np.array([[1, 2], [3, 4]], np.float32)
    F.write_latent_coords(str(tmp_path / "l2.txt"), (mu, mu + 10, mu), ["a", "b"], 2)
    assert open(tmp_path / "l2.txt").read().splitlines() == \
        ["mean1\tmean2\tsd1\tsd2\tsampleID", "1.0\t2.0\t11.0\t12.0\ta", "3.0\t4.0\t13.0\t14.0\tb"]
    mu3 = np.arange(6, dtype=np.float32).reshape(2, 3)
    F.write_latent_coords(str(tmp_path / "l3.txt"), (mu3, mu3, mu3), ["a", "b"], 3)
    assert open(tmp_path / "l3.txt").read().splitlines()[0] == "LD1\tLD2\tLD3\tsampleID"


def test_grid_order_schedule_and_argmin(tmp_path):
    assert F.grid_paramsets([32, 256], [3, 5, 8]) == [[32, 3], [32, 5], [32, 8], [256, 3], [256, 5], [256, 8]]
    sched = F.lpt_schedule([5, 3, 9, 1, 1, 7, 2, 2, 4], 8)
    assert sorted(i for s in sched for i in s) == list(range(9)) and max(len(s) for s in sched) == 2
    table = {(32, 3): 9.0, (32, 5): 7.0, (64, 3): 7.5, (64, 5): 6.0}

    def make_engine(w, d):
        return FakeEngine([table[(w, d)] + 1, table[(w, d)], table[(w, d)] + 2] + [99.0] * 200)

    # two "ranks" run their share, results are merged as all_gather_object would
    parts = []
    for r in range(2):
        got = {}
        paramsets = F.grid_paramsets([32, 64], [3, 5])
        costs = [1000.0 * w * 2 + (2 * d - 1) * w * w for w, d in paramsets]
        for i in F.lpt_schedule(costs, 2)[r]:
            w, d = paramsets[i]
            got[i] = F.grid_fit_one(make_engine, w, d, np.arange(4), np.arange(2), 40, 8)
        parts.append(got)
    rows, bw, bd = F.grid_search(make_engine, [32, 64], [3, 5], np.arange(4), np.arange(2), 40, 8, 1000,
                                 out=str(tmp_path / "g"), rank=0, world=2, gather=lambda loc: parts, log=lambda *a: None)
    assert (bw, bd) == (64, 5)
    assert [(r["width"], r["depth"]) for r in rows] == [(32, 3), (32, 5), (64, 3), (64, 5)]
    assert open(tmp_path / "g_param_grid.csv").read().splitlines()[:2] == ["width,depth,val_loss", "32,3,9.0"]
    # quarter epochs (popvae.py:354) and float quarter patience (popvae.py:277)
    eng = FakeEngine([1.0] + [2.0] * 100)
    F.fit(eng, np.arange(4), np.arange(2), int(40 / 4), patience=8 / 4, lr_patience=int(8 / 4 / 4), checkpoint=False,
          verbose=0)
    assert eng.i == 3 and eng.snapshots == []


def test_cli_parser_matches_reference_flags():
    from popvae_amd import cli
    a = cli.build_parser().parse_args(["--infile", "x.vcf"])
    ref_defaults = dict(out="vae", patience=50, max_epochs=500, batch_size=32, save_allele_counts=False,
                        save_weights=False, seed=None, train_prop=0.9, search_network_sizes=False,
                        width_range="32,64,128,256,512", depth_range="3,6,10,20", depth=6, width=128, gpu_number="0",
                        prediction_freq=5, max_SNPs=None, latent_dim=2, PCA=False, n_pc_axes=20,
                        PCA_scaler="Patterson", test=False, plot=False, metadata=None)   # popvae.py:10-145
    for k, v in ref_defaults.items():
        assert getattr(a, k) == v, k
    assert a.init_weights is None   # the one flag the reference does not have


def test_test_flag_resolves_its_input(tmp_path, monkeypatch):
    """`popvae.py --test` (popvae.py:175-180) runs on data/pabu/pabu_test_genotypes.vcf relative to the working directory;
    where that path does not exist (this repository keeps the fixture gzip-compressed under tests/golden/) it falls back to
    that copy instead of failing."""
    import os
    from popvae_amd import cli
    monkeypatch.chdir(tmp_path)
    p = cli.test_infile()
    assert p.endswith(os.path.join("tests", "golden", "pabu_test_genotypes.vcf.gz")) and os.path.exists(p)
    os.makedirs(tmp_path / "data" / "pabu")
    (tmp_path / "data" / "pabu" / "pabu_test_genotypes.vcf").write_text("##fileformat=VCFv4.2\n")
    assert cli.test_infile() == os.path.join("data", "pabu", "pabu_test_genotypes.vcf")
