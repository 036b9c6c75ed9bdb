"""Integer-path parity: native libpvae_prep (product) vs numpy legacy RNG + oracle (checker), and
oracle vs the committed golden fixture made from the reference's pabu VCF."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import prep

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pabu_prep_seed42.json")))
# the reference's own test data file (data/pabu/pabu_test_genotypes.vcf, what `popvae.py --test` runs on), committed
# gzip-compressed as a fixture so that the check also runs where /root/reference does not exist
REF_VCF = os.path.join(os.path.dirname(__file__), "golden", "pabu_test_genotypes.vcf.gz")


def test_symbols_exported():
    L = prep.lib()
    for s in prep.PREP_SYMBOLS:
        assert hasattr(L, s), s
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "pvae_prep.h")).read()
    declared = set(re.findall(r"\b(pvae_\w+)\s*\(", hdr))
    assert declared == set(prep.PREP_SYMBOLS)


@pytest.mark.parametrize("seed", [0, 1, 42, 12345, 2**32 - 1])
def test_mt19937_stream_matches_numpy(seed):
    r = prep.LegacyRNG(seed)
    rs = np.random.RandomState(seed)
    a = np.array([r.random_sample() for _ in range(2000)])
    assert np.array_equal(a, rs.random_sample(2000))


def test_survey_rng_selftest():
    """SURVEY 8a P6 check values for the seed-42 legacy stream."""
    r = prep.LegacyRNG(42)
    r.binomial(2, np.linspace(0, 1, 1001))
    assert list(r.choice(range(95), 85, replace=False)[:8]) == [18, 44, 26, 62, 37, 56, 5, 45]


@pytest.mark.parametrize("seed", [3, 42])
def test_binomial_and_choice_match_numpy(seed):
    rs = np.random.RandomState(seed)
    r = prep.LegacyRNG(seed)
    g = np.random.default_rng(seed)
    for _ in range(20):
        p = np.concatenate([g.random(997), [0.0, 1.0, 0.5, 1e-12, 1 - 1e-12]])
        g.shuffle(p)
        assert np.array_equal(r.binomial(2, p), rs.binomial(2, p))
        n = int(g.integers(1, 5000))
        k = int(g.integers(0, n + 1))
        assert np.array_equal(r.choice(range(n), k, replace=False), rs.choice(range(n), k, replace=False))
    assert r.random_sample() == rs.random_sample()
    # interval > 32 bit path is not on the reference's path (n samples / n SNPs < 2^32)


def _random_gen(seed, L0=400, N=37, miss=0.1, multi=0.05, half=0.02):
    g = np.random.default_rng(seed)
    f = g.beta(0.3, 0.3, L0)
    gen = (g.random((L0, N, 2)) < f[:, None, None]).astype(np.int8)
    tri = g.random(L0) < multi
    gen[tri] = g.integers(0, 4, size=(int(tri.sum()), N, 2))
    swap = g.random(L0) < 0.05          # {0,2}-type biallelic sites: kept by is_biallelic, derived = allele 1
    gen[swap] = np.where(gen[swap] == 1, 2, gen[swap])
    gen[g.random((L0, N)) < miss] = -1
    half = (g.random((L0, N, 2)) < half) & ((f > 0.2) & (f < 0.8))[:, None, None]
    gen[half] = -1                      # half-missing calls: alleles counted, sample counted missing
    return gen


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_filter_impute_pack_matches_oracle(seed):
    gen = _random_gen(seed)
    np.random.seed(seed)
    dc, keep = O.filter_and_impute(gen)
    tr_o, te_o = O.train_test_split(dc.shape[0], 0.8)
    rng = prep.LegacyRNG(seed)
    packed, L, keepmask, af, nimp = prep.filter_impute_pack(gen, rng)
    assert L == dc.shape[1] and np.array_equal(np.flatnonzero(keepmask), keep)
    assert np.array_equal(packed, O.pack2bit(dc))
    assert np.array_equal(prep.unpack_dc(packed, L), dc)
    tr, te = prep.train_test_split(dc.shape[0], 0.8, rng)
    assert np.array_equal(tr, tr_o) and np.array_equal(te, te_o)
    assert np.array_equal(prep.pack_dc(dc), packed)


def test_edge_cases():
    # no site survives / no missing data / train_prop == 1 (popvae.py:256-262)
    gen = np.zeros((5, 4, 2), np.int8)
    packed, L, keep, af, nimp = prep.filter_impute_pack(gen, prep.LegacyRNG(1))
    assert L == 0 and packed.shape == (4, 0) and nimp == 0
    gen = _random_gen(9, miss=0.0, half=0.0)
    np.random.seed(9)
    dc, _ = O.filter_and_impute(gen)
    rng = prep.LegacyRNG(9)
    packed, L, _, _, nimp = prep.filter_impute_pack(gen, rng)
    assert nimp == 0 and np.array_equal(packed, O.pack2bit(dc))
    tr, te = prep.train_test_split(dc.shape[0], 1, rng)
    tro, teo = O.train_test_split(dc.shape[0], 1)
    assert np.array_equal(tr, tro) and np.array_equal(te, teo) and np.array_equal(tr, te)


def test_af_above_one_is_an_error_like_the_reference():
    # all called samples 1/1 plus half-missing ./1 calls: dc_all > 2*ninds -> np.random.binomial raises
    gen = np.ones((1, 6, 2), np.int8)
    gen[0, 0, 0] = -1
    gen[0, 1] = 0
    gen[0, 2, 1] = -1
    gen[0, 3] = -1
    gen[0, 1, 0] = -1  # samples 0,1,2,3 missing; called: 4,5 -> ninds 2, dc_all = 1+1+4+... > 4
    np.random.seed(0)
    with pytest.raises(ValueError):
        O.filter_and_impute(gen)
    with pytest.raises(RuntimeError):
        prep.filter_impute_pack(gen, prep.LegacyRNG(0))


def test_max_snps_subsample_matches_oracle():
    gen = _random_gen(4)
    np.random.seed(4)
    dc, _ = O.filter_and_impute(gen)
    sub = O.subsample_snps(dc, 50)
    rng = prep.LegacyRNG(4)
    packed, L, _, _, _ = prep.filter_impute_pack(gen, rng)
    cols = rng.choice(range(L), 50, replace=False)
    assert np.array_equal(prep.unpack_dc(prep.gather_columns(packed, cols), 50), sub)


def test_oracle_unpack_roundtrip():
    g = np.random.default_rng(0)
    dc = g.integers(0, 3, (7, 77)) * 0.5
    assert np.array_equal(O.unpack2bit(O.pack2bit(dc), 77) * 0.5, dc)


def test_vcf_reader_text(tmp_path):
    txt = ("##fileformat=VCFv4.1\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\n"
           "1\t1\t.\tG\tA\t20\tPASS\tNS=3\tGT:DP\t0|1:3\t./.:0\t1/1:9\n"
           "1\t2\t.\tG\tA,T\t20\tPASS\tNS=3\tDP:GT\t3:0/2\t1:.|1\t9:10/0\n")
    p = tmp_path / "t.vcf"
    p.write_text(txt)
    gen, samples = prep.read_vcf(str(p))
    assert list(samples) == ["A", "B", "C"]
    assert gen.tolist() == [[[0, 1], [-1, -1], [1, 1]], [[0, 2], [-1, 1], [10, 0]]]
    go, so = O.read_vcf_gt(str(p))
    assert np.array_equal(go, gen)
    import gzip
    pz = tmp_path / "t.vcf.gz"
    with gzip.open(pz, "wt") as fh:
        fh.write(txt)
    gz, _ = prep.read_vcf(str(pz))
    assert np.array_equal(gz, gen)


def test_pabu_golden_native_and_oracle():
    gen, samples = prep.read_vcf(REF_VCF)
    assert gen.shape == (GOLD["n_variants"], GOLD["n_samples"], 2)
    assert hashlib.sha256(gen.tobytes()).hexdigest() == GOLD["gt_sha256"]
    assert list(samples[:3]) == GOLD["samples_first3"] and samples[-1] == GOLD["samples_last"]
    rng = prep.LegacyRNG(42)
    packed, L, keep, af, nimp = prep.filter_impute_pack(gen, rng)
    assert L == GOLD["n_snps"] == 3887
    assert hashlib.sha256(packed.tobytes()).hexdigest() == GOLD["packed_sha256"]
    tr, te = prep.train_test_split(gen.shape[1], 0.9, rng)
    assert tr.tolist() == GOLD["train"] and te.tolist() == GOLD["test"]
    assert nimp == 22384  # SURVEY 8a P3 [PROBE]
