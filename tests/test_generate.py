"""Host side of genotype generation: binomial binning on the legacy numpy stream and PLINK text."""
import numpy as np

from popvae_amd import generate, prep


def test_binomial_binning_matches_numpy_legacy_stream():
    g = np.random.default_rng(0)
    pgen = g.random((7, 501)).astype(np.float32)
    pgen[0, :5] = [0.0, 1.0, 0.5, 1e-9, 1 - 1e-7]
    np.random.seed(1234)
    want = np.copy(pgen)
    for i in range(want.shape[0]):           # the reference's loop, on numpy's own global legacy stream
        want[i, :] = np.random.binomial(2, want[i, :])
    got = generate.binomial_bin_genotypes(pgen, prep.LegacyRNG(1234))
    assert got.dtype == np.float32 and np.array_equal(got, want)
    # and the stream is left where numpy's is
    assert prep.LegacyRNG(1234).random_sample() != 0  # smoke
    r = prep.LegacyRNG(1234)
    generate.binomial_bin_genotypes(pgen, r)
    assert r.random_sample() == np.random.random_sample()


def test_population_latents_and_ped(tmp_path):
    mu = np.array([[0.0, 1.0], [2.0, 3.0], [10.0, 10.0]])
    seen = []

    def rn(m, s, n):
        seen.append((m, s, n))
        return np.full(n, m)
    z = generate.sample_population_latents(mu, ["a", "a", "b"], "a", 4, rn)
    assert z.shape == (4, 2) and seen == [(1.0, 1.0, 4), (2.0, 1.0, 4)]
    ped, mp = generate.genotypes_to_ped(np.array([[0, 1, 2], [2, 2, 0]], np.float32), ["s1", "s2"], str(tmp_path / "o"))
    assert (tmp_path / "o.ped").read_text().splitlines() == ["0 s1 0 0 0 0 1.0 1.0 1.0 2.0 2.0 2.0",
                                                           "1 s2 1 1 1 1 2.0 2.0 2.0 2.0 1.0 1.0"]
    assert (tmp_path / "o.map").read_text().splitlines() == ["1 0 0 0", "1 1 0 1", "1 2 0 2"]
    assert len(ped) == 2 and len(mp) == 3
