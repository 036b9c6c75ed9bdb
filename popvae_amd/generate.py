"""Genotype generation from the latent space.

Host-side mirror of the decoder experiments in the reference's research script
(scripts/popvae_decoder_tests_HGDP.py): ``decoder.predict`` is ``Engine.decode`` (pvae_decode, HIP);
the binning and file writing around it are integer / text work on the host.
"""
from __future__ import annotations

import numpy as np

from . import prep


def binomial_bin_genotypes(pgen, rng):
    """Derived-allele counts in {0,1,2} drawn as binomial(2, p) per site, one sample (row) at a time in row order
    (scripts/popvae_decoder_tests_HGDP.py:159-163).  ``rng`` is a ``prep.LegacyRNG``: the draws are those of
    ``np.random.binomial`` on the legacy global stream, bit for bit (float32 probabilities are widened to
    float64 first, as numpy does)."""
    pgen = np.asarray(pgen)
    out = np.empty(pgen.shape, np.float32 if pgen.dtype == np.float32 else np.float64)
    for i in range(pgen.shape[0]):
        out[i, :] = rng.binomial(2, np.ascontiguousarray(pgen[i, :], dtype=np.float64))
    return out


def generate_genotypes(engine, z, rng):
    """``decoder.predict(z)`` followed by binomial binning (script lines 156-165, 267-268)."""
    return binomial_bin_genotypes(engine.decode(z), rng)


def sample_population_latents(mu, labels, pop, n, rng_normal):
    """``n`` latent positions drawn around one population's mean with its per-axis standard deviation
    (script lines 258-265; ``np.std`` = population standard deviation).  ``rng_normal(mean, sd, n)`` supplies the
    normal draws (the reference uses ``np.random.normal``)."""
    mu = np.asarray(mu)
    sel = np.asarray(labels) == pop
    if not sel.any():
        raise ValueError(f"no samples labelled {pop!r}")
    cols = [rng_normal(float(np.mean(mu[sel, k])), float(np.std(mu[sel, k])), n) for k in range(mu.shape[1])]
    return np.stack(cols, axis=1).astype(np.float32)


def genotypes_to_ped(bingen, sample_ids, outpath=None):
    """PLINK .ped/.map text of a {0,1,2} matrix (script lines 195-223): 0 -> "1 1", 1 -> "1 2", 2 -> "2 2"; the six
    leading .ped fields are (row number, sample ID, row number x4); .map rows are (1, snp index, 0, snp index).
    The reference builds the allele columns as a float array, so pandas writes them as "1.0" / "2.0"; kept.
    Returns the two tables as lists of rows; writes ``outpath.ped`` / ``outpath.map`` when ``outpath`` is given."""
    bingen = np.asarray(bingen)
    n, L = bingen.shape
    codes = np.rint(bingen).astype(np.int64)
    if codes.min(initial=0) < 0 or codes.max(initial=0) > 2:
        raise ValueError("genotypes must be 0, 1 or 2")
    a = np.where(codes == 2, 2, 1)   # first allele
    b = np.where(codes == 0, 1, 2)   # second allele
    alleles = np.empty((n, 2 * L), np.float64)
    alleles[:, 0::2], alleles[:, 1::2] = a, b
    ped = [[i, sample_ids[i], i, i, i, i] + alleles[i].tolist() for i in range(n)]
    mp = [[1, j, 0, j] for j in range(L)]
    if outpath is not None:
        with open(outpath + ".ped", "w") as f:
            for r in ped:
                f.write(" ".join(str(x) for x in r) + "\n")
        with open(outpath + ".map", "w") as f:
            for r in mp:
                f.write(" ".join(str(x) for x in r) + "\n")
    return ped, mp


__all__ = ["binomial_bin_genotypes", "generate_genotypes", "sample_population_latents", "genotypes_to_ped", "prep"]
