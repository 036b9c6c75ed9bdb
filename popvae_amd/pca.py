"""PCA of the genotype matrix (popvae.py:528-540).

The reference calls ``allel.pca(dc.T, scaler=PCA_scaler, n_components=n_pc_axes)`` = per-site scaler followed by a
full SVD of the scaled N x L matrix, and keeps ``U[:, :n] * S[:n]``.  Here the scaler and the N x N Gram matrix of
the scaled matrix come from the GPU in float64 (``pvae_pca_gram``: f64 MFMA on the resident 2-bit matrix, no
N x L float64 copy ever exists); the eigen-decomposition of the small Gram matrix (N = samples) is host LAPACK.
``G = U S^2 U^T`` gives the same ``U S`` as the SVD, up to the sign of each axis (the sign of a singular vector
is arbitrary in LAPACK as well).
"""
from __future__ import annotations

import numpy as np


def coords_from_gram(G, n_components):
    """Top ``n_components`` principal coordinates ``U S`` (and ``explained_variance_ratio``) from a Gram matrix.
    Sign convention (the SVD leaves it open): the entry of largest magnitude of every axis is positive."""
    G = np.asarray(G, np.float64)
    n = G.shape[0]
    if n_components > n:
        raise ValueError(f"n_components {n_components} exceeds the number of samples {n}")
    w, U = np.linalg.eigh((G + G.T) * 0.5)
    order = np.argsort(w)[::-1]
    w, U = np.clip(w[order], 0.0, None), U[:, order]
    s = np.sqrt(w)
    coords = U[:, :n_components] * s[:n_components]
    for k in range(n_components):
        if coords[np.argmax(np.abs(coords[:, k])), k] < 0:
            coords[:, k] = -coords[:, k]
    evr = (w / n) / np.sum(w / n) if np.sum(w) > 0 else np.zeros_like(w)
    return coords, evr[:n_components]


def pca(engine, scaler="Patterson", n_components=20, all_reduce=None):
    """``allel.pca(...)[0]`` for the samples loaded into ``engine`` (all N, in matrix order).
    ``all_reduce``: for SNP-sharded engines, a callable summing a float64 array over the ranks in place (every rank
    holds the Gram matrix of its own SNP columns; the scaler is per site, so the sum is the global Gram matrix)."""
    G = engine.pca_gram(scaler)
    if all_reduce is not None:
        G = all_reduce(G)
    return coords_from_gram(G, n_components)


def write_pca(path, coords, samples):
    """``out_pca.txt`` (popvae.py:533-537): columns PC1..PCn, sampleID; tab separated, no index."""
    import pandas as pd
    df = pd.DataFrame(coords)
    df.columns = ["PC" + str(x + 1) for x in range(coords.shape[1])]
    df["sampleID"] = samples
    df.to_csv(path, index=False, sep="\t")
