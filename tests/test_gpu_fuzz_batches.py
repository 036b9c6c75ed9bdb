"""Differential test of the two ways a batch above 32 rows is stepped (width <= 128): the multi-tile kernels (one pass over the wide
weights per 128 rows) against 32-row tiles with gradient dumps (PVAE_NO_MULTITILE=1, the round-1 path), on random shapes -- small
SNP counts against large grids, ragged last blocks and tiles, depth 1, chunks.  Both are pinned against the oracle at fixed shapes
(test_gpu_parity.py); here the shapes are random.  Same injected noise -> losses and evaluation losses to 1e-5, Adam's first and
second moments after two steps to 1e-5 of each tensor's largest entry (the weights themselves are not compared entry by entry:
where the true gradient is ~0, Adam turns its rounding noise into a step of up to lr)."""
import os

import numpy as np
import pytest

from popvae_amd import engine, prep

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("PVAE_FUZZ_CASES", "12"))


def _case(seed):
    g = np.random.default_rng(1000 + seed)
    L = int(g.choice([33, 64, 95, 200, 777, 1500, 4000, 9000]))
    W = int(g.choice([32, 64, 96, 128]))
    D = int(g.integers(1, 5))
    K = int(g.choice([2, 3, 8]))
    B = int(g.choice([33, 48, 64, 65, 100, 128, 129, 200, 300]))
    n = int(g.integers(max(1, B // 3), B + 1))
    return g, L, W, D, K, B, n


@pytest.mark.parametrize("seed", range(N_CASES))
def test_multi_tile_equals_tile_path(seed, monkeypatch):
    g, L, W, D, K, B, n = _case(seed)
    N = B + 10
    f = np.clip(g.beta(.5, .5, L), .02, .98)
    dc = g.binomial(2, f, size=(N, L)) * 0.5
    packed = prep.pack_dc(dc)
    steps = [(g.choice(N, m, replace=False).astype(np.int32), g.normal(size=(m, K)).astype(np.float32)) for m in (n, B, max(1, n // 2))]
    outs = []
    for tiles in (False, True):
        if tiles:
            monkeypatch.setenv("PVAE_NO_MULTITILE", "1")
        else:
            monkeypatch.delenv("PVAE_NO_MULTITILE", raising=False)
        e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=B, seed=3)
        e.load_genotypes(packed)
        e.init_weights(11)
        res = []
        for rows, eps in steps[:2]:
            res.append(e.train_step(rows, eps)[0])
        res.append(e.eval_step(*steps[2])[0])
        res.append(e.eval_step(*steps[1])[0])
        outs.append((res, e.get_moments(0) + e.get_moments(1), e.get_weights()))
        e.close()
    (ra, ma, wa), (rb, mb, wb) = outs
    tag = (seed, L, W, D, K, B, n)
    for a, b in zip(ra, rb):
        assert np.isfinite(a) and abs(a - b) <= 1e-5 * abs(b), (tag, ra, rb)
    for i, (a, b) in enumerate(zip(ma, mb)):
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(b).max(), 1e-30), (tag, "moment", i, np.abs(a - b).max(), np.abs(b).max())
    for i, (a, b) in enumerate(zip(wa, wb)):
        # two Adam steps of lr = 1e-3: the bulk of the entries agree to float32 rounding of the update
        assert np.median(np.abs(a - b)) <= 1e-7 and np.abs(a - b).max() <= 2.1e-3, (tag, i, np.abs(a - b).max())
