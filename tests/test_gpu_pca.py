"""pvae_pca_gram (f64 MFMA Gram matrix + per-site scaler on the GPU) vs the oracle's allel.pca restatement."""
import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import engine, pca as P, prep
from tests.test_pca_host import _data

pytestmark = pytest.mark.gpu


def _eng(dc):
    e = engine.Engine(n_snps=dc.shape[1], width=32, depth=2, latent_dim=2, batch_size=32, seed=0)
    e.load_genotypes(prep.pack_dc(dc))
    return e


@pytest.mark.parametrize("N,L,scaler", [(100, 3000, "Patterson"), (100, 3000, "standard"), (100, 3000, "center"),
                                        (300, 777, "Patterson"),      # 3 x 3 tiles of 128 (mirror), ragged L
                                        (129, 31, "Patterson"),       # less than one SNP block, two tiles
                                        (17, 5000, "patterson")])
def test_gram_scaler_and_coordinates(N, L, scaler):
    dc = _data(N, L, N + L)
    e = _eng(dc)
    G, mean, std = e.pca_gram(scaler, want_sites=True)
    xs, m0, s0 = O.pca_scale(dc.T, scaler)
    assert np.array_equal(mean, m0)                     # integer counts -> the same float64 mean
    assert np.abs(std - s0).max() <= 2e-15 * np.abs(s0).max()
    G0 = xs.T @ xs
    assert np.array_equal(G, G.T)
    assert np.abs(G - G0).max() <= 1e-12 * np.abs(G0).max()
    n = min(10, N - 1)
    got, evr = P.pca(e, scaler, n)
    want, evr0 = O.pca(dc, scaler, n)
    sgn = np.sign(np.sum(got * want, axis=0))
    assert np.abs(got * sgn - want).max() <= 1e-8 * np.abs(want).max()
    assert np.allclose(evr, evr0, rtol=1e-8)
    e.close()


def test_zero_scale_site_is_loud():
    dc = _data(40, 200, 3)
    dc[:, 7] = 0.0                      # monomorphic: Patterson std = 0, the reference would divide by zero
    e = _eng(dc)
    with pytest.raises(RuntimeError, match="zero scale"):
        e.pca_gram("Patterson")
    G = e.pca_gram("center")            # centring alone is fine
    assert np.isfinite(G).all()
    e.close()
