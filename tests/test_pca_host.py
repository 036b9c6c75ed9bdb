"""PCA host logic and the oracle restatement of allel.pca (PARITY UNPINNED: scikit-allel is absent and the
reference holds no fixture for it; see oracle/popvae_oracle.py)."""
import numpy as np

from oracle import popvae_oracle as O
from popvae_amd import pca as P


def _data(N, L, seed):
    g = np.random.default_rng(seed)
    pop = g.integers(0, 3, N)
    f = np.clip(g.beta(.5, .5, L)[None] + 0.15 * g.normal(size=(3, L))[pop], .05, .95)
    dc = g.binomial(2, f) * 0.5
    dc[0, :] = 0.5  # no monomorphic site
    return dc


def test_patterson_scaler_formula_as_the_reference_applies_it():
    gn = np.array([[0, 0.5, 1, 1], [0.5, 0.5, 0, 0]], float)   # variants x samples, values = code / 2
    s, mean, std = O.pca_scale(gn, "Patterson")
    assert np.allclose(mean, [0.625, 0.25])
    # allel divides the mean by ploidy = 2 although the values are already halved: p = mean / 2
    assert np.allclose(std, np.sqrt(np.array([0.3125 * 0.6875, 0.125 * 0.875])))
    assert np.allclose(s[0], (gn[0] - 0.625) / std[0])
    s2, _, std2 = O.pca_scale(gn, "standard")
    assert np.allclose(std2, gn.std(axis=1))
    s3, _, std3 = O.pca_scale(gn, None)
    assert np.allclose(std3, 1) and np.allclose(s3, gn - gn.mean(axis=1, keepdims=True))


def test_gram_route_equals_svd_route_up_to_sign():
    dc = _data(60, 900, 1)
    for scaler in ("Patterson", "standard", "none"):
        want, evr = O.pca(dc, scaler, 8)
        xs = O.pca_scale(dc.T, scaler)[0].T
        got, evr2 = P.coords_from_gram(xs @ xs.T, 8)
        sgn = np.sign(np.sum(got * want, axis=0))
        assert np.abs(got * sgn - want).max() <= 1e-9 * np.abs(want).max()
        assert np.allclose(evr, evr2, rtol=1e-9)
        # sign convention: largest-magnitude entry of every axis positive
        assert all(got[np.argmax(np.abs(got[:, k])), k] > 0 for k in range(8))


def test_write_pca(tmp_path):
    c = np.array([[1.5, -2.0], [0.25, 3.0]])
    P.write_pca(str(tmp_path / "o_pca.txt"), c, ["a", "b"])
    assert (tmp_path / "o_pca.txt").read_text().splitlines() == ["PC1\tPC2\tsampleID", "1.5\t-2.0\ta", "0.25\t3.0\tb"]
