"""--save_weights / load_weights through files (popvae.py:429-435, 508): engine -> <out>_weights.hdf5 (Keras model_weights
layout, plain-Python HDF5 writer) -> a NEW engine -> identical predictions; the same through .npz."""
import numpy as np
import pytest

from popvae_amd import engine, prep, weights_io

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,D,K", [(128, 3, 2), (256, 2, 5)])
def test_weight_file_round_trip_through_the_engine(tmp_path, W, D, K):
    N, L = 50, 2500
    g = np.random.default_rng(W)
    dc = g.binomial(2, np.clip(g.beta(.5, .5, L), .02, .98), size=(N, L)) * 0.5
    packed = prep.pack_dc(dc)
    a = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=1)
    a.load_genotypes(packed)
    a.init_weights(7)
    for _ in range(3):
        a.train_epoch(g.permutation(40).astype(np.int32), np.arange(40, N, dtype=np.int32))
    rows = np.arange(N, dtype=np.int32)
    eps = g.normal(size=(N, K)).astype(np.float32)
    want = a.encode(rows, eps)
    want_eval = a.eval_step(rows[:32], eps[:32])
    for ext, save in ((".hdf5", weights_io.save_weights_hdf5), (".npz", weights_io.save_weights_npz)):
        path = save(str(tmp_path / ("m_weights" + ext)), a.get_weights(), D)
        b = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=2)
        b.load_genotypes(packed)
        b.set_weights(weights_io.load_weights_file(path))   # vae.load_weights(out + "_weights.hdf5")
        got = b.encode(rows, eps)
        for x, y in zip(got, want):
            assert np.array_equal(x, y)
        assert b.eval_step(rows[:32], eps[:32]) == want_eval
        b.close()
    a.close()
