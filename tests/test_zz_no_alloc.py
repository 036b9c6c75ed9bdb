"""pvae_train_step_noalloc (include/pvae.h): the variant of pvae_train_step whose partial-sum buffers live in the library's
file-scope device array.  Checked through the C ABI against pvae_train_step itself, bit for bit (same kernels, same
reduction order), on two calls in a row, the second with a smaller batch; then once more with a second, smaller model
on the same GPU (the array is shared by every engine of the process and keeps the last call's contents)."""
import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import engine, prep

pytestmark = pytest.mark.gpu


def _pair(L, W, D, K, seed):
    g = np.random.default_rng(seed)
    N = 64
    dc = g.binomial(2, np.clip(g.beta(0.5, 0.5, L), 0.02, 0.98), size=(N, L)) * 0.5
    w = O.init_weights(L, W, D, K, seed=seed, dtype=np.float32)
    out = []
    for _ in range(2):
        e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=3)
        e.load_genotypes(prep.pack_dc(dc))
        e.set_weights(w)
        out.append(e)
    return out, N


@pytest.mark.parametrize("L,W,D,K", [(40_000, 128, 3, 2), (9_000, 512, 2, 3)])
def test_noalloc_train_step_equals_train_step(L, W, D, K):
    (ref, na), N = _pair(L, W, D, K, seed=L % 97)
    g = np.random.default_rng(1)
    for n in (32, 11):  # two calls in a row, the second with a smaller batch
        rows = g.choice(N, n, replace=False).astype(np.int32)
        eps = g.normal(size=(n, K)).astype(np.float32)
        a = ref.train_step(rows, eps)
        b = na.train_step_noalloc(rows, eps)
        assert a == b, (n, a, b)
        assert np.isfinite(a[0])
    for i, (x, y) in enumerate(zip(ref.get_weights(), na.get_weights())):
        assert np.array_equal(x, y), ("w", i)
    for which in (0, 1):
        for i, (x, y) in enumerate(zip(ref.get_moments(which), na.get_moments(which))):
            assert np.array_equal(x, y), ("moment", which, i)
    # the engine's own noise stream (eps = NULL) and no loss read-back work the same way
    rows = np.arange(20, dtype=np.int32)
    ref.train_step(rows, None, want_loss=False)
    na.train_step_noalloc(rows, None, want_loss=False)
    assert ref.eval_step(rows, np.zeros((20, K), np.float32)) == na.eval_step(rows, np.zeros((20, K), np.float32))
    # limits are loud
    with pytest.raises(RuntimeError, match="1..32 rows"):
        na.train_step_noalloc(np.zeros(0, np.int32))
    na.set_timing(True)
    with pytest.raises(RuntimeError, match="timing"):
        na.train_step_noalloc(rows)
    na.set_timing(False)
    ref.close()
    na.close()
