"""The engine's opt-in paths (environment switches read by pvae_create) must give the step they replace:
   PVAE_SPLIT_DEC=1  the decoder pass as two launches of pvk_dec2<.., PART>, the second on a CU-masked side stream -- bit-identical;
   PVAE_XRN=1        the step head gathers the NEXT batch into compact words for the forward fused into pvk_enc_bwd -- bit-identical;
   PVAE_GATHER_AHEAD=1  the next batch is gathered by a launch of its own on a second stream, beside the forward chain -- bit-identical;
   PVAE_NO_DEC3B=1   four row tiles through the fp32 decoder training kernel instead of pvk_dec3b -- same result to 1e-4 (the
                     two kernels sum in different orders);
   PVAE_NO_DENSE16=1 widths 256 / 512: the 32-unit per-layer kernels instead of pvk_dense_fwd16 / pvk_dense_bwd16 (and no chaining of
                     the backward through dpre) -- same moments to 1e-5 at a learning rate that keeps the two runs on one trajectory.
Measured slower or equal, kept for A/B comparisons; these tests keep them working."""
import os

import numpy as np
import pytest

from popvae_amd import engine

pytestmark = pytest.mark.gpu


def _data(N, L, seed):
    g = np.random.default_rng(seed)
    f = np.clip(g.beta(0.5, 0.5, L), 0.02, 0.98)
    codes = (g.random((N, L)) < f).astype(np.uint8) + (g.random((N, L)) < f).astype(np.uint8)
    nb = (L + 31) // 32 * 8
    c = np.zeros((N, nb * 4), np.uint8)
    c[:, :L] = codes
    c = c.reshape(N, nb, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def _run(env, N, L, W, D, B, epochs, lr=1e-4):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=2, batch_size=B, seed=3, lr=lr)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    e.load_genotypes(_data(N, L, 11))
    e.init_weights(5)
    g = np.random.default_rng(2)
    n_train = int(0.9 * N)
    val = np.arange(n_train, N, dtype=np.int32)
    hist = []
    for _ in range(epochs):
        hist.append(e.train_epoch(g.permutation(n_train).astype(np.int32), val))
    w = e.get_weights()
    m = e.get_moments(0)
    e.close()
    return np.array(hist, np.float64), w, m


@pytest.mark.parametrize("env", [{"PVAE_SPLIT_DEC": "1"}, {"PVAE_SPLIT_DEC": "1", "PVAE_SPLIT_RESERVE": "1"}, {"PVAE_XRN": "1"},
                                 {"PVAE_GATHER_AHEAD": "1"}, {"PVAE_GATHER_AHEAD": "1", "PVAE_SPLIT_DEC": "1"}])
@pytest.mark.parametrize("L,W", [(5000, 128), (2077, 64)])
def test_opt_in_epoch_loops_are_bit_identical(env, L, W):
    ref = _run({}, 150, L, W, 3, 32, 3)
    got = _run(env, 150, L, W, 3, 32, 3)
    assert np.array_equal(ref[0], got[0]), (ref[0], got[0])
    for a, b in zip(ref[1] + ref[2], got[1] + got[2]):
        assert np.array_equal(a, b)


def test_fp32_decoder_pass_at_four_tiles_agrees_with_the_bf16_one():
    ref = _run({}, 300, 4000, 128, 3, 128, 2)
    got = _run({"PVAE_NO_DEC3B": "1"}, 300, 4000, 128, 3, 128, 2)
    assert np.allclose(ref[0], got[0], rtol=1e-5), (ref[0], got[0])
    for a, b in zip(ref[1], got[1]):
        assert np.max(np.abs(a - b)) <= 1e-4 * max(np.max(np.abs(a)), 1e-6) + 2e-6  # weights after ~6 Adam steps of 1e-4


@pytest.mark.parametrize("L,W,D", [(3000, 512, 4), (2077, 256, 6), (1000, 200, 2)])
def test_32_unit_per_layer_kernels_agree_with_the_16_unit_ones(L, W, D):
    # (ragged last batch: 135 training rows = 4 x 32 + 7; validation rows through the read-only pass)
    ref = _run({}, 150, L, W, D, 32, 2, lr=1e-5)
    got = _run({"PVAE_NO_DENSE16": "1"}, 150, L, W, D, 32, 2, lr=1e-5)
    assert np.allclose(ref[0], got[0], rtol=2e-6), (ref[0], got[0])
    for i, (a, b) in enumerate(zip(ref[2], got[2])):
        assert np.abs(a - b).max() <= 1e-5 * max(np.abs(a).max(), 1e-30), (i, np.abs(a - b).max(), np.abs(a).max())
