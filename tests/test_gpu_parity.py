"""HIP path (through the C ABI) vs the oracle on the same seeded inputs.  fp32 tolerance: 1e-4 relative
(north_star) on loss and on every gradient tensor (relative to the tensor's max |g|)."""
import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import engine, prep

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _problem(N, L, W, D, K, seed):
    g = np.random.default_rng(seed)
    f = np.clip(g.beta(0.5, 0.5, L), 0.02, 0.98)
    dc = g.binomial(2, f, size=(N, L)) * 0.5
    w = O.init_weights(L, W, D, K, seed=seed, dtype=np.float64)
    for t in w:  # non-zero biases so that bias paths are exercised
        if t.ndim == 1:
            t += g.normal(0, 0.05, t.shape)
    return dc, w


def _mk(dc, w, W, D, K, B=32, seed=0):
    e = engine.Engine(n_snps=dc.shape[1], width=W, depth=D, latent_dim=K, batch_size=B, seed=seed)
    e.load_genotypes(prep.pack_dc(dc))
    e.set_weights([x.astype(np.float32) for x in w])
    return e


def _relerr(a, b):
    s = max(np.abs(b).max(), 1e-30)
    return np.abs(a - b).max() / s


CASES = [
    # N, L, W, D, K, nrows
    (40, 64, 32, 2, 2, 32),
    (40, 1000, 128, 3, 2, 32),
    (50, 3887, 128, 6, 2, 32),     # pabu-shaped (C1)
    (50, 3887, 128, 6, 2, 21),     # short last batch (85 % 32 = 21)
    (40, 777, 64, 3, 3, 7),        # ragged L, W = 64, K = 3
    (40, 300, 100, 2, 5, 32),      # width padded 100 -> 128
    (40, 900, 96, 3, 2, 32),       # three block columns: pvk_dec2 on 3-wave workgroups, encoder grids in multiples of 3
    (45, 2100, 80, 2, 2, 13),      # width 80 padded to 96, ragged L, short batch
    (33, 31, 32, 1, 1, 1),         # L < one block, depth 1, single row
    (40, 700, 256, 3, 2, 32),      # two groups of 128 units (grouped wide decoder, per-layer narrow kernels)
    (40, 1000, 512, 2, 2, 19),     # four groups, short batch (C3 / grid-search width)
    (40, 500, 200, 2, 3, 32),      # width 200 padded to 256
    (40, 600, 128, 2, 16, 25),     # 16 latent dimensions: the two heads fill one 32-column tile
    # --latent_dim is free in the reference (popvae.py:78-80); the engine takes up to 64 (heads = one layer of <= 128 outputs)
    (40, 500, 128, 2, 24, 32),     # heads 48 -> 64 columns, z padded to 32
    (40, 500, 64, 3, 40, 21),      # heads 80 -> 96 columns (wider than the hidden layers), z padded to 64
    (40, 400, 128, 2, 64, 32),     # the maximum: heads exactly 128 columns
    (40, 400, 256, 2, 40, 32),     # per-layer narrow kernels: separate sampling kernels (heads are not one 32-column tile)
    (40, 300, 512, 2, 64, 17),
    (40, 400, 256, 1, 2, 32),      # depth 1 above width 128: the heads layer forms elu(P + b1) itself
    (40, 400, 512, 3, 16, 32),     # 16 latent dimensions: heads exactly 32 columns (sampling layer in the heads layer's tail)
    # the reference's default --depth_range is 3,6,10,20 (popvae.py:66-74, grid loop :280-369) and BASELINE config 5 has depth 8:
    # both narrow paths (op-table chain at width <= 128, per-layer kernels above)
    (40, 500, 128, 8, 2, 32),
    (40, 500, 64, 10, 2, 32),
    (40, 300, 128, 20, 2, 27),     # 42 ops in the chain's table (PVN_MAXOPS 64)
    (40, 300, 32, 20, 2, 32),
    (40, 400, 256, 8, 2, 32),
    (40, 300, 512, 10, 2, 32),
    (40, 300, 256, 20, 2, 9),
]


@pytest.mark.parametrize("N,L,W,D,K,n", CASES)
def test_loss_and_gradients(N, L, W, D, K, n):
    dc, w = _problem(N, L, W, D, K, seed=L + W)
    g = np.random.default_rng(1)
    rows = g.choice(N, n, replace=False).astype(np.int32)
    eps = g.normal(size=(n, K))
    e = _mk(dc, w, W, D, K)
    w32 = [x.astype(np.float32).astype(np.float64) for x in w]
    c = O.forward(w32, dc[rows], eps.astype(np.float32).astype(np.float64), D, K)
    grads = O.backward(w32, c, D, K)
    loss, recon, kl = e.grad_step(rows, eps)
    assert abs(loss - c["loss"]) <= RTOL * abs(c["loss"])
    assert abs(recon - c["recon"].mean()) <= RTOL * abs(c["recon"].mean())
    assert abs(kl - c["kl"].mean()) <= 1e-3 * abs(c["kl"].mean()) + 1e-5
    got = e.get_grads()
    for i, (a, b) in enumerate(zip(got, grads)):
        assert a.shape == b.shape
        # element by element as well: relative 2e-4 on entries that carry signal, the tensor's float32 noise floor below
        assert (np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-6 * np.abs(b).max()).all(), (i, np.abs(a - b).max())
        assert _relerr(a, b) <= RTOL, (i, _relerr(a, b))
    # eval step gives the same loss and leaves weights alone
    l2, _, _ = e.eval_step(rows, eps)
    assert abs(l2 - c["loss"]) <= RTOL * abs(c["loss"])
    for a, b in zip(e.get_weights(), w):
        assert np.array_equal(a, b.astype(np.float32))
    e.close()


def test_golden_model_steps():
    """HIP engine vs the committed golden fixture (tests/golden/model_steps.npz, logits BCE form)."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_steps.npz"))
    L, W, D, K, N = (int(x) for x in G["dims"])
    nt = 4 * D + 6
    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=0)
    e.load_genotypes(prep.pack_dc(G["codes"] * 0.5))
    e.set_weights([G[f"w0_{i}"] for i in range(nt)])
    loss, recon, kl = e.grad_step(G["rows_0"], G["eps_0"])
    ref = G["logits_loss_0"]
    assert abs(loss - ref[0]) <= RTOL * abs(ref[0]) and abs(recon - ref[1]) <= RTOL * abs(ref[1])
    for i, a in enumerate(e.get_grads()):
        assert _relerr(a, G[f"logits_grad0_{i}"]) <= RTOL, i
    for s in range(3):
        l, _, _ = e.train_step(G[f"rows_{s}"], G[f"eps_{s}"])
        assert abs(l - G[f"logits_loss_{s}"][0]) <= RTOL * abs(G[f"logits_loss_{s}"][0]), s
    for i, (a, m, v) in enumerate(zip(e.get_weights(), e.get_moments(0), e.get_moments(1))):
        assert np.abs(a - G[f"logits_w3_{i}"]).max() <= 2e-2 * 3e-3, ("w", i)   # 2 % of the 3*lr travel
        assert _relerr(m, G[f"logits_m3_{i}"]) <= 1e-3 and _relerr(v, G[f"logits_v3_{i}"]) <= 1e-3, ("mv", i)
    e.close()


def test_weight_roundtrip_and_layouts():
    dc, w = _problem(10, 333, 96, 3, 2, seed=5)
    e = _mk(dc, w, 96, 3, 2)
    for a, b in zip(e.get_weights(), w):
        assert np.array_equal(a, b.astype(np.float32))
    e.close()


@pytest.mark.parametrize("L,W,D", [(1000, 128, 3), (3887, 128, 6), (900, 256, 3), (700, 512, 2),
                                   (600, 128, 20), (600, 128, 10), (500, 256, 20), (400, 512, 8),
                                   (1100, 96, 3), (1300, 64, 2), (2000, 32, 2)])  # 3 / 2 / 1 block columns per subtile
def test_adam_training_steps_match_oracle(L, W, D):
    """Five fused-Adam train steps (incl. a short batch) vs oracle float64 Keras-Adam."""
    N, K = 64, 2
    dc, w = _problem(N, L, W, D, K, seed=11)
    e = _mk(dc, w, W, D, K)
    wo = [x.astype(np.float32).astype(np.float64) for x in w]
    opt = O.KerasAdam(wo)
    g = np.random.default_rng(2)
    for step in range(5):
        n = 32 if step != 3 else 13
        rows = g.choice(N, n, replace=False).astype(np.int32)
        eps = g.normal(size=(n, K)).astype(np.float32)
        lo, _ = O.train_step(wo, opt, dc[rows], eps.astype(np.float64), D, K)
        lg, _, _ = e.train_step(rows, eps)
        assert abs(lg - lo) <= RTOL * abs(lo), (step, lg, lo)
    # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare the UPDATE
    # (w_after - w_before) relative to lr*steps, plus the moments against the oracle's.
    # (measured, tools/adam_tol_probe.py: weights within 8e-6, first moments within 2e-5 of the oracle's after these five steps --
    # 1.9e-4 at depth 20, where 40 layers amplify the float32 rounding of the trajectory -- second moments within 3.4e-5)
    for i, (a, b, b0) in enumerate(zip(e.get_weights(), wo, w)):
        upd_err = np.abs(a.astype(np.float64) - b).max()
        assert upd_err <= 4e-3 * 5e-3, (i, upd_err)  # 0.4 % of the 5*lr travel
    m_tol = 1e-4 if D < 20 else 4e-4
    for i, (a, b) in enumerate(zip(e.get_moments(0), opt.m)):
        assert _relerr(a, b) <= m_tol, ("m", i, _relerr(a, b))
    for i, (a, b) in enumerate(zip(e.get_moments(1), opt.v)):
        assert _relerr(a, b) <= 1e-4, ("v", i, _relerr(a, b))
    e.close()


@pytest.mark.parametrize("L,W,D,B", [(1000, 128, 3, 32), (700, 512, 2, 32), (900, 256, 3, 32), (1100, 96, 2, 32), (1000, 128, 3, 96)])
def test_adam_update_step_by_step_from_the_engines_own_state(L, W, D, B):
    """The fused Adam arithmetic itself (pv_adam: v_sqrt_f32 / v_rcp_f32), isolated from trajectory drift: before each of five
    train steps the oracle is put on the ENGINE's float32 weights and moments, so that its gradient is the gradient of exactly the
    state the engine steps from; afterwards every tensor's first moment must be m + (g - m)(1 - beta_1), its second moment
    v + (g^2 - v)(1 - beta_2) and its weight w - alpha m' / (sqrt(v') + eps) (K7) -- moments to 1e-4 of the tensor's largest
    entry, the update to 1e-4 of the learning rate plus the float32 rounding of the stored weight."""
    N, K = max(64, B + 20), 2
    dc, w = _problem(N, L, W, D, K, seed=17)
    e = _mk(dc, w, W, D, K, B=B)
    g = np.random.default_rng(6)
    lr, b1, b2, eps_adam = 1e-3, 0.9, 0.999, 1e-7
    for step in range(1, 6):
        n = B if step != 4 else 13
        rows = g.choice(N, n, replace=False).astype(np.int32)
        eps = g.normal(size=(n, K)).astype(np.float32)
        w0 = [x.astype(np.float64) for x in e.get_weights()]
        m0 = [x.astype(np.float64) for x in e.get_moments(0)]
        v0 = [x.astype(np.float64) for x in e.get_moments(1)]
        c = O.forward(w0, dc[rows], eps.astype(np.float64), D, K)
        gr = O.backward(w0, c, D, K)
        loss, _, _ = e.train_step(rows, eps)
        assert abs(loss - c["loss"]) <= RTOL * abs(c["loss"]), (step, loss, c["loss"])
        alpha = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        for i, (a, gi, mi) in enumerate(zip(e.get_moments(0), gr, m0)):
            want = mi + (gi - mi) * (1 - b1)
            assert np.abs(a - want).max() <= 1e-4 * max(np.abs(want).max(), 1e-30), ("m", step, i, _relerr(a, want))
        m1 = [x.astype(np.float64) for x in e.get_moments(0)]
        v1 = [x.astype(np.float64) for x in e.get_moments(1)]
        for i, (a, gi, vi) in enumerate(zip(v1, gr, v0)):
            want = vi + (gi * gi - vi) * (1 - b2)
            assert np.abs(a - want).max() <= 2e-4 * max(np.abs(want).max(), 1e-30), ("v", step, i, _relerr(a, want))
        for i, (a, b, mi, vi) in enumerate(zip(e.get_weights(), w0, m1, v1)):
            want = -alpha * mi / (np.sqrt(vi) + eps_adam)   # from the engine's own new moments: the update arithmetic alone
            err = np.abs((a.astype(np.float64) - b) - want).max()
            assert err <= 1e-4 * lr + 1.2e-7 * max(1.0, np.abs(b).max()), ("w", step, i, err)
    e.close()


# batch_size above 32 (reference: --batch_size is free, popvae.py:26,497): 32-row tiles, gradients accumulated, ONE Adam update
BIG = [
    # N, L, W, D, K, B, nrows
    (120, 1000, 128, 3, 2, 64, 64),
    (120, 777, 64, 2, 3, 80, 80),      # ragged last tile (80 = 32 + 32 + 16)
    (120, 900, 128, 3, 2, 100, 45),    # short batch of a large batch size
    (120, 700, 256, 2, 2, 96, 96),     # grouped width (per-layer narrow kernels)
    (120, 300, 512, 2, 2, 64, 33),     # one full tile + one single-row tile
]


@pytest.mark.parametrize("N,L,W,D,K,B,n", BIG)
def test_large_batch_loss_and_gradients(N, L, W, D, K, B, n):
    dc, w = _problem(N, L, W, D, K, seed=L + B)
    g = np.random.default_rng(3)
    rows = g.choice(N, n, replace=False).astype(np.int32)
    eps = g.normal(size=(n, K)).astype(np.float32)
    e = _mk(dc, w, W, D, K, B=B)
    w32 = [x.astype(np.float32).astype(np.float64) for x in w]
    c = O.forward(w32, dc[rows], eps.astype(np.float64), D, K)
    grads = O.backward(w32, c, D, K)
    loss, recon, kl = e.grad_step(rows, eps)
    assert abs(loss - c["loss"]) <= RTOL * abs(c["loss"])
    assert abs(recon - c["recon"].mean()) <= RTOL * abs(c["recon"].mean())
    for i, (a, b) in enumerate(zip(e.get_grads(), grads)):
        assert _relerr(a, b) <= RTOL, (i, _relerr(a, b))
    l2, _, _ = e.eval_step(rows, eps)
    assert abs(l2 - c["loss"]) <= RTOL * abs(c["loss"])
    for a, b in zip(e.get_weights(), w):
        assert np.array_equal(a, b.astype(np.float32))
    e.close()


# widths up to 128 on one rank: 2 row tiles (B <= 64) or 4 in one weight pass (pvk_enc_fwd<4,T> / pvk_dec3<T> / pvk_enc_bwd<..,T>);
# width 256: 32-row tiles through the gradient dump
# above 128 rows: chunks of 128 (gradient dumps between the chunks, Adam in the last one): 200 = 128 + 72, 300 = 128 + 128 + 44
@pytest.mark.parametrize("L,W,D,B", [(1000, 128, 3, 80), (700, 256, 2, 64), (1000, 128, 3, 64), (1100, 128, 4, 128),
                                     (900, 96, 2, 100), (800, 32, 2, 48), (1000, 128, 3, 200), (900, 64, 2, 300)])
def test_large_batch_adam_steps_match_oracle(L, W, D, B):
    N, K = max(150, B + 20), 2
    dc, w = _problem(N, L, W, D, K, seed=13)
    e = _mk(dc, w, W, D, K, B=B)
    wo = [x.astype(np.float32).astype(np.float64) for x in w]
    opt = O.KerasAdam(wo)
    g = np.random.default_rng(4)
    for step in range(5):
        n = B if step != 3 else 41
        rows = g.choice(N, n, replace=False).astype(np.int32)
        eps = g.normal(size=(n, K)).astype(np.float32)
        lo, _ = O.train_step(wo, opt, dc[rows], eps.astype(np.float64), D, K)
        lg, _, _ = e.train_step(rows, eps)
        assert abs(lg - lo) <= RTOL * abs(lo), (step, lg, lo)
    for i, (a, b) in enumerate(zip(e.get_weights(), wo)):
        assert np.abs(a.astype(np.float64) - b).max() <= 4e-3 * 5e-3, i
    for i, (a, b) in enumerate(zip(e.get_moments(0), opt.m)):
        assert _relerr(a, b) <= 1e-4, ("m", i, _relerr(a, b))
    for i, (a, b) in enumerate(zip(e.get_moments(1), opt.v)):
        assert _relerr(a, b) <= 1e-4, ("v", i, _relerr(a, b))
    # 32-row steps still work on an engine made for larger batches (in-kernel Adam path)
    rows = g.choice(N, 20, replace=False).astype(np.int32)
    eps = g.normal(size=(20, K)).astype(np.float32)
    lo, _ = O.train_step(wo, opt, dc[rows], eps.astype(np.float64), D, K)
    lg, _, _ = e.train_step(rows, eps)
    assert abs(lg - lo) <= RTOL * abs(lo)
    e.close()


def test_large_batch_epoch_learns():
    N, L, W, D, K = 200, 1500, 128, 3, 2
    g = np.random.default_rng(0)
    pop = g.integers(0, 2, N)
    f = np.clip(g.beta(.5, .5, L)[None] + 0.35 * (pop[:, None] - .5) * g.normal(size=L)[None], .02, .98)
    dc = g.binomial(2, f) * 0.5
    w = O.init_weights(L, W, D, K, seed=4, dtype=np.float32)
    e = _mk(dc, w, W, D, K, B=72)
    train, val = np.arange(170, dtype=np.int32), np.arange(170, 200, dtype=np.int32)
    # the epoch's validation loss is a deterministic function of the weights apart from eps: compare its
    # scale with a stepwise evaluation of the same rows
    first = None
    for ep in range(15):
        loss, vloss = e.train_epoch(g.permutation(train).astype(np.int32), val)
        first = first or loss
    assert np.isfinite(loss) and np.isfinite(vloss) and loss < 0.9 * first
    lv, _, _ = e.eval_step(val)
    assert abs(lv - vloss) <= 0.02 * abs(vloss)
    e.close()


@pytest.mark.parametrize("B,n_train", [(72, 170), (64, 161), (64, 20), (96, 32), (80, 80), (160, 400), (260, 300)])
def test_large_batch_epoch_equals_stepwise(B, n_train):
    """pvae_train_epoch with batch_size > 32 against a loop of pvae_train_step / pvae_eval_step calls on the same rows and
    the same engine noise stream (one draw per 32-row tile on both paths): same per-batch losses, same weights and Adam
    moments.  Batch shapes: 72/72/26 and 64/64/33 (short last batch of one tile / of a full tile + one row), 20 and 32
    rows in all (n_train <= 32 < batch_size: the only batch is a single tile), 80 (no short batch).  The stepwise path is
    the one test_large_batch_adam_steps_match_oracle pins against the oracle."""
    N, L, W, D, K = max(200, n_train + 45), 900, 128, 3, 2  # 160: 128 + 32, short last batch 80; 260: 128 + 128 + 4, then 40
    dc, w = _problem(N, L, W, D, K, seed=31)
    g = np.random.default_rng(7)
    perm = g.permutation(N)[:n_train].astype(np.int32)
    val = np.arange(N - 45, N, dtype=np.int32)
    res = []
    for mode in ("epoch", "steps"):
        e = _mk(dc, w, W, D, K, B=B, seed=5)
        for _ in range(2):  # two epochs: the second starts with whatever gradient dump the first one left behind
            if mode == "epoch":
                loss, vloss = e.train_epoch(perm, val)
            else:
                tl = [e.train_step(perm[i:i + B])[0] for i in range(0, n_train, B)]
                vl = [e.eval_step(val[i:i + B])[0] for i in range(0, len(val), B)]
                loss, vloss = np.mean(tl), np.mean(vl)
        res.append((loss, vloss, e.get_weights(), e.get_moments(0), e.get_moments(1)))
        e.close()
    (la, va, wa, ma, sa), (lb, vb, wb, mb, sb) = res
    assert abs(la - lb) <= 1e-6 * abs(lb) and abs(va - vb) <= 1e-6 * abs(vb), (la, lb, va, vb)
    for i, (a, b) in enumerate(zip(wa, wb)):
        assert np.array_equal(a, b), ("w", i, np.abs(a - b).max())
    for i, (a, b) in enumerate(zip(ma + sa, mb + sb)):
        assert np.array_equal(a, b), ("m/v", i, np.abs(a - b).max())


def test_encode_matches_oracle():
    N, L, W, D, K = 70, 500, 128, 4, 2
    dc, w = _problem(N, L, W, D, K, seed=3)
    e = _mk(dc, w, W, D, K)
    rows = np.arange(N, dtype=np.int32)
    eps = np.random.default_rng(0).normal(size=(N, K)).astype(np.float32)
    mu, lv, z = e.encode(rows, eps)
    w32 = [x.astype(np.float32).astype(np.float64) for x in w]
    mo, so, zo = O.encode(w32, dc, eps.astype(np.float64), D, K)
    for a, b in ((mu, mo), (lv, so), (z, zo)):
        assert np.abs(a - b).max() <= 1e-4 * max(1.0, np.abs(b).max())
    e.close()


@pytest.mark.parametrize("L,W,D,K", [(500, 128, 4, 2), (333, 64, 2, 3), (700, 256, 3, 2), (40, 512, 2, 2)])
def test_decode_matches_oracle(L, W, D, K):
    dc, w = _problem(10, L, W, D, K, seed=9)
    e = _mk(dc, w, W, D, K)
    z = np.random.default_rng(1).normal(size=(45, K)).astype(np.float32)   # > 32 rows: two decode batches
    p = e.decode(z)
    w32 = [x.astype(np.float32).astype(np.float64) for x in w]
    po = O.decode(w32, z.astype(np.float64), D, K)
    assert p.shape == po.shape
    assert np.abs(p - po).max() <= 2e-5
    e.close()


def test_epoch_matches_stepwise_and_learns():
    N, L, W, D, K = 95, 2000, 128, 3, 2
    g = np.random.default_rng(0)
    pop = g.integers(0, 2, N)
    f = np.clip(g.beta(.5, .5, L)[None] + 0.35 * (pop[:, None] - .5) * g.normal(size=L)[None], .02, .98)
    dc = g.binomial(2, f) * 0.5
    w = O.init_weights(L, W, D, K, seed=4, dtype=np.float32)
    e = _mk(dc, w, W, D, K)
    train, val = np.arange(85, dtype=np.int32), np.arange(85, 95, dtype=np.int32)
    first = None
    for ep in range(12):
        loss, vloss = e.train_epoch(g.permutation(train).astype(np.int32), val)
        first = first or loss
    assert np.isfinite(loss) and np.isfinite(vloss) and loss < 0.9 * first
    # best-weights snapshot / restore (ModelCheckpoint + load_weights)
    e.snapshot_best()
    w_best = e.get_weights()
    e.train_epoch(train, val)
    assert not np.array_equal(e.get_weights()[0], w_best[0])
    e.restore_best()
    for a, b in zip(e.get_weights(), w_best):
        assert np.array_equal(a, b)
    e.close()


def test_engine_rng_eps_is_standard_normal():
    N, L, W, D, K = 64, 64, 32, 2, 2
    dc, w = _problem(N, L, W, D, K, seed=8)
    e = _mk(dc, w, W, D, K, seed=123)
    zs, mus, lvs = [], [], []
    for _ in range(60):
        mu, lv, z = e.encode(np.arange(32, dtype=np.int32))
        zs.append(z), mus.append(mu), lvs.append(lv)
    eps = (np.concatenate(zs) - np.concatenate(mus)) / np.exp(np.concatenate(lvs))
    assert abs(eps.mean()) < 0.06 and abs(eps.std() - 1) < 0.06
    e.close()


def test_errors_are_loud():
    dc, w = _problem(10, 64, 32, 2, 2, seed=1)
    e = _mk(dc, w, 32, 2, 2)
    with pytest.raises(RuntimeError):
        e.train_step(np.array([10], np.int32))       # row out of range
    with pytest.raises(RuntimeError):
        e.train_step(np.arange(33, dtype=np.int32) % 10)  # batch larger than batch_size
    bad = prep.pack_dc(dc).copy()
    bad[0, 0] |= 3
    with pytest.raises(RuntimeError):
        e.load_genotypes(bad)                         # reserved 'missing' code
    e.close()


def test_staged_step_with_promised_next_batch():
    """pvae_stage_c_next (the next batch's first-layer forward rides on the wide encoder backward) against the plain
    staged sequence: same losses, same weights; a broken promise (other rows, or a call in between) falls back."""
    N, L, W, D, K = 90, 2500, 128, 3, 2
    dc, w = _problem(N, L, W, D, K, seed=21)
    g = np.random.default_rng(6)
    batches = [g.choice(N, n, replace=False).astype(np.int32) for n in (32, 32, 17, 32, 32)]
    epss = [g.normal(size=(len(b), K)).astype(np.float32) for b in batches]

    def run(promise):
        e = _mk(dc, w, W, D, K)
        e.lr = 1e-3
        out = []
        for k, (b, ep) in enumerate(zip(batches, epss)):
            e.stage_a(b, ep, engine.MODE_TRAIN)
            e.stage_b(engine.MODE_TRAIN)
            nxt = None
            if promise and k + 1 < len(batches):
                # step 2 promises the WRONG rows, step 3 keeps its promise but an encode call intervenes
                nxt = batches[k + 1] if k != 2 else batches[0]
            out.append(e.stage_c(engine.MODE_TRAIN, True, next_rows=nxt)[0])
            if promise and k == 3:
                e.encode(np.arange(5, dtype=np.int32))
        wts = e.get_weights()
        e.close()
        return out, wts

    la, wa = run(False)
    lb, wb = run(True)
    assert np.allclose(la, lb, rtol=1e-6), (la, lb)
    # the two forward paths group the split-K partial sums differently (rounding-level differences in the activations);
    # Adam turns a gradient difference d on a near-zero gradient into lr * d / eps of weight difference
    for i, (a, b) in enumerate(zip(wa, wb)):
        assert np.abs(a - b).max() <= 1e-5, (i, np.abs(a - b).max())


@pytest.mark.parametrize("N,L,W,D,K,n", [(40, 900, 128, 3, 2, 32), (40, 500, 64, 2, 3, 19), (40, 600, 256, 2, 2, 32), (40, 400, 512, 2, 2, 27)])
def test_bce_from_clipped_probabilities(N, L, W, D, K, n):
    """K5: Keras' other binary cross-entropy path (probabilities clipped to [1e-7, 1 - 1e-7]) against the oracle's 'probs'
    mode -- loss and every gradient tensor through the gradient-dump kernels, and the training kernels (pvk_dec2 /
    pvk_dec_grp) through the first Adam moment.  A fifth of the output biases is pushed to +-25 so that the clip is active:
    those SNPs contribute -log(2e-7) or ~0 to the loss and pass no gradient."""
    dc, w = _problem(N, L, W, D, K, seed=L + W + 1)
    g = np.random.default_rng(2)
    w = [x.copy() for x in w]
    sat = g.random(L) < 0.2
    w[-1] = np.where(sat, np.where(g.random(L) < 0.5, 25.0, -25.0), w[-1]).astype(w[-1].dtype)
    rows = g.choice(N, n, replace=False).astype(np.int32)
    eps = g.normal(size=(n, K))
    e = _mk(dc, w, W, D, K)
    e.set_bce_mode("probs")
    w32 = [x.astype(np.float32).astype(np.float64) for x in w]
    c = O.forward(w32, dc[rows], eps.astype(np.float32).astype(np.float64), D, K, bce_mode="probs")
    grads = O.backward(w32, c, D, K, bce_mode="probs")
    c_log = O.forward(w32, dc[rows], eps.astype(np.float32).astype(np.float64), D, K)
    assert abs(c["loss"] - c_log["loss"]) > 1e-3 * abs(c["loss"])  # the two forms do differ on this input
    loss, recon, kl = e.grad_step(rows, eps)
    # The loss is compared with the oracle evaluated in float32, as Keras evaluates it: 1 - 1e-7 is 1 - 1.19e-7 there, so a
    # saturated wrong-side element costs -log(2.19e-7) = 15.33, not the -log(2e-7) = 15.42 of a float64 evaluation.  (The
    # gradients do not depend on that: such elements are outside the clip in both.)
    w_f32 = [x.astype(np.float32) for x in w]
    c32 = O.forward(w_f32, dc[rows].astype(np.float32), eps.astype(np.float32), D, K, bce_mode="probs")
    assert abs(loss - c32["loss"]) <= RTOL * abs(c32["loss"]), (loss, c32["loss"], c["loss"])
    for i, (a, b) in enumerate(zip(e.get_grads(), grads)):
        assert _relerr(a, b) <= RTOL, (i, _relerr(a, b))
    l2, _, _ = e.eval_step(rows, eps)
    assert abs(l2 - loss) <= 1e-6 * abs(loss)
    l3, _, _ = e.train_step(rows, eps)
    assert abs(l3 - loss) <= 1e-6 * abs(loss)
    for i, (a, b) in enumerate(zip(e.get_moments(0), grads)):
        assert _relerr(a.astype(np.float64) * 10.0, b) <= RTOL, ("m", i, _relerr(a.astype(np.float64) * 10.0, b))
    e.set_bce_mode("logits")
    e.set_weights(w)
    l4, _, _ = e.eval_step(rows, eps)
    assert abs(l4 - c_log["loss"]) <= RTOL * abs(c_log["loss"])
    e.close()
