"""Parity at BASELINE.json's full SNP counts (configs C2, C3, C4): the HIP path through the C ABI against

  * the float64 oracle directly (one forward/backward at 100k, 500k x width 512 and 1M SNPs: the oracle's BLAS
    needs a few seconds there; the sample count is kept small, the kernels only ever see the batch's 32 rows), and
  * size-independent properties: a repeated step is bitwise reproducible, the C-side epoch loop (first-layer
    forward fused into the previous step's wide backward) ends in the same weights as step-by-step calls, SNP
    shards add up to the unsharded result (linearity of both exchange points), and the first fused Adam
    update equals the closed form  -lr * g / (|g| + eps / sqrt(1 - beta2))  of the engine's own gradient.

fp32 tolerance 1e-4 relative (north_star) on loss and on every gradient tensor (relative to max |g|)."""
import numpy as np
import pytest

from oracle import popvae_oracle as O
from popvae_amd import engine, prep

pytestmark = pytest.mark.gpu

RTOL = 1e-4


def _codes(N, L, seed):
    g = np.random.default_rng(seed)
    f = np.clip(g.beta(0.5, 0.5, L), 0.02, 0.98).astype(np.float32)
    return ((g.random((N, L), dtype=np.float32) < f).astype(np.uint8)
            + (g.random((N, L), dtype=np.float32) < f).astype(np.uint8))


def _pack(codes):
    N, L = codes.shape
    nb = (L + 31) // 32 * 8
    c = np.zeros((N, nb * 4), np.uint8)
    c[:, :L] = codes
    c = c.reshape(N, nb, 4)
    return np.ascontiguousarray(c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6))


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def test_pack_helper_matches_front_end():
    codes = _codes(5, 1003, 1)
    assert np.array_equal(_pack(codes), prep.pack_dc(codes * 0.5))


@pytest.mark.parametrize("name,L,W,D,n", [
    ("C2", 100_000, 128, 6, 32),
    ("C4", 1_000_000, 128, 6, 32),
    ("C4-short-batch", 1_000_000, 128, 6, 13),   # 2253 % 32 = 13: the epoch's last batch at C4
    ("C3", 500_000, 512, 6, 32),
    ("C5", 200_000, 256, 5, 32),
    # BASELINE config 5 is the whole grid widths {32, 256, 512} x depths {3, 5, 8} on 200 k SNPs (popvae.py:280-369)
    ("C5-32x3", 200_000, 32, 3, 32),
    ("C5-32x5", 200_000, 32, 5, 32),
    ("C5-32x8", 200_000, 32, 8, 32),
    ("C5-256x3", 200_000, 256, 3, 32),
    ("C5-256x8", 200_000, 256, 8, 32),
    ("C5-512x3", 200_000, 512, 3, 32),
    ("C5-512x5", 200_000, 512, 5, 32),
    ("C5-512x8", 200_000, 512, 8, 32),
    # the reference's default --depth_range ends at 10 and 20 (popvae.py:66-74)
    ("C2-depth20", 100_000, 128, 20, 32),
    ("C5-256x10", 200_000, 256, 10, 19),
])
def test_full_size_loss_and_gradients_vs_oracle(name, L, W, D, n):
    N, K = 40, 2
    codes = _codes(N, L, seed=L % 1000 + W)
    w = O.init_weights(L, W, D, K, seed=7, dtype=np.float32)
    g = np.random.default_rng(5)
    for t in w:
        if t.ndim == 1:
            t += g.normal(0, 0.05, t.shape).astype(np.float32)
    rows = g.choice(N, n, replace=False).astype(np.int32)
    eps = g.normal(size=(n, K)).astype(np.float32)
    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=0)
    e.load_genotypes(_pack(codes))
    e.set_weights(w)
    loss, recon, kl = e.grad_step(rows, eps)
    got = e.get_grads()
    w64 = [x.astype(np.float64) for x in w]
    c = O.forward(w64, codes[rows] * 0.5, eps.astype(np.float64), D, K)
    grads = O.backward(w64, c, D, K)
    assert abs(loss - c["loss"]) <= RTOL * abs(c["loss"]), (loss, c["loss"])
    assert abs(recon - c["recon"].mean()) <= RTOL * abs(c["recon"].mean())
    for i, (a, b) in enumerate(zip(got, grads)):
        assert a.shape == b.shape
        assert _relerr(a, b) <= RTOL, (name, i, _relerr(a, b))
    # the same step again is bitwise reproducible (fixed-order split-K / split-N reductions)
    loss2, _, _ = e.grad_step(rows, eps)
    assert loss2 == loss
    for a, b in zip(e.get_grads(), got):
        assert np.array_equal(a, b)
    # The TRAINING pass (its own kernels: pvk_dec2 / pvk_enc_bwd's pipelined sweep, Adam fused) from zero moments:
    # the first moment it leaves behind is (1 - beta_1) * gradient, so the oracle's gradient is checked through it to the
    # same 1e-4; the loss agrees with the gradient-dump pass up to the grouping of the partial sums
    lr, eps_hat = 1e-3, 1e-7 / np.sqrt(1 - 0.999)
    l3, _, _ = e.train_step(rows, eps)
    assert abs(l3 - loss) <= 1e-6 * abs(loss)
    m1 = e.get_moments(0)
    for i, (a, b) in enumerate(zip(m1, grads)):
        assert _relerr(a.astype(np.float64) * 10.0, b) <= RTOL, (name, "m", i, _relerr(a.astype(np.float64) * 10.0, b))
    del grads, c, w64
    # ... and its weight update is the closed form of that gradient g' = m / (1 - beta_1).  (Against the gradient-dump
    # pass's g the update of a near-zero entry differs by lr * (g' - g) / eps_hat: rounding noise times 300.)
    for i, (a, b, m) in enumerate(zip(e.get_weights(), w, m1)):
        gr = m.astype(np.float64) * 10.0
        want = -lr * gr / (np.abs(gr) + eps_hat)
        # float32 weights of magnitude <= ~1: the stored update carries a rounding error of ~6e-8 * |w|
        assert np.abs((a.astype(np.float64) - b) - want).max() <= 1e-4 * lr + 1.2e-7 * max(1.0, np.abs(b).max()), (name, i)
    e.close()


@pytest.mark.parametrize("name,L,D,B,n", [("C4-B128", 1_000_000, 6, 128, 128), ("C4-B64-short", 1_000_000, 6, 64, 50),
                                          ("C2-B128-ragged", 100_000, 6, 128, 97),
                                          # above 128 rows: chunks of 128 with gradient dumps in between (200 = 128 + 72, 300 = 128 + 128 + 44)
                                          ("C4-B256-chunks", 1_000_000, 6, 256, 200), ("C2-B512-chunks", 100_000, 6, 512, 300)])
def test_full_size_batches_in_one_weight_pass(name, L, D, B, n):
    """--batch_size 33..128 at width 128 (popvae.py:26,497): 2 / 4 row tiles per weight pass (pvk_enc_fwd<4,T>, pvk_dec3<T>,
    pvk_enc_bwd<4,TRAIN,0,T>).  One training step from zero moments at full SNP counts: loss vs the float64 oracle, and every
    tensor's first moment = (1 - beta_1) * the oracle's gradient, to the same 1e-4."""
    N, K, W = max(150, n + 20), 2, 128
    codes = _codes(N, L, seed=L % 1000 + B)
    w = O.init_weights(L, W, D, K, seed=7, dtype=np.float32)
    g = np.random.default_rng(5)
    for t in w:
        if t.ndim == 1:
            t += g.normal(0, 0.05, t.shape).astype(np.float32)
    rows = g.choice(N, n, replace=False).astype(np.int32)
    eps = g.normal(size=(n, K)).astype(np.float32)
    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=B, seed=0)
    e.load_genotypes(_pack(codes))
    e.set_weights(w)
    leval, _, _ = e.eval_step(rows, eps)   # the read-only pass over 32 T rows at a time (pvk_dec_eval_mt)
    loss, recon, kl = e.train_step(rows, eps)
    w64 = [x.astype(np.float64) for x in w]
    c = O.forward(w64, codes[rows] * 0.5, eps.astype(np.float64), D, K)
    grads = O.backward(w64, c, D, K)
    assert abs(leval - c["loss"]) <= RTOL * abs(c["loss"]), (leval, c["loss"])
    assert abs(loss - c["loss"]) <= RTOL * abs(c["loss"]), (loss, c["loss"])
    assert abs(recon - c["recon"].mean()) <= RTOL * abs(c["recon"].mean())
    del c, w64
    for i, (a, b) in enumerate(zip(e.get_moments(0), grads)):
        assert _relerr(a.astype(np.float64) * 10.0, b) <= RTOL, (name, "m", i, _relerr(a.astype(np.float64) * 10.0, b))
    # a second step on other rows stays finite and is reproducible from the same state
    rows2 = g.choice(N, n, replace=False).astype(np.int32)
    l2, _, _ = e.eval_step(rows2, eps)
    l3, _, _ = e.eval_step(rows2, eps)
    assert np.isfinite(l2) and l2 == l3
    e.close()


@pytest.mark.parametrize("L,W,D", [(1_000_000, 128, 6), (500_000, 512, 6)])
def test_epoch_loop_equals_stepwise_at_full_size(L, W, D):
    """pvae_train_epoch (next step's first-layer forward rides on the wide encoder backward) against a loop of
    pvae_train_step calls: same permutation, same noise stream -> same losses and the same weights."""
    N, K, steps = 150, 2, 4
    packed = _pack(_codes(N, L, seed=3))
    perm = np.random.default_rng(1).permutation(N)[:steps * 32 - 11].astype(np.int32)  # short last batch
    outs = []
    for mode in ("epoch", "steps"):
        e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=9)
        e.load_genotypes(packed)
        e.init_weights(3)
        e.lr = 1e-4
        if mode == "epoch":
            loss, _ = e.train_epoch(perm, np.zeros(0, np.int32))
        else:
            ls = [e.train_step(perm[i:i + 32])[0] for i in range(0, len(perm), 32)]
            loss = float(np.mean(np.array(ls, np.float32)))
        wts = e.get_weights()
        outs.append((loss, [wts[0], wts[2], wts[-2], wts[-1]]))
        e.close()
        del wts
    (la, wa), (lb, wb) = outs
    assert np.isfinite(la) and abs(la - lb) <= 1e-6 * abs(lb), (la, lb)
    for i, (a, b) in enumerate(zip(wa, wb)):
        # both paths sum the same partials; allow float32 re-association only (updates are ~1e-4 per step)
        assert np.abs(a - b).max() <= 1e-6, (i, np.abs(a - b).max())


@pytest.mark.parametrize("L,W,D", [(1_000_000, 128, 6), (4_000_031, 128, 3),
                                   # BASELINE config 3 (width 512, SNP-sharded) and a C5 cell at width 256: the grouped decoder
                                   # pass (pvk_dec_grp) and the per-layer narrow kernels behind the staged calls
                                   (500_000, 512, 6), (200_000, 256, 5)])
def test_snp_shards_add_up_at_full_size(L, W, D):
    """Linearity of the two exchange points at 1M SNPs and at 4M + 31 SNPs (four times the headline shape, ragged last
    block: 64-bit tile offsets, 2 GB per wide tensor): two engines holding half of the SNP axis each, driven
    through the staged C ABI with the host adding their exchange buffers (pvae_exchange_copy), reproduce the unsharded loss and
    the unsharded gradients (wide tensors: the matching slices) -- and, at widths 256 / 512, the float64 oracle's loss."""
    from popvae_amd.dist import shard_bounds, shard_packed
    K, N, n = 2, 40, 32
    codes = _codes(N, L, seed=11)
    packed = _pack(codes)
    g = np.random.default_rng(2)
    rows = g.choice(N, n, replace=False).astype(np.int32)
    eps = g.normal(size=(n, K)).astype(np.float32)
    w = O.init_weights(L, W, D, K, seed=4, dtype=np.float32)
    full = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=0)
    full.load_genotypes(packed)
    full.set_weights(w)
    loss, recon, kl = full.grad_step(rows, eps)
    gf = full.get_grads()
    full.close()
    if W > 128:
        c = O.forward([x.astype(np.float64) for x in w], codes[rows] * 0.5, eps.astype(np.float64), D, K)
        assert abs(loss - c["loss"]) <= RTOL * abs(c["loss"]), (loss, c["loss"])
        del c
    bounds = shard_bounds(L, 2)
    engs = []
    for r, (lo, hi) in enumerate(bounds):
        e = engine.Engine(n_snps=hi - lo, width=W, depth=D, latent_dim=K, batch_size=32, seed=0,
                          n_snps_global=L, snp_offset=lo, rank=r, world=2)
        e.load_genotypes(shard_packed(packed, L, lo, hi))
        ws = list(w)
        ws[0], ws[-2], ws[-1] = (np.ascontiguousarray(w[0][lo:hi]), np.ascontiguousarray(w[-2][:, lo:hi]),
                                 np.ascontiguousarray(w[-1][lo:hi]))
        e.set_weights(ws)
        engs.append(e)

    def exchange(which):   # the all-reduce, staged through the host in rank order
        s = engs[0].exchange_read(which) + engs[1].exchange_read(which)
        for e in engs:
            e.exchange_write(which, s)

    for e in engs:
        e.stage_a(rows, eps, engine.MODE_GRAD)
    exchange(0)
    for e in engs:
        e.stage_b(engine.MODE_GRAD)
    exchange(1)
    outs = [e.stage_c(engine.MODE_GRAD, True) for e in engs]
    assert outs[0] == outs[1]
    # float32 sums of L x 32 BCE terms in two different orders (per-lane partials, then workgroups, then shards)
    assert abs(outs[0][0] - loss) <= 1e-5 * abs(loss), (outs[0][0], loss)
    for r, ((lo, hi), e) in enumerate(zip(bounds, engs)):
        gs = e.get_grads()
        ref = list(gf)
        ref[0], ref[-2], ref[-1] = gf[0][lo:hi], gf[-2][:, lo:hi], gf[-1][lo:hi]
        for i, (a, b) in enumerate(zip(gs, ref)):
            assert _relerr(a, b) <= 1e-5, (r, i, _relerr(a, b))
        e.close()


@pytest.mark.parametrize("L,W,D,steps", [(1_000_000, 128, 6, 3), (200_000, 256, 8, 8)])
def test_headline_shape_at_keras_default_lr(L, W, D, steps):
    """BASELINE's headline shape (1M SNPs, depth 6, width 128, batch 32) at the reference's own optimizer defaults
    (`vae.compile(optimizer="adam")`, popvae.py:426: lr = 1e-3).  With a fan-in of 1M the first Adam step (every first-layer
    weight moves by ~lr in the direction of its gradient's sign) throws the pre-activations out by hundreds, and the loss
    explodes from 6.9e5 to ~3e10 within one step (measured: engine 2.94208e10 / oracle 2.94260e10 at step 1, 2.3508e10 /
    2.3483e10 at step 2): a property of the model at this scale, which bench.py avoids by stepping at lr = 1e-4.  The engine
    must do what the float32 oracle does at the same settings: the same losses while they are finite (step 0 to 1e-4; later
    steps, where the float32 rounding of 128M updates of +-lr is amplified by the fan-in, to 5e-3), and non-finite at the same
    step if at all.
    Second case: the 256 x 8 cell of BASELINE config 5 (200 k SNPs), whose validation loss came out NaN from the first epoch on
    when the grid was run as a run at every default (profiles/r03_v2_grid_c5.json: an empty val_loss in _param_grid.csv, which
    is what pandas writes for a NaN).  On this test's data the cell is in the same unstable regime without leaving float32:
    the KL term jumps 0.14 -> 119 -> 3886 -> 4502 -> 19898 over the first five steps, engine and float32 oracle agree to 3e-5
    over the first eight steps and then drift apart as their rounding differences are amplified (3e-3 at step 8, per cent by
    step 15; both finite over 40 steps).  The test holds the eight steps to 2e-4; what the grid's NaN is beyond "this model is
    unstable at lr = 1e-3 and this fan-in" stays unpinned."""
    K, N, n = 2, 40, 32
    codes = _codes(N, L, seed=128)
    w = O.init_weights(L, W, D, K, seed=7, dtype=np.float32)
    e = engine.Engine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=0, lr=1e-3)
    e.load_genotypes(_pack(codes))
    e.set_weights(w)
    opt = O.KerasAdam(w, lr=1e-3)
    g = np.random.default_rng(5)
    got, want = [], []
    with np.errstate(all="ignore"):
        for step in range(steps):
            rows = g.choice(N, n, replace=False).astype(np.int32)
            eps = g.normal(size=(n, K)).astype(np.float32)
            got.append(e.train_step(rows, eps))
            lo, c = O.train_step(w, opt, (codes[rows] * np.float32(0.5)).astype(np.float32), eps, D, K)
            want.append((float(lo), float(c["recon"].mean()), float(c["kl"].mean())))
    e.close()
    for step, (a, b) in enumerate(zip(got, want)):
        print(step, "engine", a, "oracle", b)
    for step, (a, b) in enumerate(zip(got, want)):
        assert np.isfinite(a[0]) == np.isfinite(b[0]), (step, a, b)
        if np.isfinite(b[0]):
            tol = RTOL if step == 0 else 5e-3 if W == 128 else 2e-4
            assert abs(a[0] - b[0]) <= tol * abs(b[0]), (step, a, b)
            assert abs(a[1] - b[1]) <= tol * abs(b[1]), (step, a, b)
    if W != 128:
        return
    assert want[1][0] > 1e3 * want[0][0]  # the blow-up itself: what makes lr = 1e-3 unusable at this fan-in
