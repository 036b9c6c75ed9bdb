"""CPU restatement ("oracle") of the popVAE hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module -- and only as the checker / reported baseline.  Nothing under ``popvae_amd/`` imports it; the
product path fails loudly when the HIP library is missing.

What it restates (reference = /root/reference/popvae.py, cited per function):

* integer preprocessing  popvae.py:211-269   (allele counts, biallelic + "singleton" filter,
  seeded binomial imputation, x0.5 scaling, --max_SNPs subsample, train/validation split)
* model math             popvae.py:383-426   (Dense/ELU encoder, reparameterisation, decoder,
  BCE*L + KL loss) with hand-derived backward and the Keras-2.15 Adam update
* fit loop + callbacks   popvae.py:428-523   (ModelCheckpoint best-only, EarlyStopping,
  ReduceLROnPlateau, saveLDpos, history, final encode)

Pinning status
--------------
* numpy legacy RNG calls (seed / binomial / choice): the oracle calls numpy itself, i.e. the very
  library the reference calls (popvae.py:192,235,252,257,264) -> PINNED by direct call.
* scikit-allel calls (count_alleles / to_allele_counts / is_biallelic / is_missing, popvae.py:214-221)
  and everything Keras/TensorFlow computes (popvae.py:384-498): those libraries are absent from this
  image and the reference ships no golden outputs (no tests, empty out/) -> **parity unpinned** against
  the real libraries.  The restatement is pinned instead by (i) float64 finite differences,
  (ii) PyTorch-CPU autograd of the same forward (tests/test_oracle_math.py) and (iii) the published
  definitions of the Keras ops.
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------------
# Integer preprocessing (popvae.py:211-269)
# --------------------------------------------------------------------------------------------


def read_vcf_gt(path):
    """Minimal VCF GT reader standing in for ``allel.read_vcf`` (popvae.py:202-204).

    Returns (gen int8 [L0, N, 2] with -1 = missing, samples object array).  Only diploid GT is
    handled ('/' or '|' separators, '.' = missing); GT is located via the FORMAT column.
    Pure-Python loops: small fixtures only.
    """
    samples = None
    rows = []
    with open(path, "rt") as fh:
        for line in fh:
            if line.startswith("##"):
                continue
            f = line.rstrip("\n").rstrip("\r").split("\t")
            if line.startswith("#CHROM"):
                samples = np.array([s for s in f[9:] if s != ""], dtype=object)
                continue
            if len(f) < 10:
                continue
            fmt = f[8].strip().split(":")
            gi = fmt.index("GT")
            calls = np.full((len(samples), 2), -1, dtype=np.int8)
            for i in range(len(samples)):
                g = f[9 + i].split(":")[gi] if 9 + i < len(f) else "./."
                g = g.replace("|", "/").split("/")
                for k in range(2):
                    a = g[k] if k < len(g) else "."
                    calls[i, k] = -1 if (a == "." or a == "") else int(a)
            rows.append(calls)
    return np.stack(rows, axis=0), samples


def count_alleles(gen):
    """``GenotypeArray.count_alleles`` (popvae.py:214): ac_all[l, a] = #calls == a (a >= 0)."""
    A = int(gen.max()) + 1
    ac = np.zeros((gen.shape[0], A), dtype=np.int32)
    for a in range(A):
        ac[:, a] = np.sum(gen == a, axis=(1, 2))
    return ac


def to_allele_counts(gen):
    """``GenotypeArray.to_allele_counts`` (popvae.py:215): ac[l, i, a] in {0,1,2}."""
    A = int(gen.max()) + 1
    out = np.zeros(gen.shape[:2] + (A,), dtype=np.uint8)
    for a in range(A):
        out[:, :, a] = np.sum(gen == a, axis=2)
    return out


def filter_and_impute(gen, rng=np.random):
    """popvae.py:213-238.  ``rng`` must be the (already seeded) legacy numpy global module or a
    ``RandomState``; draws happen in the reference's order (sample-major, ascending site).

    Returns dc float64 [N, L] with values in {0, 0.5, 1} plus the kept-site mask over input sites.
    """
    ac_all = count_alleles(gen)  # :214
    ac = to_allele_counts(gen)  # :215
    biallel = np.count_nonzero(ac_all, axis=1) == 2  # :218 is_biallelic
    dc_all = ac_all[biallel, 1]  # :219
    dc = np.array(ac[biallel, :, 1], dtype="int_")  # :220
    missingness = np.any(gen[biallel, :, :] < 0, axis=2)  # :221 is_missing
    ninds = np.array([np.sum(x) for x in ~missingness])  # :224
    singletons = np.array([x <= 2 for x in dc_all])  # :225
    keep = np.flatnonzero(biallel)[~singletons]
    dc_all = dc_all[~singletons]
    dc = dc[~singletons, :]
    ninds = ninds[~singletons]
    missingness = missingness[~singletons, :]
    af = np.array([dc_all[x] / (ninds[x] * 2) for x in range(dc_all.shape[0])])  # :232
    for i in range(dc.shape[1]):  # :233-235
        indmiss = missingness[:, i]
        dc[indmiss, i] = rng.binomial(2, af[indmiss])
    dc = np.transpose(dc)  # :237
    dc = dc * 0.5  # :238
    return dc, keep


def subsample_snps(dc, max_SNPs, rng=np.random):
    """popvae.py:250-252 (unsorted column order, consumes the RNG before the split)."""
    return dc[:, rng.choice(range(dc.shape[1]), max_SNPs, replace=False)]


def train_test_split(ninds, train_prop, rng=np.random):
    """popvae.py:254-269 -> (train index array in permutation order, test index array)."""
    train = rng.choice(range(ninds), int(train_prop * ninds), replace=False)
    if train_prop == 1:
        test = train
    else:
        test = np.array([x for x in range(ninds) if x not in train])
    return train, test


def pack2bit(dc):
    """Pack float dc [N, L] in {0,.5,1} to 2-bit codes (code = 2*dc), 4 SNPs per byte, SNP l at
    bits 2*(l%4) of byte l//4; rows padded with code 0 to a multiple of 32 SNPs (8 bytes)."""
    codes = np.rint(np.asarray(dc) * 2).astype(np.uint8)
    N, L = codes.shape
    Lp = (L + 31) // 32 * 32
    c = np.zeros((N, Lp), dtype=np.uint8)
    c[:, :L] = codes
    c = c.reshape(N, Lp // 4, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


def unpack2bit(packed, L):
    p = np.asarray(packed, dtype=np.uint8)
    c = np.stack([(p >> s) & 3 for s in (0, 2, 4, 6)], axis=-1).reshape(p.shape[0], -1)
    return c[:, :L]


# --------------------------------------------------------------------------------------------
# Model (popvae.py:383-426)
# --------------------------------------------------------------------------------------------


def tensor_specs(L, W, D, K):
    """Keras ``vae.get_weights()`` order: encoder then decoder, kernel (in, out) then bias.
    popvae.py:394-398 (encoder), :404-409 (decoder)."""
    specs = [("enc_dense0/kernel", (L, W)), ("enc_dense0/bias", (W,))]
    for i in range(1, D):
        specs += [(f"enc_dense{i}/kernel", (W, W)), (f"enc_dense{i}/bias", (W,))]
    specs += [("z_mean/kernel", (W, K)), ("z_mean/bias", (K,)),
              ("z_log_var/kernel", (W, K)), ("z_log_var/bias", (K,))]
    specs += [("dec_dense0/kernel", (K, W)), ("dec_dense0/bias", (W,))]
    for i in range(1, D):
        specs += [(f"dec_dense{i}/kernel", (W, W)), (f"dec_dense{i}/bias", (W,))]
    specs += [("dec_out/kernel", (W, L)), ("dec_out/bias", (L,))]
    return specs


def init_weights(L, W, D, K, seed, dtype=np.float64):
    """K1: glorot_uniform kernels, zero biases.  (The TF RNG stream itself is not reproducible.)"""
    rng = np.random.default_rng(seed)
    out = []
    for name, shape in tensor_specs(L, W, D, K):
        if len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            out.append(rng.uniform(-lim, lim, size=shape).astype(dtype))
        else:
            out.append(np.zeros(shape, dtype=dtype))
    return out


def _elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))  # K2, alpha = 1


def forward(weights, X, eps, D, K, bce_mode="logits"):
    """Forward + loss for one batch (popvae.py:394-424).  X [B, L] in {0,.5,1}; eps [B, K].

    Returns a dict with every intermediate the backward needs plus loss/recon/kl.
    M4: z = mu + exp(s) * eps (NOT exp(s/2)).  M7: loss = mean_b(L*mean_l(bce) + kl).
    K5: bce_mode 'logits' = max(a,0) - a*x + log1p(exp(-|a|)); 'probs' = clipped-probability form.
    """
    w = weights
    B = X.shape[0]
    c = {"X": X, "eps": eps}
    h = X
    hs = []
    for i in range(D):
        h = _elu(h @ w[2 * i] + w[2 * i + 1])
        hs.append(h)
    c["hs"] = hs
    o = 2 * D
    mu = h @ w[o] + w[o + 1]
    s = h @ w[o + 2] + w[o + 3]
    z = mu + np.exp(s) * eps
    c.update(mu=mu, s=s, z=z)
    o = 2 * D + 4
    g = z @ w[o] + w[o + 1]  # M5: linear
    gs = [g]
    for i in range(1, D):
        g = _elu(g @ w[o + 2 * i] + w[o + 2 * i + 1])
        gs.append(g)
    c["gs"] = gs
    a = g @ w[o + 2 * D] + w[o + 2 * D + 1]
    c["a"] = a
    if bce_mode == "logits":
        bce = np.maximum(a, 0) - a * X + np.log1p(np.exp(-np.abs(a)))
    else:
        e7 = X.dtype.type(1e-7)
        p = 1.0 / (1.0 + np.exp(-a))
        pc = np.clip(p, e7, 1 - e7)
        bce = -(X * np.log(pc + e7) + (1 - X) * np.log(1 - pc + e7))
        c["p"], c["pc"] = p, pc
    L = X.shape[1]
    recon = bce.mean(axis=1) * L  # :417-418 (mean over L, times L)
    kl = -0.5 * np.sum(1 + s - mu * mu - np.exp(s), axis=1)  # :419-421
    c["recon"], c["kl"] = recon, kl
    c["loss"] = np.mean(recon + kl)  # :423
    return c


def backward(weights, c, D, K, bce_mode="logits"):
    """Hand-derived gradients of ``forward``'s loss wrt every tensor (same order as weights)."""
    w = weights
    X, eps = c["X"], c["eps"]
    B = X.shape[0]
    grads = [None] * len(w)
    a = c["a"]
    if bce_mode == "logits":
        p = 1.0 / (1.0 + np.exp(-a))
        da = (p - X) / B
    else:
        e7 = X.dtype.type(1e-7)
        p, pc = c["p"], c["pc"]
        inside = (p >= e7) & (p <= 1 - e7)
        dpc = -(X / (pc + e7) - (1 - X) / (1 - pc + e7))
        da = np.where(inside, dpc, 0.0) * p * (1 - p) / B
    o = 2 * D + 4
    gs = c["gs"]
    grads[o + 2 * D] = gs[-1].T @ da
    grads[o + 2 * D + 1] = da.sum(axis=0)
    dg = da @ w[o + 2 * D].T
    for i in range(D - 1, 0, -1):
        dpre = dg * np.where(gs[i] > 0, 1.0, gs[i] + 1.0)
        grads[o + 2 * i] = gs[i - 1].T @ dpre
        grads[o + 2 * i + 1] = dpre.sum(axis=0)
        dg = dpre @ w[o + 2 * i].T
    grads[o] = c["z"].T @ dg
    grads[o + 1] = dg.sum(axis=0)
    dz = dg @ w[o].T
    mu, s = c["mu"], c["s"]
    es = np.exp(s)
    dmu = dz + mu / B
    ds = dz * eps * es + 0.5 * (es - 1.0) / B
    o = 2 * D
    hs = c["hs"]
    grads[o] = hs[-1].T @ dmu
    grads[o + 1] = dmu.sum(axis=0)
    grads[o + 2] = hs[-1].T @ ds
    grads[o + 3] = ds.sum(axis=0)
    dh = dmu @ w[o].T + ds @ w[o + 2].T
    for i in range(D - 1, -1, -1):
        dpre = dh * np.where(hs[i] > 0, 1.0, hs[i] + 1.0)
        inp = hs[i - 1] if i > 0 else X
        grads[2 * i] = inp.T @ dpre
        grads[2 * i + 1] = dpre.sum(axis=0)
        if i > 0:
            dh = dpre @ w[2 * i].T
    return grads


class KerasAdam:
    """K7: Keras-2.15 Adam (popvae.py:426).  eps OUTSIDE the bias correction:
    alpha = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= alpha*m/(sqrt(v)+eps)."""

    def __init__(self, weights, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = [np.zeros_like(x) for x in weights]
        self.v = [np.zeros_like(x) for x in weights]

    def alpha(self):
        t = self.t
        return self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)

    def step(self, weights, grads):
        self.t += 1
        al = self.alpha()
        for w, g, m, v in zip(weights, grads, self.m, self.v):
            dt = w.dtype.type
            m += (g - m) * dt(1 - self.b1)
            v += (g * g - v) * dt(1 - self.b2)
            w -= dt(al) * m / (np.sqrt(v) + dt(self.eps))


def train_step(weights, opt, X, eps, D, K, bce_mode="logits"):
    c = forward(weights, X, eps, D, K, bce_mode)
    g = backward(weights, c, D, K, bce_mode)
    opt.step(weights, g)
    return c["loss"], c


def encode(weights, X, eps, D, K):
    """``encoder.predict`` (popvae.py:455,509): [z_mean, z_log_var, z]."""
    h = X
    for i in range(D):
        h = _elu(h @ weights[2 * i] + weights[2 * i + 1])
    o = 2 * D
    mu = h @ weights[o] + weights[o + 1]
    s = h @ weights[o + 2] + weights[o + 3]
    return mu, s, mu + np.exp(s) * eps


def decode(weights, z, D, K):
    """``decoder.predict`` (scripts/popvae_decoder_tests_HGDP.py:156)."""
    o = 2 * D + 4
    g = z @ weights[o] + weights[o + 1]
    for i in range(1, D):
        g = _elu(g @ weights[o + 2 * i] + weights[o + 2 * i + 1])
    a = g @ weights[o + 2 * D] + weights[o + 2 * D + 1]
    return 1.0 / (1.0 + np.exp(-a))


# --------------------------------------------------------------------------------------------
# Callback state machines (popvae.py:429-450) -- K10, K11, K12
# --------------------------------------------------------------------------------------------


class ModelCheckpointBest:
    """K10: save when current < best (np.less), best starts at +inf (popvae.py:429-435)."""

    def __init__(self):
        self.best = np.inf
        self.best_epoch = -1

    def on_epoch_end(self, epoch, val_loss):
        if np.less(val_loss, self.best):
            self.best = val_loss
            self.best_epoch = epoch
            return True
        return False


class EarlyStopping:
    """K11 (popvae.py:437-439): min_delta 0, patience may be a float (grid search: patience/4)."""

    def __init__(self, patience, min_delta=0.0):
        self.patience, self.min_delta = patience, abs(min_delta)
        self.wait, self.best, self.stopped_epoch = 0, np.inf, 0

    def on_epoch_end(self, epoch, val_loss):
        self.wait += 1
        if np.less(val_loss - self.min_delta, self.best):
            self.best = val_loss
            self.wait = 0
            return False
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            return True
        return False


class ReduceLROnPlateau:
    """K12 (popvae.py:441-450): factor .5, cooldown 0, min_lr 0, min_delta 0; lr kept in float32."""

    def __init__(self, patience, factor=0.5, min_lr=0.0, min_delta=0.0):
        self.patience, self.factor, self.min_lr, self.min_delta = patience, factor, min_lr, min_delta
        self.wait, self.best = 0, np.inf

    def on_epoch_end(self, epoch, val_loss, lr):
        """Returns the (possibly reduced) lr; the lr LOGGED for this epoch is the incoming one."""
        if np.less(val_loss, self.best - self.min_delta):
            self.best = val_loss
            self.wait = 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                old = np.float32(lr)
                if old > np.float32(self.min_lr):
                    lr = max(np.float32(old * np.float32(self.factor)), np.float32(self.min_lr))
                    self.wait = 0
        return np.float32(lr)


def fit(weights, dc, train, test, D, K, batch_size, max_epochs, patience, perm_fn, eps_fn,
        bce_mode="logits", lr_patience=None, prediction_freq=None, use_checkpoint=True):
    """``vae.fit`` + callbacks (popvae.py:490-498), K8/K9.

    perm_fn(epoch, n) -> permutation of range(n) (injected: the TF shuffle stream is not
    reproducible); eps_fn(kind, epoch, step, shape) -> N(0,1) draws.
    Returns history dict (loss, val_loss, lr), best weights, and saveLDpos predictions.
    """
    opt = KerasAdam(weights)
    ck = ModelCheckpointBest()
    es = EarlyStopping(patience)
    rl = ReduceLROnPlateau(int(patience / 4) if lr_patience is None else lr_patience)
    hist = {"loss": [], "val_loss": [], "lr": []}
    best = [x.copy() for x in weights]
    preds = []
    Xtr, Xva = dc[train], dc[test]
    dt = weights[0].dtype
    for epoch in range(max_epochs):
        perm = perm_fn(epoch, len(train))
        losses = []
        for st, i0 in enumerate(range(0, len(train), batch_size)):
            idx = perm[i0:i0 + batch_size]
            X = Xtr[idx].astype(dt)
            l, _ = train_step(weights, opt, X, eps_fn("train", epoch, st, (len(idx), K)).astype(dt),
                              D, K, bce_mode)
            losses.append(l)
        vl = []
        for st, i0 in enumerate(range(0, len(test), batch_size)):
            X = Xva[i0:i0 + batch_size].astype(dt)
            c = forward(weights, X, eps_fn("val", epoch, st, (X.shape[0], K)).astype(dt), D, K, bce_mode)
            vl.append(c["loss"])
        loss, val_loss = float(np.mean(losses)), float(np.mean(vl))  # K9: unweighted batch mean
        stop = False
        if use_checkpoint and ck.on_epoch_end(epoch, val_loss):
            best = [x.copy() for x in weights]
        if es.on_epoch_end(epoch, val_loss):
            stop = True
        lr_logged = np.float32(opt.lr)
        opt.lr = float(rl.on_epoch_end(epoch, val_loss, opt.lr))
        if prediction_freq and epoch % prediction_freq == 0:  # popvae.py:453-474
            mu, _, _ = encode(weights, dc.astype(dt), np.zeros((dc.shape[0], K), dt), D, K)
            preds.append((epoch, mu.copy()))
        hist["loss"].append(loss)
        hist["val_loss"].append(val_loss)
        hist["lr"].append(float(lr_logged))
        if stop:
            break
    return hist, best, preds


# ---------------------------------------------------------------------------------------------
# PCA (popvae.py:528-533 -> allel.pca).  scikit-allel is not importable here and holds no fixture for this
# call inside /root/reference: PARITY UNPINNED -- this restates its published algorithm
# (allel/stats/decomposition.py, GenotypePCA + allel/stats/preprocessing.py scalers) [UPSTREAM].
# ---------------------------------------------------------------------------------------------
def pca_scale(gn, scaler="patterson", ploidy=2):
    """gn: variants x samples (the reference passes dc.T with VALUES 0 / 0.5 / 1).  Returns (scaled, mean, std)."""
    gn = np.array(gn, dtype=np.float64)
    if isinstance(scaler, str):
        scaler = scaler.lower()
    mean = np.mean(gn, axis=1, keepdims=True)
    if scaler == "patterson":
        p = mean / ploidy
        std = np.sqrt(p * (1 - p))
    elif scaler == "standard":
        std = np.std(gn, axis=1, keepdims=True)
    else:
        std = np.ones_like(mean)
    gn -= mean
    gn /= std
    return gn, mean[:, 0], std[:, 0]


def pca(dc_values, scaler="Patterson", n_components=20):
    """allel.pca(dc.T, scaler, n_components)[0] for dc_values [N, L] in {0, 0.5, 1}: U[:, :n] * S[:n]."""
    import scipy.linalg
    gn, _, _ = pca_scale(np.asarray(dc_values).T, scaler)
    x = gn.T
    u, s, v = scipy.linalg.svd(x, full_matrices=False)
    ev = (s ** 2) / x.shape[0]
    return u[:, :n_components] * s[:n_components], (ev / np.sum(ev))[:n_components]
