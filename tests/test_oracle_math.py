"""Pins the oracle's model math (oracle/popvae_oracle.py) by float64 finite differences and by
PyTorch-CPU autograd of an independently written forward.  The Keras/TF libraries themselves are
absent from the image: parity against them is unpinned (see the oracle header)."""
import numpy as np
import pytest
import torch

from oracle import popvae_oracle as O

L, W, D, K, B = 64, 8, 3, 2, 5


def _setup(seed=0, bias_noise=True):
    rng = np.random.default_rng(seed)
    w = O.init_weights(L, W, D, K, seed=seed)
    if bias_noise:
        for t in w:
            if t.ndim == 1:
                t += rng.normal(0, 0.1, t.shape)
    X = rng.integers(0, 3, size=(B, L)) * 0.5
    eps = rng.normal(size=(B, K))
    return w, X, eps


@pytest.mark.parametrize("mode", ["logits", "probs"])
def test_finite_differences(mode):
    w, X, eps = _setup()
    c = O.forward(w, X, eps, D, K, mode)
    g = O.backward(w, c, D, K, mode)
    rng = np.random.default_rng(1)
    h = 1e-6
    for ti, t in enumerate(w):
        flat = t.reshape(-1)
        for idx in rng.choice(flat.size, size=min(6, flat.size), replace=False):
            old = flat[idx]
            flat[idx] = old + h
            lp = O.forward(w, X, eps, D, K, mode)["loss"]
            flat[idx] = old - h
            lm = O.forward(w, X, eps, D, K, mode)["loss"]
            flat[idx] = old
            fd = (lp - lm) / (2 * h)
            an = g[ti].reshape(-1)[idx]
            assert abs(fd - an) <= 1e-6 * max(1.0, abs(an)), (ti, idx, fd, an)


def _torch_forward(tw, X, eps):
    h = X
    for i in range(D):
        h = torch.nn.functional.elu(h @ tw[2 * i] + tw[2 * i + 1])
    o = 2 * D
    mu = h @ tw[o] + tw[o + 1]
    s = h @ tw[o + 2] + tw[o + 3]
    z = mu + torch.exp(s) * eps
    o = 2 * D + 4
    g = z @ tw[o] + tw[o + 1]
    for i in range(1, D):
        g = torch.nn.functional.elu(g @ tw[o + 2 * i] + tw[o + 2 * i + 1])
    a = g @ tw[o + 2 * D] + tw[o + 2 * D + 1]
    bce = torch.nn.functional.binary_cross_entropy_with_logits(a, X, reduction="none")
    recon = bce.mean(dim=1) * X.shape[1]
    kl = -0.5 * torch.sum(1 + s - mu ** 2 - torch.exp(s), dim=1)
    return torch.mean(recon + kl)


def test_against_torch_autograd():
    w, X, eps = _setup(3)
    c = O.forward(w, X, eps, D, K)
    g = O.backward(w, c, D, K)
    tw = [torch.tensor(t, dtype=torch.float64, requires_grad=True) for t in w]
    loss = _torch_forward(tw, torch.tensor(X), torch.tensor(eps))
    loss.backward()
    assert abs(loss.item() - c["loss"]) < 1e-10 * abs(c["loss"])
    for a, b in zip(g, tw):
        np.testing.assert_allclose(a, b.grad.numpy(), rtol=1e-9, atol=1e-12)


def test_bce_modes_agree_to_1e7():
    w, X, eps = _setup(5)
    a = O.forward(w, X, eps, D, K, "logits")["loss"]
    b = O.forward(w, X, eps, D, K, "probs")["loss"]
    assert abs(a - b) < 1e-5 * abs(a)


def test_keras_adam_first_steps():
    """K7: hand-computed two steps on a scalar."""
    w = [np.array([1.0])]
    opt = O.KerasAdam(w)
    g = [np.array([0.5])]
    opt.step(w, g)
    m, v = 0.05, 0.00025
    al = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    exp1 = 1.0 - al * m / (np.sqrt(v) + 1e-7)
    assert np.isclose(w[0][0], exp1, rtol=0, atol=1e-15)
    opt.step(w, [np.array([-0.25])])
    m = m + (-0.25 - m) * 0.1
    v = v + (0.0625 - v) * 0.001
    al = 1e-3 * np.sqrt(1 - 0.999 ** 2) / (1 - 0.9 ** 2)
    assert np.isclose(w[0][0], exp1 - al * m / (np.sqrt(v) + 1e-7), rtol=0, atol=1e-15)


def test_callbacks_state_machines():
    """K10-K12 on hand-constructed val_loss sequences (SURVEY 8c pin 3)."""
    seq = [10, 9, 9.5, 9.4, 9.3, 9.2, 9.1, 9.05, 9.01, 9.0, 8.0]
    es = O.EarlyStopping(patience=3)
    stops = [es.on_epoch_end(e, v) for e, v in enumerate(seq)]
    assert stops.index(True) == 4  # best at epoch 1; wait hits 3 at epoch 4
    # float patience of the grid search (popvae.py:277,335-337): 12.5 -> stop when wait == 13
    es = O.EarlyStopping(patience=12.5)
    st = [es.on_epoch_end(e, 1.0 + e) for e in range(20)]
    assert st.index(True) == 13
    # patience never triggers at epoch 0
    es = O.EarlyStopping(patience=0)
    assert es.on_epoch_end(0, 5.0) is False
    rl = O.ReduceLROnPlateau(patience=2)
    lr = np.float32(1e-3)
    lrs = []
    for e, v in enumerate([5, 6, 6, 6, 6, 4, 5, 5]):
        lrs.append(float(lr))
        lr = rl.on_epoch_end(e, v, lr)
    assert np.allclose(lrs, [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 2.5e-4, 2.5e-4, 2.5e-4])
    assert np.isclose(float(lr), 1.25e-4)
    ck = O.ModelCheckpointBest()
    saves = [ck.on_epoch_end(e, v) for e, v in enumerate([3, 3, 2, 2.5, 1])]
    assert saves == [True, False, True, False, True] and ck.best_epoch == 4


def test_fit_runs_and_improves():
    rng = np.random.default_rng(0)
    N = 40
    pop = rng.integers(0, 2, N)
    f = np.clip(rng.beta(.5, .5, L)[None, :] + 0.3 * (pop[:, None] - .5) * rng.normal(size=L)[None, :], .02, .98)
    dc = rng.binomial(2, f) * 0.5
    w = O.init_weights(L, W, D, K, seed=1)
    train, test = np.arange(32), np.arange(32, 40)
    prng = np.random.default_rng(7)
    hist, best, preds = O.fit(w, dc, train, test, D, K, 16, 30, 50,
                              lambda e, n: prng.permutation(n), lambda *a: prng.normal(size=a[-1]),
                              prediction_freq=5)
    assert hist["loss"][-1] < hist["loss"][0]
    assert len(hist["lr"]) == 30 and [e for e, _ in preds] == [0, 5, 10, 15, 20, 25]


def test_oracle_reproduces_committed_golden_steps():
    """tests/golden/model_steps.npz (tests/make_golden_model.py): three Keras-Adam steps, both BCE forms."""
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_steps.npz"))
    L, W, D, K, N = (int(x) for x in G["dims"])
    nt = 4 * D + 6
    dc = G["codes"] * 0.5
    for mode in ("logits", "probs"):
        w = [G[f"w0_{i}"].astype(np.float64) for i in range(nt)]
        opt = O.KerasAdam(w)
        for s in range(3):
            c = O.forward(w, dc[G[f"rows_{s}"]], G[f"eps_{s}"].astype(np.float64), D, K, bce_mode=mode)
            gr = O.backward(w, c, D, K, bce_mode=mode)
            np.testing.assert_allclose([c["loss"], c["recon"].mean(), c["kl"].mean()], G[f"{mode}_loss_{s}"], rtol=1e-12)
            if s == 0:
                for i, t in enumerate(gr):
                    np.testing.assert_allclose(t, G[f"{mode}_grad0_{i}"], rtol=1e-10, atol=1e-14)
            opt.step(w, gr)
        for i in range(nt):
            np.testing.assert_allclose(w[i], G[f"{mode}_w3_{i}"], rtol=1e-10, atol=1e-14)
    # the two BCE forms agree to O(1e-7) on the loss -- which is why only the logits form is built in HIP
    assert abs(G["logits_loss_0"][0] - G["probs_loss_0"][0]) <= 1e-5 * abs(G["logits_loss_0"][0])
