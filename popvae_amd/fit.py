"""``vae.fit`` and its callbacks on top of the HIP engine (host side of the hot loop).

Mirrors /root/reference/popvae.py:428-526:

* ``ModelCheckpoint(save_best_only=True, monitor="val_loss")``  :429-435 -> ``Engine.snapshot_best`` (a
  device-to-device copy in HBM; nothing is written to disk per epoch)
* ``EarlyStopping(monitor="val_loss", min_delta=0, patience)``    :437-439
* ``ReduceLROnPlateau(factor=.5, patience=int(patience/4))``      :441-450
* ``saveLDpos`` / ``LambdaCallback``                             :453-486 -> ``_training_preds.txt``
* ``vae.fit(shuffle=True, validation_data=(testgen, None))``      :490-498
* history / latent_coords writers                                :503-523

The scalar state machines follow Keras 2.15.  The batch shuffle and the
reparameterisation noise come from TensorFlow's RNG in the reference, which no other program can
reproduce; here the shuffle uses ``numpy.random.Generator(PCG64(seed))`` and the noise the engine's
Philox stream (both injectable for tests).
"""
from __future__ import annotations

import time

import numpy as np


# ------------------------------------------------------------------------------------------------
# callbacks (Keras 2.15 semantics)
# ------------------------------------------------------------------------------------------------
class ModelCheckpoint:
    """popvae.py:429-435; mode auto -> min, saves when ``current < best`` (best starts at +inf)."""

    def __init__(self, engine, verbose=1, out=None, log=print):
        self.engine, self.verbose, self.out, self.log = engine, verbose, out, log
        self.best = np.inf
        self.best_epoch = -1

    def on_epoch_end(self, epoch, logs):
        current = logs["val_loss"]
        if np.less(current, self.best):
            if self.verbose:
                self.log(f"\nEpoch {epoch + 1}: val_loss improved from {self.best:.5f} to {current:.5f}, "
                         f"keeping weights in HBM" + (f" (for {self.out}_weights)" if self.out else ""))
            self.best = current
            self.best_epoch = epoch
            self.engine.snapshot_best()
        elif self.verbose:
            self.log(f"\nEpoch {epoch + 1}: val_loss did not improve from {self.best:.5f}")


class EarlyStopping:
    """popvae.py:437-439 (and :335-337 where ``patience`` is the float ``patience / 4``)."""

    def __init__(self, patience, min_delta=0.0):
        self.patience, self.min_delta = patience, abs(min_delta)
        self.wait, self.best, self.stopped_epoch = 0, np.inf, 0
        self.stop_training = False

    def on_epoch_end(self, epoch, logs):
        current = logs["val_loss"]
        self.wait += 1
        if np.less(current - self.min_delta, self.best):
            self.best = current
            self.wait = 0
            return
        if self.wait >= self.patience and epoch > 0:
            self.stopped_epoch = epoch
            self.stop_training = True


class ReduceLROnPlateau:
    """popvae.py:441-450; cooldown 0, min_lr 0, min_delta 0.  Adds ``lr`` to the epoch's logs."""

    def __init__(self, engine, patience, factor=0.5, min_lr=0.0, min_delta=0.0, verbose=0, log=print):
        self.engine, self.patience, self.factor, self.min_lr = engine, patience, factor, min_lr
        self.min_delta, self.verbose, self.log = min_delta, verbose, log
        self.wait, self.best = 0, np.inf

    def on_epoch_end(self, epoch, logs):
        lr = np.float32(self.engine.lr)
        logs["lr"] = lr
        current = logs["val_loss"]
        if np.less(current, self.best - self.min_delta):
            self.best = current
            self.wait = 0
            return
        self.wait += 1
        if self.wait >= self.patience:
            if lr > np.float32(self.min_lr):
                new_lr = max(np.float32(lr * np.float32(self.factor)), np.float32(self.min_lr))
                self.engine.lr = new_lr
                if self.verbose:
                    self.log(f"\nEpoch {epoch + 1}: ReduceLROnPlateau reducing learning rate to {new_lr}.")
                self.wait = 0


class SaveLDpos:
    """``saveLDpos`` (popvae.py:453-486): z_mean of ALL samples, original order, every
    ``frequency`` epochs, appended to ``out_training_preds.txt``."""

    def __init__(self, engine, n_samples, samples, out, frequency, write=True):
        self.engine, self.n, self.samples, self.out, self.frequency = engine, n_samples, samples, out, frequency
        self.write = write
        self.records = []

    def on_epoch_end(self, epoch, logs):
        if epoch % self.frequency != 0:
            return
        mu = self.engine.encode(np.arange(self.n, dtype=np.int32))[0]
        self.records.append((epoch, mu))
        if self.write:
            write_training_preds(self.out + "_training_preds.txt", mu, self.samples, epoch)


# ------------------------------------------------------------------------------------------------
# writers (file formats of popvae.py:453-474, 503-505, 512-523, 372-374)
# ------------------------------------------------------------------------------------------------
def write_training_preds(path, pred, samples, epoch):
    import pandas as pd
    df = pd.DataFrame(np.asarray(pred, dtype=np.float32))
    df["sampleID"] = np.asarray(samples)
    df["epoch"] = epoch
    if epoch == 0:
        df.to_csv(path, sep="\t", index=False, mode="w", header=True)
    else:
        df.to_csv(path, sep="\t", index=False, mode="a", header=False)


def write_history(path, history):
    import pandas as pd
    pd.DataFrame(history).to_csv(path, sep="\t")


def write_latent_coords(path, pred, samples, latent_dim):
    """pred = (z_mean, z_log_var, z).  latent_dim == 2: mean1, mean2, sd1, sd2 (sd* is the raw z_log_var
    head, popvae.py:516-517); otherwise LD1..LDk (means only)."""
    import pandas as pd
    if latent_dim == 2:
        p = pd.DataFrame()
        p["mean1"] = pred[0][:, 0]
        p["mean2"] = pred[0][:, 1]
        p["sd1"] = pred[1][:, 0]
        p["sd2"] = pred[1][:, 1]
    else:
        p = pd.DataFrame(pred[0])
        p.columns = ["LD" + str(x + 1) for x in range(len(p.columns))]
    p["sampleID"] = np.asarray(samples)
    p.to_csv(path, sep="\t", index=False)


def write_param_grid(path, rows):
    import pandas as pd
    df = pd.DataFrame(rows)
    df.to_csv(path, index=False, header=True)
    return df


def plot_history(path, history):
    """popvae.py:545-559."""
    import matplotlib
    matplotlib.use("Agg")
    from matplotlib import pyplot as plt
    fig = plt.figure(figsize=(3, 1.5), dpi=200)
    plt.rcParams.update({"font.size": 7})
    ax1 = fig.add_axes([0, 0, 1, 1])
    ax1.plot(history["val_loss"][3:], "--", color="black", lw=0.5, label="Validation Loss")
    ax1.plot(history["loss"][3:], "-", color="black", lw=0.5, label="Training Loss")
    ax1.set_xlabel("Epoch")
    ax1.legend()
    fig.savefig(path, bbox_inches="tight")
    plt.close(fig)


# ------------------------------------------------------------------------------------------------
# fit
# ------------------------------------------------------------------------------------------------
def default_perm_fn(seed):
    g = np.random.default_rng(seed)
    return lambda epoch, n: g.permutation(n)


def fit(engine, train, test, max_epochs, patience, lr_patience=None, checkpoint=True, prediction_freq=None,
        n_samples=None, samples=None, out=None, perm_fn=None, verbose=1, log=print, write_preds=True):
    """One ``vae.fit`` call with the reference's callback list.

    engine   : ``Engine`` or ``ShardedEngine`` with the genotype matrix resident
    train    : row indices of the training samples (order of popvae.py:257/264)
    test     : validation rows
    Returns (history dict with loss / val_loss / lr, callbacks dict).
    """
    train = np.ascontiguousarray(train, dtype=np.int32)
    test = np.ascontiguousarray(test, dtype=np.int32)
    perm_fn = perm_fn or default_perm_fn(None)
    ck = ModelCheckpoint(engine, verbose=verbose, out=out, log=log) if checkpoint else None
    es = EarlyStopping(patience)
    rl = ReduceLROnPlateau(engine, int(patience / 4) if lr_patience is None else lr_patience, verbose=verbose, log=log)
    sp = None
    if prediction_freq:
        sp = SaveLDpos(engine, n_samples, samples, out, prediction_freq, write=write_preds and out is not None)
    history = {"loss": [], "val_loss": [], "lr": []}
    for epoch in range(int(max_epochs)):
        t0 = time.time()
        perm = np.asarray(perm_fn(epoch, len(train)))
        loss, val_loss = engine.train_epoch(train[perm], test)
        logs = {"loss": float(np.float32(loss)), "val_loss": float(np.float32(val_loss))}
        # callback order of popvae.py:495: checkpointer, earlystop, reducelr, print_predictions
        if ck:
            ck.on_epoch_end(epoch, logs)
        es.on_epoch_end(epoch, logs)
        rl.on_epoch_end(epoch, logs)
        if sp:
            sp.on_epoch_end(epoch, logs)
        for k in history:
            history[k].append(logs[k])
        if verbose:
            log(f"Epoch {epoch + 1}/{max_epochs} - {time.time() - t0:.2f}s - loss: {logs['loss']:.4f} - "
                f"val_loss: {logs['val_loss']:.4f} - lr: {logs['lr']:.4e}")
        if es.stop_training:
            break
    history["lr"] = [np.float32(x) for x in history["lr"]]
    return history, {"checkpoint": ck, "earlystop": es, "reducelr": rl, "preds": sp}


def grid_paramsets(width_range, depth_range):
    """popvae.py:280 (width-major order)."""
    return [[int(x), int(y)] for x in width_range for y in depth_range]


def grid_fit_one(make_engine, width, depth, train, test, max_epochs, patience, perm_fn=None, timings=None):
    """One cell of the --search_network_sizes grid (popvae.py:284-369): quarter patience (a float),
    quarter epochs, no checkpoint, min(val_loss) reported.  timings (a list, optional): gets (width, depth, seconds of the
    cell incl. building the model, epochs run) appended -- measurement only, the reference's files do not carry it."""
    tmp_patience = patience / 4
    t0 = time.time()
    eng = make_engine(width, depth)
    try:
        hist, _ = fit(eng, train, test, max_epochs=int(max_epochs / 4), patience=tmp_patience,
                      lr_patience=int(tmp_patience / 4), checkpoint=False, prediction_freq=None, perm_fn=perm_fn,
                      verbose=0)
    finally:
        eng.close()
    if timings is not None:
        timings.append((width, depth, time.time() - t0, len(hist["val_loss"])))
    return {"width": width, "depth": depth, "val_loss": float(np.min(hist["val_loss"]))}


def lpt_schedule(costs, n_workers):
    """Longest-processing-time-first assignment of grid cells to GPUs: returns a list of
    task-index lists, one per worker; deterministic."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * n_workers
    out = [[] for _ in range(n_workers)]
    for i in order:
        w = min(range(n_workers), key=lambda k: (load[k], k))
        out[w].append(i)
        load[w] += costs[i]
    return out


def grid_search(make_engine, width_range, depth_range, train, test, max_epochs, patience, n_snps, out=None,
                perm_fn_factory=None, rank=0, world=1, gather=None, log=print, timings=None):
    """--search_network_sizes (popvae.py:275-380).  With world > 1 the cells are spread one network per GPU
    (task parallel, no data-path collective); ``gather(list) -> list of lists`` collects the per-rank results
    (e.g. ``torch.distributed.all_gather_object``).  Returns (rows in reference order, best width, best depth)."""
    paramsets = grid_paramsets(width_range, depth_range)
    costs = [float(n_snps) * w * 2 + (2 * d - 1) * w * w for w, d in paramsets]
    mine = lpt_schedule(costs, world)[rank]
    local = {}
    for i in mine:
        w, d = paramsets[i]
        log(f"width={w}\ndepth={d}")
        pf = perm_fn_factory(i) if perm_fn_factory else None
        local[i] = grid_fit_one(make_engine, w, d, train, test, max_epochs, patience, perm_fn=pf, timings=timings)
    if world > 1:
        merged = {}
        for part in gather(local):
            merged.update(part)
        local = merged
    rows = [local[i] for i in range(len(paramsets))]
    vals = np.array([r["val_loss"] for r in rows], dtype=np.float64)
    best = rows[int(np.flatnonzero(vals == np.nanmin(vals))[0])]
    if out is not None and rank == 0:
        df = write_param_grid(out + "_param_grid.csv", rows)
        log(df)
    return rows, int(best["width"]), int(best["depth"])
