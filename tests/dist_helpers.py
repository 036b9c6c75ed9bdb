"""Helpers for the world_size > 1 tests: a CPU stand-in for one rank's engine built from the oracle's pieces
(so that popvae_amd.dist.ShardedEngine -- the product's host protocol -- can be driven over gloo without a
GPU), and the worker functions that torch.multiprocessing spawns."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import popvae_oracle as O  # noqa: E402


class OracleShardEngine:
    """One rank's share of the model in float64 NumPy, cut at the same two exchange points as the C ABI
    (pvae_stage_a / b / c).  Same constructor and methods as popvae_amd.engine.Engine."""

    def __init__(self, n_snps, width, depth, latent_dim, batch_size, seed=0, device_id=0, n_snps_global=None,
                 snp_offset=0, rank=0, world=1, **adam):
        import torch
        self.L, self.W, self.D, self.K, self.B = n_snps, width, depth, latent_dim, batch_size
        self.latent_dim = latent_dim
        self._b0 = torch.zeros(32 * width, dtype=torch.float32)
        self._b1 = torch.zeros(32 * width + 32, dtype=torch.float32)
        self._lr = np.float32(adam.get("lr", 1e-3))
        self.w = None
        self.best = None

    # -- plumbing ------------------------------------------------------------------------------
    def exchange_buffer(self, which):
        t = self._b0 if which == 0 else self._b1
        return t, t.numel()

    def load_genotypes(self, packed):
        self.dc = O.unpack2bit(packed, self.L) * 0.5

    def set_weights(self, weights):
        self.w = [np.array(x, dtype=np.float64) for x in weights]
        self.opt = O.KerasAdam(self.w, lr=float(self._lr))

    def get_weights(self):
        return [x.astype(np.float32) for x in self.w]

    def get_grads(self):
        return [x.astype(np.float32) for x in self.grads]

    lr = property(lambda s: s._lr, lambda s, v: (setattr(s, "_lr", np.float32(v)), setattr(s.opt, "lr", float(v))))

    def snapshot_best(self):
        self.best = [x.copy() for x in self.w]

    def restore_best(self):
        for a, b in zip(self.w, self.best):
            a[...] = b

    def synchronize(self):
        pass

    def close(self):
        pass

    def pca_gram(self, scaler="patterson"):
        xs = O.pca_scale(self.dc.T, scaler)[0]
        return xs.T @ xs

    # -- the three stages ------------------------------------------------------------------------
    def stage_a(self, rows, eps, mode):
        self.n = len(rows)
        self.X = self.dc[np.asarray(rows)]
        self.eps = np.zeros((self.n, self.K)) if eps is None else np.asarray(eps, dtype=np.float64)
        part = np.zeros((32, self.W))
        part[:self.n] = self.X @ self.w[0]                                # partial first-layer sums
        self._b0[:] = __import__("torch").from_numpy(part.reshape(-1).astype(np.float32))

    def _narrow_forward(self):
        w, D, n = self.w, self.D, self.n
        P = self._b0.numpy().astype(np.float64).reshape(32, self.W)[:n]
        hs = [O._elu(P + w[1])]
        for i in range(1, D):
            hs.append(O._elu(hs[-1] @ w[2 * i] + w[2 * i + 1]))
        o = 2 * D
        mu = hs[-1] @ w[o] + w[o + 1]
        s = hs[-1] @ w[o + 2] + w[o + 3]
        z = mu + np.exp(s) * self.eps
        o = 2 * D + 4
        gs = [z @ w[o] + w[o + 1]]
        for i in range(1, D):
            gs.append(O._elu(gs[-1] @ w[o + 2 * i] + w[o + 2 * i + 1]))
        self.c = dict(hs=hs, mu=mu, s=s, z=z, gs=gs)

    def stage_encode(self, n):
        self._narrow_forward()
        c = self.c
        return c["mu"].astype(np.float32), c["s"].astype(np.float32), c["z"].astype(np.float32)

    def stage_b(self, mode):
        import torch
        self._narrow_forward()
        w, D, n, c = self.w, self.D, self.n, self.c
        o = 2 * D + 4
        a = c["gs"][-1] @ w[o + 2 * D] + w[o + 2 * D + 1]
        bce = np.maximum(a, 0) - a * self.X + np.log1p(np.exp(-np.abs(a)))
        da = (1.0 / (1.0 + np.exp(-a)) - self.X) / n
        self.grads = [None] * len(w)
        self.grads[o + 2 * D] = c["gs"][-1].T @ da
        self.grads[o + 2 * D + 1] = da.sum(axis=0)
        out = np.zeros(32 * self.W + 32)
        dg = np.zeros((32, self.W))
        dg[:n] = da @ w[o + 2 * D].T
        out[:32 * self.W] = dg.reshape(-1)
        out[32 * self.W] = bce.sum()
        self._b1[:] = torch.from_numpy(out.astype(np.float32))

    def stage_c(self, mode, want_loss=True):
        from popvae_amd.engine import MODE_EVAL, MODE_TRAIN
        w, D, n, c = self.w, self.D, self.n, self.c
        buf = self._b1.numpy().astype(np.float64)
        recon_sum = buf[32 * self.W]
        kl = -0.5 * np.sum(1 + c["s"] - c["mu"] ** 2 - np.exp(c["s"]), axis=1)
        loss = (recon_sum + kl.sum()) / n
        if mode == MODE_EVAL:
            return (loss, recon_sum / n, kl.sum() / n) if want_loss else None
        g = self.grads
        dg = buf[:32 * self.W].reshape(32, self.W)[:n]
        o = 2 * D + 4
        gs, hs = c["gs"], c["hs"]
        for i in range(D - 1, 0, -1):
            dpre = dg * np.where(gs[i] > 0, 1.0, gs[i] + 1.0)
            g[o + 2 * i] = gs[i - 1].T @ dpre
            g[o + 2 * i + 1] = dpre.sum(axis=0)
            dg = dpre @ w[o + 2 * i].T
        g[o] = c["z"].T @ dg
        g[o + 1] = dg.sum(axis=0)
        dz = dg @ w[o].T
        es = np.exp(c["s"])
        dmu = dz + c["mu"] / n
        ds = dz * self.eps * es + 0.5 * (es - 1.0) / n
        o = 2 * D
        g[o], g[o + 1] = hs[-1].T @ dmu, dmu.sum(axis=0)
        g[o + 2], g[o + 3] = hs[-1].T @ ds, ds.sum(axis=0)
        dh = dmu @ w[o].T + ds @ w[o + 2].T
        for i in range(D - 1, -1, -1):
            dpre = dh * np.where(hs[i] > 0, 1.0, hs[i] + 1.0)
            g[2 * i] = (hs[i - 1] if i > 0 else self.X).T @ dpre
            g[2 * i + 1] = dpre.sum(axis=0)
            if i > 0:
                dh = dpre @ w[2 * i].T
        if mode == MODE_TRAIN:
            self.opt.step(w, g)
        return (loss, recon_sum / n, kl.sum() / n) if want_loss else None


def make_problem(N=40, L=333, W=32, D=3, K=2, seed=7):
    g = np.random.default_rng(seed)
    f = np.clip(g.beta(0.5, 0.5, L), 0.02, 0.98)
    dc = g.binomial(2, f, size=(N, L)) * 0.5
    dc[0, :] = 0.5   # no monomorphic site (the Patterson scaler divides by zero there, in the reference too)
    w = O.init_weights(L, W, D, K, seed=seed, dtype=np.float64)
    for t in w:
        if t.ndim == 1:
            t += g.normal(0, 0.05, t.shape)
    return dc, [x.astype(np.float32).astype(np.float64) for x in w]


def _init_pg(rank, world, port, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def sharded_worker(rank, world, port, out_path, use_gpu, dims):
    """Runs the sharded protocol (ShardedEngine) on one rank and saves what the parent test compares."""
    sys.path.insert(0, ROOT)
    dist = _init_pg(rank, world, port)
    from popvae_amd.dist import ShardedEngine
    N, L, W, D, K = dims
    dc, w = make_problem(N, L, W, D, K)
    packed = O.pack2bit(dc)
    if use_gpu:
        from popvae_amd.engine import Engine
        factory = Engine
    else:
        factory = OracleShardEngine
    sh = ShardedEngine(n_snps=L, width=W, depth=D, latent_dim=K, batch_size=32, seed=3, device_id=0,
                       engine_factory=factory)
    sh.load_genotypes(packed)
    sh.set_weights([x.astype(np.float32) for x in w])
    g = np.random.default_rng(11)
    res = {"lo": sh.lo, "hi": sh.hi, "native": sh.native, "comm": sh.comm_report()}
    rows = g.choice(N, 32, replace=False).astype(np.int32)
    eps = g.normal(size=(32, K)).astype(np.float32)
    res["grad_loss"] = sh.grad_step(rows, eps)[0]
    res["grads"] = sh.get_grads_shard()
    losses = []
    for step in range(4):
        n = 32 if step != 2 else 9
        rows = g.choice(N, n, replace=False).astype(np.int32)
        eps = g.normal(size=(n, K)).astype(np.float32)
        losses.append(sh.train_step(rows, eps)[0])
    res["losses"] = losses
    rows = np.arange(N, dtype=np.int32)
    eps = g.normal(size=(N, K)).astype(np.float32)
    res["encode"] = sh.encode(rows, eps)
    res["eval"] = sh.eval_step(rows[:17], eps[:17])[0]
    res["weights"] = sh.get_weights_shard()
    res["full"] = sh.get_weights_full()   # collective: whole model on rank 0 (--save_weights under sharding)
    if dims[0] > 8:
        res["pca"] = sh.pca("Patterson", 6)[0]
    np.save(out_path + f".{rank}.npy", np.array([res], dtype=object), allow_pickle=True)
    sh.close()
    dist.barrier()
    dist.destroy_process_group()


def p2p_giveup_worker(rank, world, port, out_path):
    """One-shot exchange with a rank that does not show up for an exchange: rank 0 exchanges alone, its wait for rank 1's words
    runs out (PVAE_P2P_SPIN_LOG2 keeps that short), and check_exchange -- what ShardedEngine.train_epoch / encode call -- must
    raise on BOTH ranks (the ranks agree on the outcome)."""
    sys.path.insert(0, ROOT)
    dist = _init_pg(rank, world, port)
    from popvae_amd.dist import ShardedEngine
    from popvae_amd.engine import Engine
    sh = ShardedEngine(n_snps=640, width=128, depth=2, latent_dim=2, batch_size=32, seed=3, device_id=0, engine_factory=Engine)
    res = {"native": sh.native, "path": sh.comm_report()["path"], "raised": False, "clean_before": None}
    try:
        sh.check_exchange()
        res["clean_before"] = True
    except RuntimeError:
        res["clean_before"] = False
    # (the set-up's own test exchanges ran with the default, minute-long bound: two processes start their kernels milliseconds
    # apart.)  pvae_p2p_prepare re-reads PVAE_P2P_SPIN_LOG2 on every call; the inbox is already there.
    os.environ["PVAE_P2P_SPIN_LOG2"] = "10"
    sh.eng.p2p_prepare()
    if rank == 0:
        sh.eng.comm_allreduce(0)   # nobody else takes part
        res["nan"] = bool(np.isnan(sh.eng.exchange_read(0)).all())
    try:
        sh.check_exchange()
    except RuntimeError as ex:
        res["raised"] = "gave up" in str(ex)
    np.save(out_path + f".{rank}.npy", np.array([res], dtype=object), allow_pickle=True)
    sh.close()
    dist.barrier()
    dist.destroy_process_group()


def reference_run(dims):
    """The same sequence on the unsharded float64 oracle."""
    N, L, W, D, K = dims
    dc, w = make_problem(N, L, W, D, K)
    g = np.random.default_rng(11)
    out = {}
    rows = g.choice(N, 32, replace=False)
    eps = g.normal(size=(32, K)).astype(np.float32).astype(np.float64)
    c = O.forward(w, dc[rows], eps, D, K)
    out["grad_loss"] = c["loss"]
    out["grads"] = O.backward(w, c, D, K)
    opt = O.KerasAdam(w)
    losses = []
    for step in range(4):
        n = 32 if step != 2 else 9
        rows = g.choice(N, n, replace=False)
        eps = g.normal(size=(n, K)).astype(np.float32).astype(np.float64)
        losses.append(O.train_step(w, opt, dc[rows], eps, D, K)[0])
    out["losses"] = losses
    eps = g.normal(size=(N, K)).astype(np.float32).astype(np.float64)
    out["encode"] = O.encode(w, dc, eps, D, K)
    out["eval"] = O.forward(w, dc[:17], eps[:17], D, K)["loss"]
    out["weights"] = w
    dc2 = dc.copy()
    out["pca"] = O.pca(dc2, "Patterson", 6)[0] if N > 8 else None
    return out


def compare(res_by_rank, ref, dims, rtol, wtol):
    N, L, W, D, K = dims
    for r, res in enumerate(res_by_rank):
        lo, hi = res["lo"], res["hi"]
        assert abs(res["grad_loss"] - ref["grad_loss"]) <= rtol * abs(ref["grad_loss"])
        np.testing.assert_allclose(res["losses"], ref["losses"], rtol=rtol)
        assert abs(res["eval"] - ref["eval"]) <= rtol * abs(ref["eval"])
        if ref.get("pca") is not None:
            sgn = np.sign(np.sum(res["pca"] * ref["pca"], axis=0))
            assert np.abs(res["pca"] * sgn - ref["pca"]).max() <= 1e-8 * np.abs(ref["pca"]).max()
        for a, b in zip(res["encode"], ref["encode"]):
            assert np.abs(a - b).max() <= 10 * rtol * max(1.0, np.abs(b).max())

        def sl(i, t):  # this rank's slice of an unsharded tensor
            nt = len(ref["weights"])
            if i == 0:
                return t[lo:hi]
            if i == nt - 2:
                return t[:, lo:hi]
            if i == nt - 1:
                return t[lo:hi]
            return t
        for i, (a, b) in enumerate(zip(res["grads"], ref["grads"])):
            b = sl(i, b)
            assert a.shape == b.shape, (i, a.shape, b.shape)
            assert np.abs(a - b).max() <= rtol * max(np.abs(ref["grads"][i]).max(), 1e-30), ("grad", r, i)
        for i, (a, b) in enumerate(zip(res["weights"], ref["weights"])):
            assert np.abs(a - sl(i, b)).max() <= wtol, ("weight", r, i)
    # the gathered model (rank 0) is the shards side by side
    full = res_by_rank[0]["full"]
    assert all(r["full"] is None for r in res_by_rank[1:])
    assert [t.shape for t in full] == [np.asarray(t).shape for t in ref["weights"]]
    for r, res in enumerate(res_by_rank):
        lo, hi = res["lo"], res["hi"]
        assert np.array_equal(full[0][lo:hi], res["weights"][0]) and np.array_equal(full[-2][:, lo:hi], res["weights"][-2])
        assert np.array_equal(full[-1][lo:hi], res["weights"][-1])
    # replicated (narrow) tensors are bit-identical on all ranks
    nt = len(res_by_rank[0]["weights"])
    for i in range(1, nt - 2):
        for res in res_by_rank[1:]:
            assert np.array_equal(res["weights"][i], res_by_rank[0]["weights"][i]), ("replica drift", i)
