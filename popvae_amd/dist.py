"""SNP-axis sharding of one VAE over the GPUs of a node (one process per GPU, torch.distributed).

The reference has no multi-GPU path (popvae.py:82-89,187 only select a device); this is the layout
SURVEY.md section 8(e) prescribes: rank r owns a contiguous block of SNP columns -- the matching rows of
the first encoder kernel, columns of the last decoder kernel, output-bias entries, their Adam slots and
the same columns of the genotype matrix.  The narrow layers are replicated and computed redundantly.
Per train step there are exactly two small SUM all-reduces (default: enqueued by the engine itself on its stream through
its own RCCL communicator, ``pvae_comm_init`` -- no host code inside the step; ``PVAE_COMM=torch`` or a failed
self-test selects the host-driven staged path over ``torch.distributed`` instead):

  1. partial first-layer sums  X_r . W1_r           [32 x width]        (after stage A)
  2. partial decoder feedback  dg_D (and BCE sum)   [32 x width + 32]   (after stage B)

All ranks receive bit-identical reduced buffers, so the replicated narrow layers stay bit-identical
without any gradient all-reduce.  RCCL is reached through torch.distributed (backend "nccl"); "gloo"
works too (CPU tests, or several ranks sharing one GPU).
"""
from __future__ import annotations

import numpy as np

from .engine import MODE_ENCODE, MODE_EVAL, MODE_GRAD, MODE_TRAIN, Engine


def shard_bounds(n_snps, world):
    """SNP range [lo, hi) per rank; boundaries fall on multiples of 32 SNPs (= 8 packed bytes)."""
    nblk = (n_snps + 31) // 32
    per = (nblk + world - 1) // world
    out = []
    for r in range(world):
        lo = min(r * per * 32, n_snps)
        hi = min((r + 1) * per * 32, n_snps)
        out.append((lo, hi))
    return out


def shard_packed(packed, n_snps, lo, hi):
    """Columns [lo, hi) of a packed genotype matrix (lo is a multiple of 32)."""
    assert lo % 32 == 0
    nbytes = (hi - lo + 31) // 32 * 8
    out = np.zeros((packed.shape[0], nbytes), np.uint8)
    src = packed[:, lo // 4: lo // 4 + nbytes]
    out[:, :src.shape[1]] = src
    if (hi - lo) % 4:  # clear codes beyond hi inside the last byte
        keep = (1 << (2 * ((hi - lo) % 4))) - 1
        out[:, (hi - lo) // 4] &= keep
    out[:, (hi - lo + 3) // 4:] = 0
    return out


class _DevArray:
    """Zero-copy view of an engine-owned device buffer for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2}


class ShardedEngine:
    """Same host interface as ``Engine`` for a model whose SNP axis is split over ``dist`` ranks."""

    def __init__(self, n_snps, width=128, depth=6, latent_dim=2, batch_size=32, seed=0, device_id=0,
                 process_group=None, engine_factory=Engine, force_collectives=False, **adam):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.pg = process_group
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force_collectives = bool(force_collectives) and dist.is_initialized()  # rehearsal at world == 1
        self.n_snps_global = int(n_snps)
        self.bounds = shard_bounds(self.n_snps_global, self.world)
        self.lo, self.hi = self.bounds[self.rank]
        if self.hi <= self.lo:
            raise ValueError(f"rank {self.rank} would own no SNPs ({n_snps} SNPs over {self.world} ranks)")
        self.latent_dim, self.batch_size = int(latent_dim), int(batch_size)
        self.eng = engine_factory(n_snps=self.hi - self.lo, width=width, depth=depth, latent_dim=latent_dim,
                                  batch_size=batch_size, seed=seed, device_id=device_id,
                                  n_snps_global=self.n_snps_global, snp_offset=self.lo, rank=self.rank,
                                  world=self.world, **adam)
        import os
        self.device_id = int(device_id)
        self._trace = bool(os.environ.get("PVAE_DIST_TRACE"))  # debugging aid: progress line around every host-driven exchange
        # host-driven exchanges are bracketed by host syncs (see _allreduce); PVAE_DIST_STREAM_ORDERED=1 selects the
        # stream-ordered variant under RCCL (engine on torch's current stream, no host sync inside a step)
        self._host_sync = dist.is_initialized() and not (dist.get_backend(process_group) == "nccl"
                                                         and os.environ.get("PVAE_DIST_STREAM_ORDERED"))
        self._host_staged = (dist.is_initialized() and dist.get_backend(process_group) != "nccl"
                             and hasattr(self.eng, "exchange_read"))
        self.native = self._init_native_comm()
        self._bufs = [self._wrap(*self.eng.exchange_buffer(i), device_id) for i in (0, 1)]
        try:  # test doubles (CPU stand-in engines) take no next_rows
            import inspect
            self._can_fuse = "next_rows" in inspect.signature(self.eng.stage_c).parameters
        except (TypeError, ValueError):
            self._can_fuse = False
        # With its own communicator the engine keeps its own (non-blocking) stream and torch is not involved in a step at
        # all.  Otherwise the exchanges run on a dedicated non-blocking torch stream (which is also the engine's stream in
        # the stream-ordered variant) and never touch torch's legacy default stream, whose implicit synchronisation with
        # every blocking stream of the process the dead-lock described in _allreduce hinged on.
        self._xs = None
        if not self.native and hasattr(self.eng, "set_stream") and torch.cuda.is_available():
            self._xs = torch.cuda.Stream(device=device_id)
            if not self._host_sync:
                self.eng.set_stream(self._xs.cuda_stream)

    def _init_native_comm(self):
        """Engine-owned RCCL communicator: rank 0 draws the unique id, torch.distributed carries the 128 bytes, every rank
        joins, one all-reduce of (rank + 1) on exchange buffer 0 checks the ring, and the ranks agree (MIN over a flag) on
        using it.  Anything short of that leaves the host-driven path in place."""
        import os
        dist = self._dist
        if os.environ.get("PVAE_COMM", "rccl").lower() == "torch" or not hasattr(self.eng, "comm_init"):
            return False
        if not (self.world > 1 or self.force_collectives) or not dist.is_initialized():
            return False
        # RCCL wants one device per rank: two ranks on one GPU (tests, rehearsals) would fail inside ncclCommInitRank and leave
        # half-built communicator state behind (later collectives of the process were seen to hang), so the ranks compare
        # device identities first and do not try
        import socket
        try:
            uid = str(self._torch.cuda.get_device_properties(self.device_id).uuid)
        except Exception:  # older torch: no uuid; the visibility masks + ordinal still tell devices apart
            uid = None
        ident = (socket.gethostname(), uid, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"),
                 self.device_id)
        idents = [None] * self.world
        dist.all_gather_object(idents, ident, group=self.pg)
        if len(set(idents)) < self.world:
            return False
        box = [None]
        if self.rank == 0:
            try:
                box[0] = type(self.eng).comm_unique_id()
            except Exception:
                box[0] = None
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        if box[0] is None:
            return False
        ok = 1
        try:
            self.eng.comm_init(box[0])
            n = self.eng.exchange_buffer(0)[1]
            self.eng.exchange_write(0, np.full(n, self.rank + 1, np.float32))
            self.eng.comm_allreduce(0)
            got = self.eng.exchange_read(0)
            if not np.all(got == np.float32(self.world * (self.world + 1) // 2)):
                ok = 0
        except Exception:
            ok = 0
        flag = self._torch.tensor([ok], dtype=self._torch.int32)
        backend = dist.get_backend(self.pg)
        if backend == "nccl":
            flag = flag.to(f"cuda:{self._torch.cuda.current_device()}")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
        return bool(int(flag.item()))

    def _wrap(self, ptr, n, device_id):
        if isinstance(ptr, self._torch.Tensor):  # test doubles hand out CPU tensors directly
            return ptr
        return self._torch.as_tensor(_DevArray(ptr, n), device=f"cuda:{device_id}")

    def _allreduce(self, which):
        if self.world > 1 or self.force_collectives:
            if self._trace:
                self._n_ar = getattr(self, "_n_ar", 0) + 1
                print(f"[dist r{self.rank}] all_reduce #{self._n_ar} buf {which} enter", flush=True)
            # torch.distributed runs a collective on side streams tied to the current stream by events (gloo adds a worker
            # thread and host staging).  With engine kernels still queued ahead of the exchange on torch's legacy default
            # stream this was seen to dead-lock after a few steps (two ranks on one GPU over gloo), so by default the engine
            # keeps its own stream and every exchange is bracketed by host syncs: this is the fallback path, a few tens of
            # microseconds per exchange matter less than a hang.
            # Backends without device-pointer support of their own (gloo) get host buffers: pvae_exchange_copy stages the
            # exchange buffer through the host (synchronous on the engine's stream), the collective runs on CPU tensors.
            # gloo's own staging of CUDA tensors hung intermittently here even with the engine idle.
            if self._host_staged:
                h = self._torch.from_numpy(self.eng.exchange_read(which))
                self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self.pg)
                self.eng.exchange_write(which, h.numpy())
                if self._trace:
                    print(f"[dist r{self.rank}] all_reduce #{self._n_ar} done (host-staged)", flush=True)
                return
            if self._host_sync:
                self.eng.synchronize()
            if self._xs is not None:
                with self._torch.cuda.stream(self._xs):
                    self._dist.all_reduce(self._bufs[which], op=self._dist.ReduceOp.SUM, group=self.pg)
                if self._host_sync:
                    self._xs.synchronize()
            else:
                self._dist.all_reduce(self._bufs[which], op=self._dist.ReduceOp.SUM, group=self.pg)
            if self._trace:
                print(f"[dist r{self.rank}] all_reduce #{self._n_ar} done", flush=True)

    # -- data / weights ------------------------------------------------------------------------
    def load_genotypes(self, packed_full):
        """``packed_full`` is the whole matrix (every rank slices its own columns)."""
        self.eng.load_genotypes(shard_packed(np.asarray(packed_full), self.n_snps_global, self.lo, self.hi))

    def load_genotypes_shard(self, packed_shard):
        self.eng.load_genotypes(packed_shard)

    def init_weights(self, seed):
        self.eng.init_weights(seed)

    def set_weights(self, weights):
        """Full (unsharded) Keras-order tensors; the SNP-axis ones are sliced here."""
        w = list(weights)
        w[0] = np.ascontiguousarray(w[0][self.lo:self.hi])
        w[-2] = np.ascontiguousarray(w[-2][:, self.lo:self.hi])
        w[-1] = np.ascontiguousarray(w[-1][self.lo:self.hi])
        self.eng.set_weights(w)

    def get_weights_shard(self):
        return self.eng.get_weights()

    @property
    def lr(self):
        return self.eng.lr

    @lr.setter
    def lr(self, v):
        self.eng.lr = v

    def snapshot_best(self):
        self.eng.snapshot_best()

    def restore_best(self):
        self.eng.restore_best()

    # -- steps ---------------------------------------------------------------------------------
    def _step(self, rows, eps, mode, want_loss=True, next_rows=None):
        """``next_rows`` (train steps): the batch of the NEXT call, so that its first-layer forward rides on this
        step's wide encoder backward; the partial sums it leaves are all-reduced by the next call as usual."""
        if self.native:  # the engine runs the whole step, all-reduces included, on its own stream
            fn = {MODE_TRAIN: self.eng.train_step, MODE_EVAL: self.eng.eval_step}.get(mode)
            return self.eng.grad_step(rows, eps) if fn is None else fn(rows, eps, want_loss)
        if len(rows) > 32:
            raise ValueError(f"batch of {len(rows)} rows under SNP-axis sharding: the host-driven staged path moves one 32-row "
                             "tile per exchange; batches above 32 rows need the engine-owned communicator (PVAE_COMM=rccl)")
        self.eng.stage_a(rows, eps, mode)
        self._allreduce(0)
        self.eng.stage_b(mode)
        self._allreduce(1)
        if next_rows is not None and mode == MODE_TRAIN and self._can_fuse:
            return self.eng.stage_c(mode, want_loss, next_rows=next_rows)
        return self.eng.stage_c(mode, want_loss)

    def _eps(self, n, eps):
        """Replicated layers need the SAME noise on every rank.  With eps=None every engine draws from its
        counter-based generator keyed by (seed, call index): identical seeds + identical call sequences
        give identical draws on all ranks, with no communication."""
        return eps

    def train_step(self, rows, eps=None, want_loss=True, next_rows=None):
        return self._step(rows, self._eps(len(rows), eps), MODE_TRAIN, want_loss, next_rows)

    def eval_step(self, rows, eps=None, want_loss=True):
        return self._step(rows, self._eps(len(rows), eps), MODE_EVAL, want_loss)

    def grad_step(self, rows, eps=None):
        return self._step(rows, self._eps(len(rows), eps), MODE_GRAD, True)

    def get_grads_shard(self):
        return self.eng.get_grads()

    def encode(self, rows, eps=None):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        if self.native:
            return self.eng.encode(rows, eps)
        outs = []
        for i0 in range(0, len(rows), 32):
            r = rows[i0:i0 + 32]
            e = None if eps is None else eps[i0:i0 + 32]
            self.eng.stage_a(r, self._eps(len(r), e), MODE_ENCODE)
            self._allreduce(0)
            outs.append(self.eng.stage_encode(len(r)))
        return tuple(np.concatenate([o[k] for o in outs]) for k in range(3))

    def pca_gram(self, scaler="patterson"):
        """Global Gram matrix: the scaler is per site, so the ranks' local Gram matrices simply add up."""
        G = self.eng.pca_gram(scaler)
        if self.world > 1 or self.force_collectives:
            t = self._torch.from_numpy(G)
            dev = self._bufs[0].device if isinstance(self._bufs[0], self._torch.Tensor) else "cpu"
            if str(dev) != "cpu" and not self._host_staged:
                t = t.to(dev)
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.pg)
            G = t.cpu().numpy()
        return G

    def pca(self, scaler="Patterson", n_components=20):
        from .pca import coords_from_gram
        return coords_from_gram(self.pca_gram(scaler), n_components)

    def train_epoch(self, perm, val_rows, eps_fn=None):
        """One epoch of fit (K8/K9): batches of batch_size, short last batch kept; plain batch means."""
        B = self.batch_size
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        val_rows = np.ascontiguousarray(val_rows, dtype=np.int32)
        if self.native and eps_fn is None:  # the C-side epoch loop, as on one GPU
            return self.eng.train_epoch(perm, val_rows)
        tl, vl = [], []
        fuse = B <= 32
        for i0 in range(0, len(perm), B):
            r = perm[i0:i0 + B]
            nxt = perm[i0 + B:i0 + 2 * B] if i0 + B < len(perm) else val_rows[:B]
            tl.append(self.train_step(r, None if eps_fn is None else eps_fn(len(r)),
                                      next_rows=nxt if fuse and len(nxt) else None)[0])
        for i0 in range(0, len(val_rows), B):
            r = val_rows[i0:i0 + B]
            vl.append(self.eval_step(r, None if eps_fn is None else eps_fn(len(r)))[0])
        return float(np.mean(tl)), (float(np.mean(vl)) if vl else float("nan"))

    def synchronize(self):
        self.eng.synchronize()

    def close(self):
        self.eng.close()
