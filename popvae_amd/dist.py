"""SNP-axis sharding of one VAE over the GPUs of a node (one process per GPU, torch.distributed).

The reference has no multi-GPU path (popvae.py:82-89,187 only select a device); this is the layout
SURVEY.md section 8(e) prescribes: rank r owns a contiguous block of SNP columns -- the matching rows of
the first encoder kernel, columns of the last decoder kernel, output-bias entries, their Adam slots and
the same columns of the genotype matrix.  The narrow layers are replicated and computed redundantly.
Per train step there are exactly two small SUM all-reduces (default: enqueued by the engine itself on its stream through
its own RCCL communicator, ``pvae_comm_init`` -- no host code inside the step; ``PVAE_COMM=torch`` or a failed
set-up, agreed on by all ranks, selects the host-driven, host-staged path over ``torch.distributed`` instead):

  1. partial first-layer sums  X_r . W1_r           [32 x width]        (after stage A)
  2. partial decoder feedback  dg_D (and BCE sum)   [32 x width + 32]   (after stage B)

All ranks receive bit-identical reduced buffers, so the replicated narrow layers stay bit-identical
without any gradient all-reduce.  The fallback path works over any torch.distributed backend ("nccl" = RCCL; "gloo":
CPU tests, or several ranks sharing one GPU).
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
        # The host-driven exchange is host-staged for EVERY backend (see _allreduce): the engine never shares a stream,
        # an event or a device pointer with torch.distributed.
        self._staged = hasattr(self.eng, "exchange_read")          # test doubles hand out CPU tensors instead
        self._cuda_collectives = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        self.comm_note = "single rank"
        self.native = self._init_native_comm()
        self._bufs = None if self._staged else [self.eng.exchange_buffer(i)[0] for i in (0, 1)]
        try:  # test doubles (CPU stand-in engines) take no next_rows
            import inspect
            self._can_fuse = "next_rows" in inspect.signature(self.eng.stage_c).parameters
        except (TypeError, ValueError):
            self._can_fuse = False

    def _agree(self, ok):
        """MIN over the ranks of a 0/1 flag (on the host for gloo, on the device for RCCL)."""
        flag = self._torch.tensor([int(bool(ok))], dtype=self._torch.int32)
        if self._cuda_collectives:
            flag = flag.to(f"cuda:{self._torch.cuda.current_device()}")
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN, group=self.pg)
        return bool(int(flag.item()))

    def _init_native_comm(self):
        """Engine-owned RCCL communicator, set up by agreement at every step that can fail on one rank alone:

        1. the ranks compare device identities (RCCL wants one device per rank);
        2. every rank checks that librccl can be bound (``comm_available``) and the ranks agree -- BEFORE anyone enters
           the collective ncclCommInitRank, where a rank whose peers never arrive would wait forever;
        3. rank 0 draws the unique id, torch.distributed carries the 128 bytes, every rank joins (the engine checks
           ncclCommCount / ncclCommUserRank against its configuration);
        4. one all-reduce of (rank + 1) on exchange buffer 0 checks the ring, and the ranks agree on the outcome.

        Anything short of that leaves the host-driven path in place on ALL ranks (a communicator that some rank did build
        is destroyed again).  ``PVAE_COMM_INJECT_FAIL=<rank>|all`` makes step 2 fail on that rank (tests);
        ``PVAE_COMM_ASSUME_DISTINCT=1`` skips step 1 (tests on a one-GPU box, together with the injected failure)."""
        import os
        dist = self._dist
        if not (self.world > 1 or self.force_collectives) or not dist.is_initialized():
            return False
        self.comm_note = "host-staged all-reduces over torch.distributed"
        if os.environ.get("PVAE_COMM", "rccl").lower() == "torch" or not hasattr(self.eng, "comm_init"):
            self.comm_note += " (PVAE_COMM=torch)" if hasattr(self.eng, "comm_init") else ""
            return False
        if os.environ.get("PVAE_COMM", "rccl").lower() == "p2p" and hasattr(self.eng, "p2p_prepare"):
            if self._init_p2p():
                return True
            # (not available on every rank: carry on with the RCCL communicator, then the host-staged path)
        import socket
        try:
            uid = str(self._torch.cuda.get_device_properties(self.device_id).uuid)
        except Exception:  # older torch: no uuid; the visibility masks + ordinal still tell devices apart
            uid = None
        ident = (socket.gethostname(), uid, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"),
                 self.device_id)
        idents = [None] * self.world
        dist.all_gather_object(idents, ident, group=self.pg)
        if len(set(idents)) < self.world and not os.environ.get("PVAE_COMM_ASSUME_DISTINCT"):
            self.comm_note += " (ranks share a device: RCCL wants one per rank)"
            return False
        inject = os.environ.get("PVAE_COMM_INJECT_FAIL", "")
        try:
            ok = type(self.eng).comm_available() and inject not in ("all", str(self.rank))
        except Exception:
            ok = False
        if not self._agree(ok):
            self.comm_note += " (librccl not usable on some rank)"
            return False
        box = [None]
        if self.rank == 0:
            try:
                box[0] = type(self.eng).comm_unique_id()
            except Exception:
                box[0] = None
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        if box[0] is None:
            self.comm_note += " (ncclGetUniqueId failed)"
            return False
        ok = True
        try:
            self.eng.comm_init(box[0])
            n = self.eng.exchange_buffer(0)[1]
            self.eng.exchange_write(0, np.full(n, self.rank + 1, np.float32))
            self.eng.comm_allreduce(0)
            got = self.eng.exchange_read(0)
            ok = bool(np.all(got == np.float32(self.world * (self.world + 1) // 2)))
        except Exception:
            ok = False
        if not self._agree(ok):
            try:
                self.eng.comm_destroy()
            except Exception:
                pass
            self.comm_note += " (communicator self-test failed on some rank)"
            return False
        nr, me = self.eng.comm_info()
        self.comm_note = f"engine-owned RCCL communicator, rank {me} of {nr} (ncclCommCount)"
        return True

    def _init_p2p(self):
        """One-shot exchange among the ranks' processes (PVAE_COMM=p2p; include/pvae.h pvae_p2p_*), set up by agreement like the
        RCCL communicator: every rank allocates its inbox and hands out the hipIpc handle, torch.distributed carries the handles,
        every rank maps its peers' inboxes, one exchange of (rank + 1) checks the lot.  Any failure on any rank leaves the
        host-staged path in place on all of them.  Ranks may share a device (rehearsals on a one-GPU box)."""
        dist = self._dist
        handle = None
        try:
            handle = self.eng.p2p_prepare()
        except Exception:
            handle = None
        if not self._agree(handle is not None):
            self.comm_note += " (one-shot exchange: inbox / hipIpc handle failed on some rank)"
            return False
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=self.pg)
        ok = True
        try:
            self.eng.p2p_connect(handles)
            n = self.eng.exchange_buffer(0)[1]
            for rep in range(3):  # both parities of the inbox, and one reuse
                self.eng.exchange_write(0, np.full(n, self.rank + 1 + rep, np.float32))
                self.eng.comm_allreduce(0)
                got = self.eng.exchange_read(0)
                ok = ok and bool(np.all(got == np.float32(self.world * (self.world + 1) // 2 + rep * self.world)))
            ok = ok and not self.eng.p2p_gave_up()
        except Exception:
            ok = False
        if not self._agree(ok):
            try:
                self.eng.p2p_disconnect()
            except Exception:
                pass
            self.comm_note += " (one-shot exchange: self-test failed on some rank)"
            return False
        self.comm_note = f"one-shot exchange through hipIpc-mapped inboxes, rank {self.rank} of {self.world}"
        self._p2p = True
        return True

    def check_exchange(self):
        """One-shot exchange only: an exchange whose wait for a peer's words ran out fills its buffer with NaN and raises a
        status word on THAT rank (the late peer itself still receives valid sums).  Called once per epoch / encode; the ranks
        agree on the outcome, and all of them raise: a run must not carry on -- and exit 0 -- with poisoned replicas."""
        if not getattr(self, "_p2p", False):
            return
        try:
            bad = bool(self.eng.p2p_gave_up())
        except Exception:
            bad = True
        if not self._agree(not bad):
            raise RuntimeError("one-shot exchange (PVAE_COMM=p2p): a rank gave up waiting for a peer's words "
                               f"(this rank: {'gave up' if bad else 'ok'}); activations of that step were NaN-filled. "
                               "Rerun with PVAE_COMM=rccl, or raise PVAE_P2P_SPIN_LOG2 if a rank is known to stall on the host")

    def comm_report(self):
        """For bench.py's --gpus N line: which exchange path runs and what RCCL itself says about the communicator."""
        out = {"path": "native" if self.native else ("host-staged" if (self.world > 1 or self.force_collectives) else "none"),
               "note": self.comm_note, "comm_nranks": None}
        if self.native and getattr(self, "_p2p", False):
            out["path"], out["comm_nranks"] = "p2p", self.world
        elif self.native:
            out["comm_nranks"] = self.eng.comm_info()[0]
        return out

    def _allreduce(self, which):
        """Host-driven exchange (fallback path; the default is the engine-owned communicator).  Host-staged for every
        backend: pvae_exchange_copy brings the buffer to the host (synchronous on the engine's stream), the collective runs
        on a tensor that torch owns (CPU for gloo, a fresh device tensor for RCCL), the sum goes back the same way.  The
        engine therefore never shares a stream, an event or a device pointer with torch.distributed -- the variant that did
        (zero-copy views of the engine's buffers on a torch side stream) dead-locked in rehearsals, see DESIGN.md section 6,
        and was removed."""
        if not (self.world > 1 or self.force_collectives):
            return
        if self._trace:
            self._n_ar = getattr(self, "_n_ar", 0) + 1
            print(f"[dist r{self.rank}] all_reduce #{self._n_ar} buf {which} enter", flush=True)
        if not self._staged:  # test doubles: CPU tensors, reduced in place
            self._dist.all_reduce(self._bufs[which], op=self._dist.ReduceOp.SUM, group=self.pg)
        else:
            h = self._torch.from_numpy(self.eng.exchange_read(which))
            if self._cuda_collectives:
                d = h.to(f"cuda:{self.device_id}")
                self._dist.all_reduce(d, op=self._dist.ReduceOp.SUM, group=self.pg)
                h = d.cpu()
            else:
                self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self.pg)
            self.eng.exchange_write(which, h.numpy())
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

    def get_weights_full(self, group=None):
        """The whole model in ``vae.get_weights()`` order on rank 0 (None elsewhere): the SNP-axis tensors (first
        encoder kernel's rows, last decoder kernel's columns, output bias) are gathered from the shards, the replicated
        ones are rank 0's.  Collective over ``group`` (default: this engine's process group; pass a gloo group when the
        main one is RCCL -- object collectives pickle through host memory anyway)."""
        w = self.eng.get_weights()
        if self.world == 1:
            return w
        group = self.pg if group is None else group
        parts = [None] * self.world if self.rank == 0 else None
        dst = self._dist.get_global_rank(group, 0) if group is not None else 0
        self._dist.gather_object((w[0], w[-2], w[-1]), parts, dst=dst, group=group)
        if self.rank != 0:
            return None
        w[0] = np.concatenate([p[0] for p in parts], axis=0)
        w[-2] = np.concatenate([p[1] for p in parts], axis=1)
        w[-1] = np.concatenate([p[2] for p in parts], axis=0)
        return w

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
            out = self.eng.encode(rows, eps)
            self.check_exchange()
            return out
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
            if self._cuda_collectives:
                t = t.to(f"cuda:{self.device_id}")
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
            out = self.eng.train_epoch(perm, val_rows)
            self.check_exchange()
            return out
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
        if getattr(self, "_p2p", False) and self._dist.is_initialized():
            # a peer may still be inside an exchange kernel that writes into this rank's inbox: everyone drains, then meets,
            # before anyone unmaps or frees
            try:
                self.eng.synchronize()
                self._dist.barrier(group=self.pg)
            except Exception:
                pass
        self.eng.close()
