"""ctypes binding of libpvae.so (include/pvae.h) and the ``Engine`` host object.

``Engine`` mirrors what /root/reference/popvae.py does with its Keras objects: build
(popvae.py:392-426), ``fit`` steps (:490), ``predict`` (:455,509), weight get/set and the best-weights
checkpoint (:429-435,508).  There is no CPU path: a missing library or GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None

ENGINE_SYMBOLS = [
    "pvae_version", "pvae_device_count", "pvae_create", "pvae_destroy", "pvae_last_error", "pvae_set_stream",
    "pvae_synchronize", "pvae_load_genotypes", "pvae_num_tensors", "pvae_tensor_shape", "pvae_set_tensor",
    "pvae_get_tensor", "pvae_init_weights", "pvae_get_moment", "pvae_reset_optimizer", "pvae_set_lr", "pvae_get_lr", "pvae_set_bce_mode",
    "pvae_train_step", "pvae_train_step_noalloc", "pvae_eval_step", "pvae_grad_step", "pvae_get_grad", "pvae_train_epoch", "pvae_encode",
    "pvae_decode", "pvae_pca_gram", "pvae_snapshot_best", "pvae_restore_best", "pvae_exchange_buffer", "pvae_exchange_copy", "pvae_stage_a", "pvae_stage_b",
    "pvae_stage_c", "pvae_stage_c_next", "pvae_stage_encode", "pvae_set_timing", "pvae_get_timing", "pvae_set_timing_only",
    "pvae_comm_available", "pvae_comm_info", "pvae_comm_unique_id", "pvae_comm_init", "pvae_comm_allreduce", "pvae_comm_destroy",
    "pvae_p2p_prepare", "pvae_p2p_connect", "pvae_p2p_status", "pvae_p2p_disconnect",
]


class Config(C.Structure):
    _fields_ = [
        ("n_snps_local", C.c_int64), ("n_snps_global", C.c_int64), ("snp_offset", C.c_int64),
        ("width", C.c_int32), ("depth", C.c_int32), ("latent_dim", C.c_int32), ("batch_size", C.c_int32),
        ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
        ("device_id", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("seed", C.c_uint64),
    ]


_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_fopt = C.POINTER(C.c_float)


def lib():
    """Load libpvae.so (built in-tree by popvae_amd.build); raises if absent."""
    global _lib
    if _lib is None:
        path = os.environ.get("PVAE_LIB") or os.path.join(_HERE, "libpvae.so")  # PVAE_LIB: kernel-variant experiments
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m popvae_amd.build` (hipcc --offload-arch=gfx950)")
        L = C.CDLL(path)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        L.pvae_version.restype = C.c_char_p
        L.pvae_last_error.restype = C.c_char_p
        L.pvae_last_error.argtypes = [vp]
        L.pvae_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.pvae_destroy.argtypes = [vp]
        L.pvae_set_stream.argtypes = [vp, vp, C.c_int]
        L.pvae_synchronize.argtypes = [vp]
        L.pvae_load_genotypes.argtypes = [vp, _u8p, i64, i64]
        L.pvae_num_tensors.argtypes = [vp]
        L.pvae_tensor_shape.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i64)]
        L.pvae_set_tensor.argtypes = [vp, C.c_int, _f32p, i64]
        L.pvae_get_tensor.argtypes = [vp, C.c_int, _f32p, i64]
        L.pvae_init_weights.argtypes = [vp, C.c_uint64]
        L.pvae_get_moment.argtypes = [vp, C.c_int, C.c_int, _f32p, i64]
        L.pvae_reset_optimizer.argtypes = [vp]
        L.pvae_set_lr.argtypes = [vp, C.c_float]
        L.pvae_set_bce_mode.argtypes = [vp, C.c_int]
        L.pvae_get_lr.argtypes = [vp, C.POINTER(C.c_float)]
        for f in (L.pvae_train_step, L.pvae_train_step_noalloc, L.pvae_eval_step, L.pvae_grad_step):
            f.argtypes = [vp, _i32p, i32, _fopt, _fopt]
        L.pvae_get_grad.argtypes = [vp, C.c_int, _f32p, i64]
        L.pvae_train_epoch.argtypes = [vp, _i32p, i32, _i32p, i32, _fopt, _fopt]
        L.pvae_encode.argtypes = [vp, _i32p, i32, _fopt, _fopt, _fopt, _fopt]
        L.pvae_decode.argtypes = [vp, _f32p, i32, _f32p]
        _dopt = C.POINTER(C.c_double)
        L.pvae_pca_gram.argtypes = [vp, C.c_int, _dopt, _dopt, _dopt]
        L.pvae_snapshot_best.argtypes = [vp]
        L.pvae_restore_best.argtypes = [vp]
        L.pvae_exchange_buffer.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(i64)]
        L.pvae_exchange_copy.argtypes = [vp, C.c_int, _f32p, i64, C.c_int]
        L.pvae_stage_a.argtypes = [vp, _i32p, i32, _fopt, C.c_int]
        L.pvae_stage_b.argtypes = [vp, C.c_int]
        L.pvae_stage_c.argtypes = [vp, C.c_int, _fopt]
        L.pvae_stage_c_next.argtypes = [vp, C.c_int, _fopt, C.POINTER(C.c_int32), i32]
        L.pvae_stage_encode.argtypes = [vp, i32, _fopt, _fopt, _fopt]
        L.pvae_set_timing.argtypes = [vp, C.c_int]
        L.pvae_set_timing_only.argtypes = [vp, C.c_char_p]
        L.pvae_comm_unique_id.argtypes = [C.c_char_p]
        L.pvae_comm_init.argtypes = [vp, C.c_char_p]
        L.pvae_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.pvae_comm_allreduce.argtypes = [vp, C.c_int]
        L.pvae_comm_destroy.argtypes = [vp]
        L.pvae_p2p_prepare.argtypes = [vp, C.c_char_p]
        L.pvae_p2p_connect.argtypes = [vp, C.c_char_p]
        L.pvae_p2p_status.argtypes = [vp, C.POINTER(C.c_int32)]
        L.pvae_p2p_disconnect.argtypes = [vp]
        L.pvae_get_timing.argtypes = [vp, C.c_char_p, _fopt, C.POINTER(i64)]
        _lib = L
    return _lib


def _fptr(a):
    return None if a is None else a.ctypes.data_as(_fopt)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


MODE_TRAIN, MODE_EVAL, MODE_GRAD, MODE_ENCODE = 0, 1, 2, 3


class Engine:
    """One VAE (encoder + decoder + Adam state) resident on one MI355X."""

    def __init__(self, n_snps, width=128, depth=6, latent_dim=2, batch_size=32, lr=1e-3, beta1=0.9, beta2=0.999,
                 eps=1e-7, device_id=0, seed=0, n_snps_global=None, snp_offset=0, rank=0, world=1):
        self._h = None
        L = lib()
        cfg = Config(int(n_snps), int(n_snps_global if n_snps_global is not None else n_snps), int(snp_offset),
                     int(width), int(depth), int(latent_dim), int(batch_size), lr, beta1, beta2, eps,
                     int(device_id), int(rank), int(world), int(seed) & (2**64 - 1))
        h = C.c_void_p()
        rc = L.pvae_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"pvae_create failed ({rc}): {L.pvae_last_error(None).decode()}")
        self._h = h
        self.n_snps, self.width, self.depth, self.latent_dim, self.batch_size = \
            int(n_snps), int(width), int(depth), int(latent_dim), int(batch_size)
        self.world, self.rank = int(world), int(rank)
        self._out3 = np.zeros(3, np.float32)

    # -- plumbing --------------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(f"libpvae ({rc}): {lib().pvae_last_error(self._h).decode()}")

    def close(self):
        if self._h is not None:
            lib().pvae_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr, own=False):
        """Run the engine on ``stream_ptr`` (a hipStream_t; 0 / None = the legacy default stream), or on its own
        stream again with ``own=True``."""
        self._ck(lib().pvae_set_stream(self._h, C.c_void_p(stream_ptr or 0), int(bool(own))))

    def synchronize(self):
        self._ck(lib().pvae_synchronize(self._h))

    # -- data / weights --------------------------------------------------------------------------
    def load_genotypes(self, packed):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        self.n_samples = packed.shape[0]
        self._ck(lib().pvae_load_genotypes(self._h, packed, packed.shape[0], packed.shape[1]))

    def tensor_shapes(self):
        out = []
        r, c = C.c_int64(), C.c_int64()
        for i in range(lib().pvae_num_tensors(self._h)):
            self._ck(lib().pvae_tensor_shape(self._h, i, C.byref(r), C.byref(c)))
            out.append((r.value, c.value) if c.value else (r.value,))
        return out

    def set_weights(self, weights):
        """vae.set_weights(list in Keras order)."""
        shapes = self.tensor_shapes()
        if len(weights) != len(shapes):
            raise ValueError(f"expected {len(shapes)} tensors, got {len(weights)}")
        for i, (w, s) in enumerate(zip(weights, shapes)):
            w = _f32(w)
            if tuple(w.shape) != tuple(s):
                raise ValueError(f"tensor {i}: expected shape {s}, got {w.shape}")
            self._ck(lib().pvae_set_tensor(self._h, i, w.reshape(-1), w.size))

    def _get_all(self, fn):
        out = []
        for i, s in enumerate(self.tensor_shapes()):
            a = np.empty(int(np.prod(s)), np.float32)
            self._ck(fn(i, a, a.size))
            out.append(a.reshape(s))
        return out

    def get_weights(self):
        return self._get_all(lambda i, a, n: lib().pvae_get_tensor(self._h, i, a, n))

    def get_grads(self):
        return self._get_all(lambda i, a, n: lib().pvae_get_grad(self._h, i, a, n))

    def get_moments(self, which):
        return self._get_all(lambda i, a, n: lib().pvae_get_moment(self._h, which, i, a, n))

    def init_weights(self, seed):
        self._ck(lib().pvae_init_weights(self._h, int(seed) & (2**64 - 1)))

    def reset_optimizer(self):
        self._ck(lib().pvae_reset_optimizer(self._h))

    @property
    def lr(self):
        v = C.c_float()
        self._ck(lib().pvae_get_lr(self._h, C.byref(v)))
        return np.float32(v.value)

    @lr.setter
    def lr(self, v):
        self._ck(lib().pvae_set_lr(self._h, float(v)))

    # -- steps -----------------------------------------------------------------------------------
    def set_bce_mode(self, mode):
        """'logits' (default) or 'probs': which form of Keras' binary cross-entropy the decoder pass computes (K5)."""
        self._ck(lib().pvae_set_bce_mode(self._h, {"logits": 0, "probs": 1}[mode]))

    def _step(self, fn, rows, eps, want_loss):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        eps = _f32(eps)
        if eps is not None and eps.shape != (len(rows), self.latent_dim):
            raise ValueError("eps must be [n, latent_dim]")
        self._ck(fn(self._h, rows, len(rows), _fptr(eps), _fptr(self._out3) if want_loss else None))
        return tuple(float(x) for x in self._out3) if want_loss else None

    def train_step(self, rows, eps=None, want_loss=True):
        return self._step(lib().pvae_train_step, rows, eps, want_loss)

    def train_step_noalloc(self, rows, eps=None, want_loss=True):
        """``train_step`` for callers that cannot allocate: partial sums in the library's static device array."""
        return self._step(lib().pvae_train_step_noalloc, rows, eps, want_loss)

    def eval_step(self, rows, eps=None, want_loss=True):
        return self._step(lib().pvae_eval_step, rows, eps, want_loss)

    def grad_step(self, rows, eps=None):
        return self._step(lib().pvae_grad_step, rows, eps, True)

    def train_epoch(self, perm, val_rows):
        perm = np.ascontiguousarray(perm, dtype=np.int32)
        val_rows = np.ascontiguousarray(val_rows, dtype=np.int32)
        a, b = C.c_float(), C.c_float()
        self._ck(lib().pvae_train_epoch(self._h, perm, len(perm), val_rows, len(val_rows), C.byref(a), C.byref(b)))
        return a.value, b.value

    def encode(self, rows, eps=None):
        """encoder.predict -> (z_mean, z_log_var, z), each [n, latent_dim]."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        n, K = len(rows), self.latent_dim
        eps = _f32(eps)
        mu, lv, z = (np.zeros((n, K), np.float32) for _ in range(3))
        self._ck(lib().pvae_encode(self._h, rows, n, _fptr(eps), _fptr(mu), _fptr(lv), _fptr(z)))
        return mu, lv, z

    def decode(self, z):
        """decoder.predict (scripts/popvae_decoder_tests_HGDP.py:156): z [n, latent_dim] -> probabilities [n, n_snps]."""
        z = np.ascontiguousarray(z, dtype=np.float32)
        if z.ndim != 2 or z.shape[1] != self.latent_dim:
            raise ValueError("z must be [n, latent_dim]")
        out = np.empty((z.shape[0], self.n_snps), np.float32)
        self._ck(lib().pvae_decode(self._h, z.reshape(-1), z.shape[0], out.reshape(-1)))
        return out

    def pca_gram(self, scaler="patterson", want_sites=False):
        """Gram matrix (float64 [N, N]) of the per-site scaled genotype matrix of ALL loaded samples, for
        popvae_amd.pca (popvae.py:528-533).  scaler: "patterson" | "standard" | anything else = centre only
        (allel.stats.decomposition.get_scaler).  want_sites: also return the per-site (mean, std)."""
        code = {"patterson": 1, "standard": 2}.get(scaler.lower() if isinstance(scaler, str) else None, 0)
        n = getattr(self, "n_samples", 0)
        if not n:
            raise RuntimeError("no genotypes loaded")
        G = np.empty((n, n), np.float64)
        dp = C.POINTER(C.c_double)
        mean = std = None
        if want_sites:
            mean, std = np.empty(self.n_snps, np.float64), np.empty(self.n_snps, np.float64)
        self._ck(lib().pvae_pca_gram(self._h, code, G.ctypes.data_as(dp), mean.ctypes.data_as(dp) if want_sites else None,
                                     std.ctypes.data_as(dp) if want_sites else None))
        return (G, mean, std) if want_sites else G

    def snapshot_best(self):
        self._ck(lib().pvae_snapshot_best(self._h))

    def restore_best(self):
        self._ck(lib().pvae_restore_best(self._h))

    # -- staged (SNP-sharded) step ---------------------------------------------------------------
    def exchange_buffer(self, which):
        p, n = C.c_void_p(), C.c_int64()
        self._ck(lib().pvae_exchange_buffer(self._h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    def exchange_read(self, which):
        """Host copy of exchange buffer ``which`` (transports without device-pointer support, tests)."""
        a = np.empty(self.exchange_buffer(which)[1], np.float32)
        self._ck(lib().pvae_exchange_copy(self._h, which, a, a.size, 0))
        return a

    def exchange_write(self, which, a):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        self._ck(lib().pvae_exchange_copy(self._h, which, a, a.size, 1))

    def stage_a(self, rows, eps, mode):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        eps = _f32(eps)
        self._ck(lib().pvae_stage_a(self._h, rows, len(rows), _fptr(eps), mode))

    def stage_b(self, mode):
        self._ck(lib().pvae_stage_b(self._h, mode))

    def stage_c(self, mode, want_loss=True, next_rows=None):
        """``next_rows``: the next batch, if known (train steps): its first-layer forward then rides on this step's wide
        encoder backward (pvae_stage_c_next)."""
        out = _fptr(self._out3) if want_loss else None
        if next_rows is None:
            self._ck(lib().pvae_stage_c(self._h, mode, out))
        else:
            nr = np.ascontiguousarray(next_rows, dtype=np.int32)
            self._ck(lib().pvae_stage_c_next(self._h, mode, out, nr.ctypes.data_as(C.POINTER(C.c_int32)), len(nr)))
        return tuple(float(x) for x in self._out3) if want_loss else None

    def stage_encode(self, n):
        K = self.latent_dim
        mu, lv, z = (np.zeros((n, K), np.float32) for _ in range(3))
        self._ck(lib().pvae_stage_encode(self._h, n, _fptr(mu), _fptr(lv), _fptr(z)))
        return mu, lv, z

    # -- engine-owned RCCL communicator (SNP-axis sharding with no host code inside the step) ------
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId on this rank: 128 bytes for the host to hand to every rank's ``comm_init``."""
        buf = C.create_string_buffer(128)
        rc = lib().pvae_comm_unique_id(buf)
        if rc != 0:
            raise RuntimeError("pvae_comm_unique_id failed: librccl not loadable?")
        return buf.raw

    def comm_init(self, id128):
        if len(id128) != 128:
            raise ValueError("a RCCL unique id is 128 bytes")
        self._ck(lib().pvae_comm_init(self._h, id128))

    @staticmethod
    def comm_available():
        """librccl bound?  Every rank checks this (and the ranks agree) BEFORE the collective ``comm_init``."""
        return bool(lib().pvae_comm_available())

    def comm_info(self):
        """(ranks, rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int32(), C.c_int32()
        self._ck(lib().pvae_comm_info(self._h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def comm_destroy(self):
        self._ck(lib().pvae_comm_destroy(self._h))

    def comm_allreduce(self, which):
        self._ck(lib().pvae_comm_allreduce(self._h, which))

    # one-shot exchange among the ranks' processes (opt-in alternative to the RCCL all-reduces, include/pvae.h)
    def p2p_prepare(self):
        """Allocate this rank's inbox; returns its 64-byte hipIpc handle (to be carried to every rank by the host)."""
        buf = C.create_string_buffer(64)
        self._ck(lib().pvae_p2p_prepare(self._h, buf))
        return buf.raw

    def p2p_connect(self, handles):
        """handles: the ranks' 64-byte handles in rank order; maps the peers' inboxes and switches the exchanges over."""
        blob = b"".join(handles)
        self._ck(lib().pvae_p2p_connect(self._h, blob))

    def p2p_gave_up(self):
        st = C.c_int32(0)
        self._ck(lib().pvae_p2p_status(self._h, C.byref(st)))
        return bool(st.value)

    def p2p_disconnect(self):
        self._ck(lib().pvae_p2p_disconnect(self._h))

    # -- measurement -----------------------------------------------------------------------------
    def set_timing(self, on, only=None):
        """HIP-event timing of the kernels by category; ``only="dec"`` restricts the events to one category (each
        event pair costs the stream ~5-10 us per launch)."""
        if on and only:
            self._ck(lib().pvae_set_timing_only(self._h, only.encode()))
        else:
            self._ck(lib().pvae_set_timing(self._h, int(bool(on))))

    def get_timing(self, name):
        ms, n = C.c_float(), C.c_int64()
        self._ck(lib().pvae_get_timing(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
