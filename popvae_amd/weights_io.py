"""File formats around the engine: Keras weight files and the ``.popvae.hdf5`` genotype cache.

* ``<out>_weights.hdf5`` -- what ``ModelCheckpoint`` leaves (popvae.py:429-435; ``save_weights_only`` is left at False,
  so Keras writes a full-model HDF5 whose weights sit in the ``model_weights`` group) and what ``vae.load_weights``
  reads (:508).  Layout (Keras 2.15 ``hdf5_format.save_weights_to_hdf5_group``): ``model_weights`` carries the attribute
  ``layer_names`` = the vae's layers in order (``input_1``, ``encoder``, ``decoder`` -- the nested models of
  popvae.py:400,410 are one "layer" each); every layer group carries ``weight_names`` = its variables in
  ``get_weights()`` order (``dense/kernel:0``, ``dense/bias:0``, ``dense_1/kernel:0``, ...; the reference names none of
  its Dense layers, so Keras numbers them in creation order: encoder hidden layers, z_mean, z_log_var, decoder layers,
  output layer), stored as datasets under that group.  Optimizer slots are not written (the reference never reads them).
* ``<out>_weights.npz`` -- the same tensors as arrays ``t000, t001, ...`` (get_weights() order) plus ``names``.
* ``<infile>.popvae.hdf5`` -- datasets ``derived_counts`` (float64 [N, L], values 0 / 0.5 / 1) and ``samples``
  (variable-length UTF-8 strings), popvae.py:205-209, 241-248.

HDF5 goes through popvae_amd.h5lite (plain Python; h5py is not needed).  Parity of the Keras layout is unpinned against
Keras itself (not installable here); the reader is pinned against files written by libhdf5 through h5py
(tests/golden/h5py_written_*.hdf5, made by tools/check_h5lite.py).
"""
from __future__ import annotations

import numpy as np

from . import h5lite


def keras_weight_names(depth):
    """(encoder variable names, decoder variable names) in get_weights() order for popvae.py:392-410."""
    def dn(i):
        return "dense" if i == 0 else f"dense_{i}"
    enc = [f"{dn(i)}/{v}:0" for i in range(depth + 2) for v in ("kernel", "bias")]
    dec = [f"{dn(depth + 2 + i)}/{v}:0" for i in range(depth + 1) for v in ("kernel", "bias")]
    return enc, dec


def save_weights_hdf5(path, weights, depth):
    """``weights``: list in ``vae.get_weights()`` order (``Engine.get_weights()``), float32."""
    enc, dec = keras_weight_names(depth)
    if len(weights) != len(enc) + len(dec):
        raise ValueError(f"depth {depth} has {len(enc) + len(dec)} tensors, got {len(weights)}")
    root = h5lite.Group(attrs={"keras_version": b"2.15.0", "backend": b"tensorflow"})
    mw = root.group("model_weights", attrs={"layer_names": np.array([b"input_1", b"encoder", b"decoder"]),
                                            "backend": b"tensorflow", "keras_version": b"2.15.0"})
    mw.group("input_1", attrs={"weight_names": np.array([], dtype="S1")})
    it = iter(weights)
    for model, names in (("encoder", enc), ("decoder", dec)):
        grp = mw.group(model, attrs={"weight_names": np.array([n.encode() for n in names])})
        for n in names:
            layer, var = n.split("/")
            grp.group(layer).dataset(var, np.ascontiguousarray(next(it), dtype=np.float32))
    h5lite.write(path, root)
    return path


def save_weights_npz(path, weights, depth):
    enc, dec = keras_weight_names(depth)
    np.savez(path, names=np.array(enc + dec), **{f"t{i:03d}": np.asarray(w, np.float32) for i, w in enumerate(weights)})
    return path


def _names(a):
    return [x.decode() if isinstance(x, bytes) else str(x) for x in np.asarray(a).reshape(-1)]


def load_weights_file(path):
    """``vae.load_weights(path)`` without Keras: the tensors of a Keras HDF5 weight / full-model file, or of an ``.npz``
    written by ``save_weights_npz``, as a list in ``get_weights()`` order -- ready for ``Engine.set_weights``."""
    if str(path).endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            keys = sorted(k for k in z.files if k.startswith("t") and k[1:].isdigit())
            return [np.array(z[k], dtype=np.float32) for k in keys]
    root = h5lite.read(path)
    top = root["model_weights"] if "model_weights" in root else root  # model.save(...) vs model.save_weights(...)
    if "layer_names" not in top.attrs:
        raise ValueError(f"{path}: no `layer_names` attribute -- not a Keras weight file")
    layers = _names(top.attrs["layer_names"])
    # The reference attaches its loss with vae.add_loss() on symbolic tensors built from z_mean / z_log_var (popvae.py:416-424);
    # Keras' functional add_loss may then list inner Dense / TFOpLambda / AddLoss layers in vae.layers besides the two nested
    # models, so a file the reference wrote can hold extra top-level groups that repeat encoder kernels (UNPINNED: no Keras
    # here to look).  When the nested models are there, they alone are read, in get_weights() order; otherwise every layer
    # once, by weight name.
    if "encoder" in layers and "decoder" in layers:
        layers = ["encoder", "decoder"]
    out, seen = [], set()
    for layer in layers:
        grp = top[layer]
        for wn in _names(grp.attrs.get("weight_names", [])):
            if wn in seen:
                continue
            seen.add(wn)
            out.append(np.array(grp[wn].data, dtype=np.float32))
    if len(out) < 10 or (len(out) - 6) % 4:  # 2 tensors per Dense: depth + 2 encoder layers, depth + 1 decoder layers
        raise ValueError(f"{path}: {len(out)} weight tensors; a popVAE model holds 4 * depth + 6 "
                         f"(layers read: {', '.join(layers)})")
    return out


def write_popvae_cache(path, dc, samples):
    """popvae.py:241-248: ``derived_counts`` = the float64 [N, L] matrix of 0 / 0.5 / 1, ``samples`` = vlen strings."""
    root = h5lite.Group()
    root.dataset("derived_counts", np.ascontiguousarray(dc, dtype=np.float64))
    root.dataset("samples", [str(s) for s in samples])
    h5lite.write(path, root)
    return path


def read_popvae_cache(path):
    """popvae.py:205-209 -> (dc float64 [N, L], samples object array of str)."""
    root = h5lite.read(path)
    dc = np.asarray(root["derived_counts"].data)
    s = root["samples"].data
    samples = np.array([x.decode() if isinstance(x, bytes) else str(x) for x in np.asarray(s).reshape(-1)], dtype=object)
    return dc, samples
