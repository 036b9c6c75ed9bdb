#!/usr/bin/env python3
"""Cross-check of popvae_amd/h5lite.py against libhdf5, to be run under an interpreter that HAS h5py
(in the build container: /opt/conda/bin/python3.9 tools/check_h5lite.py [--fixtures]).

1. files written by h5lite (the .popvae.hdf5 cache layout and the Keras model_weights layout) are read back with h5py;
2. files written by h5py the way the reference / Keras write them (old-style groups, vlen strings, attrs) are read
   with h5lite;
3. --fixtures: (re)generates tests/golden/h5py_written_*.hdf5, the small libhdf5-written files the CPU suite reads
   with h5lite where h5py is absent.
"""
import importlib.util
import os
import sys
import tempfile

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("h5lite", os.path.join(ROOT, "popvae_amd", "h5lite.py"))
h5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(h5)
GOLD = os.path.join(ROOT, "tests", "golden")


def keras_like(path, depth=2, L=37, W=8, K=2, seed=0):
    """What keras.Model.save(<out>_weights.hdf5) leaves for the reference's nested vae (popvae.py:392-435): model_weights
    / {input_1, encoder, decoder} with layer_names / weight_names attrs; also returns the tensors in get_weights() order."""
    g = np.random.default_rng(seed)
    shapes = [(L, W), (W,)] + [(W, W), (W,)] * (depth - 1) + [(W, K), (K,), (W, K), (K,)]
    n_enc = len(shapes)
    shapes += [(K, W), (W,)] + [(W, W), (W,)] * (depth - 1) + [(W, L), (L,)]
    tensors = [g.normal(size=s).astype(np.float32) for s in shapes]
    names = []
    for i in range(len(shapes) // 2):
        ln = "dense" if i == 0 else f"dense_{i}"
        names += [f"{ln}/kernel:0", f"{ln}/bias:0"]
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = b"2.15.0"
        f.attrs["backend"] = b"tensorflow"
        mw = f.create_group("model_weights")
        mw.attrs["layer_names"] = np.array([b"input_1", b"encoder", b"decoder"])
        mw.attrs["backend"] = b"tensorflow"
        mw.attrs["keras_version"] = b"2.15.0"
        mw.create_group("input_1").attrs["weight_names"] = np.array([], dtype="S1")
        for model, lo, hi in (("encoder", 0, n_enc), ("decoder", n_enc, len(shapes))):
            grp = mw.create_group(model)
            grp.attrs["weight_names"] = np.array([n.encode() for n in names[lo:hi]])
            for n, t in zip(names[lo:hi], tensors[lo:hi]):
                grp.create_dataset(n, data=t)
        f.create_group("optimizer_weights")
    return tensors


def main():
    tmp = tempfile.mkdtemp()
    # -- 1. h5lite writes, h5py reads ------------------------------------------------------------
    g = np.random.default_rng(1)
    dc = g.integers(0, 3, size=(7, 53)) * 0.5
    samples = [f"sample_{i}_é" for i in range(7)]
    root = h5.Group()
    root.dataset("derived_counts", dc)
    root.dataset("samples", samples)
    p1 = os.path.join(tmp, "cache.popvae.hdf5")
    h5.write(p1, root)
    with h5py.File(p1, "r") as f:  # exactly what popvae.py:205-209 does
        assert np.array_equal(np.array(f["derived_counts"]), dc) and f["derived_counts"].dtype == np.float64
        got = [s.decode() if isinstance(s, bytes) else s for s in np.array(f["samples"])]
        assert got == samples, got
        assert h5py.check_string_dtype(f["samples"].dtype) is not None
    root = h5.Group(attrs={"keras_version": b"2.15.0", "backend": b"tensorflow"})
    mw = root.group("model_weights", attrs={"layer_names": np.array([b"input_1", b"encoder"]), "backend": b"tensorflow"})
    mw.group("input_1", attrs={"weight_names": np.array([], dtype="S1")})
    enc = mw.group("encoder", attrs={"weight_names": np.array([b"dense/kernel:0", b"dense/bias:0"])})
    k, b = g.normal(size=(300, 40)).astype(np.float32), g.normal(size=40).astype(np.float32)
    d = enc.group("dense")
    d.dataset("kernel:0", k)
    d.dataset("bias:0", b)
    for i in range(40):  # more links than the default compact limit of 8
        mw.group(f"extra_{i}")
    p2 = os.path.join(tmp, "weights.hdf5")
    h5.write(p2, root)
    with h5py.File(p2, "r") as f:
        assert f.attrs["backend"] == b"tensorflow"
        assert list(f["model_weights"].attrs["layer_names"]) == [b"input_1", b"encoder"]
        assert list(f["model_weights/encoder"].attrs["weight_names"]) == [b"dense/kernel:0", b"dense/bias:0"]
        assert len(f["model_weights/input_1"].attrs["weight_names"]) == 0
        assert np.array_equal(f["model_weights/encoder/dense/kernel:0"][...], k)
        assert np.array_equal(f["model_weights/encoder/dense/bias:0"][...], b)
        assert len(f["model_weights"]) == 42
    # -- 2. h5py writes, h5lite reads ------------------------------------------------------------
    p3 = os.path.join(tmp, "ref_cache.popvae.hdf5")
    with h5py.File(p3, "w") as f:  # popvae.py:243-247 verbatim
        f.create_dataset("derived_counts", data=dc)
        f.create_dataset("samples", data=samples, dtype=h5py.string_dtype())
    r = h5.read(p3)
    assert np.array_equal(r["derived_counts"].data, dc)
    assert list(r["samples"].data) == samples
    p4 = os.path.join(tmp, "ref_weights.hdf5")
    tensors = keras_like(p4, depth=12)   # 50 datasets in one group: B-tree with several symbol nodes
    r = h5.read(p4)
    names = [n.decode() for n in r["model_weights/encoder"].attrs["weight_names"]]
    assert names[0] == "dense/kernel:0" and len(names) == 2 * 12 + 4
    flat = []
    for model in ("encoder", "decoder"):
        for n in r["model_weights"][model].attrs["weight_names"]:
            flat.append(r["model_weights"][model][n.decode()].data)
    assert len(flat) == len(tensors) and all(np.array_equal(a, b) for a, b in zip(flat, tensors))
    assert r.attrs["keras_version"] in ("2.15.0", b"2.15.0")  # h5py stores a bytes scalar as a variable-length string
    # new-style groups as libhdf5 writes them (libver latest), still compact
    p5 = os.path.join(tmp, "latest.hdf5")
    with h5py.File(p5, "w", libver="latest") as f:
        f.create_group("a").create_dataset("x", data=np.arange(5, dtype=np.int32))
        f["a"].attrs["note"] = "héllo"
        f.create_dataset("s", data=np.array([b"ab", b"cde"]))
        f.create_dataset("tiny", data=np.float64(3.5))
    r = h5.read(p5)
    assert np.array_equal(r["a/x"].data, np.arange(5)) and r["a"].attrs["note"] == "héllo"
    assert list(r["s"].data) == [b"ab", b"cde"] and float(r["tiny"].data) == 3.5
    print("h5lite <-> libhdf5", h5py.version.hdf5_version, "cross-check OK")
    if "--fixtures" in sys.argv:
        t = keras_like(os.path.join(GOLD, "h5py_written_keras_weights.hdf5"), depth=3, L=37, W=8, K=2, seed=5)
        np.savez(os.path.join(GOLD, "h5py_written_keras_weights_expected.npz"), *t)
        with h5py.File(os.path.join(GOLD, "h5py_written_cache.popvae.hdf5"), "w") as f:
            f.create_dataset("derived_counts", data=dc)
            f.create_dataset("samples", data=samples, dtype=h5py.string_dtype())
        np.savez(os.path.join(GOLD, "h5py_written_cache_expected.npz"), dc=dc, samples=np.array(samples))
        print("fixtures written to", GOLD)


if __name__ == "__main__":
    main()
