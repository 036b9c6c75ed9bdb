"""popvae_amd.h5lite / weights_io on the CPU: the plain-Python HDF5 subset against files that libhdf5 wrote (fixtures made
by tools/check_h5lite.py under an interpreter with h5py 3.3 / HDF5 1.10.6, the way popvae.py:243-247 and Keras'
ModelCheckpoint write them), write -> read round trips, and the published lookup3 test vectors."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from popvae_amd import h5lite, weights_io

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CONDA = "/opt/conda/bin/python3.9"


def test_lookup3_known_answers():
    # driver5() of Bob Jenkins' lookup3.c
    assert h5lite.lookup3(b"", 0) == 0xDEADBEEF
    assert h5lite.lookup3(b"", 0xDEADBEEF) == 0xBD5B7DDE
    assert h5lite.lookup3(b"Four score and seven years ago", 0) == 0x17770551
    assert h5lite.lookup3(b"Four score and seven years ago", 1) == 0xCD628161


def test_reads_h5py_written_popvae_cache():
    exp = np.load(os.path.join(GOLD, "h5py_written_cache_expected.npz"))
    dc, samples = weights_io.read_popvae_cache(os.path.join(GOLD, "h5py_written_cache.popvae.hdf5"))
    assert dc.dtype == np.float64 and np.array_equal(dc, exp["dc"])
    assert list(samples) == [str(s) for s in exp["samples"]]


def test_reads_h5py_written_keras_weight_file():
    exp = np.load(os.path.join(GOLD, "h5py_written_keras_weights_expected.npz"))
    want = [exp[f"arr_{i}"] for i in range(len(exp.files))]
    got = weights_io.load_weights_file(os.path.join(GOLD, "h5py_written_keras_weights.hdf5"))
    assert len(got) == len(want) == 4 * 3 + 6
    for a, b in zip(got, want):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    # the order is vae.get_weights(): [W1, b1, ..., Wmu, bmu, Ws, bs, V1, c1, ..., Vout, cout]
    L, W, K, D = 37, 8, 2, 3
    shapes = [(L, W), (W,)] + [(W, W), (W,)] * (D - 1) + [(W, K), (K,)] * 2 + [(K, W), (W,)] + [(W, W), (W,)] * (D - 1) + [(W, L), (L,)]
    assert [a.shape for a in got] == shapes


@pytest.mark.parametrize("depth", [1, 6, 20])
def test_weight_file_round_trip(tmp_path, depth):
    g = np.random.default_rng(depth)
    L, W, K = 1003, 40, 3
    shapes = [(L, W), (W,)] + [(W, W), (W,)] * (depth - 1) + [(W, K), (K,)] * 2 + [(K, W), (W,)] + [(W, W), (W,)] * (depth - 1) + [(W, L), (L,)]
    w = [g.normal(size=s).astype(np.float32) for s in shapes]
    for ext, save in ((".hdf5", weights_io.save_weights_hdf5), (".npz", weights_io.save_weights_npz)):
        p = save(str(tmp_path / ("w" + ext)), w, depth)
        got = weights_io.load_weights_file(p)
        assert len(got) == len(w) and all(np.array_equal(a, b) for a, b in zip(got, w))
    root = h5lite.read(str(tmp_path / "w.hdf5"))
    assert [x.decode() for x in root["model_weights"].attrs["layer_names"]] == ["input_1", "encoder", "decoder"]
    enc, dec = weights_io.keras_weight_names(depth)
    assert [x.decode() for x in root["model_weights/encoder"].attrs["weight_names"]] == enc
    assert enc[0] == "dense/kernel:0" and enc[-1] == f"dense_{depth + 1}/bias:0" and dec[-1] == f"dense_{2 * depth + 2}/bias:0"
    assert np.array_equal(root["model_weights/decoder"][dec[-2]].data, w[-2])


def test_popvae_cache_round_trip(tmp_path):
    g = np.random.default_rng(0)
    dc = g.integers(0, 3, size=(11, 1234)) * 0.5
    samples = [f"ind{i}" for i in range(10)] + ["Zoë_Å"]
    p = weights_io.write_popvae_cache(str(tmp_path / "x.vcf.popvae.hdf5"), dc, samples)
    dc2, s2 = weights_io.read_popvae_cache(p)
    assert np.array_equal(dc, dc2) and list(s2) == samples


def test_unsupported_input_is_loud(tmp_path):
    p = tmp_path / "junk.hdf5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(h5lite.H5Unsupported):
        h5lite.read(str(p))
    w = [np.zeros((2, 2), np.float32)]
    with pytest.raises(ValueError):
        weights_io.save_weights_hdf5(str(tmp_path / "w.hdf5"), w, 3)


@pytest.mark.skipif(not os.path.exists(CONDA), reason="no interpreter with h5py in this container")
def test_cross_check_against_libhdf5():
    """Both directions against the real library (h5py 3.3 in the build container's conda interpreter)."""
    chk = subprocess.run([CONDA, "-c", "import h5py"], capture_output=True)
    if chk.returncode != 0:
        pytest.skip("conda interpreter has no h5py")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    p = subprocess.run([CONDA, "-W", "ignore", os.path.join(here, "tools", "check_h5lite.py")], capture_output=True, text=True,
                       env=env, timeout=120)
    assert p.returncode == 0 and "cross-check OK" in p.stdout, (p.stdout[-1000:], p.stderr[-2000:])


def test_weight_file_with_extra_top_level_layers(tmp_path):
    """A file the reference wrote may list more than [input, encoder, decoder] in `layer_names`: its loss is attached with
    vae.add_loss() on tensors built from z_mean / z_log_var (popvae.py:416-424), and Keras' functional add_loss can add the inner
    Dense layers and the op layers to vae.layers, whose groups then repeat encoder kernels (UNPINNED: no Keras here).  The
    reader takes the nested models alone; a file whose tensor count is not 4 * depth + 6 is refused with a clear message."""
    g = np.random.default_rng(3)
    depth, L, W, K = 2, 50, 8, 2
    shapes = [(L, W), (W,)] + [(W, W), (W,)] * (depth - 1) + [(W, K), (K,)] * 2 + [(K, W), (W,)] + [(W, W), (W,)] * (depth - 1) + [(W, L), (L,)]
    w = [g.normal(size=s).astype(np.float32) for s in shapes]
    p = weights_io.save_weights_hdf5(str(tmp_path / "w.hdf5"), w, depth)
    root = h5lite.read(p)
    mw = root["model_weights"]
    enc, _ = weights_io.keras_weight_names(depth)
    # the z_mean head once more as a top-level layer of its own, plus two weightless op layers
    extra = mw.group("dense_2", attrs={"weight_names": np.array([enc[-4].encode(), enc[-3].encode()])})
    extra.group("dense_2").dataset("kernel:0", w[2 * depth])
    extra["dense_2"].dataset("bias:0", w[2 * depth + 1])
    for name in ("tf.math.multiply", "add_loss"):
        mw.group(name, attrs={"weight_names": np.array([], dtype="S1")})
    mw.attrs["layer_names"] = np.array([b"input_1", b"encoder", b"dense_2", b"decoder", b"tf.math.multiply", b"add_loss"])
    p2 = str(tmp_path / "w2.hdf5")   # (the tree read from `p` holds views of that file: write elsewhere)
    h5lite.write(p2, root)
    got = weights_io.load_weights_file(p2)
    assert len(got) == len(w) and all(np.array_equal(a, b) for a, b in zip(got, w))
    # without the nested models: every layer once, duplicates by weight name dropped, and the count is checked
    mw.attrs["layer_names"] = np.array([b"input_1", b"encoder", b"dense_2"])
    p3 = str(tmp_path / "w3.hdf5")
    h5lite.write(p3, root)
    with pytest.raises(ValueError, match="4 \\* depth \\+ 6"):
        weights_io.load_weights_file(p3)
