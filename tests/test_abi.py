"""CPU-side checks of the C ABI: libpvae.so loads without a GPU and exports every symbol that
include/pvae.h declares (no compute calls here)."""
import os
import re

from popvae_amd import engine


def test_library_loads_and_exports_header_symbols():
    L = engine.lib()
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "pvae.h")).read()
    declared = set(re.findall(r"\b(pvae_\w+)\s*\(", hdr))
    assert declared == set(engine.ENGINE_SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s
    assert b"gfx950" in L.pvae_version()


def test_no_cpu_fallback():
    """Without a GPU the engine must refuse to build (the product never computes on the host)."""
    L = engine.lib()
    if L.pvae_device_count() > 0:
        return
    import pytest
    with pytest.raises(RuntimeError):
        engine.Engine(n_snps=64, width=32, depth=2)


def test_product_does_not_import_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "popvae_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
