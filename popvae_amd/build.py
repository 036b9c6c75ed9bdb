"""In-tree build of the native libraries (no cmake, no JIT cache):

* ``libpvae_prep.so``  g++   host-only genotype front end (csrc/pvae_prep.cpp)
* ``libpvae.so``       hipcc --offload-arch=gfx950   HIP kernels + engine + C ABI (csrc/pvae_engine.hip)

Both land next to this file so that they travel to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(os.path.dirname(HERE), "include")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_prep(force=False):
    out = os.path.join(HERE, "libpvae_prep.so")
    srcs = [os.path.join(CSRC, "pvae_prep.cpp"), os.path.join(INC, "pvae_prep.h")]
    if force or _newer(out, srcs):
        _run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", "-o", out, srcs[0], "-lz"])
    return out


def build_engine(force=False, extra=()):
    out = os.path.join(HERE, "libpvae.so")
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".hpp"))]
    srcs.append(os.path.join(INC, "pvae.h"))
    main = os.path.join(CSRC, "pvae_engine.hip")
    if force or _newer(out, srcs):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
              "-Wall", "-Wno-unused-function", "-Wno-unused-value", *extra, "-o", out, main])
    return out


def build_all(force=False):
    return build_prep(force), build_engine(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
