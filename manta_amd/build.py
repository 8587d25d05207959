"""Build the HIP library in-tree: hipcc --offload-arch=gfx950 -> manta_amd/libmnmf.so."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmnmf.so")
SOURCES = ["engine.hip"]
DEPS = ["engine.hip", "common.hpp", "dense.hpp", "spmm.hpp", "sparse_prep.hpp", "eig.hpp", "gram_mfma.hpp",
        os.path.join("..", "..", "include", "mnmf.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.exists(os.path.join(CSRC, d)) and os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [_hipcc(), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + SOURCES
    if verbose:
        print("[manta_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    build_lib(force=True)
