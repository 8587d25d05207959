"""Build the HIP library in-tree: hipcc --offload-arch=gfx950 -> manta_amd/libmnmf.so."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmnmf.so")
OBJ = os.path.join(CSRC, "_obj")
# translation unit -> the files it includes
SOURCES = {
    "engine.hip": ["ctx.hpp", "common.hpp", "dense.hpp", "spmm.hpp", "sparse_prep.hpp", "eig.hpp", "gram_mfma.hpp", "topk.hpp", "histogram.hpp"],
    "tfidf.hip": ["ctx.hpp", "common.hpp", "histogram.hpp"],
}
HEADER = os.path.join(HERE, "..", "include", "mnmf.h")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _deps(src: str):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, d) for d in SOURCES[src]] + [HEADER]


def needs_build() -> bool:
    return any(_stale(LIB, _deps(src)) for src in SOURCES)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """Compile each translation unit that is older than its sources (objects under csrc/_obj) and link libmnmf.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, _deps(src)):
            cmd = [_hipcc()] + FLAGS + ["-c", "-o", obj, src]
            if verbose:
                print("[manta_amd] " + " ".join(cmd), flush=True)
            subprocess.run(cmd, cwd=CSRC, check=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[manta_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    import sys
    build_lib(force="--force" in sys.argv or not os.path.exists(LIB) or not needs_build())
