"""Build the HIP library in-tree: hipcc --offload-arch=gfx950 -> manta_amd/libmnmf.so."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmnmf.so")
OBJ = os.path.join(CSRC, "_obj")
_ENGINE_DEPS = ["ctx.hpp", "impl_base.hpp", "common.hpp", "dense.hpp", "spmm.hpp", "sparse_prep.hpp", "eig.hpp", "gram_mfma.hpp",
                "rk_mfma_f64.hpp", "gram_mfma_f64.hpp", "topk.hpp", "histogram.hpp"]
# object -> (source, extra flags, the files it includes).  Engine<float> and Engine<double> are separate objects so
# the four compiles run side by side.
UNITS = {
    "engine_f32.o": ("engine.hip", ["-DMNMF_ENGINE_DTYPE=32"], _ENGINE_DEPS),
    "engine_f64.o": ("engine.hip", ["-DMNMF_ENGINE_DTYPE=64"], _ENGINE_DEPS),
    "capi.o": ("capi.hip", [], ["ctx.hpp", "impl_base.hpp", "common.hpp", "dense.hpp", "eig.hpp", "sparse_prep.hpp"]),
    "tfidf.o": ("tfidf.hip", [], ["ctx.hpp", "common.hpp", "histogram.hpp"]),
}
HEADER = os.path.join(HERE, "..", "include", "mnmf.h")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC"] + os.environ.get("MNMF_EXTRA_FLAGS", "").split()


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


def _deps(obj: str):
    src, _, incs = UNITS[obj]
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, d) for d in incs] + [HEADER]


def needs_build() -> bool:
    return any(_stale(LIB, _deps(obj)) for obj in UNITS)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """Compile each translation unit that is older than its sources (objects under csrc/_obj) and link libmnmf.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    objs, jobs = [], []
    for name, (src, extra, _) in UNITS.items():
        obj = os.path.join(OBJ, name)
        objs.append(obj)
        if force or _stale(obj, _deps(name)):
            cmd = [_hipcc()] + FLAGS + extra + ["-c", "-o", obj, src]
            if verbose:
                print("[manta_amd] " + " ".join(cmd), flush=True)
            jobs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, job in jobs:
        if job.wait() != 0:
            raise subprocess.CalledProcessError(job.returncode, cmd)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[manta_amd] " + " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=CSRC, check=True)
    return LIB


if __name__ == "__main__":
    import sys
    build_lib(force="--force" in sys.argv or not os.path.exists(LIB) or not needs_build())
