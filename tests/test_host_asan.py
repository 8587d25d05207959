"""the host-only helpers of the library (segment builders, k x k eigen-solvers) built with
-fsanitize=address on the host side and run on randomised inputs (tests/native/host_asan_driver.hip).  No GPU is used;
the driver is a small executable of its own, so the sanitizer runtime never enters the Python process."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "host_asan_driver.hip")
EXE = os.path.join(HERE, "native", "_build", "host_asan_driver")
DEPS = [SRC] + [os.path.join(HERE, "..", "manta_amd", "csrc", f) for f in ("sparse_prep.hpp", "eig.hpp", "common.hpp")]


def build_driver(force=False):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    if not force and os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(d) for d in DEPS):
        return EXE
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    cmd = [hipcc, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-Xarch_host", "-fsanitize=address",
           "-Xarch_host", "-fno-omit-frame-pointer", "-o", EXE, SRC]
    subprocess.run(cmd, check=True, cwd=HERE)
    return EXE


def test_host_helpers_under_address_sanitizer():
    exe = build_driver()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK host helpers under AddressSanitizer" in r.stdout
    assert "AddressSanitizer" not in r.stderr
