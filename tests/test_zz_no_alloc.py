"""mnmf_gram_static -- the entry point that allocates nothing (include/mnmf.h) -- against mnmf_gram through the C-ABI:
two calls in a row on one context, the second with a smaller input (the static scratch keeps the first call's
partials), then a k beyond the static limit.  Tolerances are the kernel-level ones of DESIGN.md section 2 (fp64 1e-12,
fp32 2e-6 relative Frobenius): the two entry points sum in different orders."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_gram_static_matches_gram_twice_second_smaller(dtype):
    import torch

    from manta_amd import _lib
    from manta_amd.engine import Engine, MnmfError

    lib = _lib.load()
    eng = Engine(device=0, dtype=dtype)
    tol = 1e-12 if dtype == "fp64" else 2e-6
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    try:
        for rows, k, with_b in ((300_000, 50, False), (70_001, 50, True), (513, 7, False), (5, 200, True), (0, 9, False)):
            # one-row-longer buffers: a zero-row operand still has an address (torch gives an empty tensor none, and
            # both entry points refuse NULL)
            Abuf = torch.rand((rows + 1, k), generator=g, device="cuda", dtype=eng.tdtype)
            Bbuf = torch.rand((rows + 1, k), generator=g, device="cuda", dtype=eng.tdtype) if with_b else None
            A, B = Abuf[:rows], (Bbuf[:rows] if with_b else None)
            want = torch.empty((k, k), device="cuda", dtype=eng.tdtype)    # the existing entry point, mnmf_gram
            rc = lib.mnmf_gram(eng._h, rows, k, Abuf.data_ptr(), Bbuf.data_ptr() if with_b else None, want.data_ptr())
            assert rc == 0, lib.mnmf_last_error(eng._h)
            out = torch.full((k, k), float("nan"), device="cuda", dtype=eng.tdtype)
            # "allocates nothing", seen from outside torch's caching allocator: the library's own allocation counter
            # (every hipMalloc / hipHostMalloc of libmnmf.so goes through it) and the device's free bytes
            # (hipMemGetInfo) must not move over the call.  The first call per shape is a warm-up, so lazy loading of
            # the kernels' code object is not mistaken for an allocation.
            rc = lib.mnmf_gram_static(eng._h, rows, k, Abuf.data_ptr(), Bbuf.data_ptr() if with_b else None, out.data_ptr())
            assert rc == 0, lib.mnmf_last_error(eng._h)
            torch.cuda.synchronize()
            out.fill_(float("nan"))
            torch.cuda.synchronize()
            allocs, free_before = lib.mnmf_alloc_count(), torch.cuda.mem_get_info()[0]
            rc = lib.mnmf_gram_static(eng._h, rows, k, Abuf.data_ptr(), Bbuf.data_ptr() if with_b else None, out.data_ptr())
            assert rc == 0, lib.mnmf_last_error(eng._h)
            torch.cuda.synchronize()
            assert lib.mnmf_alloc_count() == allocs
            assert torch.cuda.mem_get_info()[0] == free_before
            ref = (A.double().T @ (B if with_b else A).double())
            scale = max(float(ref.norm()), 1e-300)
            assert float((out.double() - want.double()).norm()) <= tol * scale, (rows, k)
            assert float((out.double() - ref).norm()) <= tol * scale, (rows, k)
        A = torch.rand((64, 300), generator=g, device="cuda", dtype=eng.tdtype)
        out = torch.empty((300, 300), device="cuda", dtype=eng.tdtype)
        rc = lib.mnmf_gram_static(eng._h, 64, 300, A.data_ptr(), None, out.data_ptr())
        assert rc == -1   # MNMF_EINVAL
        assert b"exceeds the static scratch" in lib.mnmf_last_error(eng._h)
    finally:
        eng.close()
