"""CPU-only checks: the C-ABI library loads and exports every declared symbol, and the host-side
logic (segment builder, eigen-solver, drop-in argument handling) is right.  No GPU compute."""
import re
import os

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from manta_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mnmf.h")).read()
    declared = set(re.findall(r"\b(mnmf_[a-z0-9_]+)\s*\(", header))
    declared -= {"mnmf_allreduce_fn"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in mnmf.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert lib.mnmf_version().startswith(b"mnmf")


def _walk(rowptr, seg, long_row):
    """Python emulation of the stream kernel's row walk: which output row every nnz lands in."""
    rowptr = np.asarray(rowptr)
    nnz = int(rowptr[-1])
    rows = len(rowptr) - 1
    lens = np.diff(rowptr)
    endflag = np.zeros(nnz, bool)
    for r in range(rows):
        if 0 < lens[r] <= long_row:
            endflag[rowptr[r + 1] - 1] = True
    dest = np.full(nnz, -1, np.int64)       # row id, or -(carry+2)
    for s in range(len(seg["seg_info"])):
        a, b = seg["seg_start"][s], seg["seg_start"][s + 1]
        info = seg["seg_info"][s]
        if info < 0:
            assert not endflag[a:b].any()
            dest[a:b] = -(~info) - 2
        else:
            j = info
            start = a
            for p in range(a, b):
                if endflag[p]:
                    dest[start:p + 1] = seg["nzrow"][j]
                    j += 1
                    start = p + 1
            assert start == b, "an ordinary segment must end on a row boundary"
    return dest


@pytest.mark.parametrize("seg_nnz", [1, 32, 64, 512])
def test_segments_cover_every_nnz_once(seg_nnz):
    from manta_amd.engine import build_segments

    rng = np.random.RandomState(seg_nnz)
    lens = rng.poisson(20, 400)
    lens[[0, 5, 6, 399]] = 0                 # empty rows incl. first and last
    lens[[10, 200]] = [2000, 513]            # long rows
    lens[11] = 512                           # exactly at the threshold: not long
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seg = build_segments(rowptr, seg_nnz, 512)
    assert seg["seg_start"][0] == 0 and seg["seg_start"][-1] == rowptr[-1]
    assert np.all(np.diff(seg["seg_start"]) > 0)
    assert sorted(seg["empty_rows"].tolist()) == np.nonzero(lens == 0)[0].tolist()
    assert seg["nzrow"].tolist() == np.nonzero(lens > 0)[0].tolist()
    assert sorted(seg["fix_row"].tolist()) == [10, 200]
    dest = _walk(rowptr, seg, 512)
    truth = np.repeat(np.arange(400), lens)
    normal = dest >= 0
    assert np.array_equal(dest[normal], truth[normal])
    # chunked rows: their carry slots, in order, cover exactly the row's nnz
    for r, first, cnt in zip(seg["fix_row"], seg["fix_first"], seg["fix_count"]):
        slots = -(dest[rowptr[r]:rowptr[r + 1]]) - 2
        assert slots.min() == first and slots.max() == first + cnt - 1
        assert np.all(np.diff(slots) >= 0)
    assert seg["n_carry"] == int(seg["fix_count"].sum())
    assert not normal[rowptr[10]:rowptr[11]].any() and normal[rowptr[11]:rowptr[12]].all()


def test_segments_degenerate():
    from manta_amd.engine import build_segments

    seg = build_segments(np.zeros(6, np.int64), 64)      # all rows empty
    assert len(seg["seg_info"]) == 0 and seg["seg_start"].tolist() == [0] and len(seg["empty_rows"]) == 5
    seg = build_segments(np.array([0], np.int64), 64)    # no rows
    assert len(seg["seg_info"]) == 0


def test_lambda_max():
    from manta_amd import _lib

    lib = _lib.load()
    rng = np.random.RandomState(0)
    for n in (1, 3, 8, 52):
        d = np.abs(rng.randn(200, n))
        g = np.ascontiguousarray(d.T @ d)
        got = lib.mnmf_sym_lambda_max(g.ctypes.data, n)
        assert abs(np.sqrt(got) - np.linalg.norm(d, 2)) <= 1e-10 * np.linalg.norm(d, 2)


def test_run_nmf_argument_semantics_without_gpu():
    """Everything run_nmf decides before touching the GPU behaves like the reference
    (nmf_orchestrator.py:28-29, 105-106, 129-130)."""
    import manta_amd

    X = sp.random(30, 20, density=0.3, format="csr", random_state=1)
    with pytest.raises(ValueError, match="CSC"):
        manta_amd.find_max_rank(X)
    assert manta_amd.find_max_rank(X.tocsc()) == int(X.count_nonzero() / 50)
    calls = []

    def init(mat, rank):
        calls.append((sp.isspmatrix_csc(mat), rank))
        return np.ones((30, rank)), np.ones((rank, 20))

    with pytest.raises(ValueError, match="Unknown NMF method: bogus"):
        manta_amd.run_nmf(-1, X, init=init, nmf_method="bogus")
    assert calls == [(True, int(X.count_nonzero() / 50))]   # init ran first, on CSC, with the auto rank


def test_engine_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from manta_amd.engine import Engine

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine()
