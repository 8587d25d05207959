"""Kernel-level parity on the GPU, through the C-ABI, against numpy/scipy float64.

Tolerances: fp64 1e-12 relative Frobenius (summation order only); fp32 2e-6."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

TOL = {"fp64": 1e-12, "fp32": 2e-6}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def host(t):
    return t.detach().cpu().double().numpy()


def make_x(m, V, seed, dense_cols=2, empty_rows=(0, 7), empty_cols=(1,)):
    """Sparse matrix with empty rows/columns and a few near-dense columns (long rows of X^T)."""
    rng = np.random.RandomState(seed)
    X = sp.random(m, V, density=min(0.5, 12.0 / V), format="lil", random_state=rng, data_rvs=lambda n: rng.rand(n) + 0.1)
    for c in range(dense_cols):
        col = 2 + 3 * c
        rows = rng.rand(m) < (0.9 - 0.3 * c)
        X[np.nonzero(rows)[0], col] = rng.rand(int(rows.sum())) + 0.1
    for r in empty_rows:
        if r < m:
            X[r, :] = 0
    for c in empty_cols:
        X[:, c] = 0
    X = X.tocsr()
    X.eliminate_zeros()
    X.sort_indices()
    return X


@pytest.fixture(scope="module")
def engines():
    from manta_amd.engine import Engine

    e = {"fp32": Engine(dtype="fp32"), "fp64": Engine(dtype="fp64")}
    yield e
    for x in e.values():
        x.close()


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("k", [1, 5, 7, 10, 20, 50, 64, 100, 130, 260, 600])
def test_spmm_both_directions(engines, dtype, k):
    eng = engines[dtype]   # k = 260 (fp64) and 600 (both): wider than a wave covers -> column chunks
    X = make_x(1500, 700, seed=k)
    assert (np.diff(X.tocsc().indptr) > 512).any(), "want at least one chunked (long) row in X^T"
    eng.set_matrix(X)
    rng = np.random.RandomState(k)
    B = rng.rand(700, k)
    A = rng.rand(1500, k)
    assert rel(host(eng.spmm(B)), X @ B) <= TOL[dtype]
    assert rel(host(eng.spmm(A, transpose=True)), X.T @ A) <= TOL[dtype]
    # rows without nnz must come out exactly 0
    out = host(eng.spmm(B))
    assert np.all(out[np.diff(X.indptr) == 0] == 0)
    # signed operands (the solvers never produce them, the kernel-level entry points must still be exact)
    Bs, As = B - 0.5, 3.0 * (A - 0.5)
    assert rel(host(eng.spmm(Bs)), X @ Bs) <= TOL[dtype] * 20
    assert rel(host(eng.spmm(As, transpose=True)), X.T @ As) <= TOL[dtype] * 20


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_spmm_long_document_rows_and_ragged(engines, dtype):
    """Rows of X itself longer than the chunk threshold, single-nnz rows, a 1-row matrix."""
    eng = engines[dtype]
    rng = np.random.RandomState(3)
    X = sp.random(40, 5000, density=0.3, format="csr", random_state=rng)   # ~1500 nnz per row
    X.sort_indices()
    eng.set_matrix(X)
    B = rng.rand(5000, 12)
    assert rel(host(eng.spmm(B)), X @ B) <= TOL[dtype]
    A = rng.rand(40, 12)
    assert rel(host(eng.spmm(A, transpose=True)), X.T @ A) <= TOL[dtype]
    X1 = sp.csr_matrix(([1.5], ([0], [3])), shape=(1, 9))
    eng.set_matrix(X1)
    B = rng.rand(9, 4)
    assert rel(host(eng.spmm(B)), X1 @ B) <= TOL[dtype]
    X0 = sp.csr_matrix((5, 6))
    eng.set_matrix(X0)
    assert np.all(host(eng.spmm(rng.rand(6, 3))) == 0)


def test_set_matrix_rejects_bad_input(engines):
    import torch
    from manta_amd.engine import MnmfError

    eng = engines["fp32"]
    rowptr = torch.tensor([0, 2, 3], dtype=torch.int64)
    with pytest.raises(MnmfError, match="column index out of range"):
        eng.set_matrix(rowptr=rowptr, colidx=torch.tensor([0, 9, 1], dtype=torch.int32),
                       vals=torch.ones(3), shape=(2, 4))
    with pytest.raises(MnmfError, match="rowptr"):
        eng.set_matrix(rowptr=torch.tensor([0, 3, 2], dtype=torch.int64), colidx=torch.tensor([0, 1], dtype=torch.int32),
                       vals=torch.ones(2), shape=(2, 4))


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("rows,k", [(1, 3), (1000, 5), (5000, 50), (3001, 70), (777, 130), (300, 200),
                                     (40000, 160), (30011, 250), (2500, 136), (70001, 200)])
def test_gram(engines, dtype, rows, k):
    eng = engines[dtype]
    rng = np.random.RandomState(rows + k)
    A = rng.rand(rows, k)
    A[:, 0] *= 3.0                     # asymmetric content so a transposed write would show
    B = rng.rand(rows, k) + np.arange(k)[None, :] * 0.01
    assert rel(host(eng.gram(A)), A.T @ A) <= TOL[dtype]
    assert rel(host(eng.gram(A, B)), A.T @ B) <= TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("rows,k", [(1, 2), (999, 7), (4096, 50), (2000, 100), (500, 200), (20000, 160), (9999, 256),
                                    (3000, 260), (1500, 530), (700, 1000)])
def test_update(engines, dtype, rows, k):
    eng = engines[dtype]
    rng = np.random.RandomState(rows * 7 + k)
    A = rng.rand(rows, k)
    A[rng.rand(rows, k) < 0.3] = 0.0    # NNDSVD-style exact zeros must stay exactly zero
    num = rng.rand(rows, k)
    G = rng.rand(k, k)
    G2 = rng.rand(k, k)
    eps = 1e-10
    want = A * (num / (A @ G + eps))
    got = host(eng.update(A, num, G, eps=eps))
    assert rel(got, want) <= TOL[dtype] * 5
    assert np.all(got[A == 0] == 0)
    want = A * np.sqrt(num / (A @ G + num @ G2 + eps))
    got = host(eng.update(A, num, G, G2=G2, eps=eps, sqrt_mode=True))
    assert rel(got, want) <= TOL[dtype] * 5


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_diff_norms(engines, dtype):
    eng = engines[dtype]
    rng = np.random.RandomState(5)
    for rows, k in [(2000, 10), (300, 52), (50, 3), (30000, 136), (8001, 200), (5000, 256)]:   # incl. the wide Gram kernel
        A, B = rng.rand(rows, k), rng.rand(rows, k)
        d = np.abs(A - B)
        tol = 1e-10 if dtype == "fp64" else 1e-5
        assert abs(eng.diff_norm(A, B) - np.linalg.norm(d, 2)) <= tol * np.linalg.norm(d, 2)
        assert abs(eng.diff_norm(A, B, kind="fro") - np.linalg.norm(d, "fro")) <= tol * np.linalg.norm(d, "fro")
    assert eng.diff_norm(A, A) == 0.0


@pytest.fixture(scope="module")
def hot_engines():
    """Engines forced onto the LDS-staged persistent SpMM paths (normally used only for large matrices):
    hot rows of H^T for X.H^T with an LDS budget small enough that hot and cold rows both occur, and the
    tile stream of X^T.W (40 most frequent words, W tiles of a few hundred documents)."""
    import os
    from manta_amd.engine import Engine

    old = {k_: os.environ.get(k_) for k_ in ("MNMF_HOT_MIN_NNZ", "MNMF_HOT_LDS_BYTES", "MNMF_TILE_LDS_BYTES", "MNMF_XT_HOT_WORDS",
                                             "MNMF_TILE_MIN_ROWS", "MNMF_TILE_MIN_TILES_PCT")}
    os.environ["MNMF_HOT_MIN_NNZ"] = "0"
    os.environ["MNMF_HOT_LDS_BYTES"] = "20000"
    os.environ["MNMF_TILE_LDS_BYTES"] = "40000"
    os.environ["MNMF_XT_HOT_WORDS"] = "40"
    os.environ["MNMF_TILE_MIN_ROWS"] = "64"
    os.environ["MNMF_TILE_MIN_TILES_PCT"] = "0"
    try:
        e = {"fp32": Engine(dtype="fp32"), "fp64": Engine(dtype="fp64")}
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    yield e
    for x in e.values():
        x.close()


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("k", [5, 10, 50, 100, 130])
def test_spmm_lds_hot_rows(hot_engines, dtype, k):
    eng = hot_engines[dtype]
    X = make_x(3000, 900, seed=100 + k, dense_cols=6)
    eng.set_matrix(X)
    rng = np.random.RandomState(k)
    B = rng.rand(900, k)
    A = rng.rand(3000, k)
    assert rel(host(eng.spmm(B)), X @ B) <= TOL[dtype]
    assert rel(host(eng.spmm(A, transpose=True)), X.T @ A) <= TOL[dtype]


def test_solver_on_lds_hot_path(hot_engines, golden):
    g = golden("fixed_B_1000x1200_k10")
    eng = hot_engines["fp64"]
    eng.set_matrix(g["X"])
    W, H, _ = eng.run_nmf(g["W0"], g["H0"], fixed_iters=20, zero_threshold=1e-5)
    assert rel(host(W), g["nmf_W_20"]) <= 1e-10 and rel(host(H), g["nmf_H_20"]) <= 1e-10


@pytest.mark.parametrize("dtype,k", [("fp64", 50), ("fp32", 50), ("fp32", 37)])
def test_solver_on_compact_lds_rows(hot_engines, golden, dtype, k):
    """k = 50 lives in factor rows of 64 elements: the LDS images of the hot H^T rows and of the W tiles hold only the
    columns with data (compact rows), the lanes beyond read into the next row and must store zeros -- garbage in the
    padding of a numerator would reach the next iteration through the update kernels.  20 iterations against the CPU
    oracle."""
    from oracle import nmf_oracle as oc

    g = golden("fixed_B_1000x1200_k10")
    X = g["X"]
    rng = np.random.RandomState(k)
    W0, H0 = rng.uniform(0, X.max(), (X.shape[0], k)), rng.uniform(0, X.max(), (k, X.shape[1]))
    want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=20, check=False)
    eng = hot_engines[dtype]
    eng.set_matrix(X)
    W, H, _ = eng.run_nmf(W0, H0, fixed_iters=20, zero_threshold=1e-5)
    tol = 1e-10 if dtype == "fp64" else 2e-4
    assert rel(host(W), want["W"]) <= tol and rel(host(H), want["H"]) <= tol
    assert rel(host(eng.spmm(H0.T)), X @ H0.T) <= TOL[dtype]


@pytest.fixture(scope="module")
def small_segment_engines():
    """Engines with 8-nnz segments: every long row is cut into many segments, so its carry list is long enough
    (> 32 slots) for the block-per-row fix-up kernel, and short rows still take the wave-per-row one."""
    import os
    from manta_amd.engine import Engine

    keys = ("MNMF_X_Q", "MNMF_XT_Q")
    old = {k_: os.environ.get(k_) for k_ in keys}
    for k_ in keys:
        os.environ[k_] = "8"
    try:
        e = {"fp32": Engine(dtype="fp32"), "fp64": Engine(dtype="fp64")}
    finally:
        for k_, v in old.items():
            if v is None:
                os.environ.pop(k_, None)
            else:
                os.environ[k_] = v
    yield e
    for x in e.values():
        x.close()


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("k", [3, 10, 50, 100, 130])
def test_spmm_long_carry_lists(small_segment_engines, dtype, k):
    eng = small_segment_engines[dtype]
    X = make_x(3000, 900, seed=200 + k, dense_cols=4)
    assert (np.diff(X.tocsc().indptr) > 32 * 8 * 2).any()
    eng.set_matrix(X)
    rng = np.random.RandomState(k)
    B = rng.rand(900, k)
    A = rng.rand(3000, k)
    r1 = host(eng.spmm(A, transpose=True))
    assert rel(r1, X.T @ A) <= TOL[dtype]
    assert np.array_equal(r1, host(eng.spmm(A, transpose=True)))      # fixed summation tree: bit-reproducible
    assert rel(host(eng.spmm(B)), X @ B) <= TOL[dtype]
    Xl = sp.random(30, 4000, density=0.4, format="csr", random_state=rng)  # long rows of X itself
    Xl.sort_indices()
    eng.set_matrix(Xl)
    B = rng.rand(4000, k)
    assert rel(host(eng.spmm(B)), Xl @ B) <= TOL[dtype]


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("k", [1, 7, 10, 20, 50, 64, 100, 130])
def test_spmm_transposed_through_the_sweep_stream(dtype, k, monkeypatch):
    """X^T.A with the rare words forced through the synchronised sweep (spmm_sweep_kernel: CU-owned vocabulary slices with
    LDS accumulators, nnz in document windows; MNMF_SWEEP=2 takes it whatever the size, and tiny windows / a small LDS
    budget make a test-sized matrix span many windows and rows per partition): same products as scipy, rows without nnz
    exactly 0, bit-identical on a second call and to the default engine's sum order up to rounding."""
    from manta_amd.engine import Engine

    monkeypatch.setenv("MNMF_SWEEP", "2")
    monkeypatch.setenv("MNMF_SWEEP_WIN_BYTES", "16384")
    X = make_x(3000, 2500, seed=100 + k)
    eng = Engine(dtype=dtype)
    try:
        eng.set_matrix(X)
        rng = np.random.RandomState(k)
        A = rng.rand(3000, k)
        B = rng.rand(2500, k)
        got = host(eng.spmm(A, transpose=True))
        assert rel(got, X.T @ A) <= TOL[dtype]
        assert np.all(got[np.diff(X.tocsc().indptr) == 0] == 0)
        assert np.array_equal(got, host(eng.spmm(A, transpose=True)))
        assert rel(host(eng.spmm(B)), X @ B) <= TOL[dtype]
        As = 3.0 * (A - 0.5)
        assert rel(host(eng.spmm(As, transpose=True)), X.T @ As) <= TOL[dtype] * 20
        if k == 10:   # a short solver run on top of it
            from oracle import nmf_oracle as oc
            W0, H0 = rng.rand(3000, k) + 0.05, rng.rand(k, 2500) + 0.05
            want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=6, check=False)
            W, H, _ = eng.run_nmf(W0, H0, fixed_iters=6, zero_threshold=1e-5)
            assert rel(host(W), want["W"]) <= (1e-10 if dtype == "fp64" else 1e-4)
            assert rel(host(H), want["H"]) <= (1e-10 if dtype == "fp64" else 1e-4)
    finally:
        eng.close()


@pytest.mark.parametrize("dtype,k", [("fp32", 20), ("fp64", 50)])
def test_sweep_keeps_the_rarest_words_out_when_the_accumulators_do_not_fit(dtype, k, monkeypatch):
    """More qualifying words than accumulator rows (a tiny LDS budget here; wide rows or large vocabularies in
    practice): the rarest words stay in the blocked stream (the lower df bound is doubled until the rest fits) and the
    product is still scipy's, bit-identical run to run."""
    from manta_amd.engine import Engine

    monkeypatch.setenv("MNMF_SWEEP", "2")
    monkeypatch.setenv("MNMF_SWEEP_WIN_BYTES", "32768")
    monkeypatch.setenv("MNMF_SWEEP_LDS_BYTES", "1024")     # 2-8 rows per partition: about 500-2000 of the 2500 words
    X = make_x(3000, 2500, seed=7 + k)
    eng = Engine(dtype=dtype)
    try:
        eng.set_matrix(X)
        A = np.random.RandomState(k).rand(3000, k)
        got = host(eng.spmm(A, transpose=True))
        assert rel(got, X.T @ A) <= TOL[dtype]
        assert np.all(got[np.diff(X.tocsc().indptr) == 0] == 0)
        assert np.array_equal(got, host(eng.spmm(A, transpose=True)))
    finally:
        eng.close()
