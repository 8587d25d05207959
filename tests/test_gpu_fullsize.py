"""Full-size checks (BASELINE.json configs[1..3]; mostly configs[1]: 1M docs x 100k vocab, ~0.1 % density, k = 50)
through the C-ABI.

The CPU oracle needs minutes per iteration at this size, so parity here rests on size-independent properties and
on an independent float64 product (torch's sparse CSR matmul on the same device arrays):
  * X.B and X^T.A against torch.sparse float64 (all stream variants are active at this size: LDS-hot rows of H^T,
    the document-blocked X^T stream, the LDS tile stream of the frequent words, block-per-row fix-ups);
  * adjointness <X.B, A> = <B, X^T.A> and linearity of both products;
  * five NMF iterations: fp32 agrees with fp64 within the stated fp32 tolerance, the objective ||X - W.H||_F
    never increases (the MU guarantee the reference relies on), exact zeros of the init stay exactly zero,
    and two runs are bit-identical;
  * two iterations of NMF (configs[1]), Projective NMF (configs[2], k = 100) and NMTF (configs[3], 2M x 200k,
    k = 20) against a plain torch float64 restatement of the reference's update formulas.
Tolerances: fp64 1e-12, fp32 2e-6 relative Frobenius for single products; 1e-4 for the iterated factors."""
import pytest

pytestmark = pytest.mark.gpu

M, V, DENSITY, K = 1_000_000, 100_000, 0.001, 50


@pytest.fixture(scope="module")
def c2():
    import torch
    from manta_amd.synth import zipf_tfidf_csr

    dev = torch.device("cuda", 0)
    rowptr, col, val = zipf_tfidf_csr(M, V, DENSITY, seed=1234, device=dev, dtype=torch.float64)
    Xs = torch.sparse_csr_tensor(rowptr, col.long(), val, size=(M, V))
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    B = torch.rand((V, K), generator=g, device=dev, dtype=torch.float64)
    A = torch.rand((M, K), generator=g, device=dev, dtype=torch.float64)
    return dict(rowptr=rowptr, col=col, val=val, Xs=Xs, A=A, B=B, dev=dev)


def _rel(a, b):
    import torch
    return float(torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()))


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-6), ("fp64", 1e-12)])
def test_products_against_torch_sparse_and_adjointness(c2, dtype, tol):
    import torch
    from manta_amd.engine import Engine

    eng = Engine(dtype=dtype)
    try:
        eng.set_matrix(rowptr=c2["rowptr"], colidx=c2["col"], vals=c2["val"], shape=(M, V))
        A, B = c2["A"].to(eng.tdtype), c2["B"].to(eng.tdtype)
        XB = eng.spmm(B)
        XtA = eng.spmm(A, transpose=True)
        want_XB = c2["Xs"] @ B.double()
        assert _rel(XB, want_XB) <= tol
        del want_XB
        # X^T.A in float64 without materialising X^T: column sums through index_add over the nnz
        rows = torch.repeat_interleave(torch.arange(M, device=c2["dev"]), c2["rowptr"][1:] - c2["rowptr"][:-1])
        want = torch.zeros((V, K), device=c2["dev"], dtype=torch.float64)
        step = 1 << 24
        for s in range(0, rows.numel(), step):
            e = min(s + step, rows.numel())
            want.index_add_(0, c2["col"][s:e].long(), A.double()[rows[s:e]] * c2["val"][s:e, None])
        assert _rel(XtA, want) <= max(tol, 1e-12)
        del want, rows
        lhs = float((XB.double() * A.double()).sum())
        rhs = float((B.double() * XtA.double()).sum())
        assert abs(lhs - rhs) <= 10 * tol * abs(rhs)
        B2 = torch.flip(B, dims=[0])
        assert _rel(eng.spmm(2.0 * B - 3.0 * B2), 2.0 * XB.double() - 3.0 * eng.spmm(B2).double()) <= 10 * tol
        assert torch.equal(eng.spmm(A, transpose=True), XtA), "X^T.A must be bit-reproducible"
    finally:
        eng.close()


def test_sweep_stream_at_full_size_equals_the_blocked_stream(c2, monkeypatch):
    """X^T.A with MNMF_SWEEP=1 (rare words through the synchronised sweep stream where the engine's rule takes it:
    at this size it does, tools/bench_cfgs.sh shows the sweep kernel) against the shipped default (sweep off).  The
    sweep adds a word's nnz in document order into its accumulator row, which is the order the blocked stream uses
    for a word it does not cut -- so the two products are the same bit for bit, not merely within tolerance."""
    import torch
    from manta_amd.engine import Engine

    A = c2["A"].float()
    base = Engine(dtype="fp32")
    try:
        base.set_matrix(rowptr=c2["rowptr"], colidx=c2["col"], vals=c2["val"], shape=(M, V))
        want = base.spmm(A, transpose=True)
    finally:
        base.close()
    monkeypatch.setenv("MNMF_SWEEP", "1")
    eng = Engine(dtype="fp32")
    try:
        eng.set_matrix(rowptr=c2["rowptr"], colidx=c2["col"], vals=c2["val"], shape=(M, V))
        got = eng.spmm(A, transpose=True)
        assert torch.equal(got, want)
        assert torch.equal(eng.spmm(A, transpose=True), got)
    finally:
        eng.close()


def test_nmf_iterations_full_size(c2):
    import torch
    from manta_amd.engine import Engine

    dev = c2["dev"]
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    xmax = float(c2["val"].max())
    H0 = torch.rand((K, V), generator=g, device=dev, dtype=torch.float64) * xmax
    W0 = torch.rand((M, K), generator=g, device=dev, dtype=torch.float64) * xmax
    W0[::7, 3] = 0.0                       # NNDSVD-style exact zeros must stay zero
    H0[5, ::11] = 0.0
    x2 = float((c2["val"] ** 2).sum())

    def objective(W, H):                    # ||X - W.H||_F^2 = ||X||^2 - 2 <X.H^T, W> + <W^T W, H H^T>
        W, H = W.double(), H.double()
        XHt = c2["Xs"] @ H.t().contiguous()
        return x2 - 2.0 * float((XHt * W).sum()) + float(((W.t() @ W) * (H @ H.t())).sum())

    out = {}
    for dtype in ("fp64", "fp32"):
        eng = Engine(dtype=dtype)
        try:
            eng.set_matrix(rowptr=c2["rowptr"], colidx=c2["col"], vals=c2["val"], shape=(M, V))
            objs = [objective(W0, H0)]
            W, H = W0, H0
            for _ in range(5):
                W, H, st = eng.run_nmf(W, H, fixed_iters=1, force_check=True, finalize=False)
                objs.append(objective(W, H))
            assert all(b <= a * (1 + 1e-9) for a, b in zip(objs, objs[1:])), objs
            W5, H5, st = eng.run_nmf(W0, H0, fixed_iters=5, force_check=True, finalize=False)
            assert st.iters == 5 and st.w_norm > 0 and st.h_norm > 0
            if dtype == "fp64":
                assert _rel(W5, W) <= 1e-12 and _rel(H5, H) <= 1e-12   # one call of 5 = five calls of 1
            assert bool((W5[::7, 3] == 0).all()) and bool((H5[5, ::11] == 0).all())
            W5b, H5b, _ = eng.run_nmf(W0, H0, fixed_iters=5, force_check=True, finalize=False)
            assert torch.equal(W5, W5b) and torch.equal(H5, H5b), "runs must be bit-identical"
            out[dtype] = (W5.double(), H5.double())
        finally:
            eng.close()
    assert _rel(out["fp32"][0], out["fp64"][0]) <= 1e-4
    assert _rel(out["fp32"][1], out["fp64"][1]) <= 1e-4


# ----------------------------------------------------------------------------------------------------------------
# One or two iterations of each solver at the full size of its BASELINE.json config against a plain torch float64
# restatement of the reference's update formulas (the numpy oracle needs minutes per iteration here).
def _torch_ops(rowptr, col, val, m, V, dev):
    import torch

    Xs = torch.sparse_csr_tensor(rowptr, col.long(), val, size=(m, V))
    rows = torch.repeat_interleave(torch.arange(m, device=dev), rowptr[1:] - rowptr[:-1])

    def x_times(B):            # X @ B
        return Xs @ B.contiguous()

    def xt_times(A):           # X.T @ A without materialising X.T
        out = torch.zeros((V, A.shape[1]), device=dev, dtype=torch.float64)
        step = 1 << 24
        for s in range(0, rows.numel(), step):
            e = min(s + step, rows.numel())
            out.index_add_(0, col[s:e].long(), A[rows[s:e]] * val[s:e, None])
        return out

    return x_times, xt_times


def _inits(m, V, k, xmax, dev, seed=42):
    import torch

    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    H0 = torch.rand((k, V), generator=g, device=dev, dtype=torch.float64) * xmax
    S0 = torch.rand((k, k), generator=g, device=dev, dtype=torch.float64) * xmax
    W0 = torch.rand((m, k), generator=g, device=dev, dtype=torch.float64) * xmax
    return W0, S0, H0


@pytest.mark.parametrize("method,m,V,density,k,iters", [
    ("nmf", 1_000_000, 100_000, 0.001, 50, 2),      # configs[1]
    ("pnmf", 1_000_000, 100_000, 0.001, 100, 2),    # configs[2]
    ("nmtf", 2_000_000, 200_000, 0.0005, 20, 2),    # configs[3]
    ("nmf", 2_500_000, 500_000, 0.0002, 200, 2),    # one rank's share of configs[4] (20M x 500k, k = 200, 8 ranks)
])
def test_solver_iterations_against_torch_float64_full_size(method, m, V, density, k, iters):
    import torch
    from manta_amd.engine import Engine
    from manta_amd.synth import zipf_tfidf_csr

    dev = torch.device("cuda", 0)
    rowptr, col, val = zipf_tfidf_csr(m, V, density, seed=1234, device=dev, dtype=torch.float64)
    x_times, xt_times = _torch_ops(rowptr, col, val, m, V, dev)
    W0, S0, H0 = _inits(m, V, k, float(val.max()), dev)
    w, s, h = W0.clone(), S0.clone(), H0.clone()
    for _ in range(iters):
        if method == "nmf":                                        # nmf_basic.py:39-47, eps 1e-10
            h = h * (xt_times(w).t() / ((w.t() @ w) @ h + 1e-10))
            w = w * (x_times(h.t()) / (w @ (h @ h.t()) + 1e-10))
        elif method == "pnmf":                                     # nmf_projective_basic.py:68-79
            wtw = w.t() @ w
            pay = x_times(xt_times(w))
            w = w * torch.sqrt(pay / (w @ (w.t() @ pay) + pay @ wtw + 1e-10))
        else:                                                      # nmtf.py:150-163, eps 1e-9
            hht = h @ h.t()
            w1 = w * (x_times(h.t() @ s.t()) / (w @ s @ hht @ s.t() + 1e-9))
            P = xt_times(w1).t()                                   # w1.T @ X
            g1 = w1.t() @ w1
            s1 = s * ((P @ h.t()) / (g1 @ s @ hht + 1e-9))
            h = h * ((s1.t() @ P) / (s1.t() @ g1 @ s1 @ h + 1e-9))
            w, s = w1, s1
    for dtype, tol in (("fp64", 1e-10), ("fp32", 1e-4)):
        eng = Engine(dtype=dtype)
        try:
            eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m, V))
            kw = dict(fixed_iters=iters, force_check=True, finalize=False)
            if method == "nmf":
                W, H, st = eng.run_nmf(W0, H0, **kw)
                assert _rel(W, w) <= tol and _rel(H, h) <= tol, (dtype, _rel(W, w), _rel(H, h))
            elif method == "pnmf":
                W, H, st = eng.run_pnmf(W0, **kw)                  # finalize=False: W after `iters` updates, H = W.T @ X
                assert _rel(W, w) <= tol, (dtype, _rel(W, w))
                assert _rel(H, xt_times(w).t()) <= tol
            else:
                W, S, H, st = eng.run_nmtf(W0, S0, H0, **kw)
                assert _rel(W, w) <= tol and _rel(S, s) <= tol and _rel(H, h) <= tol, (dtype, _rel(W, w), _rel(S, s), _rel(H, h))
            assert st.iters == iters
        finally:
            eng.close()


def test_config4_shard_products_adjointness_reproducibility():
    """One rank's share of configs[4] (2.5M docs x 500k vocab, 2.5e8 nnz, k = 200; 832-byte factor rows, two vectors
    per lane, 2 GB gather tables): X.B and X^T.A against float64 references built from the same device arrays,
    adjointness, and bit-reproducibility of both products (fp32, the dtype the config is quoted in)."""
    import torch
    from manta_amd.engine import Engine
    from manta_amd.synth import zipf_tfidf_csr

    m, V, k = 2_500_000, 500_000, 200
    dev = torch.device("cuda", 0)
    rowptr, col, val = zipf_tfidf_csr(m, V, 0.0002, seed=1234, device=dev, dtype=torch.float64)
    x_times, xt_times = _torch_ops(rowptr, col, val, m, V, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    B = torch.rand((V, k), generator=g, device=dev, dtype=torch.float32)
    A = torch.rand((m, k), generator=g, device=dev, dtype=torch.float32)
    eng = Engine(dtype="fp32")
    try:
        eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m, V))
        XB = eng.spmm(B)
        assert _rel(XB, x_times(B.double())) <= 2e-6
        XtA = eng.spmm(A, transpose=True)
        want = torch.zeros((V, k), device=dev, dtype=torch.float64)   # X^T.A in column halves: bounded temporaries
        for c0 in range(0, k, 50):
            want[:, c0:c0 + 50] = xt_times(A[:, c0:c0 + 50].double())
        assert _rel(XtA, want) <= 2e-6
        del want
        lhs = float((XB.double() * A.double()).sum())
        rhs = float((B.double() * XtA.double()).sum())
        assert abs(lhs - rhs) <= 2e-5 * abs(rhs)
        assert torch.equal(eng.spmm(A, transpose=True), XtA) and torch.equal(eng.spmm(B), XB)
    finally:
        eng.close()
