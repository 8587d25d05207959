"""Solver-level parity on the GPU through the C-ABI: the three MU loops against (a) the golden
vectors produced by the unmodified reference and (b) the CPU oracle on larger seeded inputs.

Stated tolerances: relative Frobenius error fp64 <= 1e-10, fp32 <= 1e-4 (<= 200
iterations); entries that are exactly 0 in the reference are exactly 0; top-15 rankings identical
wherever adjacent reference scores differ by more than the tolerance."""
import numpy as np
import pytest

from oracle import nmf_oracle as oc

pytestmark = pytest.mark.gpu

TOL = {"fp64": 1e-10, "fp32": 1e-4}
FIXED = ["fixed_A_300x200_k5", "fixed_C_500x400_k7", "fixed_B_1000x1200_k10"]


def rel(a, b):
    d = np.linalg.norm(np.asarray(a, np.float64) - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def host(t):
    return t.detach().cpu().double().numpy()


def same_rankings(got, want, tol, top_n=15):
    """Row-wise descending order of ``got`` equals that of ``want`` where reference gaps exceed tol."""
    for g, w in zip(got, want):
        order_w = np.flip(np.argsort(w))[:top_n]
        order_g = np.flip(np.argsort(g))[:top_n]
        scale = max(np.abs(w).max(), 1e-300)
        for pos in range(top_n):
            if order_w[pos] == order_g[pos]:
                continue
            # allowed only if the reference scores of the two candidates are within tolerance (tie)
            assert abs(w[order_w[pos]] - w[order_g[pos]]) <= 10 * tol * scale, (pos, order_w, order_g)


@pytest.fixture(scope="module")
def engines():
    from manta_amd.engine import Engine

    e = {"fp32": Engine(dtype="fp32"), "fp64": Engine(dtype="fp64")}
    yield e
    for x in e.values():
        x.close()


def check_factors(got, want, dtype, zero_exact=True):
    for key in want:
        g = host(got[key])
        assert g.shape == want[key].shape
        assert np.all(np.isfinite(g))
        assert rel(g, want[key]) <= TOL[dtype], (key, rel(g, want[key]))
        if zero_exact and dtype == "fp64":
            assert np.all(g[want[key] == 0] == 0), f"{key}: reference zeros must stay exactly 0"


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("name", FIXED)
def test_nmf_fixed_iterations_vs_reference(engines, golden, name, dtype):
    g = golden(name)
    eng = engines[dtype]
    eng.set_matrix(g["X"])
    for N in g["Ns"]:
        W, H, st = eng.run_nmf(g["W0"], g["H0"], fixed_iters=int(N), zero_threshold=1e-5)
        assert st.iters == int(N)
        check_factors({"W": W, "H": H}, {"W": g[f"nmf_W_{N}"], "H": g[f"nmf_H_{N}"]}, dtype)
    N = int(g["Ns"][-1])
    W, H, _ = eng.run_nmf(g["Wr0"], g["Hr0"], fixed_iters=N, zero_threshold=1e-5)
    check_factors({"W": W, "H": H}, {"W": g["nmfr_W"], "H": g["nmfr_H"]}, dtype)
    same_rankings(host(H), g["nmfr_H"], TOL[dtype])
    same_rankings(host(W).T, g["nmfr_W"].T, TOL[dtype])


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("name", FIXED)
def test_pnmf_fixed_iterations_vs_reference(engines, golden, name, dtype):
    g = golden(name)
    eng = engines[dtype]
    eng.set_matrix(g["X"])
    for N in g["Ns"]:
        W, H, st = eng.run_pnmf(g["W0"], fixed_iters=int(N))
        assert st.iters == int(N)
        check_factors({"W": W, "H": H}, {"W": g[f"pnmf_W_{N}"], "H": g[f"pnmf_H_{N}"]}, dtype)


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("name", FIXED)
def test_nmtf_fixed_iterations_vs_reference(engines, golden, name, dtype):
    g = golden(name)
    eng = engines[dtype]
    eng.set_matrix(g["X"])
    for N in g["Ns"]:
        W, S, H, st = eng.run_nmtf(g["Wt0"], g["St0"], g["Ht0"], fixed_iters=int(N), zero_threshold=1e-5)
        assert st.iters == int(N)
        check_factors({"W": W, "S": S, "H": H},
                      {"W": g[f"nmtf_W_{N}"], "S": g[f"nmtf_S_{N}"], "H": g[f"nmtf_H_{N}"]}, dtype)
    N = int(g["Ns"][-1])
    W, S, H, _ = eng.run_nmtf(g["Wr0"], g["Sr0"], g["Hr0"], fixed_iters=N, zero_threshold=1e-5)
    check_factors({"W": W, "S": S, "H": H}, {"W": g["nmtfr_W"], "S": g["nmtfr_S"], "H": g["nmtfr_H"]}, dtype)


@pytest.mark.parametrize("method", ["nmf", "pnmf", "nmtf"])
def test_run_nmf_converged_vs_reference(golden, method):
    """The drop-in entry point, called as the pipeline calls it, from the reference's own init:
    same stopping iteration, same outputs (fp64), same keys; pnmf keeps the break-before-assign rule."""
    import manta_amd

    g = golden(f"converged_D_600x500_k6_{method}")
    stats = {}
    kw = dict(W0=g["W0"], H0=g["H0"])
    if method == "nmtf":
        kw = dict(W0=g["Wt0"], H0=g["Ht0"], S0=g["St0"])
    out = manta_amd.run_nmf(int(g["k"]), g["X"], norm_thresh=0.005, nmf_method=method, dtype="fp64",
                            stats_out=stats, **kw)
    assert sorted(out) == sorted(["W", "H"] + (["S"] if method == "nmtf" else []))
    iters = int(g["iters"]) - (1 if method == "pnmf" else 0)
    assert stats["iters"] == iters and stats["converged"]
    for key in out:
        assert out[key].dtype == np.float64 and out[key].flags["C_CONTIGUOUS"]
        want = g["out_" + key]
        assert rel(out[key], want) <= TOL["fp64"], (key, rel(out[key], want))
        assert np.all(out[key][want == 0] == 0)
    S = out.get("S")
    w_words, w_docs = oc.topic_rankings(g["out_H"], g["out_W"], S=g.get("out_S"))
    g_words, g_docs = oc.topic_rankings(out["H"], out["W"], S=S)
    assert np.array_equal(w_words, g_words)


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("method", ["nmf", "pnmf", "nmtf"])
def test_medium_zipf_vs_oracle(engines, dtype, method):
    """5000 x 8000 Zipf TF-IDF (the SURVEY calibration matrix), k = 10 and 50, 30 iterations."""
    import torch
    from manta_amd.synth import zipf_tfidf_csr, to_scipy

    rowptr, col, val = zipf_tfidf_csr(5000, 8000, 0.004, seed=77, device="cpu", dtype=torch.float64)
    X = to_scipy(rowptr, col, val, (5000, 8000))
    eng = engines[dtype]
    eng.set_matrix(X)
    rng = np.random.RandomState(42)
    for k in (10, 50):
        W0 = rng.uniform(0, X.max(), (5000, k))
        H0 = rng.uniform(0, X.max(), (k, 8000))
        S0 = rng.uniform(0, X.max(), (k, k))
        N = 30
        if method == "nmf":
            want = oc.basic_nmf(X, W0, H0, zero_threshold=1e-5, n_iter=N, check=False)
            W, H, _ = eng.run_nmf(W0, H0, fixed_iters=N, zero_threshold=1e-5)
            got = {"W": W, "H": H}
        elif method == "pnmf":
            want = oc.projective_nmf(X, W0, n_iter=N, check=False)
            W, H, _ = eng.run_pnmf(W0, fixed_iters=N)
            got = {"W": W, "H": H}
        else:
            want = oc.nmtf_core(X, W0, S0, H0, zero_threshold=1e-5, n_iter=N, check=False)
            W, S, H, _ = eng.run_nmtf(W0, S0, H0, fixed_iters=N, zero_threshold=1e-5)
            got = {"W": W, "S": S, "H": H}
        check_factors(got, want, dtype, zero_exact=False)
        same_rankings(host(got["H"]), want["H"], TOL[dtype])


def test_convergence_norms_match_oracle(engines):
    """Per-iteration spectral norms (computed from k x k Grams on the GPU) equal numpy's SVD norms."""
    g_name = "fixed_B_1000x1200_k10"
    from conftest import load_golden

    g = load_golden(g_name)
    eng = engines["fp64"]
    eng.set_matrix(g["X"])
    st_o = {}
    oc.basic_nmf(g["X"], g["W0"].copy(), g["H0"].copy(), n_iter=7, stats=st_o)
    _, _, st = eng.run_nmf(g["W0"], g["H0"], fixed_iters=7, force_check=True)
    assert abs(st.w_norm - st_o["w_norm"]) <= 1e-9 * st_o["w_norm"]
    assert abs(st.h_norm - st_o["h_norm"]) <= 1e-9 * st_o["h_norm"]
    st_o = {}
    oc.nmtf_core(g["X"], g["Wt0"].copy(), g["St0"].copy(), g["Ht0"].copy(), n_iter=5, stats=st_o)
    _, _, _, st = eng.run_nmtf(g["Wt0"], g["St0"], g["Ht0"], fixed_iters=5, force_check=True)
    for key in ("w_norm", "h_norm", "s_norm"):
        assert abs(getattr(st, key) - st_o[key]) <= 1e-9 * st_o[key]


def test_determinism(engines):
    """No atomics anywhere: two runs give bit-identical factors."""
    from conftest import load_golden

    g = load_golden("fixed_B_1000x1200_k10")
    eng = engines["fp32"]
    eng.set_matrix(g["X"])
    a = eng.run_nmf(g["Wr0"], g["Hr0"], fixed_iters=10)
    b = eng.run_nmf(g["Wr0"], g["Hr0"], fixed_iters=10)
    assert np.array_equal(host(a[0]), host(b[0])) and np.array_equal(host(a[1]), host(b[1]))


@pytest.mark.parametrize("m,V,k", [(3, 5, 2), (40, 17, 1), (900, 700, 300), (65, 33, 33), (300, 260, 530)])
@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_extreme_shapes_against_oracle(engines, dtype, m, V, k):
    """Tiny matrices, k = 1, k beyond the matrix-core kernels (k > 256), k just past a tile boundary, and k beyond
    what one wave covers of a row (fp64 > 256, fp32 > 512: the SpMM then runs in column chunks) -- the reference has
    no limit on k (nmf_basic.py:39-47)."""
    import scipy.sparse as sp

    rng = np.random.RandomState(m * 31 + k)
    X = sp.random(m, V, density=min(1.0, 6.0 / V + 0.05), format="csr", random_state=rng, data_rvs=lambda n: rng.rand(n) + 0.1)
    X.sort_indices()
    W0, H0 = rng.rand(m, k) + 0.05, rng.rand(k, V) + 0.05
    want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=7)
    eng = engines[dtype]
    eng.set_matrix(X)
    W, H, st = eng.run_nmf(W0, H0, fixed_iters=7, zero_threshold=1e-5)
    assert st.iters == 7
    check_factors({"W": W, "H": H}, want, dtype, zero_exact=False)
    want = oc.projective_nmf(X, W0.copy(), n_iter=4)
    W, H, st = eng.run_pnmf(W0, fixed_iters=4)
    check_factors({"W": W, "H": H}, want, dtype, zero_exact=False)


# ----------------------------------------------------------------------------------------------------------------
# BASELINE.json configs[0]: 20k docs x 30k vocab, ~0.3 % density, topic_count = 10, 200 iterations.
@pytest.fixture(scope="module")
def config0():
    import torch
    from manta_amd.synth import zipf_tfidf_csr, to_scipy

    m, V = 20_000, 30_000
    rowptr, col, val = zipf_tfidf_csr(m, V, 0.003, seed=1234, device="cpu", dtype=torch.float64)
    X = to_scipy(rowptr, col, val, (m, V))
    rng = np.random.RandomState(42)
    W0 = rng.uniform(0, X.max(), (m, 10))
    H0 = rng.uniform(0, X.max(), (10, V))
    want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=200, check=False)
    return X, W0, H0, want


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_config0_200_iterations_vs_oracle(engines, config0, dtype):
    """configs[0] in full: 200 MU iterations of standard NMF against the CPU oracle (nmf_basic.py:33-93) -- the stated
    tolerances at their stated horizon (fp64 1e-10, fp32 1e-4 after <= 200 iterations) and the top-15 rankings.  At
    this size the engine takes the path without the X^T tile stream and with the packed k = 10 exchange layout."""
    X, W0, H0, want = config0
    eng = engines[dtype]
    eng.set_matrix(X)
    W, H, st = eng.run_nmf(W0, H0, fixed_iters=200, zero_threshold=1e-5)
    assert st.iters == 200
    check_factors({"W": W, "H": H}, want, dtype, zero_exact=False)
    same_rankings(host(H), want["H"], TOL[dtype])
    same_rankings(host(W).T, want["W"].T, TOL[dtype])
    # with the convergence norms computed every iteration, as the reference does: same factors
    W2, H2, st2 = eng.run_nmf(W0, H0, fixed_iters=200, force_check=True, zero_threshold=1e-5)
    assert np.array_equal(host(W2), host(W)) and np.array_equal(host(H2), host(H))
    assert np.isfinite(st2.w_norm) and np.isfinite(st2.h_norm) and st2.w_norm > 0


def test_run_nmf_autorank_through_the_gpu_path(golden):
    """`topic_count = -1` (nmf_orchestrator.py:105-106, find_max_rank :14-46) through the drop-in entry point on the GPU:
    the rank the unmodified reference derived for this matrix (tests/golden/autorank_D.npz), and the factors of the
    oracle's run from the same host NNDSVD start (same numpy seed)."""
    import manta_amd

    g = golden("autorank_D")
    rank = int(g["rank"])
    np.random.seed(5)
    stats = {}
    out = manta_amd.run_nmf(-1, g["X"], norm_thresh=0.005, nmf_method="nmf", dtype="fp64", stats_out=stats)
    assert out["W"].shape == (g["X"].shape[0], rank) and out["H"].shape == (rank, g["X"].shape[1])
    np.random.seed(5)
    o_stats = {}
    want = oc.run_nmf(-1, g["X"], norm_thresh=0.005, nmf_method="nmf", stats=o_stats)
    assert stats["iters"] == o_stats["iters"] and stats["converged"]
    for key in ("W", "H"):
        assert rel(out[key], want[key]) <= TOL["fp64"], (key, rel(out[key], want[key]))
    with pytest.raises(ValueError, match="Unknown NMF method"):
        manta_amd.run_nmf(-1, g["X"], nmf_method="bogus", W0=want["W"], H0=want["H"])


FP32_STOP_SLACK = 0   # measured on MI355X: the fp32 runs stop at the reference's iterations (115 / 45 / 129)


@pytest.mark.parametrize("method", ["nmf", "pnmf", "nmtf"])
def test_run_nmf_converged_fp32_stopping_iteration(golden, method):
    """The fp32 engine under the reference's stopping rule, from the reference's own start: the reference (float64)
    stopped after 115 / 45 (44 applied) / 129 updates (tests/golden/converged_D_*, recorded from the unmodified
    run_nmf).  fp32 norms differ from float64 ones by ~1e-7 relative, so a norm that crosses 0.005 by less than that
    may cross one iteration earlier or later: the stopping iteration is asserted to within FP32_STOP_SLACK and the
    factors to the fp32 tolerance against the oracle stopped at the SAME iteration as the GPU run."""
    import manta_amd

    g = golden(f"converged_D_600x500_k6_{method}")
    stats = {}
    kw = dict(W0=g["W0"], H0=g["H0"])
    if method == "nmtf":
        kw = dict(W0=g["Wt0"], H0=g["Ht0"], S0=g["St0"])
    out = manta_amd.run_nmf(int(g["k"]), g["X"], norm_thresh=0.005, nmf_method=method, dtype="fp32", stats_out=stats, **kw)
    ref_iters = int(g["iters"]) - (1 if method == "pnmf" else 0)
    assert stats["converged"] and abs(stats["iters"] - ref_iters) <= FP32_STOP_SLACK, (stats["iters"], ref_iters)
    if stats["iters"] == ref_iters:
        want = {key: g["out_" + key] for key in out}
    elif method == "nmf":
        want = oc.basic_nmf(g["X"], g["W0"].copy(), g["H0"].copy(), zero_threshold=1e-5, n_iter=stats["iters"], check=False)
    elif method == "pnmf":
        want = oc.projective_nmf(g["X"], g["W0"].copy(), n_iter=stats["iters"], check=False)
    else:
        want = oc.nmtf_core(g["X"], g["Wt0"].copy(), g["St0"].copy(), g["Ht0"].copy(), zero_threshold=1e-5,
                            n_iter=stats["iters"], check=False)
    for key in out:
        assert rel(out[key], want[key]) <= TOL["fp32"], (key, rel(out[key], want[key]))
    print(f"fp32 {method}: stopped after {stats['iters']} updates (reference float64: {ref_iters})")


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("method", ["nmf", "pnmf", "nmtf"])
def test_non_finite_factors_are_an_error_not_convergence(engines, golden, dtype, method):
    """A factor that went NaN / inf must end in MNMF_ENAN (-4), never in `converged` with NaN factors: the norm is
    computed from a Gram whose entries are then NaN, and a clamp like max(0, NaN) = 0 would read as 'below threshold'."""
    from manta_amd.engine import MnmfError

    g = golden("fixed_A_300x200_k5")
    eng = engines[dtype]
    eng.set_matrix(g["X"])
    for bad in (float("nan"), float("inf")):
        W0 = g["Wr0"].copy()
        W0[7, 2] = bad
        with pytest.raises(MnmfError) as ei:
            if method == "nmf":
                eng.run_nmf(W0, g["Hr0"], max_iter=50)
            elif method == "pnmf":
                eng.run_pnmf(W0, max_iter=50)
            else:
                eng.run_nmtf(W0, g["Sr0"], g["Hr0"], max_iter=50)
        assert ei.value.code == -4, ei.value
    # the context stays usable
    W, H, st = eng.run_nmf(g["Wr0"], g["Hr0"], fixed_iters=3)
    assert np.all(np.isfinite(host(W)))


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_k200_medium_vs_oracle(engines, dtype):
    """k = 200 (configs[4]'s topic count: 832-byte factor rows in fp32, two vectors per lane; wide Gram and the
    two-column-group update) on a matrix the oracle finishes in seconds: 12 iterations of all three solvers."""
    import torch
    from manta_amd.synth import zipf_tfidf_csr, to_scipy

    m, V, k, N = 6000, 9000, 200, 12
    rowptr, col, val = zipf_tfidf_csr(m, V, 0.004, seed=99, device="cpu", dtype=torch.float64)
    X = to_scipy(rowptr, col, val, (m, V))
    eng = engines[dtype]
    eng.set_matrix(X)
    rng = np.random.RandomState(200)
    W0, H0, S0 = rng.uniform(0, X.max(), (m, k)), rng.uniform(0, X.max(), (k, V)), rng.uniform(0, X.max(), (k, k))
    want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=N, check=False)
    W, H, _ = eng.run_nmf(W0, H0, fixed_iters=N, zero_threshold=1e-5, force_check=True)
    check_factors({"W": W, "H": H}, want, dtype, zero_exact=False)
    same_rankings(host(H), want["H"], TOL[dtype])
    want = oc.projective_nmf(X, W0.copy(), n_iter=4, check=False)
    W, H, _ = eng.run_pnmf(W0, fixed_iters=4)
    check_factors({"W": W, "H": H}, want, dtype, zero_exact=False)
    want = oc.nmtf_core(X, W0.copy(), S0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=4, check=False)
    W, S, H, _ = eng.run_nmtf(W0, S0, H0, fixed_iters=4, zero_threshold=1e-5)
    check_factors({"W": W, "S": S, "H": H}, want, dtype, zero_exact=False)


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_wide_address_variant_of_the_spmm(golden, dtype, monkeypatch):
    """The SpMM variant for gather tables of >= 2^24 rows or >= 4 GiB (64-bit addresses, plain global loads;
    spmm.hpp WIDE) cannot be reached by a test-sized matrix, so the engine is told to take it (MNMF_FORCE_WIDE):
    both products and a short solver run must equal the default variant's bit for bit."""
    from manta_amd.engine import Engine

    g = golden("fixed_B_1000x1200_k10")
    rng = np.random.RandomState(1)
    B, A = rng.rand(1200, 10), rng.rand(1000, 10)
    res = {}
    for wide in ("0", "1"):
        monkeypatch.setenv("MNMF_FORCE_WIDE", wide)
        eng = Engine(dtype=dtype)
        try:
            eng.set_matrix(g["X"])
            W, H, _ = eng.run_nmf(g["Wr0"], g["Hr0"], fixed_iters=5)
            res[wide] = (host(eng.spmm(B)), host(eng.spmm(A, transpose=True)), host(W), host(H))
        finally:
            eng.close()
    for a, b in zip(res["0"], res["1"]):
        assert np.array_equal(a, b)
    assert rel(res["1"][0], g["X"] @ B) <= (1e-12 if dtype == "fp64" else 2e-6)
    assert rel(res["1"][1], g["X"].T @ A) <= (1e-12 if dtype == "fp64" else 2e-6)


def test_roctx_ranges_do_not_change_results(golden, tmp_path):
    """MNMF_ROCTX=1 (tracing): the marker library is loaded with dlopen on first use and every solver phase is
    bracketed by a named range (what `rocprofv3 --marker-trace` records, profiles/r03_roctx_c1_marker_stats.csv).  The
    switch is read once per process, hence a child process; its factors must equal this process's bit for bit."""
    import os
    import subprocess
    import sys

    from manta_amd.engine import Engine

    g = golden("fixed_A_300x200_k5")
    out = tmp_path / "w.npy"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the child has no conftest: hand the matrix over as arrays
    X = g["X"].tocsr()
    np.savez(tmp_path / "in.npz", data=X.data, indices=X.indices, indptr=X.indptr, shape=np.array(X.shape), W0=g["W0"], H0=g["H0"])
    code = (
        "import sys, numpy as np, scipy.sparse as sp; sys.path.insert(0, %r)\n"
        "from manta_amd.engine import Engine\n"
        "d = np.load(%r)\n"
        "X = sp.csr_matrix((d['data'], d['indices'], d['indptr']), shape=tuple(d['shape']))\n"
        "e = Engine(dtype='fp64'); e.set_matrix(X)\n"
        "W, H, s = e.run_nmf(d['W0'], d['H0'], fixed_iters=10, force_check=True)\n"
        "np.save(%r, W.detach().cpu().numpy()); e.close(); print('OK')\n" % (root, str(tmp_path / "in.npz"), str(out))
    )
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MNMF_ROCTX="1"), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    e = Engine(dtype="fp64")
    try:
        e.set_matrix(X)
        W, H, s = e.run_nmf(g["W0"], g["H0"], fixed_iters=10, force_check=True)
        assert np.array_equal(np.load(out), W.detach().cpu().numpy())
    finally:
        e.close()
