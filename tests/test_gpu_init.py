"""GPU NNDSVD initialisation (manta_amd/init_device.py) against the CPU oracle's NNDSVD
(ARPACK svds, the reference's algorithm, nmf_initialization.py:38-83).

Not bit-parity: ARPACK starts from a random vector and is itself reproducible only to ~1e-13; NNDSVD is invariant
to the sign of a singular pair.  Stated tolerances: singular values 1e-9 relative, W0 / H0 1e-6 relative Frobenius
with a float64 engine; 2e-3 with a float32 engine (the Lanczos operator is then applied in float32)."""
import numpy as np
import pytest
import scipy.sparse.linalg as sla

from oracle import nmf_oracle as oc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b)


def host(t):
    return t.detach().cpu().double().numpy()


def zipf_matrix(m, V, seed):
    import torch
    from manta_amd.synth import zipf_tfidf_csr, to_scipy

    rowptr, col, val = zipf_tfidf_csr(m, V, 0.004, seed=seed, device="cpu", dtype=torch.float64)
    return to_scipy(rowptr, col, val, (m, V))


@pytest.mark.parametrize("case", ["golden_B", "zipf"])
def test_nndsvd_device_matches_oracle(golden, case):
    from manta_amd.engine import Engine
    from manta_amd.init_device import nndsvd_device, truncated_svd_device

    if case == "golden_B":
        X, k = golden("fixed_B_1000x1200_k10")["X"], 10
    else:
        X, k = zipf_matrix(6000, 9000, 5), 12
    X = X.tocsr().astype(np.float64)
    np.random.seed(0)
    W_ref, H_ref = oc.nndsvd_init(X, k)
    s_ref = np.sort(sla.svds(X, k=k, return_singular_vectors=False))[::-1]
    eng = Engine(dtype="fp64")
    try:
        eng.set_matrix(X)
        info = {}
        u, s, v = truncated_svd_device(eng, k, info=info)
        print(case, info)
        assert info["converged"], info
        assert np.max(np.abs(host(s) - s_ref) / s_ref) <= 1e-9
        uu, vv = host(u), host(v)
        assert np.abs(uu.T @ uu - np.eye(k)).max() <= 1e-8 and np.abs(vv.T @ vv - np.eye(k)).max() <= 1e-10
        W0, H0 = nndsvd_device(eng, k)
        assert W0.shape == (X.shape[0], k) and H0.shape == (k, X.shape[1])
        assert rel(host(W0), W_ref) <= 1e-6 and rel(host(H0), H_ref) <= 1e-6, (rel(host(W0), W_ref), rel(host(H0), H_ref))
        # same sparsity pattern up to entries at the 1e-11 cut
        assert np.mean((host(W0) == 0) != (W_ref == 0)) <= 1e-3
    finally:
        eng.close()
    eng = Engine(dtype="fp32")
    try:
        eng.set_matrix(X)
        W0, H0 = nndsvd_device(eng, k, tol=1e-6)
        assert rel(host(W0), W_ref) <= 2e-3 and rel(host(H0), H_ref) <= 2e-3, (rel(host(W0), W_ref), rel(host(H0), H_ref))
    finally:
        eng.close()


def test_run_nmf_with_device_init_converges_like_host_init():
    """End to end: run_nmf with init='device' reaches the same factorisation quality as with the host NNDSVD."""
    import manta_amd

    X = zipf_matrix(4000, 5000, 9).tocsr()
    np.random.seed(0)
    a = manta_amd.run_nmf(8, X, norm_thresh=0.005, nmf_method="nmf", init="device", dtype="fp64")
    np.random.seed(0)
    b = manta_amd.run_nmf(8, X, norm_thresh=0.005, nmf_method="nmf", dtype="fp64")
    ea = np.linalg.norm(X.toarray() - a["W"] @ a["H"])
    eb = np.linalg.norm(X.toarray() - b["W"] @ b["H"])
    assert abs(ea - eb) <= 1e-3 * eb
    assert rel(a["W"], b["W"]) <= 1e-3 and rel(a["H"], b["H"]) <= 1e-3


def test_matrix_stays_resident_across_topic_counts():
    """A topic-count sweep on one matrix object uploads and preprocesses it once;
    changing the content in place invalidates the residency."""
    import manta_amd

    X = zipf_matrix(3000, 4000, 3).tocsr()
    st = {}
    a = manta_amd.run_nmf(5, X, norm_thresh=0.005, init="device", dtype="fp32", stats_out=st)
    first = st["matrix_resident"]
    b = manta_amd.run_nmf(7, X, norm_thresh=0.005, init="device", dtype="fp32", stats_out=st)
    assert st["matrix_resident"] and a["W"].shape[1] == 5 and b["W"].shape[1] == 7
    a2 = manta_amd.run_nmf(5, X, norm_thresh=0.005, init="device", dtype="fp32", stats_out=st)
    assert st["matrix_resident"] and np.array_equal(a["W"], a2["W"]) and np.array_equal(a["H"], a2["H"])
    X.data *= 2.0
    c = manta_amd.run_nmf(5, X, norm_thresh=0.005, init="device", dtype="fp32", stats_out=st)
    assert not st["matrix_resident"] and not first
    # the new content was used: the reconstruction doubled (the absolute stopping threshold moves the last iteration)
    assert rel(c["W"] @ c["H"], 2.0 * (a["W"] @ a["H"])) <= 0.05


@pytest.mark.parametrize("name,k", [("fixed_B_1000x1200_k10", 10), ("fixed_C_500x400_k7", 7), ("fixed_A_300x200_k5", 5)])
def test_nmtf_legacy_init_device_matches_reference_fixture(golden, name, k):
    """The legacy double NNDSVD (nmtf_init.py:19-41) on the device against W0/S0/H0 that the reference itself produced
    (tests/golden, keys Wt0/St0/Ht0).  Tolerance 1e-6 relative Frobenius (float64 engine)."""
    from manta_amd.engine import Engine
    from manta_amd.init_device import nmtf_nndsvd_legacy_device

    g = golden(name)
    eng = Engine(dtype="fp64")
    try:
        eng.set_matrix(g["X"].tocsr().astype(np.float64))
        W0, S0, H0 = nmtf_nndsvd_legacy_device(eng, k)
        assert W0.shape == g["Wt0"].shape and S0.shape == (k, k) and H0.shape == g["Ht0"].shape
        errs = (rel(host(W0), g["Wt0"]), rel(host(S0), g["St0"]), rel(host(H0), g["Ht0"]))
        assert max(errs) <= 1e-6, errs
    finally:
        eng.close()


def test_random_device_init_and_run_nmf_strings():
    import manta_amd
    from manta_amd.engine import Engine
    from manta_amd.init_device import random_device

    X = zipf_matrix(3000, 2500, 3).tocsr()
    eng = Engine(dtype="fp32")
    try:
        eng.set_matrix(X)
        assert eng.value_range() == (0.0, pytest.approx(float(X.max()), rel=1e-6))   # float32 engine
        W0, S0, H0 = random_device(eng, 6, seed=5, with_s=True)
        W0b, H0b = random_device(eng, 6, seed=5)
        assert W0.shape == (3000, 6) and S0.shape == (6, 6) and H0.shape == (6, 2500)
        assert (host(W0) == host(W0b)).all()                            # seeded: reproducible
        for t in (W0, S0, H0):
            assert float(t.min()) >= 0.0 and float(t.max()) <= X.max() * (1 + 1e-6)
        assert abs(float(W0.mean()) - X.max() / 2) <= 0.02 * X.max()
    finally:
        eng.close()
    dense = X.toarray()
    for method in ("nmf", "nmtf"):
        for init in ("device", "device_random"):
            st = {}
            out = manta_amd.run_nmf(5, X, norm_thresh=0.01, nmf_method=method, init=init, dtype="fp64", max_iter=200,
                                    stats_out=st)
            approx = out["W"] @ (out["S"] @ out["H"] if method == "nmtf" else out["H"])
            assert np.isfinite(approx).all() and st["iters"] >= 1
            assert np.linalg.norm(dense - approx) < np.linalg.norm(dense)
    with pytest.raises(ValueError):
        manta_amd.run_nmf(5, X, init="nonsense")


def test_device_svd_is_kept_for_the_resident_matrix(golden):
    """The same matrix factorised again with the same topic count (methods compared, a run repeated): the right singular
    vectors are kept per topic count and an exact repeat costs no iteration and returns the same triplets bit for bit;
    another topic count is computed afresh (its result must not depend on earlier calls); replacing the matrix drops
    everything kept."""
    import torch
    from manta_amd import init_device as idv
    from manta_amd.engine import Engine

    g = golden("fixed_B_1000x1200_k10")
    eng = Engine(dtype="fp64")
    try:
        eng.set_matrix(g["X"])
        info = {}
        u8, s8, v8 = idv.truncated_svd_device(eng, 8, info=info)
        assert info["steps"] > 0 and len(eng._svd_cache) == 1
        info = {}
        u8b, s8b, v8b = idv.truncated_svd_device(eng, 8, info=info)
        assert info["steps"] == 0 and info.get("cached")
        assert torch.equal(s8, s8b) and torch.equal(v8, v8b) and torch.equal(u8, u8b)
        info = {}
        u5, s5, v5 = idv.truncated_svd_device(eng, 5, info=info)
        assert info["steps"] > 0 and not info.get("cached")
        fresh = Engine(dtype="fp64")
        try:
            fresh.set_matrix(g["X"])
            uf, sf, vf = idv.truncated_svd_device(fresh, 5)
        finally:
            fresh.close()
        assert torch.equal(s5, sf) and torch.equal(v5, vf)       # independent of what was asked before
        eng.set_matrix(g["X"])
        assert not eng._svd_cache
    finally:
        eng.close()
