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
