"""Batched topic-count sweep (optimization_pipeline.py:105-136, 193-198): several standard-NMF
problems with sum(k) <= 32 share every pass over X.  The claim under test is the strong one: each problem's result is
BIT-IDENTICAL to its own single run_nmf call from the same start -- the batch only shares the SpMMs, whose columns are
independent, and runs the dense side per problem through the same kernels on the problem's own columns."""
import numpy as np
import pytest

from oracle import nmf_oracle as oc

pytestmark = pytest.mark.gpu


def host(t):
    return t.detach().cpu().double().numpy()


@pytest.fixture(scope="module")
def eng():
    from manta_amd.engine import Engine

    e = Engine(dtype="fp32")
    yield e
    e.close()


def starts_for(X, ks, seed):
    rng = np.random.RandomState(seed)
    m, V = X.shape
    return [(rng.uniform(0, X.max(), (m, k)), rng.uniform(0, X.max(), (k, V))) for k in ks]


@pytest.mark.parametrize("ks", [(5, 10, 15), (7, 1, 3, 2), (32,), (16, 16), (10, 20)])
def test_sweep_fixed_iterations_bit_identical_to_single_runs(eng, golden, ks):
    g = golden("fixed_B_1000x1200_k10")
    X = g["X"]
    eng.set_matrix(X)
    st = starts_for(X, ks, sum(ks))
    got = eng.run_nmf_sweep([a for a, _ in st], [b for _, b in st], fixed_iters=25, force_check=True, zero_threshold=1e-5)
    for (W0, H0), (W, H, s) in zip(st, got):
        W1, H1, s1 = eng.run_nmf(W0, H0, fixed_iters=25, force_check=True, zero_threshold=1e-5)
        assert np.array_equal(host(W), host(W1)) and np.array_equal(host(H), host(H1))
        assert s.iters == s1.iters == 25 and s.w_norm == s1.w_norm and s.h_norm == s1.h_norm
    # and every problem against the CPU oracle within the fp32 tolerance
    for (W0, H0), (W, H, s) in zip(st, got):
        want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=25, check=False)
        assert np.linalg.norm(host(W) - want["W"]) <= 1e-4 * np.linalg.norm(want["W"])
        assert np.linalg.norm(host(H) - want["H"]) <= 1e-4 * np.linalg.norm(want["H"])


def test_sweep_each_problem_stops_at_its_own_iteration(eng, golden):
    """Under the reference's stopping rule the problems of a batch converge at different iterations; each must return
    the factors, the iteration count and the norms of its own single run."""
    g = golden("converged_D_600x500_k6_nmf")
    X = g["X"]
    eng.set_matrix(X)
    ks = (5, 10, 15)
    st = starts_for(X, ks, 77)
    got = eng.run_nmf_sweep([a for a, _ in st], [b for _, b in st], norm_thresh=0.005, zero_threshold=1e-5)
    iters = []
    for (W0, H0), (W, H, s) in zip(st, got):
        W1, H1, s1 = eng.run_nmf(W0, H0, norm_thresh=0.005, zero_threshold=1e-5)
        assert s.converged and s1.converged and s.iters == s1.iters
        assert s.w_norm == s1.w_norm and s.h_norm == s1.h_norm
        assert np.array_equal(host(W), host(W1)) and np.array_equal(host(H), host(H1))
        iters.append(s.iters)
    assert len(set(iters)) > 1, iters     # otherwise the test does not exercise the staggered stop
    # an iteration cap below every stopping point: nobody converges, everybody returns the capped factors
    got = eng.run_nmf_sweep([a for a, _ in st], [b for _, b in st], norm_thresh=0.005, max_iter=3)
    for (W0, H0), (W, H, s) in zip(st, got):
        W1, H1, s1 = eng.run_nmf(W0, H0, norm_thresh=0.005, max_iter=3)
        assert not s.converged and s.iters == s1.iters == 4
        assert np.array_equal(host(W), host(W1)) and np.array_equal(host(H), host(H1))


def test_sweep_refuses_what_it_cannot_batch(eng, golden):
    from manta_amd.engine import Engine, MnmfError

    g = golden("fixed_A_300x200_k5")
    eng.set_matrix(g["X"])
    st = starts_for(g["X"], (20, 13), 1)
    with pytest.raises(MnmfError) as ei:
        eng.run_nmf_sweep([a for a, _ in st], [b for _, b in st], fixed_iters=2)
    assert ei.value.code == -1 and "at most 32" in str(ei.value)
    e64 = Engine(dtype="fp64")
    try:
        e64.set_matrix(g["X"])
        st = starts_for(g["X"], (9, 8), 2)            # fp64: 16 doubles per 128-byte row
        with pytest.raises(MnmfError) as ei:
            e64.run_nmf_sweep([a for a, _ in st], [b for _, b in st], fixed_iters=2)
        assert ei.value.code == -1 and "at most 16" in str(ei.value)
    finally:
        e64.close()


def test_run_nmf_sweep_drop_in_groups_and_falls_back(golden):
    """manta_amd.run_nmf_sweep == [run_nmf(k, X) for k in ks] (same starts), whatever the grouping: 5+10+15 go together,
    20 alone; pnmf runs one by one."""
    import manta_amd
    from manta_amd.nmf import sweep_batches

    assert sweep_batches([5, 10, 15, 20, 40, 3]) == [[0, 1, 2], [3], [4], [5]]
    g = golden("converged_D_600x500_k6_nmf")
    X = g["X"]
    ks = [5, 10, 15, 20]
    st = starts_for(X, ks, 5)
    stats = []
    got = manta_amd.run_nmf_sweep(X, ks, norm_thresh=0.005, dtype="fp32", starts=st, stats_out=stats)
    assert [s["batch"] for s in stats] == [[5, 10, 15]] * 3 + [[20]]
    for k, (W0, H0), out, s in zip(ks, st, got, stats):
        one = {}
        want = manta_amd.run_nmf(k, X, norm_thresh=0.005, dtype="fp32", W0=W0, H0=H0, stats_out=one)
        assert sorted(out) == ["H", "W"] and out["W"].dtype == np.float64
        assert np.array_equal(out["W"], want["W"]) and np.array_equal(out["H"], want["H"])
        assert s["iters"] == one["iters"]
    got = manta_amd.run_nmf_sweep(X, [4, 6], norm_thresh=0.005, nmf_method="pnmf", dtype="fp64", starts=starts_for(X, [4, 6], 6))
    assert [o["W"].shape[1] for o in got] == [4, 6]


@pytest.mark.parametrize("init", ["device", "device_random"])
def test_run_nmf_sweep_with_device_starts(golden, init):
    """``init`` as a string on the batched path: the starts come from the resident matrix (init_device), once per topic
    count, and every result equals the single run_nmf call with the same ``init`` (and ``seed``) bit for bit."""
    import manta_amd

    g = golden("converged_D_600x500_k6_nmf")
    X = g["X"]
    ks = [4, 6, 9]
    stats = []
    got = manta_amd.run_nmf_sweep(X, ks, init=init, norm_thresh=0.005, dtype="fp32", seed=3, stats_out=stats)
    assert [s["batch"] for s in stats] == [ks] * 3
    for k, out in zip(ks, got):
        want = manta_amd.run_nmf(k, X, init=init, norm_thresh=0.005, dtype="fp32", seed=3)
        assert np.array_equal(out["W"], want["W"]) and np.array_equal(out["H"], want["H"])
    with pytest.raises(ValueError):
        manta_amd.run_nmf_sweep(X, ks, init="host", dtype="fp32")


@pytest.mark.parametrize("ks", [(5, 10), (7, 1, 3, 2), (16,), (8, 8), (6, 4, 6)])
def test_sweep_fp64_bit_identical_to_single_runs(golden, ks):
    """The batched sweep in float64 (the drop-in's default dtype): topic counts that add up to at most 16 share one
    128-byte row; each result is bit-identical to its own run_nmf and within 1e-10 of the CPU oracle."""
    from manta_amd.engine import Engine

    e64 = Engine(dtype="fp64")
    try:
        g = golden("fixed_B_1000x1200_k10")
        X = g["X"]
        e64.set_matrix(X)
        st = starts_for(X, ks, 100 + sum(ks))
        got = e64.run_nmf_sweep([a for a, _ in st], [b for _, b in st], fixed_iters=20, force_check=True, zero_threshold=1e-5)
        for (W0, H0), (W, H, s) in zip(st, got):
            W1, H1, s1 = e64.run_nmf(W0, H0, fixed_iters=20, force_check=True, zero_threshold=1e-5)
            assert np.array_equal(host(W), host(W1)) and np.array_equal(host(H), host(H1))
            assert s.iters == s1.iters == 20 and s.w_norm == s1.w_norm and s.h_norm == s1.h_norm
        for (W0, H0), (W, H, s) in zip(st, got):
            want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=1e-5, n_iter=20, check=False)
            assert np.linalg.norm(host(W) - want["W"]) <= 1e-10 * np.linalg.norm(want["W"])
            assert np.linalg.norm(host(H) - want["H"]) <= 1e-10 * np.linalg.norm(want["H"])
        # staggered stopping under the reference's rule
        g2 = golden("converged_D_600x500_k6_nmf")
        e64.set_matrix(g2["X"])
        st = starts_for(g2["X"], (5, 10), 77)
        got = e64.run_nmf_sweep([a for a, _ in st], [b for _, b in st], norm_thresh=0.005, zero_threshold=1e-5)
        for (W0, H0), (W, H, s) in zip(st, got):
            W1, H1, s1 = e64.run_nmf(W0, H0, norm_thresh=0.005, zero_threshold=1e-5)
            assert s.converged and s1.converged and s.iters == s1.iters
            assert np.array_equal(host(W), host(W1)) and np.array_equal(host(H), host(H1))
    finally:
        e64.close()


def test_run_nmf_sweep_default_dtype_batches(golden):
    """run_nmf_sweep with the default dtype (fp64): 5 + 10 go together, 15 alone; equal to the single calls."""
    import manta_amd

    g = golden("converged_D_600x500_k6_nmf")
    X = g["X"]
    ks = [5, 10, 15]
    st = starts_for(X, ks, 9)
    stats = []
    got = manta_amd.run_nmf_sweep(X, ks, norm_thresh=0.005, starts=st, stats_out=stats)
    assert [s["batch"] for s in stats] == [[5, 10], [5, 10], [15]]
    for k, (W0, H0), out in zip(ks, st, got):
        want = manta_amd.run_nmf(k, X, norm_thresh=0.005, W0=W0, H0=H0)
        assert np.array_equal(out["W"], want["W"]) and np.array_equal(out["H"], want["H"])


def test_sweep_bit_identical_on_the_lds_paths(golden, monkeypatch):
    """The same claim with the LDS-staged SpMM paths forced on (hot rows of H^T, W tiles: normally only for large
    matrices): the tiles of X^T.W -- and with them the order of a word's partial sums -- must not depend on the number
    of columns in use, or a problem of the batch (30 columns side by side) would differ from its own run (5, 10, 15)."""
    from manta_amd.engine import Engine

    for k_, v in dict(MNMF_HOT_MIN_NNZ="0", MNMF_HOT_LDS_BYTES="20000", MNMF_TILE_LDS_BYTES="40000", MNMF_XT_HOT_WORDS="40",
                      MNMF_TILE_MIN_ROWS="64", MNMF_TILE_MIN_TILES_PCT="0").items():
        monkeypatch.setenv(k_, v)
    for dtype, ks in (("fp32", (5, 10, 15)), ("fp64", (3, 6, 5))):
        e = Engine(dtype=dtype)
        try:
            g = golden("fixed_B_1000x1200_k10")
            X = g["X"]
            e.set_matrix(X)
            st = starts_for(X, ks, 11)
            got = e.run_nmf_sweep([a for a, _ in st], [b for _, b in st], fixed_iters=15, force_check=True, zero_threshold=1e-5)
            for (W0, H0), (W, H, s) in zip(st, got):
                W1, H1, s1 = e.run_nmf(W0, H0, fixed_iters=15, force_check=True, zero_threshold=1e-5)
                assert np.array_equal(host(W), host(W1)) and np.array_equal(host(H), host(H1))
                assert s.w_norm == s1.w_norm and s.h_norm == s1.h_norm
        finally:
            e.close()


def test_sweeping_run_nmf_serves_a_callers_loop(golden):
    """``install(sweep_counts=...)`` / ``sweeping_run_nmf``: MANTA's optimisation pipeline calls run_nmf once per topic
    count with keywords (optimization_pipeline.py:193-198); the stand-in factorises the announced counts in one batch at
    the first call and answers the others from it -- every answer bit-identical to the plain call."""
    import manta_amd

    g = golden("converged_D_600x500_k6_nmf")
    X = g["X"]
    ks = [4, 6, 5]
    starts = {k: s for k, s in zip(ks, starts_for(X, ks, 21))}

    def init(csc, k):                 # deterministic per count, as nmf_initialization_nndsvd is
        return starts[k][0].copy(), starts[k][1].copy()

    f = manta_amd.sweeping_run_nmf(ks)
    outs = [f(num_of_topics=k, sparse_matrix=X, init=init, norm_thresh=0.005, nmf_method="nmf") for k in ks]
    assert sorted(f.stats) == sorted(ks) and all(f.stats[k]["batch"] == ks for k in ks)     # one batch (4 + 6 + 5 <= 16)
    for k, out in zip(ks, outs):
        want = manta_amd.run_nmf(k, X, init=init, norm_thresh=0.005)
        assert np.array_equal(out["W"], want["W"]) and np.array_equal(out["H"], want["H"])
    # a second round over the same counts is a second batch; other methods and unannounced counts go straight through
    again = f(num_of_topics=6, sparse_matrix=X, init=init, norm_thresh=0.005, nmf_method="nmf")
    assert np.array_equal(again["W"], outs[1]["W"])
    other = f(num_of_topics=3, sparse_matrix=X, init=lambda c, k: (starts[4][0][:, :3].copy(), starts[4][1][:3].copy()), norm_thresh=0.005)
    assert other["W"].shape[1] == 3
    p = f(4, X, init, 0.005, 0.00001, "pnmf")
    assert sorted(p) == ["H", "W"] and p["W"].shape[1] == 4
    with pytest.raises(ValueError):
        manta_amd.sweeping_run_nmf([])
