"""The CPU oracle must reproduce the reference's own outputs (tests/golden, made by
oracle/make_golden.py from the unmodified reference) bit for bit in float64."""
import numpy as np
import pytest

from oracle import nmf_oracle as oc

FIXED = ["fixed_A_300x200_k5", "fixed_C_500x400_k7", "fixed_B_1000x1200_k10"]


def _eq(a, b):
    assert a.shape == b.shape
    assert np.array_equal(a, b), f"max abs diff {np.max(np.abs(a - b))}"


@pytest.mark.parametrize("name", FIXED)
def test_loop_a_bit_exact(golden, name):
    g = golden(name)
    for N in g["Ns"]:
        r = oc.basic_nmf(g["X"], g["W0"].copy(), g["H0"].copy(), zero_threshold=1e-5, n_iter=int(N))
        _eq(r["W"], g[f"nmf_W_{N}"])
        _eq(r["H"], g[f"nmf_H_{N}"])
    N = int(g["Ns"][-1])
    r = oc.basic_nmf(g["X"], g["Wr0"].copy(), g["Hr0"].copy(), zero_threshold=1e-5, n_iter=N, check=False)
    _eq(r["W"], g["nmfr_W"])
    _eq(r["H"], g["nmfr_H"])


@pytest.mark.parametrize("name", FIXED)
def test_loop_b_bit_exact(golden, name):
    g = golden(name)
    for N in g["Ns"]:
        r = oc.projective_nmf(g["X"], g["W0"].copy(), n_iter=int(N), check=False)
        _eq(r["W"], g[f"pnmf_W_{N}"])
        _eq(r["H"], g[f"pnmf_H_{N}"])


@pytest.mark.parametrize("name", FIXED)
def test_loop_c_bit_exact(golden, name):
    g = golden(name)
    for N in g["Ns"]:
        r = oc.nmtf_core(g["X"], g["Wt0"].copy(), g["St0"].copy(), g["Ht0"].copy(),
                         zero_threshold=1e-5, n_iter=int(N), check=False)
        _eq(r["W"], g[f"nmtf_W_{N}"])
        _eq(r["S"], g[f"nmtf_S_{N}"])
        _eq(r["H"], g[f"nmtf_H_{N}"])
    N = int(g["Ns"][-1])
    r = oc.nmtf_core(g["X"], g["Wr0"].copy(), g["Sr0"].copy(), g["Hr0"].copy(),
                     zero_threshold=1e-5, n_iter=N, check=False)
    _eq(r["W"], g["nmtfr_W"])
    _eq(r["S"], g["nmtfr_S"])
    _eq(r["H"], g["nmtfr_H"])


@pytest.mark.parametrize("method,keys", [("nmf", "WH"), ("pnmf", "WH"), ("nmtf", "WSH")])
def test_converged_run_nmf(golden, method, keys):
    """Full dispatcher incl. NNDSVD init from the same numpy seed, convergence test,
    iteration count and (pnmf) the break-before-assign rule."""
    g = golden(f"converged_D_600x500_k6_{method}")
    np.random.seed(int(g["seed"]))
    stats = {}
    r = oc.run_nmf(int(g["k"]), g["X"], norm_thresh=0.005, nmf_method=method, stats=stats)
    assert sorted(r.keys()) == sorted(keys)
    for key in keys:
        _eq(r[key], g["out_" + key])
    iters = int(g["iters"])
    if method == "pnmf":
        # fixture counts updates tried; the oracle counts updates applied
        assert stats["iters"] == iters - 1
    else:
        assert stats["iters"] == iters


def test_init_bit_exact(golden):
    g = golden("converged_D_600x500_k6_nmtf")
    np.random.seed(int(g["seed"]))
    w, h = oc.nndsvd_init(g["X"].tocsc(), int(g["k"]))
    _eq(w, g["W0"])
    _eq(h, g["H0"])
    wt, st, ht = oc.nmtf_nndsvd_legacy_init(g["X"].tocsr(), int(g["k"]))
    _eq(wt, g["Wt0"])
    _eq(st, g["St0"])
    _eq(ht, g["Ht0"])
    # NNDSVD leaves exact zeros that MU must keep
    assert (w == 0).mean() > 0.2


def test_autorank_and_errors(golden):
    g = golden("autorank_D")
    assert oc.find_max_rank(g["X"].tocsc()) == int(g["rank"])
    with pytest.raises(ValueError):
        oc.find_max_rank(g["X"].tocsr())
    with pytest.raises(ValueError, match="Unknown NMF method"):
        oc.run_nmf(3, g["X"], init=oc.random_init, nmf_method="bogus")


def test_rankings_rule():
    H = np.array([[0.1, 0.9, 0.5], [0.7, 0.2, 0.3]])
    W = np.array([[1.0, 0.0], [0.5, 2.0], [0.2, 0.1]])
    words, docs = oc.topic_rankings(H, W, top_n=3)
    assert words.tolist() == [[1, 2, 0], [0, 2, 1]]
    assert docs.tolist() == [[0, 1, 2], [1, 2, 0]]
    S = np.array([[0.1, 0.9], [0.8, 0.2]])
    words, _ = oc.topic_rankings(H, W, S=S, top_n=3)
    assert words.tolist() == [[0, 2, 1], [1, 2, 0]]


# ---------------------------------------------------------------------------------------------------------------
# TF-IDF builders: the numpy restatement against outputs of the reference's own functions.
TFIDF_CASES = ["english_default", "english_nopivot_pagerank", "english_slope_long", "english_slope_off",
               "turkish_default", "turkish_pagerank"]


def load_tfidf_case(tag):
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"tfidf_{tag}.npz"), allow_pickle=False)
    docs = [z["tokens"][z["doc_ptr"][i]:z["doc_ptr"][i + 1]].tolist() for i in range(len(z["doc_ptr"]) - 1)]
    return z, docs


@pytest.mark.parametrize("tag", TFIDF_CASES)
def test_tfidf_oracle_reproduces_reference_bit_for_bit(tag):
    from oracle import tfidf_oracle as to

    z, docs = load_tfidf_case(tag)
    V = int(z["V"])
    w = z["pagerank"] if "pagerank" in z.files else None
    if str(z["kind"]) == "english":
        X = to.tf_idf_english(int(z["N"]), V, docs, use_pivoted_norm=bool(z["use_pivoted_norm"]), slope=float(z["slope"]),
                              pagerank_weights=w)
    else:
        X = to.tf_idf_turkish(docs, V, pagerank_weights=w)
    assert np.array_equal(X.indptr, z["out_indptr"]) and np.array_equal(X.indices, z["out_indices"])
    assert np.array_equal(X.data, z["out_data"])                     # bit-exact float64


def test_tfidf_oracle_reduceat_errors():
    from oracle import tfidf_oracle as to

    with pytest.raises(IndexError):                                   # unused last vocabulary id
        to.tf_idf_english(3, 5, [[0, 1], [2], [3]])
    with pytest.raises(IndexError):                                   # empty last document
        to.tf_idf_english(3, 4, [[0, 1], [2, 3], []])


def test_s_matrix_confusion_oracle_reproduces_reference_lines():
    """oracle.s_matrix_confusion against the debug lines the unmodified reference function emitted
    (oracle/make_golden_sconf.py; nmtf/nmtf.py:62-80): same couplings, same order, same printed strengths."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sconf.npz"), allow_pickle=False)
    for name in z["names"]:
        S = z[f"S_{name}"]
        cols, rows, vals, lines = oc.s_matrix_confusion(S)
        assert lines == z[f"lines_{name}"].tolist()
        assert np.array_equal(vals, S[rows, cols]) and np.all(np.diff(vals) <= 0)
        assert sorted(cols.tolist()) == list(range(S.shape[1]))
