"""Device-side topic rankings (mnmf_topk) against the oracle's ranking rule
(topic_extractor.py:7-23, 265-274, 294-300: ``np.flip(np.argsort(H[i, :]))``, ``W[:, i]``, NMTF's argmax(S) mapping).
Integer results: identical wherever the scores are distinct; equal scores rank the larger index first (= flipping a
stable ascending argsort), which numpy's default sort does not promise, so ties are compared against that rule."""
import numpy as np
import pytest

from oracle import nmf_oracle as oc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines():
    from manta_amd.engine import Engine

    e = {"fp32": Engine(dtype="fp32"), "fp64": Engine(dtype="fp64")}
    yield e
    for x in e.values():
        x.close()


def stable_rank(x, n):
    return np.flip(np.argsort(x, kind="stable"))[:n]


@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_topk_rows_and_columns(engines, dtype):
    eng = engines[dtype]
    rng = np.random.RandomState(3)
    A = rng.rand(7, 5000).astype(np.float32 if dtype == "fp32" else np.float64)
    A[2, 100:140] = 0.75                      # a run of equal values
    A[3, :] = 0.0                             # a whole row of ties (NNDSVD zeros)
    A[4, 17] = np.nan
    idx, val = eng.topk(A, 12)
    idx, val = idx.cpu().numpy(), val.cpu().numpy()
    for r in range(7):
        row = np.where(np.isnan(A[r]), -np.inf, A[r])
        want = stable_rank(row, 12)
        assert np.array_equal(idx[r], want), r
        assert np.array_equal(val[r], row[want])
    B = np.ascontiguousarray(A.T)             # the same data ranked by column (W[:, i])
    idx2, _ = eng.topk(B, 12, by_column=True)
    assert np.array_equal(idx2.cpu().numpy(), idx)
    short, sval = eng.topk(A[:, :5], 8)       # fewer candidates than requested: padded with -1
    assert np.all(short.cpu().numpy()[:, 5:] == -1) and np.all(np.sort(short.cpu().numpy()[0, :5]) == np.arange(5))


@pytest.mark.parametrize("method", ["nmf", "nmtf"])
def test_topic_rankings_match_oracle(engines, golden, method):
    g = golden(f"converged_D_600x500_k6_{method}")
    W, H = g["out_W"], g["out_H"]
    S = g["out_S"] if method == "nmtf" else None
    words_ref, docs_ref = oc.topic_rankings(H, W, S, top_n=15)
    words, docs = engines["fp64"].topic_rankings(W, H, S, top_n=15)
    words, docs = words.cpu().numpy(), docs.cpu().numpy()
    k = W.shape[1]
    for t in range(k):
        row = int(np.argmax(S[t, :])) if S is not None else t
        for got, ref, scores in ((words[t], words_ref[t], H[row, :]), (docs[t], docs_ref[t], W[:, t])):
            same = got == ref
            # any difference must be a tie in the reference's scores (numpy's default argsort is not stable)
            assert np.all(scores[got[~same]] == scores[ref[~same]])
            assert np.array_equal(got, stable_rank(scores, 15))


@pytest.mark.parametrize("method", ["nmf", "nmtf"])
@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_relevance_rankings_match_oracle(engines, golden, method, dtype):
    """Relevance re-ranking of the coherence step (gensim_coherence.py:40-140) on the device against the numpy
    restatement (parity unpinned: gensim is absent, the reference function cannot be run here).  Word ids identical
    wherever adjacent relevance values differ by more than 1e-9; values within 1e-9."""
    g = golden(f"converged_D_600x500_k6_{method}")
    W, H = g["out_W"], g["out_H"]
    S = g["out_S"] if method == "nmtf" else None
    for lam in (0.6, 0.0, 1.0):
        idx_ref, val_ref = oc.relevance_rankings(H, W, S, lambda_val=lam, top_n=20)
        idx, val = engines[dtype].relevance_rankings(W, H, S, lambda_val=lam, top_n=20)
        idx, val = idx.cpu().numpy(), val.cpu().numpy()
        assert idx.shape == idx_ref.shape
        valid = idx_ref >= 0
        assert np.array_equal(idx >= 0, valid)
        assert np.allclose(val[valid], val_ref[valid], rtol=0, atol=1e-9)
        diff = (idx != idx_ref) & valid
        for t, j in zip(*np.nonzero(diff)):                      # only near-ties may swap
            assert abs(val_ref[t, j] - val[t, j]) <= 1e-9
    # a topic row with fewer valid words than top_n
    H2 = H.copy()
    H2[0, 5:] = 0.0
    idx, val = engines[dtype].relevance_rankings(W, H2, None if S is None else S, top_n=10)
    idx_ref, _ = oc.relevance_rankings(H2, W, S, top_n=10)
    assert np.array_equal(idx.cpu().numpy() >= 0, idx_ref >= 0)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
def test_s_matrix_confusion_matches_reference_lines(dtype):
    """Engine.s_matrix_confusion (mnmf_s_confusion) against the lines recorded from the reference's s_matrix_confusion
    (tests/golden/sconf.npz, nmtf/nmtf.py:62-80).  fp64: the very strings; fp32: the same couplings in the same order."""
    import os
    from manta_amd.engine import Engine, confusion_lines

    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sconf.npz"), allow_pickle=False)
    eng = Engine(device=0, dtype=dtype)

    class Rec:
        def __init__(self):
            self.lines = []

        def print_debug(self, msg, tag=None):
            assert tag == "S-MATRIX"
            self.lines.append(msg)

    for name in z["names"]:
        S = z[f"S_{name}"]
        rec = Rec()
        cols, rows, vals = eng.s_matrix_confusion(S, console=rec)
        o_cols, o_rows, o_vals, o_lines = oc.s_matrix_confusion(S)
        assert np.array_equal(cols, o_cols) and np.array_equal(rows, o_rows)
        if dtype == "fp64":
            assert np.array_equal(vals, o_vals)
            assert rec.lines == z[f"lines_{name}"].tolist() == o_lines
        else:
            assert np.allclose(vals, o_vals, rtol=1e-6)
            assert len(rec.lines) == S.shape[1] and rec.lines == confusion_lines(rows, vals)
    eng.close()


@pytest.mark.gpu
def test_s_matrix_confusion_nan_and_tied_columns():
    """A NaN in S (np.argmax takes the first NaN of a column, np.argsort ranks NaN above every number) and columns with
    equal maxima: every rank is written exactly once and the order is numpy's (nmtf.py:69, 73).  Ties between equal
    values are not pinned by the reference (np.argsort is not stable): checked as sets there."""
    from manta_amd.engine import Engine

    eng = Engine(device=0, dtype="fp64")
    rng = np.random.default_rng(3)
    S = rng.random((7, 7))
    S[2, 1] = np.nan                 # column 1: best value NaN, row 2
    S[5, 1] = np.nan
    S[4, 6] = np.nan                 # a second NaN column
    S[:, 3] = S[:, 0]                # tied columns 0 and 3
    cols, rows, vals = eng.s_matrix_confusion(S)
    o_cols, o_rows, o_vals, _ = oc.s_matrix_confusion(S)
    assert sorted(cols.tolist()) == list(range(7))           # a permutation: no slot written twice, none left out
    assert set(cols[:2].tolist()) == {1, 6} and np.isnan(vals[:2]).all()
    assert rows[list(cols).index(1)] == 2 and rows[list(cols).index(6)] == 4
    assert np.array_equal(vals[2:], o_vals[2:])
    assert np.array_equal(rows[2:][np.argsort(cols[2:])], o_rows[2:][np.argsort(o_cols[2:])])
    tied = [i for i, c in enumerate(cols) if c in (0, 3)]
    assert tied[1] == tied[0] + 1 and vals[tied[0]] == vals[tied[1]]
    eng.close()
