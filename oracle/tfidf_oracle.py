"""CPU oracle for the TF-IDF builders -- TEST INFRASTRUCTURE ONLY (imported by tests/ and oracle/ scripts, never by
manta_amd).  A numpy / scipy restatement of

  tf_idf_english   manta/_functions/tfidf/tfidf_english_calculator.py:76-146
  tf_idf_turkish   manta/_functions/tfidf/tfidf_turkish_calculator.py:34-105
  tf_l, tf_L       manta/_functions/tfidf/tfidf_tf_functions.py:30-50
  idf_t, idf_p     manta/_functions/tfidf/tfidf_idf_functions.py:11-15

Pinned against the reference itself: tests/golden/tfidf_*.npz hold the reference's outputs (oracle/make_golden_tfidf.py)
and tests/test_oracle_golden.py requires this restatement to reproduce them bit for bit.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def count_matrix(docs, V):
    """The lil_matrix filled from a Counter per document (english :80-87, turkish :37-45), as canonical CSR (int)."""
    indptr, indices, data = [0], [], []
    for d in docs:
        ids, cnt = np.unique(np.asarray(d, dtype=np.int64), return_counts=True)
        if ids.size and (ids[0] < 0 or ids[-1] >= V):
            raise IndexError("column index out of bounds")
        indices.append(ids)
        data.append(cnt)
        indptr.append(indptr[-1] + ids.size)
    indices = np.concatenate(indices) if indices else np.zeros(0, np.int64)
    data = np.concatenate(data) if data else np.zeros(0, np.int64)
    return sp.csr_matrix((data.astype(np.int64), indices.astype(np.int32), np.asarray(indptr, np.int64)), shape=(len(docs), V))


def _tfidf(docs, V, idf_kind, N, use_pivoted_norm, slope, pagerank_weights):
    C = count_matrix(docs, V)
    binary = C.tocsc(copy=True)
    binary.data = np.ones_like(binary.data)
    df = np.add.reduceat(binary.data, binary.indptr[:-1])                       # english :92, turkish :49
    if idf_kind == "t":
        idf = np.log2((1 + N) / df)                                             # idf_t
        t = binary.copy()
        t.data = 1 + np.log2(t.data)                                            # tf_l
    else:
        idf = np.log2((N - df + 1) / (df + 1))                                  # idf_p
        t = binary.copy()                                                       # tf_L (on the CSC copy: "rows" = columns)
        sums = np.add.reduceat(t.data, t.indptr[:-1])
        cnts = t.indptr[1:] - t.indptr[:-1]
        payda = np.repeat(1 + np.log2((1 + sums) / (1 + cnts)), cnts)
        t.data = (1 + np.log2(t.data)) / payda
    X = t.multiply(idf).tocsr()
    X.eliminate_zeros()
    if use_pivoted_norm:
        Cr = C.tocsr()
        doc_lengths = np.add.reduceat(Cr.data, Cr.indptr[:-1])                  # english :107, turkish :64
        avg = np.mean(doc_lengths)
        if slope != -1:
            piv = (1 - slope) + slope * (doc_lengths / avg)
            X.data = X.data / np.repeat(piv, np.diff(X.indptr))
    if pagerank_weights is not None:
        X.data = X.data * np.repeat(pagerank_weights, np.diff(X.indptr))
    X.sort_indices()
    return X


def tf_idf_english(N, V, docs, use_pivoted_norm=True, slope=0.2, pagerank_weights=None):
    return _tfidf(docs, V, "t", N, use_pivoted_norm, slope, pagerank_weights)


def tf_idf_turkish(docs, V, pagerank_weights=None):
    return _tfidf(docs, V, "p", len(docs), True, 0.2, pagerank_weights)
