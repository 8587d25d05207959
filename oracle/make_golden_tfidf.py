"""Generate tests/golden/tfidf_*.npz by running the UNMODIFIED reference TF-IDF builders
(manta/_functions/tfidf/tfidf_english_calculator.py, tfidf_turkish_calculator.py).

Run in the build container only (needs /root/reference):   python oracle/make_golden_tfidf.py

The reference is imported, never copied (empty parent packages are pre-registered as in make_golden.py).  Fixtures are
data only: the token-id lists (flattened), the arguments, and the reference's CSR output.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    for name, sub in (("manta", "manta"), ("manta.utils", "manta/utils"), ("manta._functions", "manta/_functions")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    import importlib
    return importlib.import_module("manta._functions.tfidf")


class VocabSize:
    """The two tokenizer methods tf_idf_turkish touches (:36, :89)."""

    def __init__(self, V):
        self.V = V

    def get_vocab_size(self):
        return self.V

    def get_vocab(self):
        return {f"t{i}": i for i in range(self.V)}


def corpus(m, V, mean_len, seed, empty_docs=(), zipf=1.2, ensure_last=True):
    rng = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, V + 1) ** zipf
    p /= p.sum()
    docs = []
    for i in range(m):
        n = 0 if i in empty_docs else max(1, rng.poisson(mean_len))
        docs.append([int(t) for t in rng.choice(V, size=n, p=p)])
    if ensure_last:                       # the reference raises IndexError when the last vocabulary id is unused
        docs[-1].append(V - 1)
    return docs


def flat(docs):
    ptr = np.zeros(len(docs) + 1, np.int64)
    ptr[1:] = np.cumsum([len(d) for d in docs])
    return np.asarray([t for d in docs for t in d], np.int32), ptr


def save(tag, docs, V, X, **args):
    tokens, ptr = flat(docs)
    X = X.tocsr()
    X.sort_indices()
    np.savez_compressed(os.path.join(OUT, f"tfidf_{tag}.npz"), tokens=tokens, doc_ptr=ptr, V=np.int64(V),
                        out_indptr=X.indptr.astype(np.int64), out_indices=X.indices.astype(np.int32),
                        out_data=X.data.astype(np.float64), **{k: np.asarray(v) for k, v in args.items()})
    print("wrote", tag, X.shape, X.nnz)


def main():
    ref = load_reference()
    quiet = contextlib.redirect_stdout(io.StringIO())
    # english: default arguments; empty documents in the middle (the reduceat quirk moves the mean length)
    docs = corpus(400, 300, 40, seed=1, empty_docs=(0, 7, 8, 250))
    with quiet:
        X = ref.tf_idf_english(N=len(docs), vocab=list(range(300)), data=docs)
    save("english_default", docs, 300, X, N=len(docs), kind="english", use_pivoted_norm=True, slope=0.2)
    # english: N differs from len(data), no pivoted norm, PageRank weights
    docs = corpus(250, 500, 25, seed=2)
    w = 1.0 + np.random.RandomState(3).rand(len(docs))
    with quiet:
        X = ref.tf_idf_english(N=1000, vocab=list(range(500)), data=docs, use_pivoted_norm=False, pagerank_weights=w)
    save("english_nopivot_pagerank", docs, 500, X, N=1000, kind="english", use_pivoted_norm=False, slope=0.2, pagerank=w)
    # english: slope 0.5 + PageRank, one very long document with heavy repetition
    docs = corpus(120, 200, 60, seed=4, empty_docs=(5,))
    docs[10] = [int(t) for t in np.random.RandomState(5).choice(20, size=3000)] + docs[10]
    w = 1.0 + np.random.RandomState(6).rand(len(docs))
    with quiet:
        X = ref.tf_idf_english(N=len(docs), vocab=list(range(200)), data=docs, slope=0.5, pagerank_weights=w)
    save("english_slope_long", docs, 200, X, N=len(docs), kind="english", use_pivoted_norm=True, slope=0.5, pagerank=w)
    # english: slope == -1 (pivoted branch entered, normalisation skipped)
    docs = corpus(60, 80, 15, seed=7)
    with quiet:
        X = ref.tf_idf_english(N=len(docs), vocab=list(range(80)), data=docs, slope=-1)
    save("english_slope_off", docs, 80, X, N=len(docs), kind="english", use_pivoted_norm=True, slope=-1)
    # turkish: idf_p goes negative for words in more than half of the documents and is exactly 0 at df = N/2
    docs = corpus(300, 150, 30, seed=8, empty_docs=(100,), zipf=1.0)
    for i in range(len(docs)):            # word 149 in exactly half of the documents: its entries are eliminated
        docs[i] = [t for t in docs[i] if t != 149]
    for i in range(0, 300, 2):
        docs[i].append(149)
    with quiet:
        X = ref.tf_idf_turkish(docs, VocabSize(150))
    save("turkish_default", docs, 150, X, kind="turkish")
    docs = corpus(200, 400, 50, seed=9)
    w = 1.0 + np.random.RandomState(10).rand(len(docs))
    with quiet:
        X = ref.tf_idf_turkish(docs, VocabSize(400), pagerank_weights=w)
    save("turkish_pagerank", docs, 400, X, kind="turkish", pagerank=w)


if __name__ == "__main__":
    main()
