"""TF-IDF construction on the GPU (manta_amd/tfidf.py + csrc/tfidf.hip) through the C-ABI.

Bar: bit-exact.  Structure (indptr / indices) and the float64 values must equal the outputs of the reference's own
``tf_idf_english`` / ``tf_idf_turkish`` (tests/golden/tfidf_*.npz, produced by oracle/make_golden_tfidf.py) and, on
larger seeded corpora, the CPU oracle (oracle/tfidf_oracle.py, itself pinned to those fixtures)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import tfidf_oracle as to

pytestmark = pytest.mark.gpu

CASES = ["english_default", "english_nopivot_pagerank", "english_slope_long", "english_slope_off",
         "turkish_default", "turkish_pagerank"]


class VocabSize:
    def __init__(self, V):
        self.V = V

    def get_vocab_size(self):
        return self.V


def load_case(tag):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"tfidf_{tag}.npz"), allow_pickle=False)
    docs = [z["tokens"][z["doc_ptr"][i]:z["doc_ptr"][i + 1]].tolist() for i in range(len(z["doc_ptr"]) - 1)]
    return z, docs


def same_csr(X, indptr, indices, data):
    X = X.tocsr()
    assert X.has_sorted_indices or True
    assert np.array_equal(X.indptr, indptr) and np.array_equal(X.indices, indices)
    assert X.data.dtype == np.float64 and np.array_equal(X.data, data)


@pytest.mark.parametrize("tag", CASES)
def test_drop_in_functions_reproduce_reference_outputs_bit_for_bit(tag):
    from manta_amd import tfidf

    z, docs = load_case(tag)
    V = int(z["V"])
    w = z["pagerank"] if "pagerank" in z.files else None
    if str(z["kind"]) == "english":
        X = tfidf.tf_idf_english(N=int(z["N"]), vocab=list(range(V)), data=docs, use_pivoted_norm=bool(z["use_pivoted_norm"]),
                                 slope=float(z["slope"]), pagerank_weights=w)
    else:
        X = tfidf.tf_idf_turkish(docs, VocabSize(V), pagerank_weights=w)
    assert sp.isspmatrix_csr(X) and X.shape == (len(docs), V)
    same_csr(X, z["out_indptr"], z["out_indices"], z["out_data"])


def test_error_behaviour_matches_reference():
    from manta_amd import tfidf

    with pytest.raises(IndexError):                                   # unused last vocabulary id (reduceat, :92)
        tfidf.tf_idf_english(N=3, vocab=list(range(5)), data=[[0, 1], [2], [3]])
    with pytest.raises(IndexError):                                   # empty last document (reduceat, :107)
        tfidf.tf_idf_english(N=3, vocab=list(range(4)), data=[[0, 1], [2, 3], []])
    with pytest.raises(IndexError):                                   # token beyond the vocabulary
        tfidf.tf_idf_english(N=2, vocab=list(range(4)), data=[[0, 9], [3]])
    with pytest.raises(NotImplementedError):
        tfidf.tf_idf_english(N=2, vocab=list(range(4)), data=[[0], [3]], use_bm25=True)
    # an empty last document is fine without the pivoted norm (no reduceat over the rows)
    X = tfidf.tf_idf_english(N=3, vocab=list(range(4)), data=[[0, 1], [2, 3, 3], []], use_pivoted_norm=False)
    want = to.tf_idf_english(3, 4, [[0, 1], [2, 3, 3], []], use_pivoted_norm=False)
    same_csr(X, want.indptr, want.indices, want.data)


def seeded_corpus(m, V, mean_len, seed, long_docs=()):
    rng = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, V + 1) ** 1.1
    p /= p.sum()
    lens = np.maximum(1, rng.poisson(mean_len, size=m))
    lens[rng.rand(m) < 0.01] = 0                                      # empty documents in the middle
    for i, n in long_docs:
        lens[i] = n
    lens[-1] = max(lens[-1], 1)
    tokens = rng.choice(V, size=int(lens.sum()), p=p).astype(np.int32)
    ptr = np.zeros(m + 1, np.int64)
    ptr[1:] = np.cumsum(lens)
    tokens[-1] = V - 1                                                # the last vocabulary id is used
    return tokens, ptr


@pytest.mark.parametrize("kind", ["english", "turkish"])
def test_mixed_document_lengths_against_oracle(kind):
    """Documents of 0, 1, ~80, 1000-1024 (one wave), 1025-8192 (four waves) and > 8192 tokens (global-sort path)."""
    import torch
    from manta_amd import tfidf

    V = 3000
    long_docs = [(3, 1), (10, 1000), (11, 1024), (12, 1025), (13, 4096), (14, 8192), (15, 2), (16, 5000)]
    tokens, ptr = seeded_corpus(6000, V, 80, seed=21, long_docs=long_docs)
    docs = [tokens[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]
    w = 1.0 + np.random.RandomState(1).rand(len(docs))
    dev = torch.device("cuda", 0)
    for extra in (False, True):
        if extra:                                                     # one document beyond the LDS kernel's limit
            tokens = np.concatenate([np.random.RandomState(2).randint(0, V, 9000).astype(np.int32), tokens])
            ptr = np.concatenate([[0], ptr + 9000])
            docs = [tokens[ptr[i]:ptr[i + 1]] for i in range(len(ptr) - 1)]
            w = np.concatenate([[1.5], w])
        if kind == "english":
            want = to.tf_idf_english(len(docs) + 5, V, docs, slope=0.3, pagerank_weights=w)
            got = tfidf.tfidf_from_tokens(torch.from_numpy(tokens).to(dev), torch.from_numpy(ptr).to(dev), V, idf="t",
                                          N=len(docs) + 5, slope=0.3, pagerank_weights=w)
        else:
            want = to.tf_idf_turkish(docs, V, pagerank_weights=w)
            got = tfidf.tfidf_from_tokens(torch.from_numpy(tokens).to(dev), torch.from_numpy(ptr).to(dev), V, idf="p",
                                          N=len(docs), pagerank_weights=w)
        same_csr(tfidf._to_scipy(got), want.indptr, want.indices, want.data)


def test_lds_kernel_equals_global_sort_at_scale_and_feeds_the_engine():
    """300k documents x 50k words, ~100 tokens each (3e7 tokens): the per-document LDS sort against one global radix
    sort of (document, token) keys -- integer outputs must be identical -- then the device CSR goes straight into
    Engine.set_matrix and one product is checked against scipy on the downloaded matrix."""
    import torch
    from manta_amd import tfidf
    from manta_amd.engine import Engine

    dev = torch.device("cuda", 0)
    m, V = 300_000, 50_000
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    lens = torch.randint(1, 200, (m,), generator=g, device=dev)
    ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(lens, 0)
    u = torch.rand(int(ptr[-1]), generator=g, device=dev, dtype=torch.float64)
    tokens = (V * u ** 3).long().clamp(max=V - 1).to(torch.int32)    # skewed towards small ids
    tokens[-1] = V - 1
    eng = Engine(dtype="fp64")
    try:
        a = tfidf.term_counts_device(tokens, ptr, V, eng)
        b = tfidf._term_counts_global_sort(tokens, ptr, V)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert int(a[2].sum()) == tokens.numel()                      # counts add up to the corpus
        d = tfidf.tfidf_from_tokens(tokens, ptr, V, idf="t", N=m, eng=eng)
        eng.set_matrix(**d)
        B = torch.rand((V, 8), generator=g, device=dev, dtype=torch.float64)
        X = tfidf._to_scipy(d)
        got = eng.spmm(B).cpu().numpy()
        want = X @ B.cpu().numpy()
        assert np.linalg.norm(got - want) <= 1e-12 * np.linalg.norm(want)
    finally:
        eng.close()


def test_load_npz_device(tmp_path):
    import torch
    from manta_amd import tfidf

    z, docs = load_case("english_default")
    X = sp.csr_matrix((z["out_data"], z["out_indices"], z["out_indptr"]), shape=(len(docs), int(z["V"])))
    for fmt, M in (("csr", X), ("csc", X.tocsc())):
        path = str(tmp_path / f"x_{fmt}.npz")
        sp.save_npz(path, M)
        d = tfidf.load_npz_device(path, device=torch.device("cuda", 0))
        assert d["shape"] == X.shape and d["colidx"].dtype == torch.int32
        same_csr(tfidf._to_scipy(d), X.indptr, X.indices, X.data)
