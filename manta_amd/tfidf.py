"""TF-IDF construction on the GPU: the producer of X, right before the solver path.

Same functions as ``manta._functions.tfidf`` (``tf_idf_english`` tfidf_english_calculator.py:11-146,
``tf_idf_turkish`` tfidf_turkish_calculator.py:11-105): tokenised documents in, a float64 CSR TF-IDF matrix out.
What the reference computes, read off its code (not off its docstrings):

* the count matrix C (documents x vocabulary) of the token ids, canonical CSR (:80-87 / :37-45);
* df = column counts of the binary copy of C, through ``np.add.reduceat`` (:90-92 / :47-49);
* every stored entry gets the value ``idf(df[word])`` -- the term-frequency factor is applied to the *binary* copy
  and is exactly 1.0 for both variants (``tf_l``: 1 + log2(1); ``tf_L``: (1 + log2(1)) / (1 + log2(1.0))) --
  english: ``idf_t`` = log2((1 + N) / df), turkish: ``idf_p`` = log2((N - df + 1) / (df + 1));
  entries whose value is exactly 0 are dropped (``eliminate_zeros``), negative ones stay;
* pivoted normalisation: value / ((1 - slope) + slope * len(doc) / mean(len)), len = tokens of the document, again
  through ``np.add.reduceat`` (:104-117 / :62-72);
* optional PageRank weights: value * weight[doc] (:120-127 / :74-80).

``np.add.reduceat`` quirks kept because they change results: an empty document in the middle is given the first
stored count of the next document as its length (which shifts the mean), and an empty LAST document or an unused
LAST vocabulary id raises ``IndexError`` exactly like the reference.  BM25 is not offered: the reference's own BM25
branch cannot run (tfidf_bm25_turkish.py:56 uses an undefined name; the turkish function overrides the flag, :51).

The integer work (sorting and run-length encoding each document's tokens, the df histogram) and the per-entry scaling
run in libmnmf.so (csrc/tfidf.hip); idf and the pivoted norms are V- and m-element numpy expressions evaluated on the
host with the reference's own formulas, so the values are bit-identical to the reference's (tests/test_gpu_tfidf.py).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import scipy.sparse as sp
import torch

MAX_DOC_TOKENS = 8192   # csrc/tfidf.hip: longer documents go through the global-sort path below


def flatten_documents(data, device=None):
    """list of token-id lists -> (tokens int32, doc_ptr int64) tensors on ``device``."""
    lens = np.fromiter((len(d) for d in data), dtype=np.int64, count=len(data))
    doc_ptr = np.zeros(len(data) + 1, dtype=np.int64)
    np.cumsum(lens, out=doc_ptr[1:])
    tokens = np.empty(int(doc_ptr[-1]), dtype=np.int64)
    for i, d in enumerate(data):
        if lens[i]:
            tokens[doc_ptr[i]:doc_ptr[i + 1]] = d
    if tokens.size and (tokens.min() < -2 ** 31 or tokens.max() >= 2 ** 31):
        raise IndexError("token id out of range")
    return (torch.from_numpy(tokens.astype(np.int32)).to(device), torch.from_numpy(doc_ptr).to(device))


def _engine_for(device):
    from .nmf import _engine
    return _engine("fp64", device)


def _term_counts_global_sort(tokens, doc_ptr, V):
    """Count matrix through one global sort of (document, token) keys: the fallback for documents longer than
    MAX_DOC_TOKENS, and the independent check of the LDS kernel in the tests."""
    m = doc_ptr.numel() - 1
    lens = doc_ptr[1:] - doc_ptr[:-1]
    if tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= V):
        raise IndexError("token id outside [0, V)")
    doc = torch.repeat_interleave(torch.arange(m, device=tokens.device), lens)
    key, cnt = torch.unique_consecutive(torch.sort(doc * V + tokens.long()).values, return_counts=True)
    rows = torch.div(key, V, rounding_mode="floor")
    colidx = (key - rows * V).to(torch.int32)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=tokens.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=m), 0)
    df = torch.bincount(colidx.long(), minlength=V).to(torch.int32)
    return rowptr, colidx, cnt.to(torch.int32), df


def term_counts_device(tokens, doc_ptr, V: int, eng=None):
    """(rowptr int64[m+1], colidx int32[nnz] ascending per row, counts int32[nnz], df int32[V]) of the count matrix,
    all on the device (tfidf_english_calculator.py:80-92)."""
    eng = eng or _engine_for(tokens.device)
    tokens = tokens.to(device=eng.device, dtype=torch.int32).contiguous()
    doc_ptr = doc_ptr.to(device=eng.device, dtype=torch.int64).contiguous()
    m = int(doc_ptr.numel()) - 1
    if m < 0 or int(doc_ptr[0]) != 0 or int(doc_ptr[-1]) != tokens.numel():
        raise ValueError("doc_ptr does not describe tokens")
    if m and int((doc_ptr[1:] - doc_ptr[:-1]).max()) > MAX_DOC_TOKENS:
        return _term_counts_global_sort(tokens, doc_ptr, V)
    lib, h = eng._lib, eng._h
    row_nnz = torch.zeros(max(m, 1), dtype=torch.int64, device=eng.device)
    ws_cols = torch.empty(max(tokens.numel(), 1), dtype=torch.int32, device=eng.device)     # encoded rows at token offsets
    ws_counts = torch.empty(max(tokens.numel(), 1), dtype=torch.int32, device=eng.device)
    rc = lib.mnmf_tfidf_count_terms(h, m, V, doc_ptr.data_ptr(), tokens.data_ptr(), row_nnz.data_ptr(),
                                    ws_cols.data_ptr(), ws_counts.data_ptr())
    if rc != 0:
        msg = lib.mnmf_last_error(h).decode()
        if "token id" in msg:
            raise IndexError(msg)                        # the reference's lil assignment raises IndexError too
        eng._check(rc)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=eng.device)
    if m:
        torch.cumsum(row_nnz[:m], 0, out=rowptr[1:])
    nnz = int(rowptr[-1])
    colidx = torch.empty(nnz, dtype=torch.int32, device=eng.device)
    counts = torch.empty(nnz, dtype=torch.int32, device=eng.device)
    df = torch.zeros(V, dtype=torch.int32, device=eng.device)
    eng._check(lib.mnmf_tfidf_fill_terms(h, m, V, doc_ptr.data_ptr(), rowptr.data_ptr(), nnz, ws_cols.data_ptr(),
                                         ws_counts.data_ptr(), colidx.data_ptr(), counts.data_ptr(), df.data_ptr()))
    return rowptr, colidx, counts, df


def _reduceat_lengths(doc_ptr, rowptr, counts):
    """``np.add.reduceat(matris.data, matris.indptr[:-1])`` (tfidf_english_calculator.py:107): the token count of each
    document, except that an empty document gets the first stored count of the next one, and IndexError when the last
    document is empty."""
    lens = (doc_ptr[1:] - doc_ptr[:-1]).cpu().numpy()
    nnz = int(counts.numel())
    empty = np.flatnonzero(lens == 0)
    if empty.size:
        at = rowptr[:-1].cpu().numpy()[empty]
        if nnz == 0 or at.max() >= nnz:
            raise IndexError(f"index {nnz} out-of-bounds in add.reduceat [0, {nnz})")
        lens = lens.copy()
        lens[empty] = counts[torch.from_numpy(at).to(counts.device)].cpu().numpy()
    return lens


def tfidf_from_tokens(tokens, doc_ptr, V: int, *, idf: str, N, use_pivoted_norm=True, slope=0.2,
                      pagerank_weights=None, eng=None):
    """Device TF-IDF of a flattened corpus.  ``idf``: "t" (english, idf_t with the caller's N) or "p" (turkish, idf_p).
    Returns a dict of device tensors ``rowptr, colidx, vals`` (float64) and ``shape`` that ``Engine.set_matrix``
    accepts as keywords."""
    eng = eng or _engine_for(tokens.device)
    rowptr, colidx, counts, df = term_counts_device(tokens, doc_ptr, V, eng)
    m, nnz = int(rowptr.numel()) - 1, int(colidx.numel())
    df_h = df.cpu().numpy().astype(np.int64)
    if m > 0 and (nnz == 0 or df_h[-1] == 0):                       # reduceat over the CSC pointer (:92 / :49)
        raise IndexError(f"index {nnz} out-of-bounds in add.reduceat [0, {nnz})")
    df_h[df_h == 0] = 1                                             # reduceat's value for an empty column; never used
    with np.errstate(divide="ignore", invalid="ignore"):
        if idf == "t":
            idf_h = np.log2((1 + N) / df_h)                         # idf_t, tfidf_idf_functions.py:11-12
        elif idf == "p":
            idf_h = np.log2((N - df_h + 1) / (df_h + 1))            # idf_p, :14-15
        else:
            raise ValueError("idf must be 't' or 'p'")
    pivot_d = None
    if use_pivoted_norm:
        doc_lengths = _reduceat_lengths(doc_ptr.to(eng.device), rowptr, counts)
        if slope != -1 and m > 0:
            avg = np.mean(doc_lengths)
            pivot_d = torch.from_numpy((1 - slope) + slope * (doc_lengths / avg)).to(eng.device)
    weight_d = None
    if pagerank_weights is not None:
        w = np.asarray(pagerank_weights, dtype=np.float64)
        if w.shape != (m,):
            raise ValueError("operands could not be broadcast together: pagerank_weights must have one entry per document")
        weight_d = torch.from_numpy(np.ascontiguousarray(w)).to(eng.device)
    idf_d = torch.from_numpy(idf_h.astype(np.float64)).to(eng.device)
    zero_words = np.flatnonzero(idf_h == 0)
    if zero_words.size:                                             # eliminate_zeros (:100 / :57): rare, done with torch
        keep = idf_d[colidx.long()] != 0
        rows = torch.repeat_interleave(torch.arange(m, device=eng.device), rowptr[1:] - rowptr[:-1])[keep]
        colidx = colidx[keep].contiguous()
        rowptr = torch.zeros(m + 1, dtype=torch.int64, device=eng.device)
        rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=m), 0)
        nnz = int(colidx.numel())
    vals = torch.empty(nnz, dtype=torch.float64, device=eng.device)
    eng._check(eng._lib.mnmf_tfidf_scale(eng._h, m, rowptr.data_ptr(), colidx.data_ptr(), idf_d.data_ptr(),
                                         pivot_d.data_ptr() if pivot_d is not None else None,
                                         weight_d.data_ptr() if weight_d is not None else None, vals.data_ptr()))
    torch.cuda.synchronize(eng.device)
    return dict(rowptr=rowptr, colidx=colidx, vals=vals, shape=(m, V))


def _to_scipy(d):
    return sp.csr_matrix((d["vals"].cpu().numpy(), d["colidx"].cpu().numpy(), d["rowptr"].cpu().numpy()), shape=d["shape"])


def tf_idf_english(N=None, vocab=None, data=None, output_dir=None, fieldname=None, lemmatize=False, use_bm25=False,
                   k1=1.2, b=0.75, use_pivoted_norm=True, slope=0.2, pagerank_weights=None, *, device=None,
                   return_device: bool = False):
    """Drop-in for ``tf_idf_english`` (tfidf_english_calculator.py:11).  ``data``: one list of vocabulary indices per
    document; ``vocab``: the vocabulary (only its length is used); ``N``: the document count used by idf_t.
    ``return_device=True`` returns the device CSR dict for ``Engine.set_matrix(**d)`` instead of a scipy matrix."""
    del output_dir, fieldname, lemmatize, k1, b
    if use_bm25:
        raise NotImplementedError("BM25 is not available: the reference's BM25 branch itself cannot run "
                                  "(tfidf_bm25_turkish.py:56 references an undefined name)")
    tokens, doc_ptr = flatten_documents(data, device=_engine_for(device).device)
    d = tfidf_from_tokens(tokens, doc_ptr, len(vocab), idf="t", N=N, use_pivoted_norm=use_pivoted_norm, slope=slope,
                          pagerank_weights=pagerank_weights, eng=_engine_for(device))
    return d if return_device else _to_scipy(d)


def tf_idf_turkish(veri, tokenizer, use_bm25=False, k1=1.2, b=0.75, pagerank_weights=None, *, device=None,
                   return_device: bool = False):
    """Drop-in for ``tf_idf_turkish`` (tfidf_turkish_calculator.py:11): ``veri`` is one list of token ids per document,
    the vocabulary size comes from ``tokenizer.get_vocab_size()``; ``use_bm25`` is ignored like in the reference
    (:51), the pivoted normalisation with slope 0.2 is always on (:60-61)."""
    del use_bm25, k1, b
    tokens, doc_ptr = flatten_documents(veri, device=_engine_for(device).device)
    d = tfidf_from_tokens(tokens, doc_ptr, int(tokenizer.get_vocab_size()), idf="p", N=len(veri),
                          use_pivoted_norm=True, slope=0.2, pagerank_weights=pagerank_weights, eng=_engine_for(device))
    return d if return_device else _to_scipy(d)


def load_npz_device(path: str, device=None):
    """A cached ``*_tfidf_matrix.npz`` (scipy ``save_npz``; cache_manager.py:55,104) straight into the device CSR dict
    that ``Engine.set_matrix`` accepts: the arrays are read without unpickling and a CSC file is transposed on the GPU."""
    with np.load(path, allow_pickle=False) as z:
        fmt = z["format"].item()
        fmt = fmt.decode() if isinstance(fmt, bytes) else str(fmt)
        shape = tuple(int(x) for x in z["shape"])
        data, indices, indptr = z["data"], z["indices"], z["indptr"]
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    data_d = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    idx_d = torch.from_numpy(indices.astype(np.int64)).to(dev)
    ptr_d = torch.from_numpy(indptr.astype(np.int64)).to(dev)
    if fmt == "csr":
        rowptr, colidx = ptr_d, idx_d
        rows = torch.repeat_interleave(torch.arange(shape[0], device=dev), ptr_d[1:] - ptr_d[:-1])
        key = rows * shape[1] + colidx
        if key.numel() > 1 and not bool((key[1:] >= key[:-1]).all()):        # unsorted column indices inside a row
            order = torch.sort(key, stable=True).indices
            colidx, data_d = colidx[order], data_d[order]
    elif fmt == "csc":
        cols = torch.repeat_interleave(torch.arange(shape[1], device=dev), ptr_d[1:] - ptr_d[:-1])
        order = torch.sort(idx_d * shape[1] + cols, stable=True).indices      # row-major order of the entries
        rows = idx_d[order]
        colidx, data_d = cols[order], data_d[order]
        rowptr = torch.zeros(shape[0] + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=shape[0]), 0)
    else:
        raise ValueError(f"unsupported sparse format in {path}: {fmt}")
    return dict(rowptr=rowptr, colidx=colidx.to(torch.int32), vals=data_d, shape=shape)
