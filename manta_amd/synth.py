"""Deterministic synthetic TF-IDF matrices shaped like MANTA's English TF-IDF output.

Values follow the reference's pipeline: binary term frequency times
``idf_t = log2((1 + N) / df_j)`` (tfidf_idf_functions.py:12, tfidf_english_calculator.py:90-101)
divided by the pivoted length norm ``0.8 + 0.2 * len_i / avg_len``
(tfidf_english_calculator.py:106-120).  Structure: tokens drawn from a Zipf(s)
vocabulary, lognormal document lengths, duplicates inside a document merged.
Vocabulary ids are shuffled with a seeded permutation so column order carries no
frequency information (as in a real vocabulary).

Runs on any torch device (CPU here, the GPU for the full-size bench matrices) and
returns CSR arrays (int64 rowptr, int32 colidx, values) on that device.
"""
from __future__ import annotations

import math

import torch


def _draws_for_unique(p: torch.Tensor, target_unique: float) -> float:
    """Tokens to draw from distribution ``p`` so that E[#distinct] == target_unique."""
    lo, hi = 1.0, 64.0 * max(target_unique, 1.0)
    logq = torch.log1p(-p.double().clamp(max=1 - 1e-12))
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        uniq = float((1.0 - torch.exp(mid * logq)).sum())
        if uniq < target_unique:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def zipf_tfidf_csr(m: int, V: int, density: float, *, seed: int = 1234, zipf_s: float = 1.1,
                   sigma: float = 0.5, device="cpu", dtype=torch.float64, chunk_docs: int = 1 << 18):
    """Return ``(rowptr int64[m+1], colidx int32[nnz], vals dtype[nnz])`` on ``device``.

    ``density`` is the target fraction of non-zeros (mean distinct terms per
    document ~= density * V).
    """
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    ranks = torch.arange(1, V + 1, dtype=torch.float64, device=dev)
    p = ranks.pow(-zipf_s)
    p = p / p.sum()
    cdf = torch.cumsum(p, 0)
    cdf[-1] = 1.0
    mean_draws = _draws_for_unique(p, density * V)
    mu = math.log(mean_draws) - 0.5 * sigma * sigma
    lens = torch.exp(mu + sigma * torch.randn(m, generator=g, device=dev, dtype=torch.float64))
    lens = lens.round().clamp(min=5, max=max(5.0, 4.0 * mean_draws)).to(torch.int64)
    perm = torch.randperm(V, generator=g, device=dev)

    keys = []
    for d0 in range(0, m, chunk_docs):
        d1 = min(m, d0 + chunk_docs)
        l = lens[d0:d1]
        doc = torch.repeat_interleave(torch.arange(d0, d1, device=dev, dtype=torch.int64), l)
        u = torch.rand(doc.numel(), generator=g, device=dev, dtype=torch.float64)
        word = torch.searchsorted(cdf, u).clamp(max=V - 1)
        word = perm[word]
        keys.append(torch.unique(doc * V + word))  # sorted, duplicates merged
        del doc, u, word
    key = torch.cat(keys) if len(keys) > 1 else keys[0]
    del keys
    rows = torch.div(key, V, rounding_mode="floor")
    cols = (key - rows * V)
    del key
    counts = torch.bincount(rows, minlength=m)
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    rowptr[1:] = torch.cumsum(counts, 0)
    df = torch.bincount(cols, minlength=V).double()
    idf = torch.log2((1.0 + m) / df.clamp(min=1.0))
    norm = 0.8 + 0.2 * lens.double() / lens.double().mean()
    vals = idf[cols] / norm[rows]
    # the reference calls eliminate_zeros() (tfidf_english_calculator.py:101); idf is 0 only
    # when df == m + 1, which cannot happen, so no entry is dropped here.
    return rowptr, cols.to(torch.int32), vals.to(dtype)


def to_scipy(rowptr, colidx, vals, shape):
    import numpy as np
    import scipy.sparse as sp

    return sp.csr_matrix((vals.cpu().numpy().astype(np.float64), colidx.cpu().numpy(),
                          rowptr.cpu().numpy().astype(np.int64)), shape=shape)
