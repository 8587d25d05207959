"""Host-side factor initialisation with the reference's semantics.

The GPU path consumes W0/H0[/S0]; producing them on the device is a "next" row of SURVEY.md
section 8(f).  These are the same algorithms the reference runs on the host:

* NNDSVD over ARPACK ``svds``          manta/_functions/nmf/nmf_initialization.py:38-83
* uniform random in [X.min(), X.max()]  manta/_functions/nmf/nmf_initialization.py:86-104
* NMTF legacy double NNDSVD            manta/_functions/nmf/nmtf/nmtf_init.py:19-41
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
from scipy.sparse import linalg as sla


def _theoretical_rank(X) -> int:
    m, n = X.shape
    return int(math.ceil(X.nnz / (m + n)))


def nmf_initialization_nndsvd(in_mat, rank: int):
    if rank < 0:
        rank = _theoretical_rank(in_mat)
    u, s, v = sla.svds(in_mat, k=rank)
    order = np.argsort(s)[::-1]
    s, u, v = s[order], u[:, order], v[order, :]
    w = np.zeros((in_mat.shape[0], rank))
    h = np.zeros((rank, in_mat.shape[1]))
    w[:, 0] = math.sqrt(s[0]) * abs(u[:, 0])
    h[0, :] = math.sqrt(s[0]) * abs(v[0, :].T)
    for i in range(1, rank):
        uu, vv = u[:, i], v[i, :]
        uup, uun = np.multiply(uu >= 0, uu), np.multiply(uu < 0, -uu)
        vvp, vvn = np.multiply(vv >= 0, vv), np.multiply(vv < 0, -vv)
        n_uup, n_uun = np.linalg.norm(uup, 2), np.linalg.norm(uun, 2)
        n_vvp, n_vvn = np.linalg.norm(vvp, 2), np.linalg.norm(vvn, 2)
        termp, termn = n_uup * n_vvp, n_uun * n_vvn
        if termp >= termn:
            w[:, i] = math.sqrt(s[i] * termp) / n_uup * uup
            h[i, :] = math.sqrt(s[i] * termp) / n_vvp * vvp.T
        else:
            w[:, i] = math.sqrt(s[i] * termn) / n_uun * uun
            h[i, :] = math.sqrt(s[i] * termn) / n_vvn * vvn.T
    w[w < 1e-11] = 0
    h[h < 1e-11] = 0
    return w, h


def nmf_initialization_random(in_mat, rank: int):
    if rank < 0:
        rank = _theoretical_rank(in_mat)
    lo, hi = in_mat.min(), in_mat.max()
    m, n = in_mat.shape
    return np.random.uniform(lo, hi, (m, rank)), np.random.uniform(lo, hi, (rank, n))


def nmtf_initialization_nndsvd_legacy(in_mat, rank: int):
    wt, ht = nmf_initialization_nndsvd(in_mat, rank + 1)
    w, s_w = nmf_initialization_nndsvd(sp.csr_matrix(wt), rank)
    s_h, h = nmf_initialization_nndsvd(sp.csr_matrix(ht), rank)
    return w, np.sqrt(s_w @ s_h), h


def nmtf_initialization_random(in_mat, rank: int):
    lo, hi = in_mat.min(), in_mat.max()
    m, n = in_mat.shape
    return (np.random.uniform(lo, hi, (m, rank)), np.random.uniform(lo, hi, (rank, rank)),
            np.random.uniform(lo, hi, (rank, n)))
