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


def _l2_rows(a):
    """Euclidean norm of every row of a C-contiguous 2-D array, each as ``sqrt(row . row)`` through BLAS ddot: the
    rounding of ``np.linalg.norm(x, 2)`` on a contiguous vector, which the fixtures are compared against bit for bit."""
    return np.sqrt(np.array([r.dot(r) for r in a], dtype=np.float64))


def nmf_initialization_nndsvd(in_mat, rank: int):
    """NNDSVD start (Boutsidis & Gallopoulos) from the ``rank`` leading singular triplets of ARPACK ``svds``, with
    the reference's conventions (nmf_initialization.py:38-83): component 0 is ``sqrt(sigma) * |u|, |v|``; every other
    component keeps the dominant of its positive / negative parts scaled by ``sqrt(sigma * |u+-| * |v+-|)``; entries
    below 1e-11 become 0.  All components are split and scaled at once, one singular vector per contiguous row; every
    element sees the reference's operations in the reference's order, so the result is bit-identical
    (tests/test_host_logic.py::test_product_nndsvd_matches_reference_fixture)."""
    if rank < 0:
        rank = _theoretical_rank(in_mat)
    u, sigma, vt = sla.svds(in_mat, k=rank)
    desc = np.argsort(sigma)[::-1]
    sigma = sigma[desc]
    ut = np.ascontiguousarray(u.T[desc])            # rank x m
    vt = np.ascontiguousarray(vt[desc])             # rank x n
    u_pos, u_neg = np.maximum(ut, 0.0), np.maximum(-ut, 0.0)
    v_pos, v_neg = np.maximum(vt, 0.0), np.maximum(-vt, 0.0)
    nu_pos, nu_neg, nv_pos, nv_neg = _l2_rows(u_pos), _l2_rows(u_neg), _l2_rows(v_pos), _l2_rows(v_neg)
    mass_pos, mass_neg = nu_pos * nv_pos, nu_neg * nv_neg
    keep_pos = mass_pos >= mass_neg                 # ties go to the positive part, as `termp >= termn` does
    with np.errstate(divide="ignore", invalid="ignore"):   # a one-signed vector gives 0/0 exactly as in the reference
        root = np.sqrt(sigma * np.where(keep_pos, mass_pos, mass_neg))
        scale_w = root / np.where(keep_pos, nu_pos, nu_neg)
        scale_h = root / np.where(keep_pos, nv_pos, nv_neg)
        wt = scale_w[:, None] * np.where(keep_pos[:, None], u_pos, u_neg)
        ht = scale_h[:, None] * np.where(keep_pos[:, None], v_pos, v_neg)
    lead = math.sqrt(sigma[0])                      # the leading pair is one-signed up to orientation
    wt[0] = lead * np.abs(ut[0])
    ht[0] = lead * np.abs(vt[0])
    wt[wt < 1e-11] = 0
    ht[ht < 1e-11] = 0
    return np.ascontiguousarray(wt.T), ht


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
