"""CPU oracle for MANTA's NMF multiplicative-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``manta_amd/`` may import this file:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

This is a numpy/scipy restatement of the three solver loops of the reference
(paths relative to /root/reference):

* loop A  standard MU NMF      manta/_functions/nmf/nmf_basic.py:33-93
* loop B  projective NMF       manta/_functions/nmf/nmf_projective_basic.py:61-121
* loop C  NMTF (W.S.H)         manta/_functions/nmf/nmtf/nmtf.py:149-216
* dispatcher / auto-rank       manta/_functions/nmf/nmf_orchestrator.py:14-46, 50-140, 143-181
* NNDSVD / random init         manta/_functions/nmf/nmf_initialization.py:38-104
* NMTF legacy double NNDSVD    manta/_functions/nmf/nmtf/nmtf_init.py:19-41
* top-word ranking rule        manta/_functions/common_language/topic_extractor.py:7-23, 265-274, 294-300

Parity status: PINNED.  The reference ships no golden vectors, so
the pin is the reference itself run in the build container:
``oracle/make_golden.py`` imports the unmodified reference solvers, runs them on
seeded inputs and stores inputs + outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` requires this restatement to reproduce those
outputs bit for bit (float64, same numpy/scipy expression order).

All arithmetic is float64 unless ``dtype`` says otherwise (the fp32 mode is used
only to calibrate the GPU fp32 tolerance and for the CPU baseline).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sp
from scipy.sparse import linalg as sla

# constants hard-coded in the reference
NMF_EPS = 1e-10          # nmf_basic.py:27
NMF_MAX_ITER = 10_000    # nmf_basic.py:26  (-> at most 10 001 updates, :57)
PNMF_EPS = 1e-10         # nmf_projective_basic.py:50
PNMF_MAX_ITER = 1000     # nmf_projective_basic.py:37
PNMF_DELTA = 0.005       # nmf_projective_basic.py:38
PNMF_ZERO = 1e-10        # nmf_projective_basic.py:43
NMTF_EPS = 1e-9          # nmtf.py:128
NMTF_MAX_ITER = 1000     # nmtf.py:204  (-> at most 1001 updates)


def spectral_norm(a: np.ndarray) -> float:
    """``np.linalg.norm(a, 2)`` -- largest singular value (nmf_basic.py:52-53)."""
    return float(np.linalg.norm(a, 2))


# --------------------------------------------------------------------------- loop A
def basic_nmf(X, w, h, norm_thresh=0.005, zero_threshold=0.0001, n_iter=None,
              check=True, stats=None):
    """Standard Lee-Seung MU NMF, H first, then W with the new H (nmf_basic.py:33-93).

    ``n_iter``: if given, apply exactly that many (H,W) update pairs and ignore
    the convergence test (the reference is driven the same way through its
    injectable ``norm_func``).  ``check=False`` skips the per-iteration spectral
    norms (only meaningful together with ``n_iter``).
    """
    X = X.tocsr()
    eps = NMF_EPS
    i = 0
    w_norm = h_norm = float("inf")
    while True:
        w_old, h_old = w, h
        wT_w = w.T @ w
        numerator_h = w.T @ X
        denominator_h = wT_w @ h + eps
        h = h * (numerator_h / denominator_h)

        h_hT = h @ h.T
        numerator_w = X @ h.T
        denominator_w = w @ h_hT + eps
        w = w * (numerator_w / denominator_w)

        if check:
            w_norm = spectral_norm(np.abs(w - w_old))
            h_norm = spectral_norm(np.abs(h - h_old))
        if n_iter is not None:
            if i + 1 >= n_iter:
                break
        else:
            if i >= NMF_MAX_ITER:
                break
            if w_norm < norm_thresh and h_norm < norm_thresh:
                break
        i += 1
    if stats is not None:
        stats.update(iters=i + 1, w_norm=w_norm, h_norm=h_norm)
    w = np.array(w, copy=True)
    h = np.array(h, copy=True)
    w[w < zero_threshold] = 0
    h[h < zero_threshold] = 0
    return {"W": w, "H": h}


# --------------------------------------------------------------------------- loop B
def projective_nmf(X, W, maxiter=PNMF_MAX_ITER, delta=PNMF_DELTA,
                   zero_threshold=PNMF_ZERO, n_iter=None, check=True, stats=None):
    """Projective NMF, Yuan-Oja sqrt MU (nmf_projective_basic.py:61-121).

    Reference quirk kept: the convergence test breaks BEFORE ``W = w_new``
    (:101-107), so the iterate that met the threshold is discarded.
    ``n_iter``: apply exactly that many updates, no convergence test.
    """
    X = X.tocsr()
    eps = PNMF_EPS
    i = 0
    w_norm = float("inf")
    limit = maxiter if n_iter is None else n_iter
    while i < limit:
        wtw = W.T @ W
        vtw = X.T @ W
        pay = X @ vtw
        wt_pay = W.T @ pay
        payda = W @ wt_pay + pay @ wtw + eps
        w_new = W * np.sqrt(pay / payda)
        if check:
            w_norm = spectral_norm(np.abs(w_new - W))
        if n_iter is None and w_norm < delta:
            break
        W = w_new
        i += 1
    if stats is not None:
        stats.update(iters=i, w_norm=w_norm)
    W = np.array(W, copy=True)
    W[W < zero_threshold] = 0
    # sklearn.preprocessing.normalize(W, norm='l2', axis=0) (nmf_projective_basic.py:114):
    # on W.T, norms = sqrt(einsum('ij,ij->i')), norms < 10*eps -> 1, rows divided in place.
    Wt = W.T
    norms = np.sqrt(np.einsum("ij,ij->i", Wt, Wt))
    norms[norms < 10 * np.finfo(norms.dtype).eps] = 1.0
    Wt /= norms[:, None]
    w = Wt.T
    h = w.T @ X
    return {"W": w, "H": np.asarray(h)}


# --------------------------------------------------------------------------- loop C
def nmtf_core(X, w, s, h, norm_thresh=0.005, zero_threshold=0.00001, n_iter=None,
              check=True, stats=None):
    """NMTF W.S.H multiplicative updates (nmtf.py:149-216, the live ``_core_nmtf_test``)."""
    X = X.tocsr()
    eps = NMTF_EPS
    i = 0
    w_norm = h_norm = s_norm = float("inf")
    while True:
        numerator_w = X @ (h.T @ s.T)
        denominator_w = w @ s @ (h @ h.T) @ s.T + eps
        w1 = w * (numerator_w / denominator_w)

        numerator_s = (w1.T @ X @ h.T)
        denominator_s = (w1.T @ w1) @ s @ (h @ h.T) + eps
        s1 = s * (numerator_s / denominator_s)

        numerator_h = s1.T @ (w1.T @ X)
        denominator_h = s1.T @ (w1.T @ w1) @ s1 @ h + eps
        h1 = h * (numerator_h / denominator_h)

        if check:
            w_norm = float(np.linalg.norm(np.abs(w1 - w), "fro"))
            h_norm = float(np.linalg.norm(np.abs(h1 - h), "fro"))
            s_norm = float(np.linalg.norm(np.abs(s1 - s), "fro"))
        w, h, s = w1, h1, s1
        i += 1
        if n_iter is not None:
            if i >= n_iter:
                break
        else:
            if w_norm < norm_thresh and h_norm < norm_thresh and s_norm < norm_thresh:
                break
            if i > NMTF_MAX_ITER:
                break
    if stats is not None:
        stats.update(iters=i, w_norm=w_norm, h_norm=h_norm, s_norm=s_norm)
    w = np.array(w, copy=True)
    h = np.array(h, copy=True)
    s = np.array(s, copy=True)
    w[w < zero_threshold] = 0
    h[h < zero_threshold] = 0
    s[s < zero_threshold] = 0
    return {"W": w, "S": s, "H": h}


# --------------------------------------------------------------------------- inits
def nndsvd_init(X, rank):
    """NNDSVD via ARPACK svds (nmf_initialization.py:38-83)."""
    if rank < 0:
        m, n = X.shape
        rank = int(math.ceil(X.nnz / (m + n)))
    u, s, v = sla.svds(X, k=rank)
    idx = np.argsort(s)[::-1]
    s, u, v = s[idx], u[:, idx], v[idx, :]
    w = np.zeros((X.shape[0], rank))
    h = np.zeros((rank, X.shape[1]))
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


def random_init(X, rank):
    """Uniform(X.min(), X.max()) W0/H0 from numpy's global RNG (nmf_initialization.py:86-104)."""
    if rank < 0:
        m, n = X.shape
        rank = int(math.ceil(X.nnz / (m + n)))
    lo, hi = X.min(), X.max()
    m, n = X.shape
    return np.random.uniform(lo, hi, (m, rank)), np.random.uniform(lo, hi, (rank, n))


def nmtf_nndsvd_legacy_init(X, rank):
    """Double NNDSVD used by the live NMTF path (nmtf_init.py:19-41)."""
    wt, ht = nndsvd_init(X, rank + 1)
    w, s_w = nndsvd_init(sp.csr_matrix(wt), rank)
    s_h, h = nndsvd_init(sp.csr_matrix(ht), rank)
    s = np.sqrt(s_w @ s_h)
    return w, s, h


# --------------------------------------------------------------------------- dispatcher
def find_max_rank(X_csc) -> int:
    """int(nnz / (m + n)); CSC required (nmf_orchestrator.py:14-46)."""
    if not sp.isspmatrix_csc(X_csc):
        raise ValueError("Input matrix must be in CSC (Compressed Sparse Column) format.")
    m, n = X_csc.shape
    return int(X_csc.count_nonzero() / (m + n))


def run_nmf(num_of_topics, sparse_matrix, init=nndsvd_init, norm_thresh=0.001,
            zero_threshold=0.00001, nmf_method="nmf", stats=None):
    """Dispatcher with the reference's semantics (nmf_orchestrator.py:143-181, 50-140).

    * ``init`` is always called, also for nmtf (whose result is then discarded, :109).
    * pnmf ignores the caller's thresholds (:120).
    * nmtf ignores ``init`` and uses the legacy double NNDSVD (nmtf.py:224).
    """
    X = sparse_matrix.tocsc()
    topic_count = num_of_topics
    if topic_count == -1:
        topic_count = find_max_rank(X)
    w, h = init(X, topic_count)
    Xr = X.tocsr()
    if nmf_method == "pnmf":
        return projective_nmf(Xr, w, stats=stats)
    if nmf_method == "nmf":
        return basic_nmf(Xr, w, h, norm_thresh=norm_thresh, zero_threshold=zero_threshold, stats=stats)
    if nmf_method == "nmtf":
        w, s, h = nmtf_nndsvd_legacy_init(Xr, topic_count)
        return nmtf_core(Xr, w, s, h, norm_thresh=norm_thresh, zero_threshold=zero_threshold, stats=stats)
    raise ValueError(f"Unknown NMF method: {nmf_method}. Supported methods are: 'nmf', 'pnmf', 'nmtf'")


# --------------------------------------------------------------------------- ranking
def topic_rankings(H, W, S=None, top_n=15):
    """Per-topic ordered word / doc ids before any string filtering
    (topic_extractor.py:265-274 for NMTF, :294-300 otherwise)."""
    k = W.shape[1] if S is not None else H.shape[0]
    words, docs = [], []
    for t in range(k):
        row = int(np.argmax(S[t, :])) if S is not None else t
        words.append(np.flip(np.argsort(H[row, :]))[:top_n].copy())
        docs.append(np.flip(np.argsort(W[:, t]))[:top_n].copy())
    return np.array(words), np.array(docs)


def s_matrix_confusion(S):
    """The NMTF topic-relationship summary the reference logs after every nmtf run (nmtf/nmtf.py:62-80): per column
    of S its argmax row and that value (:67-71), columns ordered by value descending (:73-75).  Returns
    (columns, rows, values, log lines); the lines are the reference's own (:77-78), which number the couplings by
    rank.  Pinned to tests/golden/sconf.npz (lines recorded from the unmodified reference function)."""
    S = np.asarray(S)
    rows = np.array([int(np.argmax(S[:, i])) for i in range(S.shape[1])])
    vals = np.array([S[rows[i], i] for i in range(S.shape[1])])
    order = np.argsort(vals)[::-1]
    cols, rows, vals = order, rows[order], vals[order]
    lines = [f"Topic {r} has connection with Topic {rows[r]}, strength: {vals[r]}" for r in range(len(rows))]
    return cols, rows, vals, lines


def relevance_rankings(H, W, S=None, lambda_val=0.6, top_n=15):
    """LDAvis-style relevance ranking used by the coherence step, before any vocabulary string filtering
    (manta/utils/analysis/gensim_coherence.py:40-103 and :106-140): per topic the word ids in descending relevance
    (-1 where fewer than top_n words have weight > 1e-10) and the relevance values.
    PARITY UNPINNED: that module imports gensim, which is absent here, so this restatement could not be run against
    the reference function; it follows its lines one by one."""
    H, W = np.asarray(H), np.asarray(W)
    topic_weights = W.sum(axis=0)                                              # :71
    term_frequency = np.sum(H * topic_weights.reshape(-1, 1), axis=0)          # :72
    overall = term_frequency / (term_frequency.sum() + 1e-10)                  # :73
    k = W.shape[1] if S is not None else H.shape[0]
    idx = np.full((k, top_n), -1, dtype=np.int64)
    val = np.full((k, top_n), -np.inf)
    for t in range(k):
        row = int(np.argmax(np.asarray(S)[t, :])) if S is not None else t      # :24-37, :82-84
        v = H[row]
        p = v / (v.sum() + 1e-10)                                              # :121
        lift = np.clip(p / (overall + 1e-10), 1e-10, None)                     # :124-125
        rel = lambda_val * np.log(p + 1e-10) + (1 - lambda_val) * np.log(lift)  # :127-130
        masked = np.where(v > 1e-10, rel, -np.inf)                             # :134-135
        top = np.argsort(masked)[::-1][:top_n]                                 # :137
        ok = masked[top] > -np.inf
        idx[t, :ok.sum()] = top[ok]
        val[t, :ok.sum()] = rel[top[ok]]
    return idx, val


# --------------------------------------------------------------------------- fp32 emulation (tolerance calibration / CPU baseline)
def basic_nmf_f32(X, w, h, n_iter):
    """Loop A carried in float32 (not in the reference; calibrates the GPU fp32 tolerance)."""
    X = X.tocsr().astype(np.float32)
    w = w.astype(np.float32)
    h = h.astype(np.float32)
    eps = np.float32(NMF_EPS)
    for _ in range(n_iter):
        h = h * ((w.T @ X) / ((w.T @ w) @ h + eps))
        w = w * ((X @ h.T) / (w @ (h @ h.T) + eps))
    return {"W": w, "H": h}
