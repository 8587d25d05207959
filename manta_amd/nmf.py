"""Drop-in ``run_nmf`` for MANTA on MI355X.

Same signature, return contract and exceptions as
``manta._functions.nmf.run_nmf`` (manta/_functions/nmf/nmf_orchestrator.py:143-181 and its
``_nmf_cpu``, :50-140); the multiplicative-update loops run in libmnmf.so on the GPU.
``install()`` rebinds the name inside an installed MANTA so ``run_topic_analysis(nmf_method=...,
topic_count=...)`` (manta/__init__.py:60-84) uses this engine unchanged.
"""
from __future__ import annotations

import os
from typing import Callable, Optional

import numpy as np
import scipy.sparse as sp

from .init import (nmf_initialization_nndsvd, nmf_initialization_random,  # noqa: F401  (re-exported)
                   nmtf_initialization_nndsvd_legacy, nmtf_initialization_random)

_ENGINES: dict = {}


def find_max_rank(sparse_matrix) -> int:
    """``int(nnz / (m + n))``; CSC required (nmf_orchestrator.py:14-46)."""
    if not sp.isspmatrix_csc(sparse_matrix):
        raise ValueError("Input matrix must be in CSC (Compressed Sparse Column) format.")
    m, n = sparse_matrix.shape
    return int(sparse_matrix.count_nonzero() / (m + n))


def _device_index(device) -> int:
    """None, 0, "cuda:0" and torch.device("cuda", 0) are the same GPU: one cached engine (and one resident copy of
    X and X^T) per GPU and dtype, whatever form the caller names it in."""
    import torch

    if device is None:
        return torch.cuda.current_device() if torch.cuda.is_available() else 0
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.index is not None:
        return d.index
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


def _engine(dtype, device):
    from .engine import Engine

    key = (dtype, _device_index(device))
    if key not in _ENGINES:
        _ENGINES[key] = Engine(device=key[1], dtype=dtype)
    return _ENGINES[key]


def _matrix_key(mat):
    """Cheap identity of a scipy sparse matrix: the object, its buffers and a content checksum, so that an engine
    can recognise the matrix it already holds (OptimizationPipeline.run_optimization calls run_nmf once per topic
    count on the same TF-IDF matrix, optimization_pipeline.py:105-136, 193-198)."""
    try:
        data = mat.data
        idx = mat.indices if hasattr(mat, "indices") else mat.col
        return (id(mat), mat.format, mat.shape, int(mat.nnz), data.ctypes.data, idx.ctypes.data,
                float(data.sum()), int(idx[:: max(1, idx.size // 65536)].sum()))
    except Exception:
        return None


def run_nmf(num_of_topics: int, sparse_matrix, init: Callable = nmf_initialization_nndsvd, norm_thresh=0.001,
            zero_threshold=0.00001, nmf_method: str = "nmf", console=None, *, dtype: Optional[str] = None,
            device=None, max_iter: Optional[int] = None, fixed_iters: Optional[int] = None, check_every: int = 1,
            W0=None, H0=None, S0=None, seed: int = 0, stats_out: Optional[dict] = None) -> dict:
    """Factorise ``sparse_matrix`` (docs x vocab) into ``{"W": (m,k), "H": (k,V)[, "S": (k,k)]}``,
    float64 C-contiguous ndarrays, exactly as the reference does.

    Reference semantics kept on purpose (nmf_orchestrator.py:105-133):
    ``num_of_topics == -1`` -> auto rank; ``init`` is always called (also for nmtf, where its result is
    discarded and the legacy double NNDSVD is used, nmtf.py:224); pnmf ignores ``norm_thresh`` /
    ``zero_threshold`` (:120); unknown methods raise ``ValueError``.

    Keyword-only extras (not in the reference): ``dtype`` "fp64" (default; env MANTA_AMD_DTYPE) or
    "fp32"; ``device``; ``max_iter`` / ``fixed_iters`` / ``check_every``; explicit ``W0``/``H0``/``S0``;
    for nmtf either all three of ``W0``/``S0``/``H0`` or none (``ValueError`` otherwise);
    ``init="device"`` computes the NNDSVD start on the GPU (``manta_amd.init_device``: block Lanczos over the
    engine's SpMM kernels instead of ARPACK on the host; agrees with the host path to ~1e-6, not bit for bit; for
    nmtf it is the legacy double NNDSVD, nmtf_init.py:19-41); ``init="device_random"`` draws the uniform start of
    ``nmf_initialization_random`` on the GPU (same distribution, its own stream seeded with ``seed`` -- unlike numpy's
    global generator, the same seed gives the same start on every call);
    ``stats_out`` dict receives iteration count, norms and timings (nmtf: also ``s_confusion``, the strongest coupling
    per S column in rank order -- what the reference's ``s_matrix_confusion`` logs, nmtf.py:62-80, 214).
    The engine keeps the last matrix resident: a second call with the same matrix object (unchanged content), as in
    a topic-count sweep, skips the upload and the stream preprocessing (``stats_out["matrix_resident"]``).
    """
    del console  # progress output is not reproduced
    dtype = dtype or os.environ.get("MANTA_AMD_DTYPE", "fp64")
    device_init = isinstance(init, str)
    if device_init and init not in ("device", "device_random"):
        raise ValueError("init must be a callable (csc, rank) -> (W0, H0) or one of the strings 'device', 'device_random'")
    have_init = W0 is not None and H0 is not None
    if nmf_method == "nmtf" and (W0 is not None or H0 is not None) and (S0 is None or W0 is None or H0 is None):
        raise ValueError("nmtf: explicit starts need all of W0, S0 and H0 (the reference derives the three together, "
                         "nmtf.py:224); pass none of them to get its legacy double NNDSVD")
    key = _matrix_key(sparse_matrix)
    # X (and its streams) already on the GPU?  The marker lives on the engine and Engine.set_matrix clears it, so
    # another user of the cached engine (tfidf, a direct set_matrix) cannot leave a stale one behind.
    resident = key is not None and \
        getattr(_ENGINES.get((dtype, _device_index(device))), "_resident_key", None) == key
    host_init = (not device_init and not have_init) or \
        (nmf_method == "nmtf" and not device_init and (S0 is None or not have_init))
    # the reference converts to CSC first (:168) and to CSR for the solvers (:116); the CSC copy is only made here when
    # something on the host reads it (a host init): at C2 each scipy conversion of 1e8 nnz costs about a second
    csc = sparse_matrix.tocsc() if host_init else None
    topic_count = num_of_topics
    if topic_count == -1:                                                  # :105-106 (same value in any format)
        topic_count = find_max_rank(csc) if csc is not None else \
            int(sparse_matrix.count_nonzero() / (sparse_matrix.shape[0] + sparse_matrix.shape[1]))
    if have_init:
        w, h = W0, H0
    elif not device_init:
        w, h = init(csc, topic_count)                                      # :109
    if nmf_method not in ("nmf", "pnmf", "nmtf"):
        raise ValueError(f"Unknown NMF method: {nmf_method}. Supported methods are: 'nmf', 'pnmf', 'nmtf'")  # :130
    csr = None
    if not resident or host_init:
        csr = (csc if csc is not None else sparse_matrix).tocsr()          # :116
    eng = _engine(dtype, device)
    if not resident:
        eng.set_matrix(csr)              # clears the marker
        eng._resident_key = key
    s_dev = None
    if device_init and not (have_init and (nmf_method != "nmtf" or S0 is not None)):
        from . import init_device as idv                                   # starts computed on the matrix already resident
        if init == "device_random":
            w, s_dev, h = idv.random_device(eng, topic_count, seed=seed, with_s=True)
        elif nmf_method == "nmtf":
            w, s_dev, h = idv.nmtf_nndsvd_legacy_device(eng, topic_count)  # nmtf.py:224
        else:
            w, h = idv.nndsvd_device(eng, topic_count)
    extra = dict(check_every=check_every)
    if max_iter is not None:
        extra["max_iter"] = max_iter
    if fixed_iters:
        extra["fixed_iters"] = fixed_iters
    if nmf_method == "pnmf":
        W, H, st = eng.run_pnmf(w, **extra)                                # thresholds ignored, :120
        out = {"W": W, "H": H}
    elif nmf_method == "nmf":
        W, H, st = eng.run_nmf(w, h, norm_thresh=norm_thresh, zero_threshold=zero_threshold, **extra)  # :122
        out = {"W": W, "H": H}
    else:
        if s_dev is not None:
            s = s_dev
        elif S0 is None or W0 is None or H0 is None:
            w, s, h = nmtf_initialization_nndsvd_legacy(csr, topic_count)  # nmtf.py:224
        else:
            s = S0
        W, S, H, st = eng.run_nmtf(w, s, h, norm_thresh=norm_thresh, zero_threshold=zero_threshold, **extra)  # :127
        out = {"W": W, "S": S, "H": H}
        if stats_out is not None:                                          # the summary the reference logs, nmtf.py:214
            stats_out["s_confusion"] = eng.s_matrix_confusion(S)
    out = {k_: np.ascontiguousarray(eng._host(v)) for k_, v in out.items()}
    if any(v is None for v in out.values()):
        raise RuntimeError(f"NMF computation failed for method: {nmf_method}")  # :133
    if stats_out is not None:
        stats_out.update(matrix_resident=resident, iters=st.iters, converged=st.converged, w_norm=st.w_norm, h_norm=st.h_norm,
                         s_norm=st.s_norm, ms_total=st.ms_total, ms_kernels=st.ms_kernels)
    return out


SWEEP_LINE = 32   # fp32 factor columns that share one padded 128-byte row (mnmf_run_nmf_sweep); fp64: 16
SWEEP_LINES = {"fp32": 32, "fp64": 16}


def sweep_batches(ks, limit=SWEEP_LINE):
    """Greedy grouping of topic counts, in the given order, into batches whose counts add up to at most ``limit``;
    a count above the limit is a batch of its own.  Returns lists of positions into ``ks``."""
    batches, cur, tot = [], [], 0
    for i, k in enumerate(ks):
        if cur and tot + k > limit:
            batches.append(cur)
            cur, tot = [], 0
        cur.append(i)
        tot += k
    if cur:
        batches.append(cur)
    return batches


def run_nmf_sweep(sparse_matrix, topic_counts, init: Callable = nmf_initialization_nndsvd, norm_thresh=0.001,
                  zero_threshold=0.00001, nmf_method: str = "nmf", *, dtype: Optional[str] = None, device=None,
                  max_iter: Optional[int] = None, fixed_iters: Optional[int] = None, check_every: int = 1,
                  starts=None, seed: int = 0, stats_out: Optional[list] = None) -> list:
    """``[run_nmf(k, sparse_matrix, ...) for k in topic_counts]`` -- the loop of OptimizationPipeline.run_optimization
    (optimization_pipeline.py:105-136, 193-198) -- with the matrix uploaded once and, for standard NMF, the topic
    counts that fit one padded 128-byte factor row (sum <= 32 in fp32, <= 16 in fp64, the drop-in's default dtype)
    iterated TOGETHER: one pass over X per SpMM serves the whole batch.  Every entry of the returned list equals the
    single ``run_nmf`` call bit for bit (same start).

    ``starts``: optional list of ``(W0, H0)`` per topic count; otherwise ``init`` is called once per count, in order,
    as the reference's loop would (``init="device"`` / ``"device_random"`` as in ``run_nmf``: the starts are then
    computed on the resident matrix, ``seed`` feeding the random one).  Other methods and counts that do not fit a
    batch run one by one."""
    ks = [int(k) for k in topic_counts]
    dtype = dtype or os.environ.get("MANTA_AMD_DTYPE", "fp64")
    device_init = isinstance(init, str)
    if device_init and init not in ("device", "device_random"):
        raise ValueError("init must be a callable (csc, rank) -> (W0, H0) or one of the strings 'device', 'device_random'")
    results = [None] * len(ks)
    stats = [dict() for _ in ks]
    batched = nmf_method == "nmf" and dtype in SWEEP_LINES and all(k > 0 for k in ks)
    if nmf_method not in ("nmf", "pnmf", "nmtf"):
        raise ValueError(f"Unknown NMF method: {nmf_method}. Supported methods are: 'nmf', 'pnmf', 'nmtf'")
    extra = dict(check_every=check_every)
    if max_iter is not None:
        extra["max_iter"] = max_iter
    if fixed_iters:
        extra["fixed_iters"] = fixed_iters
    if batched:
        csc = sparse_matrix.tocsc() if (starts is None and not device_init) else None
        if starts is None and not device_init:
            starts = [init(csc, k) for k in ks]                            # nmf_orchestrator.py:109, once per call
        key = _matrix_key(sparse_matrix)
        eng = _engine(dtype, device)
        if key is None or eng._resident_key != key:
            eng.set_matrix((csc if csc is not None else sparse_matrix).tocsr())
            eng._resident_key = key
        if starts is None:                                                 # device starts, on the matrix now resident
            from . import init_device as idv
            if init == "device_random":
                starts = [idv.random_device(eng, k, seed=seed, with_s=True)[::2] for k in ks]   # run_nmf's draw: W, (S), H
            else:
                starts = [idv.nndsvd_device(eng, k) for k in ks]
        for batch in sweep_batches(ks, SWEEP_LINES[dtype]):
            if len(batch) == 1:
                i = batch[0]
                W, H, st = eng.run_nmf(starts[i][0], starts[i][1], norm_thresh=norm_thresh, zero_threshold=zero_threshold, **extra)
                outs = [(W, H, st)]
            else:
                outs = eng.run_nmf_sweep([starts[i][0] for i in batch], [starts[i][1] for i in batch],
                                         norm_thresh=norm_thresh, zero_threshold=zero_threshold, **extra)
            for i, (W, H, st) in zip(batch, outs):
                results[i] = {"W": np.ascontiguousarray(eng._host(W)), "H": np.ascontiguousarray(eng._host(H))}
                stats[i].update(iters=st.iters, converged=st.converged, w_norm=st.w_norm, h_norm=st.h_norm,
                                ms_total=st.ms_total, batch=[ks[j] for j in batch])
    else:
        for i, k in enumerate(ks):
            kw = {} if starts is None else dict(W0=starts[i][0], H0=starts[i][1])
            results[i] = run_nmf(k, sparse_matrix, init=init, norm_thresh=norm_thresh, zero_threshold=zero_threshold,
                                 nmf_method=nmf_method, dtype=dtype, device=device, max_iter=max_iter,
                                 fixed_iters=fixed_iters, check_every=check_every, seed=seed, stats_out=stats[i], **kw)
    if stats_out is not None:
        stats_out[:] = stats
    return results


def sweeping_run_nmf(topic_counts, init=None):
    """A ``run_nmf`` for callers that loop over topic counts themselves -- OptimizationPipeline.run_optimization calls
    ``run_nmf(num_of_topics=k, sparse_matrix=X, norm_thresh=0.005, nmf_method=...)`` once per candidate
    (optimization_pipeline.py:105-136, 193-198) -- and cannot be handed ``run_nmf_sweep``: told the counts of the sweep
    up front, the first standard-NMF call with one of them factorises ALL of them through ``run_nmf_sweep`` (one pass over
    X per SpMM for the counts that share a padded factor row) and keeps the results; the later calls with the same
    matrix and arguments are answered from there.  Every answer equals the plain ``run_nmf`` call bit for bit (the
    starts are drawn per count, in the order of ``topic_counts``, as the loop would).  Anything else -- another method,
    explicit starts, a count that was not announced, a result already handed out -- goes to ``run_nmf``.

    ``init``: as in ``install``.  The returned function has ``.stats``, the per-count statistics of the last batch."""
    import functools
    import inspect

    planned = [int(k) for k in topic_counts]
    if not planned or any(k <= 0 for k in planned):
        raise ValueError("sweeping_run_nmf: topic_counts must be positive integers")
    sig = inspect.signature(run_nmf)
    state = {"key": None, "results": {}}

    @functools.wraps(run_nmf)
    def target(*args, **kw):
        b = sig.bind(*args, **kw)
        explicit_init = "init" in b.arguments
        b.apply_defaults()
        a = b.arguments
        if init is not None and not explicit_init:
            a["init"] = init
        k = a["num_of_topics"]
        plain = (a["nmf_method"] == "nmf" and isinstance(k, (int, np.integer)) and int(k) in planned and
                 a["W0"] is None and a["H0"] is None and a["stats_out"] is None)
        if not plain:
            return run_nmf(**a)
        mkey = _matrix_key(a["sparse_matrix"])
        init_id = a["init"] if isinstance(a["init"], str) else id(a["init"])
        key = (mkey, init_id, a["norm_thresh"], a["zero_threshold"], a["dtype"], a["device"], a["max_iter"],
               a["fixed_iters"], a["check_every"], a["seed"])
        if mkey is None or state["key"] != key or int(k) not in state["results"]:
            stats = []
            outs = run_nmf_sweep(a["sparse_matrix"], planned, init=a["init"], norm_thresh=a["norm_thresh"],
                                 zero_threshold=a["zero_threshold"], nmf_method="nmf", dtype=a["dtype"], device=a["device"],
                                 max_iter=a["max_iter"], fixed_iters=a["fixed_iters"], check_every=a["check_every"],
                                 seed=a["seed"], stats_out=stats)
            state["key"] = key if mkey is not None else None
            state["results"] = {}
            for kk, o in zip(planned, outs):
                state["results"].setdefault(kk, o)      # a count announced twice: the first factorisation
            target.stats = dict(zip(planned, stats))
        return state["results"].pop(int(k))             # handed out once: the caller owns the arrays

    target.stats = {}
    return target


def install(tfidf: bool = False, init=None, sweep_counts=None) -> list:
    """Rebind ``run_nmf`` inside an installed MANTA.  Callers bind the name at import time
    (modeling_pipeline.py:10, optimization_pipeline.py:16), so the attribute is replaced on each
    importing module.  ``tfidf=True`` also rebinds ``tf_idf_english`` / ``tf_idf_turkish`` where they are bound
    (english_entry.py:10, turkish_entry.py:4).  ``init="device"``: the rebound function computes the NNDSVD start on
    the GPU whenever the caller passes no ``init`` of its own (MANTA's pipelines never do, modeling_pipeline.py:62-67):
    the host ``svds`` the reference uses takes 1-12 s on matrices whose whole factorisation takes 0.02-0.6 s here
    (tools/e2e_small.py); the start then agrees with the reference's to ~1e-6 instead of bit for bit.
    ``sweep_counts=[5, 10, 15, ...]``: the topic counts a following ``run_optimization`` will try -- its per-count
    ``run_nmf`` calls are then served by one batched factorisation (``sweeping_run_nmf``).
    Returns the module names that were patched."""
    import functools
    import importlib

    target = run_nmf
    if init is not None and init not in ("device", "device_random"):
        raise ValueError("install(init=...): 'device' or 'device_random'")
    if sweep_counts is not None:
        target = sweeping_run_nmf(sweep_counts, init=init)
    elif init is not None:

        @functools.wraps(run_nmf)
        def target(num_of_topics, sparse_matrix, *args, **kw):
            if not args:
                kw.setdefault("init", init)
            return run_nmf(num_of_topics, sparse_matrix, *args, **kw)

    patched = []
    if tfidf:
        from . import tfidf as _t
        for name in ("manta._functions.tfidf", "manta._functions", "manta._functions.english.english_entry",
                     "manta._functions.turkish.turkish_entry"):
            try:
                mod = importlib.import_module(name)
            except Exception:
                continue
            for fn in ("tf_idf_english", "tf_idf_turkish"):
                if hasattr(mod, fn):
                    setattr(mod, fn, getattr(_t, fn))
                    patched.append(f"{name}.{fn}")
    for name in ("manta._functions.nmf.nmf_orchestrator", "manta._functions.nmf", "manta._functions",
                 "manta.pipeline.modeling_pipeline", "manta.pipeline.optimization_pipeline"):
        try:
            mod = importlib.import_module(name)
        except Exception:
            continue
        if hasattr(mod, "run_nmf"):
            setattr(mod, "run_nmf", target)
            patched.append(name)
    return patched
