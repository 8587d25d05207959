"""Generate tests/golden/*.npz by running the UNMODIFIED reference solvers.

Run in the build container only (needs /root/reference):

    python oracle/make_golden.py

The reference is imported, never copied: empty parent packages (`manta`,
`manta.utils`, `manta._functions`) are pre-registered so that only the solver
modules and the console manager execute (the real parent ``__init__`` files pull
gensim / nltk / wordcloud, absent here -- SURVEY.md section 8c).  Exactly-N-iteration
control uses the solvers' own injectable ``norm_func`` (a counting callable), so
no reference line is changed.

Fixtures are data only: inputs (CSR arrays, W0/H0/S0) and the reference's outputs.
"""
from __future__ import annotations

import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def load_reference():
    for name, sub in (("manta", "manta"), ("manta.utils", "manta/utils"),
                      ("manta._functions", "manta/_functions")):
        mod = types.ModuleType(name)
        mod.__path__ = [os.path.join(REF, sub)]
        sys.modules[name] = mod
    sys.path.insert(0, REF)
    import importlib

    ref = types.SimpleNamespace()
    ref.orch = importlib.import_module("manta._functions.nmf.nmf_orchestrator")
    ref.basic = importlib.import_module("manta._functions.nmf.nmf_basic")
    ref.pnmf = importlib.import_module("manta._functions.nmf.nmf_projective_basic")
    ref.nmtf = importlib.import_module("manta._functions.nmf.nmtf.nmtf")
    ref.init = importlib.import_module("manta._functions.nmf.nmf_initialization")
    ref.nmtf_init = importlib.import_module("manta._functions.nmf.nmtf.nmtf_init")
    return ref


class CountingNorm:
    """+inf for the first ``n_calls`` calls... i.e. never converged, then 0.0 (converged)."""

    def __init__(self, n_inf_calls):
        self.n = n_inf_calls
        self.calls = 0

    def __call__(self, a, ord=None):
        self.calls += 1
        return float("inf") if self.calls <= self.n else 0.0


def make_matrix(m, V, density, seed, empty_rows=(), empty_cols=()):
    import torch
    from manta_amd.synth import zipf_tfidf_csr, to_scipy

    rowptr, col, val = zipf_tfidf_csr(m, V, density, seed=seed, device="cpu", dtype=torch.float64)
    X = to_scipy(rowptr, col, val, (m, V)).tolil()
    for r in empty_rows:
        X[r, :] = 0
    for c in empty_cols:
        X[:, c] = 0
    X = X.tocsr()
    X.eliminate_zeros()
    X.sort_indices()
    return X


def csr_arrays(X):
    return dict(rowptr=X.indptr.astype(np.int64), colidx=X.indices.astype(np.int32),
                vals=X.data.astype(np.float64), shape=np.array(X.shape, dtype=np.int64))


def fixed_iteration_cases(ref, tag, X, k, Ns, seed):
    rng = np.random.RandomState(seed)
    np.random.seed(seed)
    W0, H0 = ref.init.nmf_initialization_nndsvd(X.tocsc(), k)
    Wn, Sn, Hn = ref.nmtf_init.nmtf_initialization_nndsvd_legacy(X.tocsr(), k)
    # random positive init too (dense, no structural zeros)
    Wr = rng.uniform(0.0, X.max(), (X.shape[0], k))
    Hr = rng.uniform(0.0, X.max(), (k, X.shape[1]))
    Sr = rng.uniform(0.0, X.max(), (k, k))
    out = dict(csr_arrays(X), k=np.int64(k), Ns=np.array(Ns, dtype=np.int64),
               W0=W0, H0=H0, Wt0=Wn, St0=Sn, Ht0=Hn, Wr0=Wr, Hr0=Hr, Sr0=Sr)
    Xr = X.tocsr()
    for N in Ns:
        # loop A: 2 norm calls per iteration; iteration N must see 0.0 on both
        r = ref.basic._basic_nmf(Xr, W0.copy(), H0.copy(), None, log=False, norm_thresh=0.005,
                                 zero_threshold=1e-5, norm_func=CountingNorm(2 * (N - 1)))
        out[f"nmf_W_{N}"], out[f"nmf_H_{N}"] = r["W"], r["H"]
        # loop B: options maxiter=N and a never-converging norm -> exactly N updates
        r = ref.pnmf.projective_nmf(Xr, r=k, options={"maxiter": N}, W_mat=W0.copy(), h_mat=H0.copy(),
                                    norm_func=CountingNorm(10 ** 9))
        out[f"pnmf_W_{N}"], out[f"pnmf_H_{N}"] = r["W"], np.asarray(r["H"])
        # loop C: 3 norm calls per iteration
        w, s, h = ref.nmtf._core_nmtf_test(Xr, Wn.copy(), Sn.copy(), Hn.copy(), None, log=False,
                                           norm_thresh=0.005, zero_threshold=1e-5,
                                           norm_func=CountingNorm(3 * (N - 1)))
        out[f"nmtf_W_{N}"], out[f"nmtf_S_{N}"], out[f"nmtf_H_{N}"] = w, s, h
    # random-init loop A / C at the last N (dense factors)
    N = Ns[-1]
    r = ref.basic._basic_nmf(Xr, Wr.copy(), Hr.copy(), None, log=False, norm_thresh=0.005,
                             zero_threshold=1e-5, norm_func=CountingNorm(2 * (N - 1)))
    out["nmfr_W"], out["nmfr_H"] = r["W"], r["H"]
    w, s, h = ref.nmtf._core_nmtf_test(Xr, Wr.copy(), Sr.copy(), Hr.copy(), None, log=False,
                                       norm_thresh=0.005, zero_threshold=1e-5,
                                       norm_func=CountingNorm(3 * (N - 1)))
    out["nmtfr_W"], out["nmtfr_S"], out["nmtfr_H"] = w, s, h
    np.savez_compressed(os.path.join(OUT, f"fixed_{tag}.npz"), **out)
    print("wrote", tag, {k_: getattr(v, "shape", v) for k_, v in out.items() if k_.endswith(f"_{Ns[-1]}")})


def converged_case(ref, tag, X, k, method, seed):
    """Full run_nmf(...) as the pipeline calls it (modeling_pipeline.py:62-67).

    The solver entry points are wrapped (not edited) so that the norm function they
    receive counts its calls; iterations = calls / (norm calls per iteration).
    """
    captured = {}
    calls = {"n": 0}

    def counting_norm(*a, **kw):
        calls["n"] += 1
        return np.linalg.norm(*a, **kw)

    def init_capture(in_mat, rank):
        w, h = ref.init.nmf_initialization_nndsvd(in_mat, rank)
        captured["W0"], captured["H0"] = w.copy(), h.copy()
        return w, h

    real_legacy = ref.nmtf.nmtf_initialization_nndsvd_legacy
    real_basic, real_pnmf, real_core = ref.orch._basic_nmf, ref.orch.projective_nmf, ref.nmtf._core_nmtf_test

    def legacy_capture(in_mat, rank):
        w, s, h = real_legacy(in_mat, rank)
        captured["Wt0"], captured["St0"], captured["Ht0"] = w.copy(), s.copy(), h.copy()
        return w, s, h

    def basic_wrap(*a, **kw):
        kw["norm_func"] = counting_norm
        return real_basic(*a, **kw)

    def pnmf_wrap(*a, **kw):
        kw["norm_func"] = counting_norm
        return real_pnmf(*a, **kw)

    def core_wrap(*a, **kw):
        kw["norm_func"] = counting_norm
        return real_core(*a, **kw)

    ref.nmtf.nmtf_initialization_nndsvd_legacy = legacy_capture
    ref.orch._basic_nmf, ref.orch.projective_nmf, ref.nmtf._core_nmtf_test = basic_wrap, pnmf_wrap, core_wrap
    np.random.seed(seed)
    try:
        res = ref.orch.run_nmf(num_of_topics=k, sparse_matrix=X, init=init_capture,
                               norm_thresh=0.005, nmf_method=method)
    finally:
        ref.nmtf.nmtf_initialization_nndsvd_legacy = real_legacy
        ref.orch._basic_nmf, ref.orch.projective_nmf, ref.nmtf._core_nmtf_test = real_basic, real_pnmf, real_core
    out = dict(csr_arrays(X), k=np.int64(k), seed=np.int64(seed), method=np.array(method))
    out.update(captured)
    for key in ("W", "H", "S"):
        if key in res:
            out["out_" + key] = np.asarray(res[key])
    per_iter = {"nmf": 2, "pnmf": 1, "nmtf": 3}[method]
    # nmf/nmtf: updates applied; pnmf: updates TRIED (the last one is discarded, break-before-assign)
    out["iters"] = np.int64(calls["n"] // per_iter)
    np.savez_compressed(os.path.join(OUT, f"converged_{tag}_{method}.npz"), **out)
    print("wrote converged", tag, method, "iters", int(out["iters"]))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = load_reference()
    # A: tiny, with empty rows/cols, k=5
    XA = make_matrix(300, 200, 0.06, seed=11, empty_rows=(0, 17, 299), empty_cols=(3, 199))
    fixed_iteration_cases(ref, "A_300x200_k5", XA, 5, [1, 5, 50], seed=101)
    # C: k not a multiple of 4
    XC = make_matrix(500, 400, 0.03, seed=12)
    fixed_iteration_cases(ref, "C_500x400_k7", XC, 7, [20], seed=102)
    # B: larger, k=10
    XB = make_matrix(1000, 1200, 0.012, seed=13)
    fixed_iteration_cases(ref, "B_1000x1200_k10", XB, 10, [20], seed=103)
    # converged runs through run_nmf, as the pipeline calls it
    XD = make_matrix(600, 500, 0.03, seed=14)
    for method in ("nmf", "pnmf", "nmtf"):
        converged_case(ref, "D_600x500_k6", XD, 6, method, seed=104)
    # auto-rank
    Xcsc = XD.tocsc()
    rank = ref.orch.find_max_rank(Xcsc)
    np.savez_compressed(os.path.join(OUT, "autorank_D.npz"), **csr_arrays(XD), rank=np.int64(rank))
    print("auto rank", rank)


if __name__ == "__main__":
    main()
