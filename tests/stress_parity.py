"""Randomised parity hunt (not collected by pytest; run by hand on the GPU box: python tests/stress_parity.py [cases] [seed]).
Random shapes, densities, topic counts, methods and dtypes -- including empty rows / columns, Zipf-skewed columns and
matrices large enough for the LDS-hot, tile and blocked streams -- a few MU iterations through the C-ABI against the
CPU oracle, to the stated tolerances (fp64 1e-10, fp32 1e-4)."""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nmf_oracle as oc          # noqa: E402
from manta_amd.engine import Engine           # noqa: E402

TOL = {"fp64": 1e-10, "fp32": 1e-4}


def rel(a, b):
    n = np.linalg.norm(b)
    d = np.linalg.norm(np.asarray(a, np.float64) - b)
    return d / n if n > 0 else d


def random_matrix(rng, big):
    if big:
        m, V = int(rng.randint(20_000, int(os.environ.get("STRESS_MAX_DOCS", 120_000)))), int(rng.randint(5_000, 60_000))
        per_doc = rng.randint(20, 120)
    else:
        m, V = int(rng.randint(1, 3000)), int(rng.randint(1, 3000))
        per_doc = rng.randint(1, max(2, min(V, 60)))
    nnz = int(min(m * per_doc, int(os.environ.get("STRESS_MAX_NNZ", 12_000_000))))
    rows = rng.randint(0, m, nnz)
    if rng.rand() < 0.7:                      # Zipf-skewed vocabulary
        p = 1.0 / np.arange(1, V + 1) ** rng.uniform(0.8, 1.3)
        cols = rng.choice(V, nnz, p=p / p.sum())
    else:
        cols = rng.randint(0, V, nnz)
    X = sp.csr_matrix((rng.rand(nnz) + 0.05, (rows, cols)), shape=(m, V))   # duplicates are summed
    X.sort_indices()
    return X


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.RandomState(seed)
    engines = {d: Engine(dtype=d) for d in ("fp64", "fp32")}
    worst = {"fp64": 0.0, "fp32": 0.0}
    t0 = time.time()
    for case in range(n_cases):
        big = case % int(os.environ.get("STRESS_BIG_EVERY", 6)) == int(os.environ.get("STRESS_BIG_EVERY", 6)) - 1
        X = random_matrix(rng, big)
        m, V = X.shape
        k = int(rng.choice([1, 2, 3, 5, 7, 10, 16, 17, 20, 32, 33, 50, 64, 65, 70, 100, 112, 128, 130])) if not big else int(rng.choice([5, 10, 20, 50, 70, 100]))
        k = min(k, 130)
        method = ["nmf", "pnmf", "nmtf"][rng.randint(3)]
        dtype = ["fp64", "fp32"][rng.randint(2)]
        iters = int(rng.randint(1, 6))
        W0, H0, S0 = rng.rand(m, k) + 0.05, rng.rand(k, V) + 0.05, rng.rand(k, k) + 0.05
        eng = engines[dtype]
        eng.set_matrix(X)
        if method == "nmf":
            want = oc.basic_nmf(X, W0.copy(), H0.copy(), zero_threshold=0.0, n_iter=iters, check=False)
            W, H, _ = eng.run_nmf(W0, H0, fixed_iters=iters, zero_threshold=0.0)
            got = {"W": W, "H": H}
        elif method == "pnmf":
            want = oc.projective_nmf(X, W0.copy(), n_iter=iters, check=False)
            W, H, _ = eng.run_pnmf(W0, fixed_iters=iters)
            got = {"W": W, "H": H}
        else:
            want = oc.nmtf_core(X, W0.copy(), S0.copy(), H0.copy(), zero_threshold=0.0, n_iter=iters, check=False)
            W, S, H, _ = eng.run_nmtf(W0, S0, H0, fixed_iters=iters, zero_threshold=0.0)
            got = {"W": W, "S": S, "H": H}
        errs = {key: rel(got[key].detach().cpu().double().numpy(), want[key]) for key in want}
        e = max(errs.values())
        worst[dtype] = max(worst[dtype], e)
        flag = "" if e <= TOL[dtype] and np.isfinite(e) else "   <-- FAIL"
        print(f"case {case:3d} {method:4s} {dtype} {m:6d} x {V:6d} nnz {X.nnz:9d} k {k:3d} iters {iters}: max rel err {e:.2e}{flag}", flush=True)
        if flag:
            raise SystemExit(1)
    print(f"{n_cases} cases in {time.time() - t0:.0f} s; worst fp64 {worst['fp64']:.2e}, fp32 {worst['fp32']:.2e}")


if __name__ == "__main__":
    main()
