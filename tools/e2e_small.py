"""End-to-end drop-in call at small / medium sizes (what most MANTA corpora look like): manta_amd.run_nmf on a host
scipy matrix, timed as a user sees it, split into solver time and everything around it.
python tools/e2e_small.py [docs vocab density k]..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import manta_amd
from manta_amd.synth import zipf_tfidf_csr, to_scipy

cases = [(20_000, 30_000, 0.003, 10), (100_000, 50_000, 0.002, 20)]
for (M, V, dens, K) in cases:
    rowptr, col, val = zipf_tfidf_csr(M, V, dens, seed=1234, device=torch.device("cuda", 0), dtype=torch.float64)
    X = to_scipy(rowptr.cpu(), col.cpu(), val.cpu(), (M, V))
    print(f"X: {X.shape}, nnz {X.nnz}, k {K}", flush=True)
    for method in ("nmf", "pnmf", "nmtf"):
        for dtype in ("fp64", "fp32"):
            for init in ("device", "nndsvd"):
                for rep in (1, 2):
                    st = {}
                    kw = dict(init=init) if init == "device" else {}
                    np.random.seed(0)
                    t = time.time()
                    out = manta_amd.run_nmf(K, X, norm_thresh=0.005, nmf_method=method, dtype=dtype, stats_out=st, **kw)
                    dt = time.time() - t
                    print(f"  {method:4s} {dtype} init={init:7s} call {rep}: {dt:6.3f} s wall; {st['iters']:5d} iterations in {st['ms_total'] / 1e3:6.3f} s; "
                          f"around the solver {dt - st['ms_total'] / 1e3:6.3f} s", flush=True)
