"""End-to-end drop-in call at BASELINE.json configs[1]: manta_amd.run_nmf on a host scipy matrix (upload, stream
preparation, GPU NNDSVD, MU iterations to the reference's stopping rule, download), timed as a user sees it."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import manta_amd
from manta_amd.synth import zipf_tfidf_csr, to_scipy

M, V, K = 1_000_000, 100_000, 50
rowptr, col, val = zipf_tfidf_csr(M, V, 0.001, seed=1234, device=torch.device("cuda", 0), dtype=torch.float64)
X = to_scipy(rowptr.cpu(), col.cpu(), val.cpu(), (M, V))
del rowptr, col, val
print(f"X: {X.shape}, nnz {X.nnz}", flush=True)
for dtype in ("fp32", "fp64"):
    for rep in ("first call", "second call (matrix resident)"):
        st = {}
        t = time.time()
        out = manta_amd.run_nmf(K, X, norm_thresh=0.005, nmf_method="nmf", init="device", dtype=dtype, stats_out=st)
        dt = time.time() - t
        print(f"{dtype} {rep}: {dt:.2f} s wall; iterations {st['iters']} in {st['ms_total'] / 1e3:.2f} s, converged {st['converged']}, "
              f"resident {st['matrix_resident']}, W {out['W'].shape} {out['W'].dtype}", flush=True)
