"""Where the time around the solver goes in a drop-in call with init="device" (python tools/e2e_pieces.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from manta_amd import nmf as N, init_device as idv
from manta_amd.synth import zipf_tfidf_csr, to_scipy

def T(label, f):
    torch.cuda.synchronize(); t = time.time(); r = f(); torch.cuda.synchronize(); print(f"    {label:34s} {time.time() - t:7.3f} s", flush=True); return r

for (M, V, dens, K) in [(20_000, 30_000, 0.003, 10), (100_000, 50_000, 0.002, 20)]:
    rowptr, col, val = zipf_tfidf_csr(M, V, dens, seed=1234, device=torch.device("cuda", 0), dtype=torch.float64)
    X = to_scipy(rowptr.cpu(), col.cpu(), val.cpu(), (M, V))
    for dtype in ("fp64", "fp32"):
        print(f"X {X.shape} nnz {X.nnz} k {K} {dtype}")
        for rep in range(2):
            print(f"  pass {rep + 1}")
            T("_matrix_key", lambda: N._matrix_key(X))
            csr = T("tocsr", lambda: X.tocsr())
            eng = N._engine(dtype, None)
            T("set_matrix", lambda: eng.set_matrix(csr))
            w, h = T("nndsvd_device", lambda: idv.nndsvd_device(eng, K))
            W, H, st = T("run_nmf (engine)", lambda: eng.run_nmf(w, h, norm_thresh=0.005, zero_threshold=1e-5))
            T("download + float64", lambda: [np.ascontiguousarray(eng._host(v)) for v in (W, H)])
            print(f"    solver itself                      {st.ms_total / 1e3:7.3f} s ({st.iters} iterations)")
