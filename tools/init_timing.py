"""GPU NNDSVD (manta_amd.init_device) at BASELINE.json configs[1] next to ARPACK svds on a bounded host sample."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manta_amd.engine import Engine
from manta_amd.init_device import nndsvd_device
from manta_amd.synth import zipf_tfidf_csr, to_scipy
import scipy.sparse.linalg as sla

M, V, K = 1_000_000, 100_000, 50
dev = torch.device("cuda", 0)
rowptr, col, val = zipf_tfidf_csr(M, V, 0.001, seed=1234, device=dev, dtype=torch.float64)
for dtype in ("fp32", "fp64"):
    eng = Engine(dtype=dtype)
    eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(M, V))
    for which in ("first call (loads the BLAS / solver kernels)", "second call"):
        info = {}
        torch.cuda.synchronize(); t = time.time()
        W0, H0 = nndsvd_device(eng, K, info=info)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"gpu nndsvd {dtype}, {which}: {dt:.2f} s, {info}", flush=True)
    eng.close()
rows = 100_000
X = to_scipy(rowptr[: rows + 1].clone(), col[: int(rowptr[rows])], val[: int(rowptr[rows])], (rows, V))
t = time.time(); sla.svds(X, k=K); dt = time.time() - t
print(f"host ARPACK svds on the first {rows} documents (k={K}, {os.cpu_count()} cpus visible): {dt:.1f} s", flush=True)
