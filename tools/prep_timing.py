"""Wall time of set_matrix and of the one-off stream preparation (first product in each direction) at C2 size."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_amd.engine import Engine
from manta_amd.synth import zipf_tfidf_csr

m, V, k = 1_000_000, 100_000, 50
dev = torch.device("cuda", 0)
rowptr, col, val = zipf_tfidf_csr(m, V, 0.001, seed=1234, device=dev, dtype=torch.float32)
for rep in range(2):
    eng = Engine(dtype="fp32")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m, V))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    B = torch.rand((V, k), device=dev); A = torch.rand((m, k), device=dev)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.spmm(B); torch.cuda.synchronize(); t3 = time.perf_counter()
    eng.spmm(A, transpose=True); torch.cuda.synchronize(); t4 = time.perf_counter()
    eng.spmm(B); eng.spmm(A, transpose=True); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"rep {rep}: set_matrix {1e3*(t1-t0):.1f} ms, first X.B {1e3*(t3-t2):.1f} ms, first X^T.A {1e3*(t4-t3):.1f} ms, steady pair {1e3*(t5-t4):.1f} ms")
    eng.close()
