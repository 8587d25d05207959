"""Cost of the all-reduce hook itself (Python callback + two staging copies, no communication) on a 1/8 shard of
C2: what the N = 8 run pays per iteration on top of the wire time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from manta_amd import _lib
from manta_amd.engine import Engine
from manta_amd.synth import zipf_tfidf_csr

dev = torch.device("cuda", 0)
m, V, k = 125_000, 100_000, 50
rowptr, col, val = zipf_tfidf_csr(m, V, 0.001, seed=1234, device=dev, dtype=torch.float64)
eng = Engine(dtype="fp32")
eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m, V))
g = torch.Generator(device=dev); g.manual_seed(1)
W0 = torch.rand((m, k), generator=g, device=dev); H0 = torch.rand((k, V), generator=g, device=dev)

def run(n=100):
    eng.run_nmf(W0, H0, fixed_iters=5, force_check=True, finalize=False)
    torch.cuda.synchronize(); t = time.time()
    eng.run_nmf(W0, H0, fixed_iters=n, force_check=True, finalize=False)
    torch.cuda.synchronize(); return n / (time.time() - t)

print("no hook      :", round(run(), 1), "it/s")
stage = torch.empty(V * 64 + 64 * 64 + 16, device=dev)
def hook(_u, ptr, count, _dt, _st):
    nb = count * 4
    eng._lib.mnmf_memcpy_d2d(eng._h, stage.data_ptr(), ptr, nb)
    eng._lib.mnmf_memcpy_d2d(eng._h, ptr, stage.data_ptr(), nb)
    return 0
cb = _lib.ALLREDUCE_FN(hook)
eng._check(eng._lib.mnmf_set_allreduce(eng._h, cb, None, 2))
print("staging hook :", round(run(), 1), "it/s")
