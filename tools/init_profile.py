"""One GPU NNDSVD at C2 size (fp32 engine) for rocprofv3 --stats: where the initialisation's time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manta_amd.engine import Engine
from manta_amd.init_device import nndsvd_device
from manta_amd.synth import zipf_tfidf_csr
M, V, K = 1_000_000, 100_000, 50
dev = torch.device("cuda", 0)
rowptr, col, val = zipf_tfidf_csr(M, V, 0.001, seed=1234, device=dev, dtype=torch.float32)
eng = Engine(dtype="fp32")
eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(M, V))
info = {}
torch.cuda.synchronize(); t = time.time()
nndsvd_device(eng, K, info=info)
torch.cuda.synchronize(); print(f"nndsvd {time.time() - t:.2f} s {info}")
