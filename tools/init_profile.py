"""Steps and time of the device truncated SVD over its parameters (python tools/init_profile.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manta_amd import nmf as N, init_device as idv
from manta_amd.synth import zipf_tfidf_csr

for (M, V, dens, K) in [(20_000, 30_000, 0.003, 10), (100_000, 50_000, 0.002, 20), (1_000_000, 100_000, 0.001, 50)]:
    rowptr, col, val = zipf_tfidf_csr(M, V, dens, seed=1234, device=torch.device("cuda", 0), dtype=torch.float64)
    for dtype in ("fp64",):
        eng = N._engine(dtype, None)
        eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(M, V))
        for block in (8, 16):
            for rr in (1, 2, 3):
                info = {}
                kw = dict(info=info, block=block, rr_every=rr, seed=1)
                eng._svd_cache = None
                idv.truncated_svd_device(eng, K, **dict(kw, info={}, seed=2))
                torch.cuda.synchronize(); t = time.time()
                u, s, v = idv.truncated_svd_device(eng, K, **kw)
                torch.cuda.synchronize(); dt = time.time() - t
                print(f"{M}x{V} k={K} {dtype} block={block} rr_every={rr}: {dt:.3f} s, steps {info['steps']}, restarts {info['restarts']}, "
                      f"{dt / max(info['steps'], 1) * 1e3:.2f} ms/step", flush=True)
