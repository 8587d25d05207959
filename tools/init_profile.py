"""Steps and time of the device NNDSVD (python tools/init_profile.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manta_amd import nmf as N, init_device as idv
from manta_amd.synth import zipf_tfidf_csr

for (M, V, dens, K) in [(20_000, 30_000, 0.003, 10), (100_000, 50_000, 0.002, 20), (1_000_000, 100_000, 0.001, 50)]:
    rowptr, col, val = zipf_tfidf_csr(M, V, dens, seed=1234, device=torch.device("cuda", 0), dtype=torch.float64)
    for dtype in ("fp64", "fp32"):
        eng = N._engine(dtype, None)
        eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(M, V))
        for tol in (None, 1e-6, 1e-4):
            for block in (16, 32):
                info = {}
                kw = dict(info=info, block=block)
                if tol is not None: kw["tol"] = tol
                idv.truncated_svd_device(eng, K, **dict(kw, info={}))
                torch.cuda.synchronize(); t = time.time()
                u, s, v = idv.truncated_svd_device(eng, K, **kw)
                torch.cuda.synchronize(); dt = time.time() - t
                print(f"{M}x{V} k={K} {dtype} tol={tol} block={block}: {dt:.3f} s, steps {info['steps']}, restarts {info['restarts']}, dim {info['dim']}, "
                      f"residual {info['max_residual']:.1e}, {dt / info['steps'] * 1e3:.2f} ms/step", flush=True)
