"""Topic-count sweep at BASELINE.json configs[0] (20k x 30k, density 0.003): three/four small-k NMF solves as separate
run_nmf calls on the resident matrix against one batched mnmf_run_nmf_sweep call (200 iterations each, norms every iteration).
usage: sweep_timing.py [c1|c2] [fp32|fp64]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_amd.engine import Engine  # noqa: E402
from manta_amd.synth import zipf_tfidf_csr  # noqa: E402

m, V, dens = (20_000, 30_000, 0.003) if len(sys.argv) < 2 or sys.argv[1] == "c1" else (1_000_000, 100_000, 0.001)
N = 200 if m < 100_000 else 30
dev = torch.device("cuda", 0)
rowptr, col, val = zipf_tfidf_csr(m, V, dens, seed=1234, device=dev, dtype=torch.float64)
DT = sys.argv[2] if len(sys.argv) > 2 else "fp32"
eng = Engine(dtype=DT)
eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m, V))
g = torch.Generator(device=dev)
g.manual_seed(1)
out = {"matrix": f"{m}x{V} density {dens}", "dtype": DT, "iterations": N, "cases": []}
for ks in (((5, 10, 15), (8, 8, 8, 8), (10, 20)) if DT == "fp32" else ((5, 10), (4, 4, 4, 4), (8, 8))):
    st = [(torch.rand((m, k), generator=g, device=dev), torch.rand((k, V), generator=g, device=dev)) for k in ks]
    kw = dict(fixed_iters=N, force_check=True)
    for _ in range(2):                                  # warm both paths
        eng.run_nmf_sweep([a for a, _ in st], [b for _, b in st], **kw)
        for a, b in st:
            eng.run_nmf(a, b, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sep = [eng.run_nmf(a, b, **kw) for a, b in st]
    torch.cuda.synchronize()
    t_sep = time.perf_counter() - t0
    t0 = time.perf_counter()
    bat = eng.run_nmf_sweep([a for a, _ in st], [b for _, b in st], **kw)
    torch.cuda.synchronize()
    t_bat = time.perf_counter() - t0
    same = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(sep, bat))
    out["cases"].append({"ks": ks, "separate_ms": round(t_sep * 1e3, 2), "batched_ms": round(t_bat * 1e3, 2),
                         "speedup": round(t_sep / t_bat, 2), "bit_identical": same,
                         "separate_loop_ms": round(sum(s[2].ms_total for s in sep), 2), "batched_loop_ms": round(bat[0][2].ms_total, 2)})
print(json.dumps(out))
