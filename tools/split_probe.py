"""Would X.H^T gain from two passes -- frequent + warm words (their rows of H^T fit an XCD's L2) and the cold rest?
Times the engine's own X.H^T on the column-filtered halves of the configs[1] matrix against the whole matrix.
python tools/split_probe.py [warm row counts...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from manta_amd.engine import Engine
from manta_amd.synth import zipf_tfidf_csr
import bench

method, m, V, density, k = bench.CONFIGS["c2"]
dev = torch.device("cuda", 0)
rowptr, col, val = zipf_tfidf_csr(m, V, density, seed=1234, device=dev, dtype=torch.float64)
val = val.float()
df = torch.bincount(col.long(), minlength=V)
order = torch.argsort(df, descending=True, stable=True)
rank = torch.empty_like(order); rank[order] = torch.arange(V, device=dev)
crank = rank[col.long()]
rows = torch.repeat_interleave(torch.arange(m, device=dev), rowptr[1:] - rowptr[:-1])
B = torch.rand((V, k), device=dev)

def timed(eng, n=12):
    for _ in range(3): eng.spmm(B)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.spmm(B)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def engine_for(mask):
    r = rows[mask]; c = col[mask]; v = val[mask]
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(torch.bincount(r, minlength=m), 0)
    e = Engine(device=0, dtype="fp32")
    e.set_matrix(rowptr=rp, colidx=c, vals=v, shape=(m, V))
    return e, int(mask.sum())

e, n = engine_for(torch.ones_like(col, dtype=torch.bool))
print(f"whole matrix: nnz {n:.3e}  {timed(e):.3f} ms", flush=True); e.close()
for tw in [int(x) for x in sys.argv[1:]] or [4096, 8192, 12288, 16384, 24576]:
    ea, na = engine_for(crank < tw); ta = timed(ea); ea.close()
    eb, nb = engine_for(crank >= tw); tb = timed(eb); eb.close()
    print(f"warm rows {tw:6d} ({tw * 256 / 2**20:.1f} MB): frequent+warm nnz {na:.3e} {ta:.3f} ms | cold nnz {nb:.3e} {tb:.3f} ms | sum {ta + tb:.3f} ms", flush=True)
