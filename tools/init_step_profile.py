import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from manta_amd import nmf as N, init_device as idv
from manta_amd.synth import zipf_tfidf_csr
M, V, dens, K = 100_000, 50_000, 0.002, 20
rowptr, col, val = zipf_tfidf_csr(M, V, dens, seed=1234, device=torch.device("cuda", 0), dtype=torch.float64)
eng = N._engine("fp64", None)
eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(M, V))
dev = eng.device
def T(label, f, n=20):
    f(); torch.cuda.synchronize(); t = time.time()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); print(f"{label:40s} {(time.time() - t) / n * 1e3:8.3f} ms", flush=True); return r
Vj = torch.randn(V, 16, device=dev, dtype=torch.float64)
Q = torch.linalg.qr(torch.randn(V, 100, device=dev, dtype=torch.float64))[0]
Y = T("spmm X.Vj", lambda: eng.spmm(Vj))
Z = T("spmm Xt.Y", lambda: eng.spmm(Y, transpose=True))
T("tsgemm Q^T Z (V x 100, V x 16)", lambda: idv._tsgemm(Q, Z))
Tm = torch.randn(100, 100, device=dev, dtype=torch.float64); Tm = Tm + Tm.t()
T("eigh 100 (device)", lambda: torch.linalg.eigh(Tm))
Tc = Tm.cpu()
T("eigh 100 (host) incl. copies", lambda: [x.to(dev) for x in torch.linalg.eigh(Tm.cpu())])
S = torch.randn(100, 20, device=dev, dtype=torch.float64)
T("AQ @ S - (Q @ S) * theta", lambda: Q @ S - (Q @ S) * S[0])
T("norm + float(max)", lambda: float(torch.linalg.norm(Z, dim=0).max()))
T("orth_against(16 cols)", lambda: idv._orthonormal_columns(Z - Q @ idv._tsgemm(Q, Z)))
T("cat", lambda: torch.cat([Q, Vj], dim=1))
T("cholesky_ex 16 + checks", lambda: (lambda L, bad: (int(bad), float(L.diagonal().min())))(*torch.linalg.cholesky_ex(idv._tsgemm(Z, Z))))
