"""What would X.H^T cost if hot and cold nnz did not share gather instructions?  ("What comes next" 1.)
Splits the configs[1] matrix by column frequency -- the `n_hot` most frequent words / the rest -- and times X.H^T of each
part alone through the engine's own kernels: the cold part with the plain stream (every lane group gathers at every
step: full exec masks), the hot part with the LDS-hot kernel (no gathers at all).

python tools/r3_cold_probe.py [n_hot]   (run with MNMF_HOT_LDS_BYTES=0 for the plain kernel on the cold part)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from manta_amd.engine import Engine  # noqa: E402
from manta_amd.synth import zipf_tfidf_csr  # noqa: E402


def main():
    n_hot = int(sys.argv[1]) if len(sys.argv) > 1 else 748
    which = sys.argv[2] if len(sys.argv) > 2 else "cold"
    m, V, k = 1_000_000, 100_000, 50
    dev = torch.device("cuda:0")
    rowptr, col, val = zipf_tfidf_csr(m, V, 0.001, seed=1234, device=dev, dtype=torch.float64)
    df = torch.bincount(col.long(), minlength=V)
    order = torch.argsort(df, descending=True)
    is_hot = torch.zeros(V, dtype=torch.bool, device=dev)
    is_hot[order[:n_hot]] = True
    keep = is_hot[col.long()] if which == "hot" else ~is_hot[col.long()]
    if which == "cold_even":   # half of the cold vocabulary: what one of two XCD groups would gather from
        rank = torch.empty(V, dtype=torch.long, device=dev)
        rank[order] = torch.arange(V, device=dev)
        keep &= (rank[col.long()] % 2 == 0)
    rows = torch.repeat_interleave(torch.arange(m, device=dev), rowptr[1:] - rowptr[:-1])
    cnt = torch.bincount(rows[keep], minlength=m)
    rp = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    rp[1:] = torch.cumsum(cnt, 0)
    c2, v2 = col[keep].contiguous(), val[keep].to(torch.float32).contiguous()
    print(which, "part:", int(keep.sum()), "of", col.numel(), "nnz", flush=True)
    eng = Engine(device=0, dtype="fp32")
    eng.set_matrix(rowptr=rp, colidx=c2, vals=v2, shape=(m, V))
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    W0 = torch.rand((m, k), generator=g, device=dev, dtype=torch.float32)
    H0 = torch.rand((k, V), generator=g, device=dev, dtype=torch.float32)
    eng.run_nmf(W0, H0, fixed_iters=3, finalize=False)
    st = eng.run_nmf(W0, H0, fixed_iters=10, finalize=False, profile=True)[-1]
    print({n: round(v / 10, 4) for n, v in st.phase_ms.items() if v > 0}, flush=True)


if __name__ == "__main__":
    main()
