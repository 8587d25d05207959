"""Document sharding for multi-GPU runs (one process per GPU, torch.distributed / RCCL).

Rows of X and W are split into contiguous blocks of (nearly) equal nnz; H (and S) are replicated.
Per iteration only the k x V partial of X^T.W (+ the k x k Gram riding in the same buffer) and the
k x k Gram of the update step are summed across ranks -- see DESIGN.md "Multi-GPU".  The reference
has no counterpart (single process); results are independent of the rank count up to summation order.
"""
from __future__ import annotations

import numpy as np


def partition_rows_by_nnz(rowptr, world: int) -> np.ndarray:
    """Row boundaries b[0..world] (b[0]=0, b[world]=m) so every rank gets ~nnz/world non-zeros."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    m = len(rowptr) - 1
    nnz = int(rowptr[-1])
    targets = (np.arange(1, world, dtype=np.float64) * nnz / world)
    cuts = np.searchsorted(rowptr, targets, side="left").astype(np.int64)
    b = np.concatenate([[0], np.clip(cuts, 0, m), [m]]).astype(np.int64)
    return np.maximum.accumulate(b)


def shard_csr(X, rank: int, world: int):
    """Rows of scipy CSR ``X`` owned by ``rank`` and their [lo, hi) range."""
    X = X.tocsr()
    b = partition_rows_by_nnz(X.indptr, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    return X[lo:hi], lo, hi


def run_sharded(method: str, X_local, W0_local, H0, S0=None, *, dtype="fp64", device=None, group=None,
                collectives=True, single_rank_hook=False, **opts):
    """Run one solver on this rank's document shard; returns (W_local, H[, S], stats).
    Requires an initialised torch.distributed process group (nccl on GPUs).  ``single_rank_hook``: install the
    collective hook even in a group of one (rehearsal of the RCCL calls on a one-GPU box)."""
    from .engine import Engine

    eng = Engine(device=device, dtype=dtype)
    try:
        eng.set_matrix(X_local)
        eng.set_process_group(group, collectives=collectives, single_rank_hook=single_rank_hook)
        if method == "nmf":
            return eng.run_nmf(W0_local, H0, **opts)
        if method == "pnmf":
            return eng.run_pnmf(W0_local, **opts)
        if method == "nmtf":
            return eng.run_nmtf(W0_local, S0, H0, **opts)
        raise ValueError(f"Unknown NMF method: {method}. Supported methods are: 'nmf', 'pnmf', 'nmtf'")
    finally:
        eng.close()
