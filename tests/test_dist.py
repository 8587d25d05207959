"""world_size-2 coverage of the document-sharded path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dist_launch import launch  # noqa: E402


def test_partition_by_nnz():
    from manta_amd.dist import partition_rows_by_nnz, shard_csr

    rng = np.random.RandomState(0)
    lens = rng.poisson(30, 1000)
    lens[:100] = 300
    rowptr = np.concatenate([[0], np.cumsum(lens)])
    for world in (1, 2, 4, 8):
        b = partition_rows_by_nnz(rowptr, world)
        assert b[0] == 0 and b[-1] == 1000 and np.all(np.diff(b) >= 0)
        per = np.diff(rowptr[b])
        assert per.sum() == rowptr[-1]
        assert per.max() <= rowptr[-1] / world + 300
    X = sp.random(50, 30, density=0.2, format="csr", random_state=1)
    parts = [shard_csr(X, r, 3)[0] for r in range(3)]
    assert (sp.vstack(parts) != X).nnz == 0
    # more ranks than rows with nnz: empty shards are legal
    b = partition_rows_by_nnz(np.array([0, 5]), 4)
    assert b[0] == 0 and b[-1] == 1


@pytest.mark.parametrize("method", ["nmf", "pnmf", "nmtf"])
def test_sharded_schedule_gloo_cpu(method):
    launch(["emulate", method], 29611 + ["nmf", "pnmf", "nmtf"].index(method))


@pytest.mark.parametrize("method", ["nmf", "nmtf"])
def test_reduce_scatter_schedule_gloo_cpu(method):
    """reduce-scatter -> slice update -> all-gather: 1200 vocabulary rows do not divide... they do by
    2; the odd split is covered by the three-rank case below."""
    launch(["emulate_rsag", method], 29651 + ["nmf", "nmtf"].index(method))


def test_reduce_scatter_schedule_three_ranks_uneven_slices():
    launch(["emulate_rsag", "nmf"], 29655, nproc=3)      # 1200 / 3 = 400; 1000 docs by nnz
    launch(["emulate_rsag", "nmtf"], 29656, nproc=7)     # 1200 = 7 * 172 - 4: the last slice is padded


@pytest.mark.parametrize("nproc", [2, 3, 4])
def test_early_exchange_schedule_gloo_cpu(nproc):
    """The early exchange of a sharded NMF iteration (two row ranges, the cold one first) in numpy over gloo: equal to
    the unsharded oracle to 1e-11 with 2, 3 (hot range rounded down to 96 -> 96, 1104 cold rows = 3 x 368) and 4 ranks."""
    launch(["emulate_early", "nmf"], 29681 + nproc, nproc=nproc)
