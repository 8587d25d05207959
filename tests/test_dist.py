"""world_size-2 coverage of the document-sharded path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def launch(args, port, nproc=2, **extra_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER] + args
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "OK" in r.stdout


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


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,dtype", [(2, "fp64"), (3, "fp32")])   # 4 ranks on ONE GPU: minutes of queue stalls, see below
def test_sharded_engine_early_exchange(nproc, dtype):
    """The real engine through the early exchange: the tile stream of X^T.W forced on for the test-sized matrix (it is
    what leaves the cold rows final early), ranks sharing the one GPU over gloo; results within the tolerances of the
    unsharded oracle, H bit-identical across ranks.  MNMF_EARLY_EXCHANGE=0 (the default) must give the plain sliced schedule."""
    tile = dict(MNMF_HOT_MIN_NNZ="0", MNMF_TILE_MIN_ROWS="1", MNMF_TILE_MIN_TILES_PCT="0", MNMF_XT_HOT_WORDS="100",
                MNMF_TILE_LDS_BYTES="4096", MNMF_EARLY_CHECK="1")
    launch(["engine", "nmf", dtype, "check"], 29690 + nproc, nproc=nproc, MNMF_EARLY_EXCHANGE="1", **tile)
    if nproc == 2:
        # The plain sliced schedule on the same streams.  Only with two ranks: it runs the tile stream and the blocked
        # stream on two HIP streams per process, and four processes with three streams each on ONE GPU oversubscribe
        # its hardware queues -- cross-stream waits then stall (the early exchange with 4 ranks took 124 s instead of 4,
        # the plain schedule did not finish; neither with MNMF_XT_OVERLAP=0).  A one-process-per-GPU run has no such
        # sharing; four ranks are covered by the gloo emulation above and by test_sharded_engine_four_ranks.
        launch(["engine", "nmf", dtype, "check"], 29695 + nproc, nproc=nproc, MNMF_EARLY_EXCHANGE="0", **tile)


@pytest.mark.gpu
def test_rccl_early_exchange_in_a_group_of_one():
    """The early exchange with the RCCL hook in a group of one rank (MNMF_RS_AG=2): the collectives on the communication
    stream, ordered against the solver's stream by events, leave the oracle's result."""
    launch(["nccl1", "nmf", "fp32", "check"], 29699, nproc=1, MNMF_RS_AG="2", HSA_ENABLE_IPC_MODE_LEGACY="0",
           MNMF_HOT_MIN_NNZ="0", MNMF_TILE_MIN_ROWS="1", MNMF_TILE_MIN_TILES_PCT="0", MNMF_XT_HOT_WORDS="100",
           MNMF_TILE_LDS_BYTES="4096", MNMF_EARLY_CHECK="1", MNMF_EARLY_EXCHANGE="1")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp64", "fp32"])
@pytest.mark.parametrize("method", ["nmf", "pnmf", "nmtf"])
def test_sharded_engine_two_ranks_one_gpu(method, dtype):
    """collective hook: nmf / nmtf take the reduce-scatter -> slice update -> all-gather schedule"""
    launch(["engine", method, dtype, "check"], 29631 + ["nmf", "pnmf", "nmtf"].index(method) + (3 if dtype == "fp32" else 0))


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["nmf", "nmtf"])
def test_sharded_engine_all_reduce_hook_only(method):
    """the all-reduce-only hook (mnmf_set_allreduce): all-reduce + replicated H update, as in round 1"""
    launch(["engine_ar", method, "fp64"], 29641 + ["nmf", "nmtf"].index(method))


@pytest.mark.gpu
def test_sharded_engine_three_ranks_padded_slices():
    launch(["engine", "nmf", "fp32", "check"], 29645, nproc=3)
    launch(["engine", "nmtf", "fp64"], 29646, nproc=7)


@pytest.mark.gpu
def test_sharded_engine_four_ranks():
    """the result must not depend on the number of ranks beyond summation order -- 2, 3 and 7 ranks above,
    4 here; every case is checked against the unsharded oracle within the dtype's tolerance and H must be bit-identical
    on all ranks.  (The 8-rank run is the driver's.)"""
    launch(["engine", "nmf", "fp32", "check"], 29647, nproc=4)
    launch(["engine", "pnmf", "fp64", "check"], 29648, nproc=4)


@pytest.mark.gpu
@pytest.mark.parametrize("method,dtype", [("nmf", "fp32"), ("nmtf", "fp64"), ("pnmf", "fp32")])
def test_rccl_calls_in_a_group_of_one(method, dtype):
    """The nccl (= RCCL) branch of the hook on the one GPU this box has: a group of one rank, hook forced on, the
    engine forced through reduce-scatter -> slice update -> all-gather (MNMF_RS_AG=2).  The sums are trivial; what is
    checked is that RCCL accepts the in-place calls on tensors aliasing the engine's buffers and that they are ordered
    against the engine's kernels (the result must still be the oracle's)."""
    launch(["nccl1", method, dtype, "check"], 29661 + ["nmf", "pnmf", "nmtf"].index(method), nproc=1, MNMF_RS_AG="2",
           HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.gpu
def test_solver_called_under_another_torch_stream():
    """Engine created on the default stream, solver (with the RCCL hook of a group of one) called under a side stream:
    the torch side of the call and the hook's collectives run on the engine's stream, so the result is still the
    oracle's."""
    launch(["nccl1_stream", "nmf", "fp32", "check"], 29671, nproc=1, MNMF_RS_AG="2", HSA_ENABLE_IPC_MODE_LEGACY="0")


@pytest.mark.gpu
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts --gpus 1) must start its ranks as a
    child torch.distributed.run and print one JSON line with n_gpus = 2.  Two ranks share the one GPU here, so the
    exchange goes through gloo; the command line is otherwise the driver's."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--config", "c1", "--backend", "gloo", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["parallelism"] == "doc-shard x2"


@pytest.mark.gpu
def test_bench_rehearses_the_rccl_schedule_on_one_gpu():
    """`python bench.py --rehearse-rccl`: an nccl group of one rank, the collective hook forced on, the whole sharded
    schedule through RCCL -- one JSON line, marked as a rehearsal, n_gpus 1."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="2", MASTER_PORT="29671", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--config", "c1",
           "--rehearse-rccl", "--no-cpu"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "RCCL rehearsal" in d["config"]["parallelism"]
