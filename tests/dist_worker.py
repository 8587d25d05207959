"""Worker for the world_size-2 tests (launched with torch.distributed.run).

mode=emulate (CPU, gloo): every rank runs the engine's per-rank schedule in numpy on its document
shard and sums exactly what engine.hip all-reduces, at the same points; the result must equal the
unsharded oracle.  This pins the sharding algorithm (what is exchanged, and when).
mode=emulate_rsag (CPU, gloo): the reduce-scatter -> slice update -> all-gather schedule of nmf / nmtf
(engine.hip use_slices(); pnmf keeps its all-reduce): vocabulary rows cut into world equal slices (zero padded),
each rank sums and updates its slice only, the slices are gathered.
mode=engine / engine_ar (GPU): the real HIP engine, two ranks on one GPU, exchange staged through gloo; with the
collective hook (reduce-scatter / all-gather) or the all-reduce-only hook.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden  # noqa: E402
from manta_amd.dist import shard_csr  # noqa: E402
from oracle import nmf_oracle as oc  # noqa: E402


def allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a))
    dist.all_reduce(t)
    return t.numpy()


def reduce_scatter_rows(a, rank, world):
    """a: (Vp, k) partial on every rank, Vp = world * Vs -> this rank's (Vs, k) slice of the sum."""
    Vs = a.shape[0] // world
    parts = [torch.from_numpy(np.ascontiguousarray(a[r * Vs:(r + 1) * Vs])) for r in range(world)]
    out = torch.empty_like(parts[0])
    for r in range(world):      # gloo has no reduce_scatter: one reduce per destination, same result
        buf = parts[r].clone()
        dist.reduce(buf, dst=r)
        if r == rank:
            out = buf
    return out.numpy()


def all_gather_rows(mine, world):
    t = torch.from_numpy(np.ascontiguousarray(mine))
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    return np.concatenate([p.numpy() for p in parts])


def emulate_rsag(method, X, W, H, S, N):
    """engine.hip with use_slices(): what is exchanged, by which collective, and who updates what."""
    rank, world = dist.get_rank(), dist.get_world_size()
    k, V = H.shape
    Vs = -(-V // world)
    Vp = Vs * world
    lo = Vs * rank
    n_own = max(0, min(Vs, V - lo))

    def pad(a):                                  # (V, k) -> (Vp, k), zero rows behind
        return np.concatenate([a, np.zeros((Vp - V, a.shape[1]))])

    Ht = pad(H.T.copy())                         # the engine keeps H transposed
    if method == "nmf":
        for _ in range(N):
            wtw = allreduce(W.T @ W)                                              # small all-reduce
            num = reduce_scatter_rows(pad(X.T @ W), rank, world)                  # this rank's slice of X^T W, summed
            mine = Ht[lo:lo + Vs].copy()
            mine[:n_own] = mine[:n_own] * (num[:n_own] / (mine[:n_own] @ wtw + oc.NMF_EPS))   # (wtw @ h).T = h.T @ wtw
            Ht = all_gather_rows(mine, world)
            Hn = Ht[:V]
            W = W * ((X @ Hn) / (W @ (Hn.T @ Hn) + oc.NMF_EPS))                   # h @ h.T replicated from the gathered h
        return {"W": W, "H": Ht[:V].T.copy()}
    assert method == "nmtf"
    for _ in range(N):
        Hn = Ht[:V]
        hhT = Hn.T @ Hn
        W1 = W * ((X @ (Hn @ S.T)) / (W @ (S @ hhT @ S.T) + oc.NMTF_EPS))
        g = allreduce(W1.T @ W1)
        Pt = reduce_scatter_rows(pad(X.T @ W1), rank, world)                      # slice of (w1.T X).T
        own = Ht[lo:lo + Vs]
        num_s = allreduce(Pt[:n_own].T @ own[:n_own])                             # slice share of (w1.T X) @ h.T
        S1 = S * (num_s / (g @ S @ hhT + oc.NMTF_EPS))
        mine = own.copy()
        mine[:n_own] = own[:n_own] * ((Pt[:n_own] @ S1) / (own[:n_own] @ (S1.T @ g @ S1) + oc.NMTF_EPS))
        Ht = all_gather_rows(mine, world)
        W, S = W1, S1
    return {"W": W, "S": S, "H": Ht[:V].T.copy()}


def emulate_early(X, W, H, N, hot=96):
    """engine.hip run_nmf with the early exchange: the vocabulary rows are two ranges -- [hot, Vp),
    the rows of the less frequent words, exchanged first (on the GPU: beside the tile stream that is still producing the
    hot rows), then [0, hot) -- each cut into one slice per rank: reduce-scatter, this rank's rows updated, all-gather.
    `hot` is rounded down to a multiple of the rank count, as the engine does."""
    rank, world = dist.get_rank(), dist.get_world_size()
    k, V = H.shape
    hot -= hot % world
    Vp = world * (-(-V // world))
    Ht = np.concatenate([H.T.copy(), np.zeros((Vp - V, k))])
    for _ in range(N):
        wtw = allreduce(W.T @ W)
        part = np.concatenate([X.T @ W, np.zeros((Vp - V, k))])
        new = np.empty_like(Ht)
        for base, rows in ((hot, Vp - hot), (0, hot)):            # cold range first
            Vr = rows // world
            num = reduce_scatter_rows(part[base:base + rows], rank, world)
            lo = base + Vr * rank
            n_own = max(0, min(Vr, V - lo))
            mine = Ht[lo:lo + Vr].copy()
            mine[:n_own] = mine[:n_own] * (num[:n_own] / (mine[:n_own] @ wtw + oc.NMF_EPS))
            new[base:base + rows] = all_gather_rows(mine, world)
        Ht = new
        Hn = Ht[:V]
        W = W * ((X @ Hn) / (W @ (Hn.T @ Hn) + oc.NMF_EPS))
    return {"W": W, "H": Ht[:V].T.copy()}


def emulate(method, X, W, H, S, N):
    """Per-rank schedule of engine.hip (run_nmf / run_pnmf / run_nmtf) with its exchange points."""
    if method == "nmf":
        for _ in range(N):
            ex = allreduce(np.concatenate([(X.T @ W).ravel(), (W.T @ W).ravel()]))   # [X^T W | W^T W]
            k, V = H.shape
            num_h, wtw = ex[:V * k].reshape(V, k).T, ex[V * k:].reshape(k, k)
            H = H * (num_h / (wtw @ H + oc.NMF_EPS))
            W = W * ((X @ H.T) / (W @ (H @ H.T) + oc.NMF_EPS))
        return {"W": W, "H": H}
    if method == "pnmf":
        k = W.shape[1]
        V = X.shape[1]
        for _ in range(N):
            ex = allreduce(np.concatenate([(X.T @ W).ravel(), (W.T @ W).ravel()]))   # [vtw | wtw]
            vtw, wtw = ex[:V * k].reshape(V, k), ex[V * k:].reshape(k, k)
            pay = X @ vtw
            wt_pay = vtw.T @ vtw                 # = W.T @ X @ vtw = W.T @ pay over all ranks: no second exchange
            W = W * np.sqrt(pay / (W @ wt_pay + pay @ wtw + oc.PNMF_EPS))
        W = W.copy()
        W[W < oc.PNMF_ZERO] = 0
        nrm = np.sqrt(np.diag(allreduce(W.T @ W)))
        nrm[nrm < 10 * np.finfo(np.float64).eps] = 1.0
        W = W / nrm
        return {"W": W, "H": allreduce(np.ascontiguousarray((X.T @ W))).T}
    k = W.shape[1]
    V = X.shape[1]
    for _ in range(N):
        hhT = H @ H.T
        W1 = W * ((X @ (H.T @ S.T)) / (W @ (S @ hhT @ S.T) + oc.NMTF_EPS))
        ex = allreduce(np.concatenate([(X.T @ W1).ravel(), (W1.T @ W1).ravel()]))      # [P^T | w1^T w1]
        Pt, g = ex[:V * k].reshape(V, k), ex[V * k:].reshape(k, k)
        S1 = S * ((Pt.T @ H.T) / (g @ S @ hhT + oc.NMTF_EPS))
        H1 = H * ((S1.T @ Pt.T) / ((S1.T @ g @ S1) @ H + oc.NMTF_EPS))
        W, S, H = W1, S1, H1
    return {"W": W, "S": S, "H": H}


def main():
    mode, method = sys.argv[1], sys.argv[2]
    if os.environ.get("MNMF_TEST_DUMP_AFTER"):      # where is every rank, if the run is still going after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["MNMF_TEST_DUMP_AFTER"]), exit=True)
    if mode.startswith("nccl1"):
        # one rank, RCCL backend: the collectives are trivial, but they are the real calls (in place, on tensors that
        # alias the engine's buffers, ordered against the engine's stream) -- all a one-GPU box can rehearse of them
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = load_golden("fixed_B_1000x1200_k10")
    X = g["X"]
    N = 6
    W0, H0, S0 = g["Wr0"], g["Hr0"], g["Sr0"]
    Xl, lo, hi = shard_csr(X, rank, world)
    if method == "nmf":
        want = oc.basic_nmf(X, W0, H0, zero_threshold=0.0, n_iter=N, check=False)
    elif method == "pnmf":
        want = oc.projective_nmf(X, W0, n_iter=N, check=False)
    else:
        want = oc.nmtf_core(X, W0, S0, H0, zero_threshold=0.0, n_iter=N, check=False)
    if mode == "emulate":
        got = emulate(method, Xl.tocsr(), W0[lo:hi].copy(), H0.copy(), S0.copy(), N)
        tol = 1e-11
    elif mode == "emulate_early":
        got = emulate_early(Xl.tocsr(), W0[lo:hi].copy(), H0.copy(), N)
        tol = 1e-11
    elif mode == "emulate_rsag":
        got = emulate_rsag(method, Xl.tocsr(), W0[lo:hi].copy(), H0.copy(), S0.copy(), N)
        tol = 1e-11
    else:
        from manta_amd.dist import run_sharded

        dtype = sys.argv[3]
        kw = dict(fixed_iters=N, zero_threshold=0.0) if method != "pnmf" else dict(fixed_iters=N)
        if len(sys.argv) > 4 and sys.argv[4] == "check":      # with the convergence Grams exchanged every iteration
            kw["force_check"] = True
        if mode.endswith("_stream"):
            # "nccl1_stream": the engine lives on the stream it was created under (the default one); the solver is then
            # called under ANOTHER torch stream -- uploads, the hook's collectives and the engine's kernels must still be
            # ordered (Engine runs its torch side on its own stream, the hook on the stream the library hands it)
            from manta_amd.engine import Engine
            eng = Engine(device=0, dtype=dtype)
            with torch.cuda.stream(torch.cuda.Stream()):
                eng.set_matrix(Xl)
                eng.set_process_group(None, collectives=True, single_rank_hook=True)
                if method == "nmf":
                    res = eng.run_nmf(W0[lo:hi], H0, **kw)
                elif method == "pnmf":
                    res = eng.run_pnmf(W0[lo:hi], **kw)
                else:
                    res = eng.run_nmtf(W0[lo:hi], S0, H0, **kw)
                res = tuple(r.clone() if isinstance(r, torch.Tensor) else r for r in res)
            eng.close()
        else:
            res = run_sharded(method, Xl, W0[lo:hi], H0, S0, dtype=dtype, device=0, collectives=(mode != "engine_ar"),
                              single_rank_hook=mode.startswith("nccl1"), **kw)
        keys = ["W", "S", "H"] if method == "nmtf" else ["W", "H"]
        got = {k_: v.cpu().double().numpy() for k_, v in zip(keys, res[:-1])}
        tol = 1e-10 if dtype == "fp64" else 1e-4
    for key in want:
        ref = want[key][lo:hi] if key == "W" else want[key]
        err = np.linalg.norm(got[key] - ref) / np.linalg.norm(ref)
        assert err <= tol, f"rank {rank} {method} {key}: rel err {err}"
    # every rank must hold the same replicated H
    if not mode.startswith("nccl1"):
        h = torch.from_numpy(np.ascontiguousarray(got["H"]))
        hs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(hs, h)
        assert all(torch.equal(hs[0], x) for x in hs), "H differs between ranks"
        dist.barrier()
    if rank == 0:
        print(f"OK {mode} {method}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
