#!/usr/bin/env python3
"""Headline benchmark: NMF multiplicative-update iterations/second on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one full MU iteration (H update, W update, and -- like the reference, nmf_basic.py:52-53
-- both convergence norms) of standard NMF on a synthetic Zipf TF-IDF matrix, BASELINE.json config 1:
1M docs x 100k vocab, ~0.1 % density, k = 50, fp32.  Inputs are resident in HBM before the timed region.
For N > 1 the documents (rows of X and W) are sharded over the ranks (total work fixed: strong scaling);
the k x V partial of X^T.W and the k x k Grams are all-reduced with RCCL every iteration.

Prints ONE JSON line on rank 0 (see the contract in the task statement) including
  roofline     -- dominant kernel's algorithmic bytes / its HIP-event time, against 8 TB/s
  cpu_baseline -- the CPU oracle (numpy/scipy port of the reference loop) timed on this box's host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (method, m, V, density, k)
    "c1": ("nmf", 20_000, 30_000, 0.003, 10),
    "c2": ("nmf", 1_000_000, 100_000, 0.001, 50),
    "c3": ("pnmf", 1_000_000, 100_000, 0.001, 100),
    "c4": ("nmtf", 2_000_000, 200_000, 0.0005, 20),
    # one rank's share of configs[4] (20M x 500k, k = 200 over 8 GPUs); with --gpus 8 it is sharded again
    "c5shard": ("nmf", 2_500_000, 500_000, 0.0002, 200),
}
HBM_PEAK_GBS = 8000.0
L1_GATHER_PEAK_GBS = 28000.0   # measured on MI355X with tools/l1_probe.hip (profiles/r01_l1_gather_probe.txt)
LDS_READ_PEAK_GBS = 150000.0   # ds_read_b128, every CU streaming (MI355X_MICROARCH.md, LDS)


def source_fingerprint():
    """Hash of the kernel sources the library is built from: counter summaries under profiles/ carry the fingerprint
    of the build they were measured on, and a summary of another build is not quoted as this build's traffic."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "manta_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(config, dtype, direction):
    """HBM-side bytes per launch of the dominant SpMM from the committed rocprofv3 counter passes
    (tools/prof_pmc.sh -> tools/pmc_summary.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950,
    plus WRITE_SIZE).  Counters cannot be read from inside this process, so the figure is the one measured on the
    same command and committed under profiles/; None when there is no summary for this config."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_bench_{config}_pmc_summary.json")))
    if not files or dtype != "fp32":
        return None, None
    try:
        doc = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None, None
    # summaries written since round 2 name the build they were measured on; anything else is not this build's traffic
    if not isinstance(doc, dict) or doc.get("source_sha") != source_fingerprint():
        return None, "stale: " + os.path.relpath(files[-1], ROOT) + " was measured on another build of the kernels"
    rows = doc["kernels"]
    total, hit = 0.0, False
    for r in rows:
        name = r.get("kernel", "")
        mm = re.search(r"spmm_stream_kernel<float, \d+, \d+, (\d)", name)
        # X^T.W is several launches per iteration: the blocked stream, the tile stream, the sweep and the carry fix-ups
        if (mm and int(mm.group(1)) == direction) or (direction == 1 and ("spmm_fixup" in name or "spmm_sweep" in name)):
            total += r["fetch_MB_corrected"] + r["write_MB"]
            hit = True
    if hit:
        return int(total * 1e6), os.path.relpath(files[-1], ROOT)
    return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--docs", dest="m", type=int, default=0)
    ap.add_argument("--vocab", dest="V", type=int, default=0)
    ap.add_argument("--topics", dest="k", type=int, default=0)
    ap.add_argument("--density", type=float, default=0.0)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "fp64"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo only rehearses the N>1 path with several ranks on one GPU (exchange staged through the host)")
    ap.add_argument("--rehearse-rccl", action="store_true",
                    help="--gpus 1 only: an RCCL group of one rank with the collective hook forced on, so that the whole "
                         "sharded schedule (reduce-scatter, slice update, all-gather) runs and its cost on one GPU shows")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (decided
    before anything touches the GPU; this process only relays the ranks' output and exit code)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args))
    import numpy as np
    import torch
    import torch.distributed as dist

    from manta_amd.engine import Engine
    from manta_amd.synth import zipf_tfidf_csr

    method, m, V, density, k = CONFIGS[args.config]
    m, V, k = args.m or m, args.V or V, args.k or k
    density = args.density or density
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    elif args.rehearse_rccl:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29688")
        os.environ["MNMF_RS_AG"] = "2"      # read at mnmf_create: slices also in a group of one
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def all_reduce(t, op=None):
        """all-reduce of a small device tensor on whichever backend is in use"""
        op = op or dist.ReduceOp.SUM
        if args.backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    tdt = torch.float32 if args.dtype == "fp32" else torch.float64
    # ---- this rank's shard of the documents (seed 1234 + rank), df / avg length combined over ranks
    m_lo = (m * rank) // world
    m_hi = (m * (rank + 1)) // world
    m_loc = m_hi - m_lo
    t_gen = time.time()
    rowptr, col, val = zipf_tfidf_csr(m_loc, V, density, seed=1234 + rank, device=dev, dtype=torch.float64)
    if world > 1:
        # values were computed from local df; redo idf with the global df so the matrix is one TF-IDF
        df = torch.bincount(col.long(), minlength=V).double()
        df_g = df.clone()
        all_reduce(df_g)
        scale = torch.log2((1.0 + m) / df_g.clamp(min=1.0)) / torch.log2((1.0 + m_loc) / df.clamp(min=1.0)).clamp(min=1e-30)
        val = val * scale[col.long()]
    val = val.to(tdt)
    nnz_loc = int(col.numel())
    nnz = nnz_loc
    if world > 1:
        t = torch.tensor([nnz_loc], device=dev, dtype=torch.int64)
        all_reduce(t)
        nnz = int(t.item())
    xmax = float(val.max())
    if world > 1:
        t = torch.tensor([xmax], device=dev, dtype=torch.float64)
        all_reduce(t, op=dist.ReduceOp.MAX)
        xmax = float(t.item())
    t_gen = time.time() - t_gen

    eng = Engine(device=dev_index, dtype=args.dtype)
    t_prep = time.time()
    eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m_loc, V))
    torch.cuda.synchronize()
    t_prep = time.time() - t_prep
    if world > 1 or args.rehearse_rccl:
        eng.set_process_group(single_rank_hook=args.rehearse_rccl)

    # ---- seeded uniform(0, max X) init (the reference's nmf_initialization_random), seed 42
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    H0 = torch.rand((k, V), generator=g, device=dev, dtype=torch.float64).mul_(xmax).to(tdt)
    S0 = torch.rand((k, k), generator=g, device=dev, dtype=torch.float64).mul_(xmax).to(tdt)
    g.manual_seed(4242 + rank)
    W0 = torch.rand((m_loc, k), generator=g, device=dev, dtype=torch.float64).mul_(xmax).to(tdt)

    def run(n, profile=False, check=True):
        kw = dict(fixed_iters=n, force_check=check, check_every=1, finalize=False, profile=profile)
        if method == "nmf":
            return eng.run_nmf(W0, H0, **kw)[-1]
        if method == "pnmf":
            return eng.run_pnmf(W0, **kw)[-1]
        return eng.run_nmtf(W0, S0, H0, **kw)[-1]

    if args.warmup > 0:
        run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    st = run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / args.steps
    value = args.steps / dt

    # ---- throughput mode (no per-iteration norms) and per-kernel HIP-event times, outside the timed region
    barrier()
    t1 = time.perf_counter()
    run(args.steps, check=False)
    barrier()
    dt_nocheck = time.perf_counter() - t1
    stp = run(args.steps, profile=True)
    barrier()

    sv = 4 if args.dtype == "fp32" else 8
    xb = nnz_loc * (sv + 4)
    phases = {n: v / args.steps for n, v in stp.phase_ms.items()}
    alg_bytes = {
        # stream X once, read the dense factor once (compulsory), write the product once
        "spmm_x": xb + (m_loc + 1) * 8 + V * k * sv + m_loc * k * sv,
        "spmm_xt": xb + (V + 1) * 8 + m_loc * k * sv + V * k * sv,
    }
    dom = max(alg_bytes, key=lambda n: phases.get(n, 0.0))
    dom_ms = phases[dom]
    achieved = alg_bytes[dom] / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    ld = eng.row_stride(k)                                      # the engine's padded row width (mnmf_row_stride)
    custom = bool(args.m or args.V or args.k or args.density)   # size overrides: the committed counter passes do not apply
    traffic, traffic_src = pmc_traffic(args.config, args.dtype, 1 if dom == "spmm_xt" else 0) \
        if world == 1 and not custom else (None, None)
    # secondary bound: every nnz pulls one ld-wide factor row through the vector memory path;
    # tools/l1_probe.hip measured 28 TB/s for that access (16 B per lane, 4 rows per wave load) with every row in L1
    gather_bytes = nnz_loc * ld * sv
    roofline = {"bound": "hbm", "kernel": "spmm_stream_kernel (" + ("X^T.W: tile stream + blocked stream + sweep + carry fix-ups" if dom == "spmm_xt" else "X.H^T") + ")",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": alg_bytes[dom], "avg_launch_ms": round(dom_ms, 4),
                "gather_bound": {"bytes_per_launch": gather_bytes, "achieved_GBps": round(gather_bytes / (dom_ms * 1e-3) / 1e9, 1),
                                 "l1_hit_peak_GBps": L1_GATHER_PEAK_GBS,
                                 "frac": round(gather_bytes / (dom_ms * 1e-3) / 1e9 / L1_GATHER_PEAK_GBS, 4)},
                # the same gather if every factor row came out of LDS (ds_read_b128 at the chip-wide rate)
                "lds_bound": {"bytes_per_launch": gather_bytes, "lds_read_peak_GBps": LDS_READ_PEAK_GBS,
                              "frac": round(gather_bytes / (dom_ms * 1e-3) / 1e9 / LDS_READ_PEAK_GBS, 4)},
                "phase_ms_per_iter": {n: round(v, 4) for n, v in phases.items() if v > 0},
                "iteration_bytes_model": st.bytes_model,
                "iteration_frac_of_hbm": round(st.bytes_model * value / world / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline: the oracle (numpy/scipy port of the reference loop) on a bounded row sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(method, rowptr, col, val, m_loc, V, k, W0, S0, H0, args.cpu_seconds)

    if rank == 0:
        out = {
            "metric": "nmf_mu_iterations_per_sec", "value": round(value, 3), "unit": "iter/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "f64", "data": "synthetic",
            "config": {"workload": f"{method} MU, {m} docs x {V} vocab, density {density}, k={k} "
                                   + ("(custom size)" if custom else "(one eighth of BASELINE.json configs[4])" if args.config == "c5shard"
                                      else f"(BASELINE.json configs[{sorted(CONFIGS).index(args.config)}])"),
                       "nnz": nnz, "method": method, "parallelism": f"doc-shard x{world}" + (" (RCCL rehearsal: group of one)" if args.rehearse_rccl else ""),
                       "convergence_check": "every iteration (as the reference)",
                       "init": "uniform(0, max X), seed 42", "matrix": "Zipf(1.1) TF-IDF, seed 1234+rank"},
            "value_without_convergence_check": round(args.steps / dt_nocheck, 3),
            "roofline": roofline, "cpu_baseline": cpu,
            "setup_s": {"generate": round(t_gen, 2), "set_matrix": round(t_prep, 2)},
        }
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def cpu_baseline(method, rowptr, col, val, m, V, k, W0, S0, H0, budget_s):
    """Time the CPU oracle on the first rows of the same matrix; extrapolate linearly in rows."""
    import numpy as np
    import scipy.sparse as sp

    from oracle import nmf_oracle as oc

    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    ms = min(m, 100_000)
    end = int(rowptr[ms].item())
    X = sp.csr_matrix((val[:end].double().cpu().numpy(), col[:end].cpu().numpy(), rowptr[: ms + 1].cpu().numpy()),
                      shape=(ms, V))
    W = W0[:ms].double().cpu().numpy()
    H = H0.double().cpu().numpy()
    S = S0.double().cpu().numpy()
    def call(n):
        t0 = time.perf_counter()
        if method == "nmf":
            oc.basic_nmf(X, W, H, zero_threshold=0.0, n_iter=n, check=True)
        elif method == "pnmf":
            oc.projective_nmf(X, W, zero_threshold=0.0, n_iter=n, check=True)
        else:
            oc.nmtf_core(X, W, S, H, zero_threshold=0.0, n_iter=n, check=True)
        return time.perf_counter() - t0

    t_first = call(1)
    iters = max(2, min(20, int(budget_s / max(t_first, 1e-3))))
    el = call(iters)
    per_iter_sample = el / iters
    per_iter_full = per_iter_sample * (m / ms)
    return {"value": round(1.0 / per_iter_full, 5), "unit": "iter/s", "cores": threads, "kind": "port",
            "sample": f"{iters} iterations of the CPU oracle (float64, per-iteration norms as shipped) on the first "
                      f"{ms} of {m} documents; seconds/iteration scaled by {m / ms:.1f}x (linear in rows) to the full matrix",
            "sample_seconds_per_iter": round(per_iter_sample, 3)}


if __name__ == "__main__":
    main()
