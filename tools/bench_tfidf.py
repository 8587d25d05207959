"""Throughput of the device TF-IDF builder on a synthetic corpus shaped like BASELINE.json configs[1]
(1M documents x 100k words, ~100 tokens per document), tokens resident in HBM, next to the CPU oracle on a sample (timed by tests/cpu_tfidf_baseline.py in its own process).

    python tools/bench_tfidf.py [--docs 1000000] [--vocab 100000] [--len 100] [--cpu-docs 20000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--len", type=int, default=100)
    ap.add_argument("--cpu-docs", type=int, default=20_000)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from manta_amd import tfidf
    from manta_amd.engine import Engine

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    lens = torch.randint(1, 2 * a.len, (a.docs,), generator=g, device=dev)
    ptr = torch.zeros(a.docs + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(lens, 0)
    u = torch.rand(int(ptr[-1]), generator=g, device=dev, dtype=torch.float64)
    tokens = (a.vocab * u ** 3).long().clamp(max=a.vocab - 1).to(torch.int32)
    tokens[-1] = a.vocab - 1
    eng = Engine(dtype="fp32")
    out = {"docs": a.docs, "vocab": a.vocab, "tokens": int(tokens.numel())}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.reps):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / a.reps, r

    t_counts, r = timed(lambda: tfidf.term_counts_device(tokens, ptr, a.vocab, eng))
    nnz = int(r[1].numel())
    t_sort, _ = timed(lambda: tfidf._term_counts_global_sort(tokens, ptr, a.vocab))
    t_full, d = timed(lambda: tfidf.tfidf_from_tokens(tokens, ptr, a.vocab, idf="t", N=a.docs, eng=eng))
    # algorithmic bytes of the two document passes: tokens read twice (4 B), row_nnz written, colidx + counts written
    alg = 2 * 4 * tokens.numel() + 8 * a.docs + 8 * nnz
    out.update(nnz=nnz, term_counts_ms=round(t_counts * 1e3, 2), global_sort_ms=round(t_sort * 1e3, 2),
               tfidf_total_ms=round(t_full * 1e3, 2), tokens_per_s=round(tokens.numel() / t_full),
               term_counts_alg_GBps=round(alg / t_counts / 1e9, 1))
    # CPU leg: the oracle on a bounded sample of the same distribution, in a process of its own under tests/
    # (only test code imports the oracle)
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(here, "tests", "cpu_tfidf_baseline.py"), "--docs", str(min(a.cpu_docs, a.docs)),
                        "--vocab", str(a.vocab), "--len", str(a.len)], capture_output=True, text=True)
    if r.returncode == 0:
        out.update(json.loads(r.stdout.strip().splitlines()[-1]))
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
