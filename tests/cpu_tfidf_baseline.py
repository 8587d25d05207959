"""CPU leg of tools/bench_tfidf.py: the TF-IDF oracle (oracle/tfidf_oracle.py) timed on a bounded sample of the same
synthetic corpus.  Lives under tests/ because only test code may import the oracle; not collected by pytest.

    python tests/cpu_tfidf_baseline.py --docs 20000 --vocab 100000 --len 100   ->  one JSON line
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=20_000)
    ap.add_argument("--vocab", type=int, default=100_000)
    ap.add_argument("--len", type=int, default=100)
    a = ap.parse_args()
    from oracle import tfidf_oracle as to

    rng = np.random.RandomState(1)
    lens = rng.randint(1, 2 * a.len, size=a.docs)
    docs = [np.minimum((a.vocab * rng.rand(n) ** 3).astype(np.int64), a.vocab - 1) for n in lens]   # skewed ids
    docs[-1] = np.append(docs[-1], a.vocab - 1)
    t = time.perf_counter()
    to.tf_idf_english(a.docs, a.vocab, docs)
    dt = time.perf_counter() - t
    print(json.dumps({"cpu_oracle_docs": a.docs, "cpu_oracle_tokens": int(lens.sum()) + 1, "cpu_oracle_s": round(dt, 3),
                      "cpu_oracle_tokens_per_s": round((int(lens.sum()) + 1) / dt), "cores": 1}))


if __name__ == "__main__":
    main()
