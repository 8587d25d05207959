"""Generate tests/golden/sconf.npz: the log lines of the UNMODIFIED reference s_matrix_confusion
(manta/_functions/nmf/nmtf/nmtf.py:62-80) for a few S matrices.  Build container only (needs /root/reference):

    python oracle/make_golden_sconf.py

The function returns nothing; what it computes is only visible in its debug lines, so a recording console
(`print_debug(msg, tag=...)`) is passed in.  Fixture = the S matrices and the recorded lines.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, load_reference  # noqa: E402


class Recorder:
    def __init__(self):
        self.lines = []

    def print_debug(self, msg, tag=None):
        self.lines.append(str(msg))


def main():
    ref = load_reference()
    rng = np.random.RandomState(7)
    conv = np.load(os.path.join(OUT, "converged_D_600x500_k6_nmtf.npz"), allow_pickle=False)
    cases = {
        "converged": np.asarray(conv["out_S"], dtype=np.float64),      # S of the converged reference run (k = 6)
        "random20": rng.uniform(0, 3, (20, 20)),                        # C4's k
        "sparse9": np.where(rng.uniform(size=(9, 9)) < 0.6, 0.0, rng.uniform(0, 1, (9, 9))),   # thresholded zeros
    }
    cases["sparse9"][:, 4] = 0.0                                        # a column of zeros: np.argmax answers row 0
    out = {}
    for name, S in cases.items():
        rec = Recorder()
        ref.nmtf.s_matrix_confusion(S, console=rec)
        out[f"S_{name}"] = S
        out[f"lines_{name}"] = np.array(rec.lines)
    out["names"] = np.array(sorted(cases))
    np.savez_compressed(os.path.join(OUT, "sconf.npz"), **out)
    print("wrote sconf.npz", {k: len(v) for k, v in out.items() if k.startswith("lines_")})


if __name__ == "__main__":
    main()
