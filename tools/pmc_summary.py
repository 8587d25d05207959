"""Summarise rocprofv3 outputs of tools/prof_pmc.sh: per-kernel average duration and HBM traffic.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes for wide (16 B per lane) reads (MI355X_MICROARCH.md, section HBM), so the fetch side
is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n[-70:]


def counter(sub, cname):
    acc = defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname and "mnmf::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


dur = {}
for f in glob.glob(f"{root}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "mnmf::" in r["Name"]:
            dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
fetch = counter("fetch", "FETCH_SIZE")
write = counter("write", "WRITE_SIZE")
rows = []
for k, (calls, us) in sorted(dur.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
    fb = 2.0 * fetch.get(k, 0.0) * 1024
    wb = write.get(k, 0.0) * 1024
    rows.append(dict(kernel=k, calls=calls, avg_us=round(us, 1), fetch_MB_corrected=round(fb / 1e6, 1),
                     write_MB=round(wb / 1e6, 1), hbm_GBps=round((fb + wb) / (us * 1e-6) / 1e9, 1) if us else 0))
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import source_fingerprint  # noqa: E402
print(json.dumps({"source_sha": source_fingerprint(), "kernels": rows}, indent=1))
