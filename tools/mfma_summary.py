"""Matrix-core and LDS counters per kernel from tools/prof_round.sh (passes `mfma` and `lds`).

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles per SIMD x SIMDs), with kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3
sums that counter over the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back') and 1024 SIMDs; mfma_flops = MOPS x 512
(the counters count operations / 512).  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import source_fingerprint  # noqa: E402


def short(name):
    return name.split("(")[0].replace("void ", "")[-80:]


def collect(sub):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(f"{root}/{sub}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "mnmf::" in r["Kernel_Name"]:
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


dur = {}
for f in glob.glob(f"{root}/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "mnmf::" in r["Name"]:
            dur[short(r["Name"])] = float(r["AverageNs"]) / 1e3
mf, lds = collect("mfma"), collect("lds")
rows = []
for k in sorted(dur, key=lambda x: -dur[x]):
    m, l = mf.get(k, {}), lds.get(k, {})
    row = {"kernel": k, "avg_us": round(dur[k], 1)}
    gui = m.get("GRBM_GUI_ACTIVE", 0.0)
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    mops = m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
    if busy > 0 and gui > 0:
        row["mfma_busy_cycles"] = busy
        row["kernel_cycles"] = round(gui / 8)
        row["mfma_util"] = round(busy / (gui / 8 * 1024), 4)
        row["mfma_gflop_per_launch"] = round(mops * 512 / 1e9, 3)
        row["mfma_tflops"] = round(mops * 512 / (dur[k] * 1e-6) / 1e12, 1)
    if l.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        row["lds_idx_active"] = l["SQ_LDS_IDX_ACTIVE"]
        row["lds_bank_conflict"] = l.get("SQ_LDS_BANK_CONFLICT", 0.0)
        row["lds_conflict_share"] = round(l.get("SQ_LDS_BANK_CONFLICT", 0.0) / l["SQ_LDS_IDX_ACTIVE"], 4)
    if l.get("SQ_WAVE_CYCLES", 0) > 0:
        row["wait_any_share"] = round(l.get("SQ_WAIT_ANY", 0.0) / l["SQ_WAVE_CYCLES"], 3)
        row["wait_inst_share"] = round(l.get("SQ_WAIT_INST_ANY", 0.0) / l["SQ_WAVE_CYCLES"], 3)
    if len(row) > 2:
        rows.append(row)
print(json.dumps({"source_sha": source_fingerprint(), "kernels": rows}, indent=1))
