"""Per-kernel averages of every counter collected by tools/prof_deep.sh."""
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{root}/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmm_stream" not in k and "spmm_sweep" not in k and "dense_rk" not in k and "fixup" not in k:
            continue
        acc[k.split("(")[0].replace("void mnmf::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:45s} {sum(v)/len(v):18.1f}   (n={len(v)})")
