# usage: tools/kstats.sh <tag> [bench args...]: per-kernel average durations of the mnmf kernels for one bench run
R=$GRAFT_REPO_ROOT; T=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$T -o k --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu "$@" > $R/gpurun_out/ks_$T.log 2>&1 || exit 1
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/ks_$T/k_kernel_stats.csv")):
    if "mnmf::" in r["Name"] and float(r["Percentage"]) > 0.3:
        print("$T", r["Name"][11:75].ljust(66), r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
