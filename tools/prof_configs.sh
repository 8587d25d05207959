# per-config kernel stats (rocprofv3 --kernel-trace --stats) + MFMA counter pass; usage: tools/prof_configs.sh <tag> "c1 c3 ..." [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; CFGS=$2; shift; shift
for c in $CFGS; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_${T}_$c -o k --output-format csv -- python3 $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu "$@" > $R/gpurun_out/ks_${T}_$c.log 2>&1 || exit 1
  grep '^{' $R/gpurun_out/ks_${T}_$c.log | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$c', d['value'], d['roofline']['phase_ms_per_iter'])"
  python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/ks_${T}_$c/k_kernel_stats.csv")):
    if "mnmf::" in r["Name"] and float(r["Percentage"]) > 1.0:
        print("  ", r["Name"][11:90].ljust(80), r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
done
