# usage: tools/r3_kstats_cfg.sh <tag> <config> [bench args]: kernel stats of one bench config (mnmf kernels with >= 8 calls)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$1; C=$2; shift; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ks_$T -o k --output-format csv -- python3 $R/bench.py --config $C --steps 20 --warmup 2 --no-cpu "$@" > $R/gpurun_out/ks_$T.log 2>&1 || exit 1
grep '^{' $R/gpurun_out/ks_$T.log | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$T', d['value'], d['roofline']['phase_ms_per_iter'])"
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/ks_$T/k_kernel_stats.csv")):
    if "mnmf::" in r["Name"] and int(r["Calls"]) >= 8:
        print("  ", r["Name"][11:100].ljust(90), r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
