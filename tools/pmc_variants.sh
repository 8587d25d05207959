# FETCH_SIZE + duration of the two SpMM kernels under tuning variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
run() {
  i=$((i+1)); OUT=gpurun_out/pv_$i; rm -rf $OUT; mkdir -p $OUT
  echo "== $*"
  for v in "$@"; do export "$v"; done
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/log 2>&1
  for v in "$@"; do unset "${v%%=*}"; done
  python3 - $OUT <<'PY'
import csv,glob,sys
from collections import defaultdict
acc=defaultdict(list); dur=defaultdict(list)
for f in glob.glob(sys.argv[1]+"/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "spmm_stream" in r["Kernel_Name"] or "spmm_fixup" in r["Kernel_Name"]:
            k=r["Kernel_Name"].split("(")[0][-40:]
            acc[k].append(float(r["Counter_Value"])); dur[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in acc: print("  ",k, "fetch_GB(x2)=%.2f"%(2*sum(acc[k])/len(acc[k])*1024/1e9), "us=%.0f"%(sum(dur[k])/len(dur[k])))
PY
}
run MNMF_DUMMY=1
run MNMF_XT_Q=128
run MNMF_XT_Q=64
run MNMF_XT_Q=128 MNMF_XT_BLOCK_BYTES=1048576
run MNMF_XT_Q=256 MNMF_XT_BLOCK_BYTES=1048576
run MNMF_XCD_AFFINITY=0
