cd $GRAFT_REPO_ROOT
one() { echo "== $*"; env "$@" timeout -k 10 120 python -u bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
for df in 0 1000 3000 10000 1000000000; do
  for ord in 1 0; do one MNMF_SWEEP_MAX_DF=$df MNMF_SWEEP_ORDER=$ord; done
done
one MNMF_SWEEP_MAX_DF=1000000000 MNMF_XT_OVERLAP=0
one MNMF_SWEEP_MAX_DF=3000 MNMF_XT_OVERLAP=0
