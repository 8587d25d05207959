cd $GRAFT_REPO_ROOT
one() { echo "== $*"; env $1 timeout -k 10 200 python -u bench.py ${@:2} --steps 12 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
for args in "$@"; do
  for b in 4194304 8388608 16777216; do one MNMF_XT_BLOCK_BYTES=$b $args; done
done
