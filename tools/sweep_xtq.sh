cd $GRAFT_REPO_ROOT
one() { echo "== $*"; env $1 timeout -k 10 200 python -u bench.py ${@:2} --steps ${STEPS:-30} --warmup 5 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['config']['nnz'], d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
for args in "$@"; do
  for t in 128 64 32; do one MNMF_XT_Q=$t $args; done
done
