# usage: tools/r3_cfgs.sh "<configs>" [bench args]: one clean bench line per config (value, no-check value, phases)
cd $GRAFT_REPO_ROOT
CFGS=$1; shift
for c in $CFGS; do
  timeout -k 10 400 python -u bench.py --config $c --steps ${STEPS:-20} --warmup 3 --no-cpu "$@" 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$c', '$*', d['value'], d['value_without_convergence_check'], d['roofline']['phase_ms_per_iter'])" || exit 1
done
