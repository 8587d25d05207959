# usage: tools/r3_env.sh "<bench args>" "VAR=.. VAR=.." "VAR=.." ... : one bench line (value + phases) per environment set
cd $GRAFT_REPO_ROOT
BARGS=$1; shift
for e in "$@"; do
  env $e timeout -k 10 200 python -u bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu $BARGS 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$e', d['value'], d['roofline']['phase_ms_per_iter'])" || exit 1
done
