# usage: tools/sweep_env.sh VAR "v1 v2 ..." [bench args...]: one bench line per value of an environment tunable
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python -u bench.py --steps ${STEPS:-15} --warmup 2 --no-cpu "$@" 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$VAR=$v', d['value'], d['ms_per_step'], d['roofline']['phase_ms_per_iter'])" || exit 1
done
