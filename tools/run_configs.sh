# usage: tools/run_configs.sh "c1 c3 c4" [extra bench args]; prints value, value without checks, ms/step, phases, cpu
for c in ${1:-c1 c3 c4}; do
  echo "== $c"
  timeout -k 10 400 python -u bench.py --config $c --steps ${STEPS:-10} --warmup 2 ${CPU:---no-cpu} 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['value_without_convergence_check'], d['ms_per_step'], d['roofline']['phase_ms_per_iter'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], d['setup_s'])" || exit 1
done
