run() { echo "== $*"; env "$@" timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['value_without_convergence_check'], d['roofline']['phase_ms_per_iter'])"; }
run MNMF_LD_ALIGN=4
run MNMF_LD_ALIGN=16
run MNMF_LD_ALIGN=32
run MNMF_LD_ALIGN=64
