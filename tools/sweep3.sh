run() { echo "== $*"; env "$@" timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['value_without_convergence_check'], d['roofline']['phase_ms_per_iter'])"; }
run MNMF_PERMUTE_VOCAB=0
run MNMF_HOT_LDS_BYTES=0
run MNMF_HOT_LDS_BYTES=65536
run MNMF_HOT_LDS_BYTES=131072
run MNMF_HOT_LDS_BYTES=155648
