run() { echo "== $*"; env "$@" timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['value_without_convergence_check'], d['roofline']['phase_ms_per_iter'])"; }
run MNMF_DUMMY=1
run MNMF_LIB=$PWD/manta_amd/libmnmf_nont.so
run MNMF_DUMMY=1
run MNMF_LIB=$PWD/manta_amd/libmnmf_nont.so
