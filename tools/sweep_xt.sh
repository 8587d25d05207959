set -e
run() { echo "== $*"; env "$@" timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['roofline']['phase_ms_per_iter'])"; }
run MNMF_XCD_AFFINITY=0
run MNMF_XCD_AFFINITY=1
run MNMF_XCD_AFFINITY=1 MNMF_XT_Q=128
run MNMF_XCD_AFFINITY=1 MNMF_XT_Q=64
run MNMF_XCD_AFFINITY=1 MNMF_XT_Q=128 MNMF_XT_BLOCK_BYTES=1048576
run MNMF_XCD_AFFINITY=1 MNMF_XT_Q=128 MNMF_XT_BLOCK_BYTES=4194304
run MNMF_XCD_AFFINITY=0 MNMF_XT_Q=128 MNMF_XT_BLOCK_BYTES=33554432
