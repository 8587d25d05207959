cd $GRAFT_REPO_ROOT
one() { echo "== $*"; env "$@" timeout -k 10 120 python -u bench.py --config c1 --steps 200 --warmup 20 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['value'], d['value_without_convergence_check'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
one MNMF_HOT_LDS_BYTES=131072
one MNMF_HOT_LDS_BYTES=65536
one MNMF_HOT_LDS_BYTES=32768
one MNMF_HOT_LDS_BYTES=16384
one MNMF_HOT_MIN_NNZ=100000000
one MNMF_X_Q=128
one MNMF_X_Q=64
one MNMF_XT_Q=64
one MNMF_XT_Q=32
one MNMF_XT_BLOCK_BYTES=262144
one MNMF_XT_BLOCK_BYTES=1048576
