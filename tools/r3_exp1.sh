# round 3, experiment 1: back-to-back segment queues + wide chunk loads; kernel/solver parity first, then bench variants
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3e1_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r3e1_tests.log; [ $rc -eq 0 ] || exit $rc
run() { tag=$1; shift; env "$@" timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu $BARGS 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$tag', d['value'], d['roofline']['phase_ms_per_iter'])" || exit 1; }
BARGS=""
run default X=1
run xthq-1 MNMF_XTH_Q=-1
run xthq-2 MNMF_XTH_Q=-2
run xthq192 MNMF_XTH_Q=192
run xthq48 MNMF_XTH_Q=48
run noverlap MNMF_XT_OVERLAP=0
run noverlap-1 MNMF_XT_OVERLAP=0 MNMF_XTH_Q=-1
run tile1024 MNMF_TILE_THREADS=1024
BARGS="--dtype fp64"
run f64 X=1
run f64_onevec MNMF_TWO_VEC=0
