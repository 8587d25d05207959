# usage: tools/gpu_quick.sh <tag> "<configs>": kernel + solver parity tests, then kernel stats per config
cd $GRAFT_REPO_ROOT
T=$1; CFGS=${2:-c2}
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -m gpu -q -x --timeout 300 --timeout-method thread -rf > gpurun_out/tq_$T.log 2>&1
rc=$?; tail -4 gpurun_out/tq_$T.log; [ $rc -eq 0 ] || exit 1
bash tools/prof_configs.sh $T "$CFGS" || exit 1
