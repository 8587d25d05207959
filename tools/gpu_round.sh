# usage: tools/gpu_round.sh <tag>: GPU test-suite (+ the not-yet-collected no-alloc test), then bench variants.
# A crash / timeout of the tests (exit code > 1) ends the call: nothing more is started on the GPU.
cd $GRAFT_REPO_ROOT
T=${1:-r}
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 400 --timeout-method thread -rf > gpurun_out/tests_$T.log 2>&1
rc=$?
tail -25 gpurun_out/tests_$T.log
echo "pytest rc=$rc"
[ $rc -le 1 ] || exit $rc
timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err || exit 1
cat gpurun_out/bench_$T.json
MNMF_LDS_PAD=4 timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/bench_${T}_pad4.json 2> gpurun_out/bench_${T}_pad4.err || exit 1
cat gpurun_out/bench_${T}_pad4.json
