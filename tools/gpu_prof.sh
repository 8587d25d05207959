# usage: tools/gpu_prof.sh <tag> [bench args]: no-alloc test alone in a fresh process, then rocprofv3 kernel stats of the bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py "tests/test_gpu_kernels.py::test_spmm_both_directions" -m gpu -q --timeout 120 --timeout-method thread -rf > gpurun_out/tz_$T.log 2>&1
rc=$?; tail -5 gpurun_out/tz_$T.log; [ $rc -le 1 ] || exit $rc
bash tools/kstats.sh $T "$@" || exit 1
MNMF_XT_HOT_WORDS=1024 bash tools/kstats.sh ${T}_h1024 "$@" || exit 1
MNMF_XT_HOT_WORDS=768 MNMF_XT_OVERLAP=0 bash tools/kstats.sh ${T}_h768_noov "$@" || exit 1
MNMF_XT_OVERLAP=0 bash tools/kstats.sh ${T}_noov "$@" || exit 1
