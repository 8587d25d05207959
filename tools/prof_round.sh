# usage: tools/prof_round.sh <tag> <config> [bench args]
# rocprofv3 passes of the bench command, one run per pass: kernel trace + stats, FETCH_SIZE, WRITE_SIZE, matrix-core
# counters, LDS counters.  Outputs under gpurun_out/prof_<tag>; summaries are made by tools/pmc_summary.py (traffic)
# and tools/mfma_summary.py (matrix cores, LDS).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; C=$2; shift; shift
OUT=gpurun_out/prof_$T
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --config $C --steps 10 --warmup 2 --no-cpu "$@" > $OUT/trace.log 2>&1 || exit 1
run() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 bench.py --config $C --steps 3 --warmup 1 --no-cpu $EXTRA > $OUT/$name.log 2>&1 || exit 1
  echo "pass $name done"
}
EXTRA="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES
python3 tools/pmc_summary.py $OUT > $OUT/pmc_summary.json
python3 tools/mfma_summary.py $OUT > $OUT/mfma_lds_summary.json
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
