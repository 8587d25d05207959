# SQ-side counters for the LDS-heavy kernels (one run per pass)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/deep_$1
shift
rm -rf $OUT; mkdir -p $OUT
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu > $OUT/$name.log 2>&1
  echo "pass $name done"
}
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_VALU
run lds SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY
run ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
