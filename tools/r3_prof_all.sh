# usage: tools/r3_prof_all.sh "<configs>": per config the rocprofv3 passes of tools/prof_round.sh (kernel trace + stats,
# FETCH_SIZE, WRITE_SIZE, matrix-core and LDS counters), the summaries copied to profiles/r03_bench_<cfg>_*, then the
# clean bench line of the same command (its roofline.traffic is read from the summary just written).
cd $GRAFT_REPO_ROOT
for c in $1; do
  bash tools/prof_round.sh r3_$c $c || exit 1
  O=gpurun_out/prof_r3_$c
  cp $O/pmc_summary.json profiles/r03_bench_${c}_pmc_summary.json
  cp $O/mfma_lds_summary.json profiles/r03_bench_${c}_mfma_lds_summary.json
  cp $O/kernel_stats.csv profiles/r03_bench_${c}_kernel_stats.csv
  cp profiles/r03_bench_${c}_* gpurun_out/
  if [ "$c" = "c2" ]; then CPU=""; else CPU="--no-cpu"; fi
  timeout -k 10 600 python -u bench.py --config $c --steps ${STEPS:-20} --warmup 3 $CPU 2>/dev/null | grep '^{' > gpurun_out/r03_bench_$c.json || exit 1
  python3 -c "
import json; d=json.load(open('gpurun_out/r03_bench_$c.json')); r=d['roofline']
print('$c', d['value'], 'frac', r['frac'], 'traffic', r['traffic'], r['traffic_source'], r['phase_ms_per_iter'])"
done
