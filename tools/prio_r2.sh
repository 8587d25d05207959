cd $GRAFT_REPO_ROOT
for p in -1 0 1; do
  echo "== side prio $p"
  MNMF_SIDE_PRIO=$p bash tools/bench_cfgs.sh "c2" || exit 1
  MNMF_SIDE_PRIO=$p bash tools/bench_cfgs.sh "c2" --rehearse-rccl || exit 1
done
MNMF_SIDE_PRIO=-1 bash tools/bench_cfgs.sh "c3 c4 c5shard c1"
MNMF_SIDE_PRIO=0 bash tools/bench_cfgs.sh "c3 c4 c5shard c1"
