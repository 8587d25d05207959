# usage: tools/ab_libs.sh "<lib tags>" [env...]: bench C2 (overlapped and tile kernel alone) for library variants manta_amd/libmnmf_<tag>.so
cd $GRAFT_REPO_ROOT
for v in $1; do
  for ov in 1 0; do
    MNMF_LIB=$GRAFT_REPO_ROOT/manta_amd/libmnmf_$v.so MNMF_XT_OVERLAP=$ov timeout -k 10 120 python -u bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print('$v overlap=$ov', d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1
  done
done
