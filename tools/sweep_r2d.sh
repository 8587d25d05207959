cd $GRAFT_REPO_ROOT
one() { echo "== $*"; env "$@" timeout -k 10 120 python -u bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
export MNMF_SWEEP=1
for bm in 4194304:4 2097152:2 1048576:1 524288:1; do
  for q in 128 64 32; do
    one MNMF_XT_BLOCK_BYTES=${bm%%:*} MNMF_XT_MIN_PIECE=${bm##*:} MNMF_XT_Q=$q
  done
done
