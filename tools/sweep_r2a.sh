# rare-word blocking threshold sweep + MFMA counter names
cd $GRAFT_REPO_ROOT
timeout -k 5 60 rocprofv3 -L 2>/dev/null | grep -i "mfma" | head -40 > gpurun_out/mfma_counters.txt
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -q --timeout 120 --timeout-method thread 2>&1 | tail -2
for v in "512 4" "256 4" "128 2" "64 1" "0 1" "0 2" "0 3"; do
  set -- $v
  echo "== floor $1 min_piece $2"
  MNMF_XT_LONG_FLOOR=$1 MNMF_XT_MIN_PIECE=$2 timeout -k 10 120 python -u bench.py --steps 10 --warmup 3 --no-cpu 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['roofline']['phase_ms_per_iter'])" || exit 1
done
