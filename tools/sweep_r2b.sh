cd $GRAFT_REPO_ROOT
one() { echo "== $*"; env "$@" timeout -k 10 120 python -u bench.py --steps 15 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
one MNMF_X_Q=512
for v in 384 768 1024; do one MNMF_XT_HOT_WORDS=$v; done
for v in 64 256; do one MNMF_XT_Q=$v; done
for v in 256 1024 2048; do one MNMF_X_Q=$v; done
for v in 64 160 256; do one MNMF_XTH_Q=$v; done
for v in 384 768 1024; do one MNMF_TILE_THREADS=$v; done
for v in 2097152 8388608; do one MNMF_XT_BLOCK_BYTES=$v; done
for v in 98304 147456; do one MNMF_HOT_LDS_BYTES=$v; done
one MNMF_XT_HOT_WORDS=768 MNMF_TILE_THREADS=768
one MNMF_XT_HOT_WORDS=1024 MNMF_TILE_THREADS=1024
