# usage: tools/sweep_knobs.sh <bench args...>: one knob at a time around the defaults, value / X^T.W ms / X.H^T ms
cd $GRAFT_REPO_ROOT
ARGS="$*"
one() { echo "== $*"; env "$@" timeout -k 10 160 python -u bench.py $ARGS --steps 12 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print(d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; }
one MNMF_X_Q=512
for v in 256 1024; do one MNMF_XT_HOT_WORDS=$v; done
for v in 64 256; do one MNMF_XT_Q=$v; done
for v in 128 256 1024; do one MNMF_X_Q=$v; done
for v in 48 192; do one MNMF_XTH_Q=$v; done
for v in 384 768 1024; do one MNMF_TILE_THREADS=$v; done
for v in 2097152 8388608; do one MNMF_XT_BLOCK_BYTES=$v; done
for v in 8 32; do one MNMF_LD_ALIGN=$v; done
one MNMF_XT_OVERLAP=0
one MNMF_XCD_AFFINITY=0
one MNMF_PERMUTE_VOCAB=0
one MNMF_XT_LONG_FLOOR=64
one MNMF_XT_LONG_FLOOR=256
one MNMF_XT_MIN_PIECE=2
one MNMF_XT_MIN_PIECE=8
