# exact-length segments by matrix size: value / X^T.W / X.H^T for MNMF_EXACT_SEG = 0, 1, 2
cd $GRAFT_REPO_ROOT
one() { tag=$1; shift; for e in 0 1 2; do MNMF_EXACT_SEG=$e timeout -k 10 200 python -u bench.py "$@" --steps 30 --warmup 3 --no-cpu 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); p=d['roofline']['phase_ms_per_iter']; print('$tag exact=$e', d['value'], p['spmm_xt'], p['spmm_x'])" || exit 1; done; }
one "100kx50k k20 f64" --docs 100000 --vocab 50000 --topics 20 --density 0.002 --dtype fp64
one "100kx50k k20 f32" --docs 100000 --vocab 50000 --topics 20 --density 0.002
one "250kx100k k50 f32" --docs 250000 --vocab 100000 --topics 50 --density 0.001
one "500kx100k k50 f32" --docs 500000 --vocab 100000 --topics 50 --density 0.001
one "125kx100k k50 f32" --docs 125000 --vocab 100000 --topics 50 --density 0.001
one "c4" --config c4
one "c3" --config c3
