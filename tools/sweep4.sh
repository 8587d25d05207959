run() { echo "== $*"; env "$@" timeout -k 10 200 python -u bench.py --steps 10 --warmup 2 --no-cpu --vocab 480 --density 0.208 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['nnz'], d['value'], d['roofline']['phase_ms_per_iter'])"; }
run MNMF_HOT_LDS_BYTES=0
run MNMF_HOT_LDS_BYTES=155648
python - <<'PY'
import torch
from manta_amd.synth import zipf_tfidf_csr
rp,c,v = zipf_tfidf_csr(1000000,100000,0.001,seed=1234,device='cuda')
df = torch.bincount(c.long(), minlength=100000).sort(descending=True).values.double()
cs = df.cumsum(0)/df.sum()
for t in (100,208,315,481,630,1000,2000,4000,8000,16000): print(t, float(cs[t-1]))
print('rows>512', int((df>512).sum()), 'nnz share', float(df[df>512].sum()/df.sum()))
PY
