"""Host-side cost of the collective hook: an RCCL group of one rank on one GPU, C2-sized problem; times every
torch.distributed call made by the hook and the whole hook (python tools/hook_timing.py [config])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29689")
os.environ["MNMF_RS_AG"] = "2"
import torch, torch.distributed as dist
from manta_amd.engine import Engine
from manta_amd.synth import zipf_tfidf_csr
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
method, m, V, density, k = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rowptr, col, val = zipf_tfidf_csr(m, V, density, seed=1234, device=dev, dtype=torch.float64)
val = val.float()
eng = Engine(device=0, dtype="fp32")
eng.set_matrix(rowptr=rowptr, colidx=col, vals=val, shape=(m, V))
acc = {}
def timed(name, fn):
    def w(*a, **kw):
        t0 = time.perf_counter(); r = fn(*a, **kw); dt = time.perf_counter() - t0
        c = acc.setdefault(name, [0, 0.0]); c[0] += 1; c[1] += dt
        return r
    return w
for n in ("all_reduce", "reduce_scatter_tensor", "all_gather_into_tensor"):
    setattr(dist, n, timed(n, getattr(dist, n)))
eng._alias = timed("alias", eng._alias)
eng.set_process_group(single_rank_hook=True)
xmax = float(val.max())
H0 = torch.rand((k, V), device=dev) * xmax; W0 = torch.rand((m, k), device=dev) * xmax
kw = dict(fixed_iters=5, force_check=True, check_every=1, finalize=False)
eng.run_nmf(W0, H0, **kw); torch.cuda.synchronize(); acc.clear()
N = 20
t0 = time.perf_counter(); eng.run_nmf(W0, H0, **dict(kw, fixed_iters=N)); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{cfg}: {N / dt:.1f} it/s, {dt / N * 1e3:.3f} ms per iteration")
for n, (c, t) in acc.items():
    print(f"  {n:28s} {c / N:5.1f} calls/iteration  {t / c * 1e6:8.1f} us each  {t / N * 1e3:7.3f} ms/iteration")
dist.destroy_process_group()
