# achievable HBM bandwidth on this box for simple streaming kernels (calibration for DESIGN.md)
import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for mb in (64, 256, 448, 1024):
    n = mb*1024*1024//4
    a=torch.rand(n,device='cuda'); b=torch.rand(n,device='cuda'); c=torch.empty_like(a)
    ms=t(lambda: torch.mul(a,b,out=c)); print(f"mul  {mb} MB x3: {ms:.3f} ms  {3*mb/1024/ms*1000/1000:.2f} TB/s")
    ms=t(lambda: c.copy_(a)); print(f"copy {mb} MB x2: {ms:.3f} ms  {2*mb/1024/ms:.2f} TB/s")
    ms=t(lambda: a.sum()); print(f"sum  {mb} MB x1: {ms:.3f} ms  {mb/1024/ms:.2f} TB/s")
    ms=t(lambda: c.zero_()); print(f"fill {mb} MB x1: {ms:.3f} ms  {mb/1024/ms:.2f} TB/s")
