"""HBM write vs read bandwidth with plain torch kernels (context for the write-drain-bound GEMM epilogues, DESIGN_LOG.md section 4)."""
import torch
dev = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / n
for mb in (19, 38, 256, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    w = t(lambda: a.fill_(1.0)); c = t(lambda: b.copy_(a)); r = t(lambda: a.sum())
    print(f"{mb:5d} MiB: fill {mb/1024/w/1.024**-1/1e3*1.048576:6.2f} TB/s ({w*1e6:7.1f} us)   copy (r+w) {2*mb*1.048576/1e6/c:6.2f} TB/s ({c*1e6:7.1f} us)   "
          f"sum (read) {mb*1.048576/1e6/r:6.2f} TB/s ({r*1e6:7.1f} us)", flush=True)
