"""What makes fc1 slower inside the step than back to back?  The same launch timed alone (HIP events around it) after
(a) nothing, (b) a 1 GB fill (L2 / Infinity Cache contents replaced, few pages touched per byte), (c) one word per 4 KiB page of
8 GB (translation caches replaced, almost no data), (d) both.  Per-workgroup epilogue cycles come from PRIMX_GEMM_PROF=1 runs of
the same script (the timeline is printed by the library)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops

dev, dt = "cuda:0", torch.float16
M, N, K = 4096, 4608, 1152
A = torch.randn(M, K, device=dev).to(dt)
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
b = torch.randn(N, device=dev).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
big = torch.empty(1 << 28, device=dev)                      # 1 GB of fp32
pages = torch.zeros(1 << 31, device=dev, dtype=torch.float32)   # 8 GB
stride_view = pages[::1024]                                  # one word per 4 KiB


def timed(pre, reps=12):
    tot = 0.0
    for _ in range(reps):
        pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.linear(A, W, b, out=out, act=1)
        e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e) * 1e3
    return tot / reps


for _ in range(3):
    ops.linear(A, W, b, out=out, act=1)
torch.cuda.synchronize()
print(f"back to back                       {timed(lambda: None):7.1f} us")
print(f"after a 1 GB fill                  {timed(lambda: big.fill_(0.5)):7.1f} us")
print(f"after one word per 4 KiB of 8 GB   {timed(lambda: stride_view.sum()):7.1f} us")
print(f"after both                         {timed(lambda: (big.fill_(0.5), stride_view.sum())):7.1f} us")
print(f"back to back again                 {timed(lambda: None):7.1f} us")
