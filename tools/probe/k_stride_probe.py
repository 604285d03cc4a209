"""Does the row stride of the K-contiguous operands matter (L2 channel mapping)?  4096 x 1152 GEMM at K = 4608 (stride
9216 B = 36 x 256) against K = 4480 / 4736 (35 / 37 x 256): time per 64-wide k-tile."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops
dev, dt = "cuda:0", torch.float16
M, N = 4096, 1152
for K in (4480, 4608, 4736, 4608, 4480, 4736, 1152, 1088, 1216):
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    for _ in range(3): ops.linear(A, W, b, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): ops.linear(A, W, b, out=out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 30
    print(f"K={K:5d} stride {K*2:5d} B = {K*2/256:6.2f} x 256   {us:7.2f} us   {us/(K//64):6.3f} us per k-tile   {2.0*M*N*K/us/1e6:6.0f} TF", flush=True)
