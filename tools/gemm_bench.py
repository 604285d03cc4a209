"""Micro-benchmark of the GEMM entry points at the DiT-XL shapes (T = 4096 tokens) - HIP-event timing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops

dev = "cuda:0"
dt = torch.float16
reps = int(os.environ.get("REPS", "20"))
shapes = [("proj", 4096, 1152, 1152), ("fc2", 4096, 1152, 4608), ("qkv", 4096, 3456, 1152), ("fc1", 4096, 4608, 1152),
          ("kv", 2740, 2304, 768), ("big_proj", 32768, 1152, 1152), ("big_fc1", 32768, 4608, 1152), ("n128", 4096, 1024, 1152), ("adaLN", 2, 28 * 9 * 1152 + 2 * 1152, 1152)]
only = os.environ.get("ONLY")
for name, M, N, K in shapes:
    if only and name not in only.split(","):
        continue
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    for _ in range(3):
        ops.linear(A, W, b, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.linear(A, W, b, out=out)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f"{name:9s} M={M:6d} N={N:5d} K={K:5d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
