"""Context only (NOT part of the product path): the vendor GEMM (torch F.linear -> hipBLASLt) at the large-batch shapes, so that its
kernel names, counters and clock can be read next to the hand-written kernels' (tools/gpu/r6_s1.sh)."""
import os
import torch
import torch.nn.functional as F

dev, dt = "cuda:0", torch.float16
reps = int(os.environ.get("REPS", "16"))
for name, M, N, K in (("fc1 b8", 32768, 4608, 1152), ("proj b8", 32768, 1152, 1152), ("fc2 b8", 32768, 1152, 4608), ("qkv b8", 32768, 3456, 1152)):
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(4)]
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev).to(dt)
    for i in range(3):
        F.linear(As[i % 4], W, b)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        F.linear(As[i % 4], W, b)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f"torch F.linear {name:8s} {M}x{N}x{K}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s", flush=True)
    del As
