"""Micro-benchmark of the big-tile GEMM kernels at the large-batch shapes (T = 32768 tokens: BASELINE configs[2] / [3] / [4] per GPU)
and at the headline's T = 4096: Linear + GELU (fc1), gate-residual (proj, fc2), plain Linear - HIP-event timing over rotating
buffers (8 output / input sets, so that nothing stays resident from one launch to the next), and the kernel the library selected.
    PRIMX_LIB=<other build> python tools/gemm_bench_big.py      # same-box A/B against another library of the same ABI
(written for the round-4 persistent-kernel experiment, whose PRIMX_GEMM_W switch is gone with the kernel)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import _lib, ops

dev, dt = "cuda:0", torch.float16
reps = int(os.environ.get("REPS", "16"))
NB = 4
shapes = [("fc1+gelu", "lin", 32768, 4608, 1152), ("proj", "gr", 32768, 1152, 1152), ("fc2", "gr", 32768, 1152, 4608),
          ("fc1+gelu", "lin", 8192, 4608, 1152), ("fc1+gelu", "lin", 4096, 4608, 1152), ("proj", "gr", 8192, 1152, 1152), ("fc2", "gr", 8192, 1152, 4608)]
only = os.environ.get("ONLY")          # e.g. ONLY=32768: the shapes with that M
for name, kind, M, N, K in shapes:
    if only and str(M) not in only.split(","):
        continue
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(NB)]
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev).to(dt)
    if kind == "lin":
        outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(NB)]
        fn = lambda i: ops.linear(As[i % NB], W, b, out=outs[i % NB], act=1)
    else:
        gate = (torch.randn(2, N, device=dev) * 0.5).to(dt)
        xs = [torch.randn(M, N, device=dev) for _ in range(NB)]
        fn = lambda i: ops.linear_gate_residual(As[i % NB], W, b, gate, xs[i % NB], M // 2)
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f"{name:9s} {M:6d}x{N:5d}x{K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {_lib.load().primx_last_gemm_kernel().decode()}", flush=True)
    del As, W
    torch.cuda.empty_cache()
