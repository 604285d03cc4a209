"""K-sweep of the T144 GEMM at M=4096, N=1152: intercept = fixed cost per launch, slope = cost per 64-wide k-tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops
dev, dt = "cuda:0", torch.float16
M, N = 4096, int(os.environ.get("N", "1152"))
for K in (64, 128, 256, 512, 1152, 2304, 4608):
    A = torch.randn(M, K, device=dev).to(dt); W = torch.randn(N, K, device=dev).to(dt); b = torch.randn(N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    x = torch.zeros(M, N, device=dev); gate = torch.randn(2, N, device=dev).to(dt)
    res = {}
    for name, fn in (("linear", lambda: ops.linear(A, W, b, out=out)),
                     ("gate_res", lambda: ops.linear_gate_residual(A, W, b, gate, x, 2048))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50): fn()
        e.record(); torch.cuda.synchronize()
        res[name] = s.elapsed_time(e) * 1e3 / 50
    print(f"K={K:5d} ({K//64:3d} k-tiles)  linear {res['linear']:7.2f} us   gate_residual {res['gate_res']:7.2f} us", flush=True)
# an empty-ish kernel for the launch floor
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
z = torch.zeros(1024, device=dev, dtype=dt)
s.record()
for _ in range(50): ops.cast16(torch.zeros(1024, device=dev), dt) if False else ops.silu_cast(x[:1], dt)
e.record(); torch.cuda.synchronize()
print(f"tiny kernel back-to-back: {s.elapsed_time(e)*1e3/50:.2f} us per launch (includes a torch.empty)")
