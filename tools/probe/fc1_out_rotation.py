"""Does the fc1 epilogue's speed depend on whether the OUTPUT lines are resident in the Infinity Cache?  The same GEMM
(4096 x 4608 x 1152, GELU) with its output rotating over n buffers of 37.7 MB: n = 1 rewrites the same lines every launch (what
an isolated micro-benchmark does), n = 8 (300 MB) never finds them in the 256 MB cache.  Run with PRIMX_GEMM_PROF=1 for the
per-workgroup epilogue cycles, without for HIP-event times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops
dev, dt = "cuda:0", torch.float16
M, N, K = 4096, 4608, 1152
A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
for n in (1, 2, 4, 8, 16):
    outs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(n)]
    for i in range(2 * n): ops.linear(A, W, b, out=outs[i % n], act=1)
    torch.cuda.synchronize()
    reps = 4 * n if os.environ.get("PRIMX_GEMM_PROF") else 40
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps): ops.linear(A, W, b, out=outs[i % n], act=1)
    e.record(); torch.cuda.synchronize()
    print(f"== {n} output buffer(s) ({n * 37.7:.0f} MB): {s.elapsed_time(e) * 1e3 / reps:.1f} us per launch", flush=True, file=sys.stderr)
