"""Micro-benchmark of primx_attention at the DiT-XL shapes (HIP-event timing)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops
from topia_xl_amd._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT

dev, dt, reps = "cuda:0", torch.float16, int(os.environ.get("REPS", "20"))
for name, B, H, nq, nkv, dh in [("self_b1", 2, 16, 2048, 2048, 72), ("cross_b1", 2, 16, 2048, 1370, 72),
                                ("self_b8", 16, 16, 2048, 2048, 72), ("cross_b8", 16, 16, 2048, 1370, 72),
                                ("self_n4096", 2, 16, 4096, 4096, 72), ("self_n4096_b4", 8, 16, 4096, 4096, 72)]:
    q = torch.randn(B, nq, H, dh, device=dev).to(dt)
    k = torch.randn(B, nkv, H, dh, device=dev).to(dt)
    v = torch.randn(B, nkv, H, dh, device=dev).to(dt)
    Q, K, Vt = ops.pack_heads(q, HEADS_ROWS, 128, "q"), ops.pack_heads(k, HEADS_KROWS, 64, "k"), ops.pack_heads(v, HEADS_VT, 64)
    out = torch.empty(B, nq, H * dh, device=dev, dtype=dt)
    for _ in range(3):
        ops.attention(Q, K, Vt, nq, nkv, dh, dh ** -0.5, out=out)
    torch.cuda.synchronize()
    if B == 2:   # accuracy of this build: two heads against float64
        qd, kd, vd = (t[0, :, :2].double().permute(1, 0, 2) for t in (q, k, v))
        ref = (torch.softmax(qd @ kd.transpose(1, 2) * dh ** -0.5, -1) @ vd).permute(1, 0, 2)
        got = out[0].view(nq, H, dh)[:, :2].double()
        print(f"   max abs err vs float64 (2 heads): {float((got - ref).abs().max()):.3e}   rel-L2 {float((got - ref).norm() / ref.norm()):.3e}")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.attention(Q, K, Vt, nq, nkv, dh, dh ** -0.5, out=out)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    print(f"{name:11s} B={B:3d} nq={nq} nkv={nkv}  {us:8.1f} us  {4.0*B*H*nq*nkv*dh/us/1e6:7.1f} TFLOP/s (algorithmic, dh=72)", flush=True)
