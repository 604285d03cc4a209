"""Micro-benchmark of primx_linear_heads at the DiT-XL shapes (qkv, to_q, batched kv)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops
from topia_xl_amd._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT
dev, dt, H, dh = "cuda:0", torch.float16, 16, 72
def run(name, M, rpb, K, kinds, n_rep):
    N = n_rep * len(kinds) * H * dh
    A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt); b = torch.randn(N, device=dev).to(dt)
    B = M // rpb
    dsts = [ops.alloc_heads(n_rep * B, H, rpb, dh, k, dt, dev, 128) for k in kinds]
    n_pad = dsts[0].shape[3] if kinds[0] == HEADS_VT else dsts[0].shape[2]
    f = lambda: ops.linear_heads(A, W, b, rpb, H, dh, kinds, dsts, n_pad, n_rep=n_rep, rep_batches=B)
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / 20
    print(f"{name:8s} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
run("qkv", 4096, 2048, 1152, [HEADS_ROWS, HEADS_KROWS, HEADS_VT], 1)
run("to_q", 4096, 2048, 1152, [HEADS_ROWS], 1)
run("kv_all", 2740, 1370, 768, [HEADS_KROWS, HEADS_VT], 28)
run("k_only", 2740, 1370, 768, [HEADS_ROWS], 28)
run("v_only", 2740, 1370, 768, [HEADS_VT], 28)
run("kk_all", 2740, 1370, 768, [HEADS_KROWS, HEADS_KROWS], 28)
run("qkk", 4096, 2048, 1152, [HEADS_ROWS, HEADS_KROWS, HEADS_KROWS], 1)
run("qkv_b4", 16384, 2048, 1152, [HEADS_ROWS, HEADS_KROWS, HEADS_VT], 1)
run("to_q_b4", 16384, 2048, 1152, [HEADS_ROWS], 1)
run("kv_pad", 3072, 1536, 768, [HEADS_KROWS, HEADS_VT], 28)   # 1370 conditioning rows padded to 1536 (dit.py)
