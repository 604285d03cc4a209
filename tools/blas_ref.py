"""Context only (NOT part of the product path): what the vendor libraries that ship in this image reach on the headline GEMM
and attention shapes - torch.matmul (hipBLASLt / rocBLAS) and F.scaled_dot_product_attention - so that the roofline
fractions of the hand-written kernels can be read against an independent implementation on the same box."""
import torch
import torch.nn.functional as F

dev, dt = "cuda:0", torch.float16


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


for name, M, N, K in (("fc1", 4096, 4608, 1152), ("fc2", 4096, 1152, 4608), ("qkv", 4096, 3456, 1152), ("proj", 4096, 1152, 1152),
                      ("kv", 2740, 64512, 768), ("fc1 b8", 32768, 4608, 1152), ("proj b8", 32768, 1152, 1152)):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    b = torch.randn(N, device=dev).to(dt)
    us = t(lambda: F.linear(A, W, b))
    print(f"torch F.linear {name:8s} {M}x{N}x{K}: {us:7.2f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF/s", flush=True)
for name, B, H, Nq, Nk, d in (("self", 2, 16, 2048, 2048, 72), ("cross", 2, 16, 2048, 1370, 72), ("self d64", 2, 16, 2048, 2048, 64),
                              ("self d128", 2, 16, 2048, 2048, 128)):
    q = torch.randn(B, H, Nq, d, device=dev).to(dt)
    k = torch.randn(B, H, Nk, d, device=dev).to(dt)
    v = torch.randn(B, H, Nk, d, device=dev).to(dt)
    try:
        us = t(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"torch SDPA {name:9s} {B * H}x{Nq}x{Nk}x{d}: {us:7.2f} us  {4.0 * B * H * Nq * Nk * d / us / 1e6:6.0f} TF/s", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"torch SDPA {name}: {type(ex).__name__}: {ex}", flush=True)
