"""Sustained HBM bandwidth of plain device-wide kernels on this box: write-only (fill), read + write (copy), read-only (sum) at
sizes beyond the 256 MB Infinity Cache - the yardstick for the end-of-kernel write bursts of the GEMM epilogues (DESIGN.md
section 4 item 11: fc1's 37.7 MB leave at ~2.2 TB/s inside the step)."""
import torch

dev = "cuda:0"
for mb in (38, 256, 2048):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device=dev)
    b = torch.empty(n, device=dev)
    for name, fn, bytes_ in (("fill  (write)", lambda: a.fill_(1.0), 4 * n), ("copy  (read+write)", lambda: b.copy_(a), 8 * n),
                             ("sum   (read)", lambda: a.sum(), 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / reps
        print(f"{mb:5d} MB  {name:20s} {us:9.1f} us  {bytes_ / us / 1e6:6.2f} TB/s")
