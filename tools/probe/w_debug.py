import os, sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__; __graft_entry__.build()
from topia_xl_amd import ops, _lib
dev = "cuda:0"
torch.manual_seed(0)
for (M, N, K, act) in ((520, 288 * 96, 1152, 0), (512, 288 * 96, 1152, 0), (2048, 4608, 1152, 0), (8192, 4608, 256, 0), (8192, 4608, 1152, 1)):
    A = torch.randn(M, K, device=dev).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half(); b = torch.randn(N, device=dev).half()
    ref = (A.double() @ W.double().t() + b.double())
    if act: ref = torch.nn.functional.gelu(ref.half().float(), approximate="tanh").double()
    for rep in range(2):
        got = ops.linear(A, W, b, act=act).double()
        err = (got - ref).abs()
        bad = err > 0.05
        mt, nt = (M + 255) // 256, N // 288
        tiles = []
        for mi in range(mt):
            for ni in range(nt):
                blk = bad[mi * 256:(mi + 1) * 256, ni * 288:(ni + 1) * 288]
                if blk.any():
                    rows = blk.any(1).nonzero().flatten(); cols = blk.any(0).nonzero().flatten()
                    tiles.append((mi, ni, int(blk.sum()), int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max())))
        print(M, N, K, "act", act, "rep", rep, _lib.load().primx_last_gemm_kernel().decode(), "bad elements", int(bad.sum()), "bad tiles", len(tiles), tiles[:6], flush=True)
