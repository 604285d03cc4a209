"""Timing of the PrimSDF field query at mesh-extraction scale: the 256^3 marching-cubes lattice (inference.py:106-116)
against 2048 primitives with 8^3 payloads."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd.primsdf import PrimSDF

dev = "cuda:0"
gen = torch.Generator().manual_seed(5)
P, S, R = 2048, 8, int(os.environ.get("RES", "256"))
m = PrimSDF(num_prims=P, prim_shape=S).eval()
m.srt_param.data = torch.cat([0.03 + 0.05 * torch.rand(P, 1, generator=gen), 1.6 * torch.rand(P, 3, generator=gen) - 0.8], dim=1)
m.feat_param.data = torch.randn(P, 6 * S ** 3, generator=gen) * 0.5 + 0.3
m.to(dev)
xx = torch.linspace(-1, 1, R, device=dev)
pts = torch.stack(torch.meshgrid(xx, xx, xx, indexing="ij"), dim=-1).reshape(-1, 3)
chunks = torch.split(pts, 1 << 21)
for c in chunks[:2]:
    m.query(c)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
cov = 0
for c in chunks:
    out = m.query(c)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e)
n = pts.shape[0]
print(f"PrimSDF query {R}^3 = {n/1e6:.1f} M points x {P} primitives: {ms:8.2f} ms  ({n / ms / 1e3:8.1f} M points/s, "
      f"{n * P / ms / 1e9:6.2f} T point-primitive tests/s)", flush=True)
