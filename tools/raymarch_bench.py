"""Timing of the forward ray marcher at preview scale: 518 x 518 rays, 2048 primitives with 8^3 RGBA payloads, the
shipped marcher settings (volradius 10000, dt 1.0 -> 1e-4 of the normalised volume per step; inference_dit.yml:73-75)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd.raymarch import RayMarcher

dev = "cuda:0"
g = torch.Generator().manual_seed(11)
K, S, R = 2048, 8, int(os.environ.get("RES", "518"))
volradius, dt = 10000.0, float(os.environ.get("DT", "1.0"))
half = 0.03 + 0.05 * torch.rand(1, K, 1, generator=g)                      # half extents in the [-1, 1] volume
pos = (1.2 * torch.rand(1, K, 3, generator=g) - 0.6) * volradius
rot = torch.eye(3).expand(1, K, 3, 3).contiguous()
scale = (1.0 / half).expand(-1, -1, 3).contiguous()
rgba = torch.rand(1, K, 4, S, S, S, generator=g)
rgba[:, :, 3] *= 40.0
Rt = torch.tensor([[[1.0, 0, 0, 0], [0, -1.0, 0, 0], [0, 0, -1.0, 5 * volradius]]])
Kc = torch.tensor([[[2084.95 * R / 1024, 0, R / 2], [0, 2084.95 * R / 1024, R / 2], [0, 0, 1]]])
m = RayMarcher(R, R, volradius, dt=dt).eval()
args = [t.to(dev) for t in (rgba, pos, rot, scale, Kc, Rt)]
out = m(*args)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3):
    out = m(*args)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 3
a = out["rgba_image"][0, 3]
print(f"raymarch {R}x{R}, K={K}, dt/volradius={dt/volradius:g}: {ms:9.2f} ms per view   (coverage {float((a > 0).float().mean()):.2f}, "
      f"saturated {float((a >= 0.999).float().mean()):.2f})", flush=True)
