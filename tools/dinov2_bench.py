"""Timing of the DINOv2 conditioner forward (ViT-B/14 + 4 registers, 518 x 518 -> 1370 x 768 tokens), HIP events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import dinov2

dev = "cuda:0"
m = dinov2.vit_base(img_size=518, init_values=1.0).eval()
torch.manual_seed(0)
with torch.no_grad():   # random weights of trained-network-like magnitude (no checkpoints offline)
    for name, p in m.named_parameters():
        if p.dim() > 1 and "token" not in name and "pos_embed" not in name:
            p.copy_(torch.randn_like(p) * p[0].numel() ** -0.5)
        elif name.endswith("weight") or name.endswith("gamma"):
            p.copy_(1.0 + 0.1 * torch.randn_like(p))
        else:
            p.copy_(0.1 * torch.randn_like(p))
m.to(dev)
D, depth, nt, npatch = 768, 12, 1374, 1369
flops_img = depth * (2 * nt * D * 3 * D + 4 * nt * nt * D + 2 * nt * D * D + 16 * nt * D * D) + 2 * npatch * 588 * D
for B in (1, 8):
    x = torch.randn(B, 3, 518, 518, device=dev)
    for _ in range(3):
        m.conditioner_tokens(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        m.conditioner_tokens(x)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"dinov2_vitb14_reg  B={B}: {ms:7.3f} ms per forward  ({B / ms * 1e3:7.1f} images/s, {B * flops_img / ms / 1e9:6.1f} TFLOP/s algorithmic)", flush=True)
