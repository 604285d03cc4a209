"""VAE.decode benchmark: one sample = 2048 primitives (1x4^3 -> 6x8^3), per-op breakdown via HIP events."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
import topia_xl_amd as pkg
from topia_xl_amd import ops

dev = "cuda:0"
CFG = dict(in_channels=6, latent_channels=1, out_channels=6, down_channels=[32, 256], mid_attention=True,
           up_channels=[256, 32], layers_per_block=2, gradient_checkpointing=False)
with torch.device(dev):
    vae = pkg.VAE(**CFG).eval()
P = int(os.environ.get("P", "2048"))
z = torch.randn(P, 1, 4, 4, 4, device=dev)
for _ in range(2):
    out = vae.decode(z)
torch.cuda.synchronize()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    out = vae.decode(z, denormalize=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
flops = 2.2428e9 * P
print(f"VAE.decode P={P}: {dt*1e3:.2f} ms  = {flops/dt/1e12:.1f} TFLOP/s algorithmic (2.2428 GF/prim), {P/dt:.0f} prims/s, finite={bool(torch.isfinite(out).all())}")
# per-op breakdown
names = ["groupnorm_silu", "conv3d_k3", "conv_in", "convtranspose_k2s2", "linear_residual", "linear_heads", "attention", "vae_output"]
acc = {}
orig = {n: getattr(ops, n) for n in names}
def wrap(n):
    f = orig[n]
    def g(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = f(*a, **k); e.record()
        acc.setdefault(n, []).append((s, e))
        return r
    return g
for n in names:
    setattr(ops, n, wrap(n))
vae.decode(z, denormalize=True)
torch.cuda.synchronize()
for n, ev in sorted(acc.items(), key=lambda kv: -sum(s.elapsed_time(e) for s, e in kv[1])):
    print(f"  {n:20s} calls={len(ev):3d}  total {sum(s.elapsed_time(e) for s, e in ev):8.3f} ms")
