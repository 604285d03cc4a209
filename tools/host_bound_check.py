"""Is the DDIM step host-bound, and what would a hipGraph replay of a forward buy?  (a) the time Python needs to ENQUEUE steps against
the time the GPU needs to run them; (b) one PLANNED forward_with_cfg (the path a sampling loop runs: LayerNorm fold, modulation from
the loop's table) eager against its hipGraph replay - the replay removes every host-side gap, what is left between two kernels is
the packet processor's own dependent-launch cost."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__
__graft_entry__.build()
import topia_xl_amd as pkg

dev = torch.device("cuda:0")
with torch.device(dev):
    model = pkg.DiT(**bench.XL).eval()
with torch.no_grad():
    for p in model.parameters():
        p.normal_(0.0, 0.02)
x = torch.randn(1, 2048, 68, device=dev)
y = torch.randn(1, 1370, 768, device=dev)
d = pkg.create_diffusion("ddim25", noise_schedule="squaredcos_cap_v2", parameterization="v")
kw = dict(y=y, cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
stream = bench.step_stream(d, model, x, kw)
for _ in range(28):
    next(stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(25):
    next(stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"sampling loop: enqueue {1e3 * (t1 - t0) / 25:.2f} ms/step   total {1e3 * (t2 - t0) / 25:.3f} ms/step   (the enqueue time includes "
      "back-pressure: HIP blocks the host once a few hundred launches are queued)")
# without back-pressure: ONE step at a time, the queue empty when its enqueue starts - with the forward's blocks issued by the library
# from one foreign call (primx_dit_blocks_fold, the default) and by the Python block loop (DiT.blocks_call = False)
for bc in (True, False, True, False):
    model.blocks_call = bc
    one = []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        next(stream)
        one.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    one.sort()
    print(f"blocks_call={bc}: one step enqueued into an empty queue: median {1e3 * one[len(one) // 2]:.2f} ms, min {1e3 * one[0]:.2f} ms of host "
          "time (the GPU needs ~8.3)", flush=True)
model.blocks_call = True
stream.close()

t = torch.full((1,), 480, device=dev, dtype=torch.int64)
model.plan_timesteps(t)
model.select_planned_timestep(0)


def fwd():
    return model.forward_with_cfg(x, t, **kw)


for _ in range(3):
    out = fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    out = fwd()
torch.cuda.synchronize()
eager = 1e3 * (time.perf_counter() - t0) / 20
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2):
        fwd()
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out_g = fwd()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    rep = 1e3 * (time.perf_counter() - t0) / 20
    print(f"planned forward_with_cfg: eager {eager:.3f} ms   hipGraph replay {rep:.3f} ms   max|graph - eager| = "
          f"{float((out_g.float() - out.float()).abs().max()):.3e}")
except Exception as ex:
    print(f"planned forward_with_cfg: eager {eager:.3f} ms   graph capture failed:", repr(ex)[:300])
