"""Is the DDIM step host-bound?  Compare the time Python needs to ENQUEUE steps with the time the GPU needs to run them."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__
__graft_entry__.build()
import topia_xl_amd as pkg
dev = torch.device("cuda:0")
with torch.device(dev):
    model = pkg.DiT(**bench.XL).eval()
bench.random_init_(model, 42)
x = torch.randn(1, 2048, 68, device=dev); y = torch.randn(1, 1370, 768, device=dev)
d = pkg.create_diffusion("ddim25", noise_schedule="squaredcos_cap_v2", parameterization="v")
kw = dict(y=y, cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
stream = bench.step_stream(d, model, x, kw)
for _ in range(3): next(stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): next(stream)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step   total {1e3*(t2-t0)/10:.2f} ms/step")
# CUDA-graph replay of forward_with_cfg
t = torch.full((1,), 960, device=dev, dtype=torch.int64)
out = model.forward_with_cfg(x, t, **kw); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(2): model.forward_with_cfg(x, t, **kw)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out_g = model.forward_with_cfg(x, t, **kw)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay of forward_with_cfg: {1e3*(time.perf_counter()-t0)/10:.2f} ms   max|graph-eager| = {float((out_g.float()-out.float()).abs().max()):.3e}")
except Exception as ex:
    print("graph capture failed:", repr(ex)[:300])
