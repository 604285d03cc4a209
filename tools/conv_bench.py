"""Micro-benchmark of the register-resident 4^3 convolution (csrc/conv3.hip) at the decode shape: P primitives x 64 voxels x
256 -> 256 channels, random activations (DVFS-honest).  REPS launches; prints us per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__  # noqa: E402

__graft_entry__.build()
from topia_xl_amd import ops  # noqa: E402

P = int(os.environ.get("P", 2048))
REPS = int(os.environ.get("REPS", 10))
x = torch.randn(P, 64, 256, device="cuda", dtype=torch.float16)
w = torch.randn(256, 6912, device="cuda", dtype=torch.float16) * 0.012
wp = ops.pack_conv3(w, 256)
b = torch.zeros(256, device="cuda", dtype=torch.float16)
for _ in range(3):
    ops.conv3d_k3(x, w, b, 4, Wp=wp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(REPS):
    ops.conv3d_k3(x, w, b, 4, Wp=wp)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / REPS * 1e3
print(f"conv3 s4 256->256 x{P}: {us:.1f} us  ({2.0 * P * 64 * 256 * 6912 / us / 1e6:.0f} TF nominal, {2.0 * P * 64 * 256 * 6912 * 5 / 6 / us / 1e6:.0f} TF executed)")
