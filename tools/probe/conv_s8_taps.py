"""Per-tap check of the 8^3 activation-resident convolution: weight nonzero for ONE tap at a time, against F.conv3d."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import ops
from topia_xl_amd.vae import _conv_weight_as_gemm
torch.manual_seed(0)
P, S, Cin, Cout = 2, 8, 256, 32
x = torch.randn(P, Cin, S, S, S).half()
xcl = x.reshape(P, Cin, -1).permute(0, 2, 1).contiguous().cuda()
wfull = (torch.randn(Cout, Cin, 3, 3, 3) * 0.02).half()
for tap in list(range(27)) + [-1]:
    w = torch.zeros_like(wfull)
    if tap >= 0:
        w.view(Cout, Cin, 27)[:, :, tap] = wfull.view(Cout, Cin, 27)[:, :, tap]
    else:
        w = wfull
    ref = F.conv3d(x.double(), w.double(), None, padding=1)
    wk = _conv_weight_as_gemm(w, torch.float16).cuda()
    wp = ops.pack_conv3(wk, Cin)
    got = ops.conv3d_k3(xcl, wk, None, S, Wp=wp).float().cpu().permute(0, 2, 1).reshape(P, Cout, S, S, S).double()
    err = ((got - ref).norm() / ref.norm()).item()
    bad = (got - ref).abs().amax(dim=(0, 1))    # [z, y, x]
    print(f"tap {tap:2d} (dz,dy,dx)=({tap//9-1},{tap//3%3-1},{tap%3-1}) rel {err:.3e}  bad planes z: {[round(float(bad[z].max()),3) for z in range(8)]}")
