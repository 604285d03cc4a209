"""gemm192d_dma_kernel (PRIMX_GEMM_D192, csrc/gemm.hip) against the default kernels: the same sums in the same order and the same rounding
points, so the results must agree to the last bit.  Run once with PRIMX_GEMM_D192=2 SAVE=path and once without, or let this script
spawn the reference run itself."""
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from topia_xl_amd import _lib, ops

dev = "cuda:0"
SHAPES = [(32768, 4608, 1152, "lin"), (32768, 1152, 1152, "gr"), (32768, 1152, 4608, "gr"), (8192, 4608, 1152, "lin"), (16384, 1152, 832, "gr"),
          (24576, 2304, 1152, "lin0")]


def run():
    outs = {}
    for dt in (torch.float16, torch.bfloat16):
        for (M, N, K, kind) in SHAPES:
            g = torch.Generator(device="cpu").manual_seed(M + N + K)
            A = torch.randn(M, K, generator=g).to(dev).to(dt)
            W = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(dt)
            b = torch.randn(N, generator=g).to(dev).to(dt)
            if kind.startswith("lin"):
                o = ops.linear(A, W, b if kind == "lin" else None, act=1 if kind == "lin" else 0)
            else:
                gate = (torch.randn(M // 2048, N, generator=g) * 0.5).to(dev).to(dt)
                o = torch.randn(M, N, generator=g).to(dev)
                ops.linear_gate_residual(A, W, b, gate, o, 2048)
            name = _lib.load().primx_last_gemm_kernel().decode()
            outs[(str(dt), M, N, K, kind)] = (o.cpu(), name)
            print(dt, M, N, K, kind, name, flush=True)
    return outs


if os.environ.get("SAVE"):
    torch.save(run(), os.environ["SAVE"])
    sys.exit(0)
with tempfile.TemporaryDirectory() as d:
    f = os.path.join(d, "ref.pt")
    env = {k: v for k, v in os.environ.items() if k != "PRIMX_GEMM_D192"}
    env["SAVE"] = f
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = torch.load(f)
os.environ["PRIMX_GEMM_D192"] = os.environ.get("PRIMX_GEMM_D192", "2")
got = run()
bad = 0
for k, (o, name) in got.items():
    ro, rname = ref[k]
    same = torch.equal(o, ro)
    nd = int((o != ro).sum())
    print(("OK  " if same else "DIFF"), k, name, "vs", rname, "" if same else f"{nd} elements differ, max abs {float((o.double() - ro.double()).abs().max()):.3e}")
    bad += not same
    if "192d" not in name:
        print("   (the launch did not take gemm192d)")
print("bit-identical" if not bad else f"{bad} case(s) differ")
sys.exit(1 if bad else 0)
