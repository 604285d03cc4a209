"""Forward ray marcher: HIP kernels vs the oracle restatement of the reference's CUDA kernels
(parity unpinned - the reference extension cannot be built here, see oracle/raymarch_ref.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import raymarch_ref

DEV = "cuda:0"


def _scene(seed=3, N=2, K=24, S=8, H=40, W=36):
    g = torch.Generator().manual_seed(seed)
    pos = 0.55 * (2 * torch.rand(N, K, 3, generator=g) - 1)
    rv = torch.randn(N, K, 3, generator=g)
    th = rv.norm(dim=-1, keepdim=True).clamp(min=1e-6)
    ax = rv / th
    Kx = torch.zeros(N, K, 3, 3)
    Kx[..., 0, 1], Kx[..., 0, 2], Kx[..., 1, 0] = -ax[..., 2], ax[..., 1], ax[..., 2]
    Kx[..., 1, 2], Kx[..., 2, 0], Kx[..., 2, 1] = -ax[..., 0], -ax[..., 1], ax[..., 0]
    rot = torch.eye(3) + torch.sin(th)[..., None] * Kx + (1 - torch.cos(th))[..., None] * (Kx @ Kx)
    scale = 1.0 / (0.12 + 0.2 * torch.rand(N, K, 3, generator=g))          # inverse half extents
    rgba = torch.rand(N, K, 4, S, S, S, generator=g)
    rgba[:, :, 3] = 6.0 * rgba[:, :, 3] ** 2                                # opacities large enough to saturate some rays
    ang = torch.tensor([0.3, -0.8])[:N]
    R = torch.stack([torch.tensor([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]) for a in ang])
    campos = torch.stack([R[i].t() @ torch.tensor([0.0, 0.1, -3.0]) for i in range(N)])
    RT = torch.cat([R, (-R @ campos[..., None])], dim=-1)
    Kc = torch.tensor([[[1.4 * W, 0, W / 2], [0, 1.4 * W, H / 2], [0, 0, 1]]]).repeat(N, 1, 1)
    return rgba, pos, rot, scale, Kc, RT, H, W


@pytest.mark.gpu
def test_raydirs_and_march_against_oracle():
    import __graft_entry__
    __graft_entry__.build()
    from topia_xl_amd import raymarch as rm
    rgba, pos, rot, scale, Kc, RT, H, W = _scene()
    volradius, dt = 1.0, 0.02
    m = rm.RayMarcher(H, W, volradius, dt=dt).eval()
    out = m(rgba.to(DEV), pos.to(DEV), rot.to(DEV), scale.to(DEV), Kc.to(DEV), RT.to(DEV))
    img = out["rgba_image"].cpu()
    assert img.shape == (2, 4, H, W) and out["pixel_coords"].shape == (2, H, W, 2)
    # rays
    campos, camrot, focal, princpt = raymarch_ref.convert_camera(RT, Kc)
    pc = out["pixel_coords"].cpu()
    rp, rd, tmm = raymarch_ref.compute_raydirs(campos, camrot, focal, princpt, pc, volradius)
    grp, grd, gtm = rm.compute_raydirs(campos.to(DEV), camrot.to(DEV), focal.to(DEV), princpt.to(DEV), pc.to(DEV), volradius)
    assert (grp.cpu() - rp).abs().max() < 1e-6 and (grd.cpu() - rd).abs().max() < 1e-5
    fin = torch.isfinite(tmm) & torch.isfinite(gtm.cpu())
    assert (gtm.cpu()[fin] - tmm[fin]).abs().max() < 1e-4
    # image: march the oracle on the DEVICE's rays (so that step alignment is compared, not ray rounding)
    ref = raymarch_ref.raymarch(grp.cpu(), grd.cpu(), gtm.cpu(), dt / volradius, pos / volradius, rot, scale,
                                rgba.permute(0, 1, 3, 4, 5, 2).contiguous(), 8.0, 8.0)
    ref = ref.permute(0, 3, 1, 2)
    err = (img - ref).abs()
    assert float(ref[:, 3].max()) > 0.99 and float((ref[:, 3] > 0).float().mean()) > 0.2    # saturated and empty rays both present
    # fast-math exp/pow in the fade and one-ulp step-boundary decisions: a handful of pixels may differ visibly
    assert float(err.mean()) < 2e-4, float(err.mean())
    assert float((err > 5e-3).float().mean()) < 2e-3, float((err > 5e-3).float().mean())


def test_cpu_tensors_are_refused():
    from topia_xl_amd import raymarch as rm
    rgba, pos, rot, scale, Kc, RT, H, W = _scene(N=1, K=2, H=8, W=8)
    with pytest.raises(RuntimeError):
        rm.RayMarcher(H, W, 1.0).eval()(rgba, pos, rot, scale, Kc, RT)
