"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the forward ray marcher.

PARITY PINNED against the reference's own executable PyTorch statement of these kernels: tests/golden/raymarch.npz is
produced by tests/golden/make_golden_raymarch.py, which RUNS the torch march loop of mvpraymarch.py:301-461 (seed 1112)
and the torch ray-direction code of utils/utils.py:73-148 (seed 1113) unmodified; tests/test_raymarch.py checks this
file against it (max-abs 1.1e-6 on the image, 1.2e-7 on directions).  The CUDA extension itself (nvcc, sm_70, 32-lane
warp intrinsics) cannot be built or run here; this file restates the kernels' arithmetic from their sources:

  compute_raydirs      dva/mvp/extensions/utils/utils_kernel.cu:15-56
  convert_camera       dva/ray_marcher.py:22-30
  hit interval         dva/mvp/extensions/mvpraymarch/utils.h:749-770  (per ray: union of its own local slab intervals)
  marching / start     dva/mvp/extensions/mvpraymarch/mvpraymarch_subset_kernel.h:58-86
  transform            .../primtransf.h:113-126   y = (R^T-rows combination of (x - pos)) * scale, valid iff |y| < 1 strictly
  sample + fade        .../primsampler.h:37-62, utils.h:406-500  (trilinear, align_corners, zeros; alpha *= exp(-fs * sum |y|^fe))
  accumulation         .../primaccum.h:63-78
Dense formulation: every ray against every primitive in index order (the hit list of the kernel only removes
primitives whose box the ray never enters, which `valid` rejects anyway).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def convert_camera(Rt: Tensor, K: Tensor):
    """(camera centre -R^T t, R, diag focal, principal point)   dva/ray_marcher.py:22-30 + 195."""
    rot = Rt[:, :3, :3]
    centre = -(rot.transpose(1, 2) @ Rt[:, :3, 3:4])[..., 0]
    return centre, rot, torch.stack([K[:, 0, 0], K[:, 1, 1]], dim=-1), K[:, :2, 2]


def compute_raydirs(viewpos, viewrot, focal, princpt, pixelcoords, volradius):
    raypos = (viewpos / volradius)[:, None, None, :].expand(-1, pixelcoords.shape[1], pixelcoords.shape[2], -1)
    pc = (pixelcoords - princpt[:, None, None, :]) / focal[:, None, None, :]
    d = torch.cat([pc, torch.ones_like(pc[..., :1])], dim=-1)
    d = torch.einsum("nhwi,nij->nhwj", d, viewrot)                  # row0 * d.x + row1 * d.y + row2 * d.z
    d = d / d.norm(dim=-1, keepdim=True)
    t1, t2 = (-1.0 - raypos) / d, (1.0 - raypos) / d
    tmin = torch.minimum(t1, t2).amax(-1)
    tmax = torch.maximum(t1, t2).amin(-1)
    return raypos.contiguous(), d, torch.stack([tmin.clamp(min=0.0), tmax], dim=-1)


def raymarch(raypos, raydir, tminmax, stepsize, primpos, primrot, primscale, template_chlast, fadescale, fadeexp):
    """raypos / raydir [N,H,W,3], tminmax [N,H,W,2], prim* [N,K,..], template_chlast [N,K,D,H,W,4] -> rgba [N,H,W,4]."""
    N, H, W, _ = raypos.shape
    K = primpos.shape[1]
    out = torch.zeros(N, H, W, 4)
    tpl = template_chlast.permute(0, 1, 5, 2, 3, 4).contiguous()    # [N,K,4,D,H,W] for F.grid_sample
    for n in range(N):
        rp, rd = raypos[n].reshape(-1, 3), raydir[n].reshape(-1, 3)
        tmin0, tmax0 = tminmax[n].reshape(-1, 2).unbind(-1)
        R, pos, scl = primrot[n], primpos[n], primscale[n]

        def local(x):                                                # [rays, K, 3]
            return torch.einsum("rki,kij->rkj", x[:, None, :] - pos[None], R) * scl[None]

        r0 = local(rp)
        r1 = torch.einsum("ri,kij->rkj", rd, R) * scl[None]
        ird = 1.0 / r1
        a, b = (-1.0 - r0) * ird, (1.0 - r0) * ird
        trmin, trmax = torch.minimum(a, b).amax(-1), torch.maximum(a, b).amin(-1)
        hit = trmin <= trmax
        rtmin = torch.where(hit, trmin, torch.full_like(trmin, float("inf"))).amin(1).clamp(min=tmin0)
        rtmax = torch.where(hit, trmax, torch.full_like(trmax, float("-inf"))).amax(1)
        rtmax = torch.minimum(rtmax, tmax0)
        incs = torch.floor((rtmin - tmin0) / stepsize)
        incs = torch.where(torch.isfinite(incs), incs, torch.zeros_like(incs))
        active = torch.isfinite(rtmin)
        t = tmin0 + incs * stepsize
        x = rp + rd * tmin0[:, None]
        x = x + rd * incs[:, None] * stepsize
        acc = torch.zeros(rp.shape[0], 4)
        sat = torch.zeros(rp.shape[0], dtype=torch.bool)
        while bool(((t <= rtmax + 1e-5) & ~sat & active).any()):
            y = local(x)
            for k in range(K):
                yk = y[:, k]
                m = (yk.abs() < 1).all(-1) & ~sat & (t < rtmax + 1e-5) & active
                if not bool(m.any()):
                    continue
                ys = yk[m]
                s = F.grid_sample(tpl[n, k][None], ys.reshape(1, -1, 1, 1, 3), mode="bilinear", padding_mode="zeros",
                                  align_corners=True).reshape(4, -1).t()
                fade = torch.exp(-fadescale * (ys.abs() ** fadeexp).sum(-1))
                alpha = s[:, 3] * fade
                newalpha = acc[m, 3] + alpha * stepsize
                contrib = newalpha.clamp(max=1.0) - acc[m, 3]
                upd = acc[m]
                upd[:, :3] += s[:, :3] * contrib[:, None]
                upd[:, 3] += contrib
                acc[m] = upd
                idx = m.nonzero()[:, 0]
                sat[idx[newalpha >= 1.0]] = True
            t = t + stepsize
            x = x + rd * stepsize
        out[n] = acc.reshape(H, W, 4)
    return out
