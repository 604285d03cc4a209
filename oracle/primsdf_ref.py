"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of PrimSDF.forward (models/primsdf.py).

  prim_weight        primsdf.py:103-107   relu(1 - ||(x - pos) / scale||_inf), normalised by (sum + 1e-6)
  grid_sample_feat   primsdf.py:66-101    trilinear sample (align_corners=True, x -> W) of every covering primitive's
                                          [6, S, S, S] volume, weighted sum; eval mode: nearest primitive / nearest grid
                                          point fill of the SDF channel for uncovered points
  forward            primsdf.py:52-64     sdf, clip(tex), clip(mat)
Written densely (every point against every primitive) instead of with the reference's masked gather: same arithmetic
per (point, primitive) pair.  Pinned by tests/golden/primsdf.npz (outputs of the real module).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def primsdf_forward(srt: Tensor, feat: Tensor, x: Tensor, S: int, training: bool = False) -> Dict[str, Tensor]:
    P, C = srt.shape[0], feat.shape[1] // S ** 3
    pos, scale = srt[:, 1:4], srt[:, 0:1]
    sp = (x[:, None, :] - pos[None]) / scale[None]                                   # [n, P, 3]
    w = F.relu(1 - sp.abs().amax(dim=-1))
    wn = w / (w.sum(dim=-1, keepdim=True) + 1e-6)
    out = torch.zeros(x.shape[0], C)
    vol = feat.reshape(P, C, S, S, S)
    for p in range(P):                                                              # small P in the tests
        m = wn[:, p] > 0
        if m.any():
            g = sp[m, p].reshape(1, -1, 1, 1, 3)
            s = F.grid_sample(vol[p:p + 1], g, mode="bilinear", padding_mode="zeros", align_corners=True)
            out[m] += s.reshape(C, -1).t() * wn[m, p][:, None]
    if not training:
        un = w.sum(1) <= 0
        if un.any():
            xu = x[un]
            near = torch.norm(xu[:, None, :] - pos[None], p=2, dim=-1).argmin(1)
            xx = torch.linspace(-1, 1, S)
            mx, my, mz = torch.meshgrid(xx, xx, xx, indexing="ij")
            local = torch.stack((mz, my, mx), dim=-1).reshape(-1, 3)
            cand = pos[near][:, None, :] + scale[near][..., None] * local[None]
            d = torch.norm(xu[:, None, :] - cand, p=2, dim=-1)
            dmin, j = d.min(1)
            s0 = feat[:, :S ** 3][near, j]
            out[un, 0] = s0 + dmin * torch.sign(s0)
    return {"sdf": out[:, 0:1], "tex": out[:, 1:4].clip(0, 1), "mat": out[:, 4:6].clip(0, 1)}
