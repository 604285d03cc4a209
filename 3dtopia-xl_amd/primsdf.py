"""PrimSDF field query on MI355X - SURVEY.md section 8(f) row N3 (models/primsdf.py, queried by the mesh / texture
extraction of inference.py:106-116, 180-193).

`PrimSDF` mirrors the reference module's parameters (`srt_param` [P, 1 + 3] = scale + translation, `feat_param`
[P, 6 * S^3]), its properties (`pos`, `scale`, `feat`, `feat_geo`, `feat_tex`, `feat_mat`) and `forward(x) ->
{'sdf', 'tex', 'mat'}` including the inference-time fill of uncovered points; the evaluation itself is one HIP kernel
(`primx_primsdf_query`).  The fitting path (`_init_param`, which needs trimesh and a mesh SDF) is training-side and not
part of this package.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _lib, ops


class PrimSDF(nn.Module):
    def __init__(self, mesh_obj=None, f_sdf=None, geo_fn=None, asset_list=None, num_prims=1024, dim_feat=6, prim_shape=8,
                 init_scale=0.05, sdf2alpha_var=0.005, auto_scale_init=True, init_sampling="uniform"):
        super().__init__()
        if f_sdf is not None or geo_fn is not None or asset_list is not None:
            raise NotImplementedError("primitive fitting (PrimSDF._init_param) is training-side; load fitted parameters")
        self.num_prims, self.dim_feat, self.prim_shape = num_prims, dim_feat, prim_shape
        self.sdf2alpha_var = sdf2alpha_var
        self.srt_param = nn.Parameter(torch.zeros(num_prims, 1 + 3))
        self.feat_param = nn.Parameter(torch.zeros(num_prims, dim_feat * prim_shape ** 3))
        s3 = prim_shape ** 3
        self.geo_start_index, self.geo_end_index = 0, s3
        self.tex_start_index, self.tex_end_index = s3, 4 * s3
        self.mat_start_index, self.mat_end_index = 4 * s3, 6 * s3
        xx = torch.linspace(-1, 1, prim_shape)
        meshx, meshy, meshz = torch.meshgrid(xx, xx, xx, indexing="ij")
        self.local_grid = torch.stack((meshz, meshy, meshx), dim=-1).reshape(-1, 3)   # (primsdf.py:35-41)
        self.__dict__["_lin"] = {}

    # reference properties (primsdf.py:112-137)
    pos = property(lambda self: self.srt_param[:, 1:4])
    scale = property(lambda self: self.srt_param[:, 0:1])
    feat = property(lambda self: self.feat_param)
    feat_geo = property(lambda self: self.feat_param[:, self.geo_start_index:self.geo_end_index])
    feat_tex = property(lambda self: self.feat_param[:, self.tex_start_index:self.tex_end_index])
    feat_mat = property(lambda self: self.feat_param[:, self.mat_start_index:self.mat_end_index])

    def sdf2alpha(self, sdf):
        return torch.exp(-(sdf / self.sdf2alpha_var) ** 2)

    def _linspace(self, device) -> torch.Tensor:
        key = (str(device), self.prim_shape)
        if key not in self._lin:
            self._lin[key] = torch.linspace(-1, 1, self.prim_shape).to(device)   # the reference's own table (primsdf.py:35)
        return self._lin[key]

    def query(self, x: torch.Tensor) -> torch.Tensor:
        """x: [n, 3] fp32 on the HIP device -> [n, dim_feat] = [sdf, clipped tex (3), clipped mat (2)]."""
        if not x.is_cuda:
            raise RuntimeError("PrimSDF.query needs HIP device tensors; there is no CPU path")
        if self.dim_feat != 6 or self.srt_param.shape[0] == 0:
            raise NotImplementedError("the query kernel covers the 6-channel PrimX payload")
        x = x.float().contiguous()
        n = x.shape[0]
        out = torch.empty(n, self.dim_feat, dtype=torch.float32, device=x.device)
        if n == 0:
            return out
        srt = self.srt_param.detach().float().contiguous()
        feat = self.feat_param.detach().float().contiguous()
        _lib.check(_lib.load().primx_primsdf_query(
            ops._dev(x, "x", torch.float32), ops._dev(srt, "srt", torch.float32), ops._dev(feat, "feat", torch.float32),
            ops._dev(self._linspace(x.device), "lin", torch.float32), out.data_ptr(), n, srt.shape[0], self.prim_shape,
            self.dim_feat, 0 if self.training else 1, ops._stream()), "primx_primsdf_query")
        return out

    @ops.on_input_device
    def forward(self, x: torch.Tensor):
        out = self.query(x)
        return {"sdf": out[:, 0:1], "tex": out[:, 1:4], "mat": out[:, 4:6]}
