"""Forward volumetric ray marcher - SURVEY.md section 8(f) row N2 (dva/ray_marcher.py:RayMarcher and the CUDA extension
behind it).  `RayMarcher` keeps the reference's constructor and `forward(prim_rgba, prim_pos, prim_rot, prim_scale, K, RT)`
-> {"rgba_image": [B, 4, H, W], "pixel_coords"}; the two kernels are `primx_compute_raydirs` and `primx_raymarch`
(csrc/raymarch.hip).  Inference only: no backward pass, no training-time random ray subsampling."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import _lib, ops


def convert_camera_parameters(Rt: torch.Tensor, K: torch.Tensor):
    """World-to-camera [R | t] and pinhole K -> the quantities the ray generator consumes (dva/ray_marcher.py:22-30):
    camera centre c = -R^T t, the rotation itself, the 2 x 2 focal block and the principal point.  (Two tiny batched
    products on a handful of floats: plumbing.)"""
    rot = Rt[:, :3, :3]
    centre = -torch.einsum("nji,nj->ni", rot, Rt[:, :3, 3])
    return {"campos": centre, "camrot": rot, "focal": K[:, :2, :2], "princpt": K[:, :2, 2]}


@ops.on_input_device
def compute_raydirs(viewpos, viewrot, focal, princpt, pixelcoords, volradius):
    N, H, W = pixelcoords.shape[0], pixelcoords.shape[1], pixelcoords.shape[2]
    dev = viewpos.device
    raypos = torch.empty(N, H, W, 3, dtype=torch.float32, device=dev)
    raydir = torch.empty_like(raypos)
    tminmax = torch.empty(N, H, W, 2, dtype=torch.float32, device=dev)
    # the dense fp32 copies must stay referenced until the launch is enqueued: a temporary dropped earlier returns its
    # block to the caching allocator, and the NEXT temporary's copy kernel would overwrite it ahead of our kernel
    keep = [t.float().contiguous() for t in (viewpos, viewrot, focal, princpt, pixelcoords)]
    ptrs = [ops._dev(t, name, torch.float32) for t, name in zip(keep, ("viewpos", "viewrot", "focal", "princpt", "pixelcoords"))]
    _lib.check(_lib.load().primx_compute_raydirs(*ptrs, float(volradius), raypos.data_ptr(), raydir.data_ptr(),
                                                 tminmax.data_ptr(), N, H, W, ops._stream()), "primx_compute_raydirs")
    del keep
    return raypos, raydir, tminmax


def mvpraymarch(raypos, raydir, stepsize, tminmax, primtransf, template, fadescale=8.0, fadeexp=8.0):
    """template: [N, K, TD, TH, TW, 4] channels-last RGBA; primtransf = (pos [N,K,3], rot [N,K,3,3], scale [N,K,3])."""
    primpos, primrot, primscale = (t.float().contiguous() for t in primtransf)
    template = template.float().contiguous()
    N, H, W = raypos.shape[:3]
    K, TD, TH, TW = template.shape[1:5]
    if template.shape[-1] != 4:
        raise RuntimeError("mvpraymarch: the template must be channels-last RGBA")
    out = torch.empty(N, H, W, 4, dtype=torch.float32, device=raypos.device)
    _lib.check(_lib.load().primx_raymarch(
        ops._dev(raypos, "raypos", torch.float32), ops._dev(raydir, "raydir", torch.float32),
        ops._dev(tminmax, "tminmax", torch.float32), float(stepsize), ops._dev(primpos, "primpos", torch.float32),
        ops._dev(primrot, "primrot", torch.float32), ops._dev(primscale, "primscale", torch.float32),
        ops._dev(template, "template", torch.float32), out.data_ptr(), N, H, W, K, TD, TH, TW, float(fadescale),
        float(fadeexp), ops._stream()), "primx_raymarch")
    return out


class RayMarcher(nn.Module):
    def __init__(self, image_height, image_width, volradius, fadescale=8.0, fadeexp=8.0, dt=1.0, ray_subsample_factor=1,
                 accum=2, termthresh=0.99, blocksize=None, with_t_img=True, chlast=False, assets=None):
        super().__init__()
        self.image_height, self.image_width = image_height, image_width
        self.volradius, self.dt = volradius, dt
        self.fadescale, self.fadeexp = fadescale, fadeexp
        self.accum, self.termthresh = accum, termthresh          # carried; the forward path uses additive accumulation
        self.ray_subsample_factor = ray_subsample_factor
        self.__dict__["_coords"] = {}

    def resize(self, h: int, w: int):
        self.image_height, self.image_width = h, w

    def _pixel_coords(self, B: int, factor: int, device) -> torch.Tensor:
        key = (self.image_height, self.image_width, factor, str(device))
        c = self._coords.get(key)
        if c is None:
            ys, xs = torch.meshgrid(torch.arange(self.image_height, dtype=torch.float32, device=device),
                                    torch.arange(self.image_width, dtype=torch.float32, device=device), indexing="ij")
            c = torch.stack([xs, ys], dim=-1)
            if factor > 1:   # resize_pixel_coords (ray_marcher.py:57-76)
                sw, sh = self.image_width // factor, self.image_height // factor
                x0 = y0 = factor // 2
                c = c[y0:y0 + factor * sh:factor, x0:x0 + factor * sw:factor, :]
            self.__dict__["_coords"] = {key: c.contiguous()}
        return c[None].expand(B, -1, -1, -1).contiguous()

    @ops.on_input_device
    def forward(self, prim_rgba, prim_pos, prim_rot, prim_scale, K, RT, ray_subsample_factor: Optional[int] = None):
        if self.training:
            raise NotImplementedError("the accelerated ray marcher is forward / inference only: call .eval()")
        if not prim_rgba.is_cuda:
            raise RuntimeError("RayMarcher.forward needs HIP device tensors; there is no CPU path")
        B = prim_rgba.shape[0]
        cam = convert_camera_parameters(RT.float(), K.float())
        factor = self.ray_subsample_factor if ray_subsample_factor is None else ray_subsample_factor
        pixel_coords = self._pixel_coords(B, factor, prim_rgba.device)
        focal = torch.diagonal(cam["focal"], dim1=1, dim2=2).contiguous()
        raypos, raydir, tminmax = compute_raydirs(cam["campos"], cam["camrot"], focal, cam["princpt"], pixel_coords,
                                                  self.volradius)
        rgba = mvpraymarch(raypos, raydir, self.dt / self.volradius, tminmax,
                           (prim_pos / self.volradius, prim_rot, prim_scale),
                           prim_rgba.permute(0, 1, 3, 4, 5, 2), self.fadescale, self.fadeexp)
        return {"rgba_image": rgba.permute(0, 3, 1, 2), "pixel_coords": pixel_coords}


def generate_colored_boxes(template: torch.Tensor, prim_rot: torch.Tensor, alpha: float = 10000.0, seed: int = 123456):
    """Debug templates for the bounding-box preview (dva/ray_marcher.py:232-279, used by dva/visualize.py:26): every
    primitive becomes an opaque box of one random colour, shaded by the face its voxel is nearest to.  template:
    [B, K, 4, S, S, S] -> same shape.  Elementwise preparation of a marcher input (plumbing, plain torch); colours follow
    the reference's `np.random.seed(seed)` stream so previews look the same."""
    import numpy as np

    B, K, S = template.shape[0], template.shape[1], template.shape[-1]
    dev = template.device
    lin = torch.linspace(-1.0, 1.0, S, device=dev)
    zz, yy, xx = torch.meshgrid(lin, lin, lin, indexing="ij")
    ax, ay, az = xx.abs(), yy.abs(), zz.abs()
    # outward unit normal of the dominant face(s), in the marcher's (x, -y, -z) convention; ties share the normal
    n = torch.stack([torch.where((ax >= ay) & (ax >= az), xx.sign(), torch.zeros_like(xx)),
                     -torch.where((ay >= ax) & (ay >= az), yy.sign(), torch.zeros_like(xx)),
                     -torch.where((az >= ax) & (az >= ay), zz.sign(), torch.zeros_like(xx))], dim=-1)
    n = n / n.pow(2).sum(-1, keepdim=True).sqrt()
    light = torch.full((3,), -3.0, device=dev)
    light = light / light.norm()
    shade = 1.4 * (n * light).sum(-1).clamp(min=0.2)                                   # [S, S, S]
    colours = torch.as_tensor(np.random.RandomState(seed).rand(K, 3) * 255.0, dtype=template.dtype, device=dev)
    out = template.clone()
    out[:, :, :3] = colours[None, :, :, None, None, None] * shade[None, None, None]
    out[:, :, 3] = alpha
    return out
