"""Post-sampling driver of the hot path: denoised latents -> per-primitive PBR voxel payload.

Mirrors what the reference's CLI / app do between the sampler and the ray-marcher / mesh export
(inference.py:326-348, app.py:117-139):  de-normalise with the per-channel latent statistics, split
(scale + xyz | 4^3 VAE latent), decode every sample's primitives, apply the decoder's inverse normalisation
(SDF / 5, colour+material (v + 1) / 2) and concatenate to ``[B, N_prim, 4 + 6 * 8^3]`` - the ``recon_param``
tensor the renderer and ``PrimSDF`` consume (``srt_param`` = [:, :, :4], ``feat_param`` = [:, :, 4:]).

MI355X-first differences: the reference decodes one sample at a time "to avoid oom" (inference.py:334-340) and
runs four full-tensor elementwise passes afterwards; here all B * N_prim primitives go through ONE decoder call
(288 GB of HBM: 8 samples x 2048 primitives peak at ~9 GB of 16-bit activations), the inverse normalisation is
fused into the decoder's output kernel and the latent de-normalisation + split is one kernel.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops


_STATS_CACHE: dict = {}


def _latent_stats(latent_mean, latent_std, dev):
    """fp32 device tensors of the per-channel statistics.  The reference re-uploads them with every call
    (inference.py:328-332); here a Python list / tuple is uploaded once per device (two pageable host-to-device copies and
    their synchronisation per decode otherwise), tensors are used as they are."""
    def one(v):
        if isinstance(v, torch.Tensor):
            return v.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        key = (tuple(float(a) for a in v), str(dev))
        t = _STATS_CACHE.get(key)
        if t is None:
            if len(_STATS_CACHE) > 16:
                _STATS_CACHE.clear()
            t = _STATS_CACHE[key] = torch.tensor(key[0], dtype=torch.float32, device=dev)
        return t
    return one(latent_mean), one(latent_std)


@ops.on_input_device
def latents_to_primitives(samples: torch.Tensor, vae, latent_mean: Optional[Sequence[float]] = None,
                          latent_std: Optional[Sequence[float]] = None, latent_nf: float = 1.0,
                          max_prims_per_call: int = 8 * 2048) -> torch.Tensor:
    """samples: (B, N_prim, 68) fp32 sampler output -> recon_param (B, N_prim, 4 + C_out * (2S)^3) fp32."""
    if not samples.is_cuda:
        raise RuntimeError("latents_to_primitives needs HIP device tensors; there is no CPU path")
    B, N, C = samples.shape
    dev = samples.device
    if latent_mean is None:
        # the reference's non-per-channel branch (inference.py:336-344) is dead for the shipped config (yml:64-65)
        raise NotImplementedError("per-channel latent_mean / latent_std are required (configs/inference_dit.yml:64-65)")
    mean, std = _latent_stats(latent_mean, latent_std, dev)      # device copies are made once per (values, device)
    if mean.numel() != C or std.numel() != C:
        raise AssertionError("latent_mean / latent_std must have one entry per latent channel")
    srt, z = ops.latent_denorm(samples.float().contiguous(), mean, std, latent_nf, 4)
    S = round((C - 4) ** (1.0 / 3.0))
    if S ** 3 != C - 4:
        raise AssertionError("latent channels minus 4 must be a cube")
    z = z.reshape(B * N, 1, S, S, S)
    outs = []
    for lo in range(0, B * N, max_prims_per_call):
        outs.append(vae.decode(z[lo:lo + max_prims_per_call].contiguous(), denormalize=True))
    dec = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    feat = dec.reshape(B, N, -1)
    return torch.cat([srt, feat], dim=-1)


# ---------------------------------------------------------------------------------------------------------------------
# On-disk formats of the hot path: the two checkpoints the CLI loads and the `denoised.pt`
# it writes between sampling and mesh extraction.
def load_checkpoints(model=None, vae=None, dit_checkpoint_path: Optional[str] = None,
                     vae_checkpoint_path: Optional[str] = None, packed_dtype: Optional[torch.dtype] = None) -> None:
    """`model.load_state_dict(torch.load(p)['ema'])` / `vae.load_state_dict(torch.load(p)['model_state_dict'])`, strict,
    as inference.py:257-262 does (fp16 `.pt` files load into the fp32 parameters; the packed 16-bit copies the kernels
    read are rebuilt lazily on the first forward).

    `packed_dtype` (torch.float16 / torch.bfloat16) takes the direct route for the DiT instead: the checkpoint is memory-
    mapped and its tensors are copied straight into the packed 16-bit blob on the model's device (`DiT.pack_from_state_dict`)
    - no 3.6 GB of fp32 parameters, no repack on the first forward; the model is then packed-only (its fp32 route raises).
    A path ending in `.primxpk` is a `DiT.save_packed` file and is mapped as is (`DiT.load_packed`)."""
    if model is not None and dit_checkpoint_path:
        if dit_checkpoint_path.endswith(".primxpk"):
            model.load_packed(dit_checkpoint_path)
        elif packed_dtype is not None:
            model.pack_from_state_dict(torch.load(dit_checkpoint_path, map_location="cpu", mmap=True)["ema"], packed_dtype)
        else:
            model.load_state_dict(torch.load(dit_checkpoint_path, map_location="cpu")["ema"], strict=True)
    if vae is not None and vae_checkpoint_path:
        vae.load_state_dict(torch.load(vae_checkpoint_path, map_location="cpu")["model_state_dict"], strict=True)


def save_denoised(path: str, recon_param: torch.Tensor, index: int = 0) -> None:
    """`{'model_state_dict': {'srt_param': [N, 4], 'feat_param': [N, 6 * 8^3]}}` of sample `index` (inference.py:351-352):
    the file PrimSDF (mesh extraction) and the viewer load."""
    p = recon_param[index].detach().cpu()
    torch.save({"model_state_dict": {"srt_param": p[:, :4].contiguous(), "feat_param": p[:, 4:].contiguous()}}, path)


def primsdf_from_denoised(path: str, device=None):
    """A `PrimSDF` holding the primitives of a `denoised.pt` (what the GLB export builds before querying the field)."""
    from .primsdf import PrimSDF
    sd = torch.load(path, map_location="cpu")["model_state_dict"]
    n, s3 = sd["srt_param"].shape[0], sd["feat_param"].shape[1] // 6
    m = PrimSDF(num_prims=n, dim_feat=6, prim_shape=round(s3 ** (1.0 / 3.0)))
    m.load_state_dict(sd, strict=True)
    return m.to(device).eval() if device is not None else m.eval()


def primitives_to_marcher_inputs(recon_param: torch.Tensor, volradius: float, sdf2alpha_var: float = 0.005):
    """recon_param [B, N, 4 + 6 S^3] -> (prim_rgba [B,N,4,S,S,S] in 0..255, prim_pos, prim_rot, prim_scale) exactly as the
    preview renderer prepares them (dva/visualize.py:215-239): alpha = 255 exp(-(sdf / 0.005)^2), rgb = 255 tex, identity
    rotations, inverse scales.  Elementwise tensor preparation around the marcher (plumbing)."""
    B, N, C = recon_param.shape
    S = round(((C - 4) / 6) ** (1.0 / 3.0))
    s3 = S ** 3
    geo = recon_param[:, :, 4:4 + s3]
    tex = recon_param[:, :, 4 + s3:4 + 4 * s3]
    alpha = torch.exp(-(geo / sdf2alpha_var) ** 2).reshape(B, N, 1, S, S, S) * 255
    rgb = tex.reshape(B, N, 3, S, S, S) * 255
    pos = recon_param[:, :, 1:4].reshape(B, N, 3) * volradius
    rot = torch.eye(3, device=recon_param.device, dtype=recon_param.dtype)[None, None].repeat(B, N, 1, 1)
    scale = 1.0 / recon_param[:, :, 0:1].reshape(B, N, 1).repeat(1, 1, 3)
    return torch.cat([rgb, alpha], dim=2), pos, rot, scale


def preview_camera(volradius: float, height: int, width: int, device):
    """The fixed preview camera of dva/visualize.py:240-285 (looking down -z from 5 volume radii, 1024-pixel intrinsics
    rescaled to the image size)."""
    Rt = torch.tensor([[[1.0, 0.0, 0.0, 0.0], [0.0, -1.0, 0.0, 0.0], [0.0, 0.0, -1.0, 5.0 * volradius]]], device=device)
    K = torch.tensor([[[2084.9526697685183, 0.0, 512.0], [0.0, 2084.9526697685183, 512.0], [0.0, 0.0, 1.0]]], device=device)
    K[:, 0:1, :] *= height / 1024.0
    K[:, 1:2, :] *= width / 1024.0
    return K, Rt
