"""Generate the golden vectors from the REAL reference (run in the build container only):

    python tests/golden/make_golden.py

Imports the unmodified 3DTopia-XL modules from /root/reference through oracle/ref_import.py (xformers
stand-in documented there), loads deterministic synthetic weights (oracle/synth.py - the reference's
own init is all-zero in the adaLN / final layers, which would make parity vacuous), runs them in fp32
on the CPU and stores inputs' seeds + outputs as small .npz fixtures next to this file.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# (name, DiT kwargs, heads, N tokens, L cond tokens, batch)
DIT_CASES = [
    ("dit_dh64", dict(in_channels=68, condition_channels=96, hidden_size=384, depth=2), 6, 256, 5, 2),
    ("dit_dh72", dict(in_channels=68, condition_channels=64, hidden_size=288, depth=2), 4, 128, 70, 1),
]
VAE_CFG = dict(in_channels=6, latent_channels=1, out_channels=6, down_channels=[32, 256], mid_attention=True,
               up_channels=[256, 32], layers_per_block=2, gradient_checkpointing=False)
SEED = 1234


def gen_schedule(diffusion_pkg):
    out = {}
    gd = sys.modules["models.diffusion.gaussian_diffusion"]
    for n in (5, 25, 50, 100, 200):
        d = diffusion_pkg.create_diffusion(timestep_respacing=f"ddim{n}", noise_schedule="squaredcos_cap_v2",
                                           parameterization="v", diffusion_steps=1000)
        out[f"ddim{n}_map"] = np.array(d.timestep_map, dtype=np.int64)
        for attr in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                     "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                     "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
                     "posterior_mean_coef2"):
            out[f"ddim{n}_{attr}"] = np.asarray(getattr(d, attr), dtype=np.float64)
    out["cos1000_betas"] = gd.get_named_beta_schedule("squaredcos_cap_v2", 1000)
    out["lin1000_betas"] = gd.get_named_beta_schedule("linear", 1000)
    out["lin250_betas"] = gd.get_named_beta_schedule("linear", 250)
    respace = sys.modules["models.diffusion.respace"]
    out["sections_300_10_15_20"] = np.array(sorted(respace.space_timesteps(300, [10, 15, 20])), dtype=np.int64)
    out["sections_1000_str"] = np.array(sorted(respace.space_timesteps(1000, "7,3,11")), dtype=np.int64)
    full = diffusion_pkg.create_diffusion(timestep_respacing="", noise_schedule="linear", parameterization="eps",
                                          learn_sigma=False, diffusion_steps=50)
    out["full50_map"] = np.array(full.timestep_map, dtype=np.int64)
    out["full50_alphas_cumprod"] = full.alphas_cumprod
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **out)


def gen_dit(dit_mod, diffusion_pkg):
    for name, cfg, heads, N, L, B in DIT_CASES:
        model = dit_mod.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
        sd = synth.dit_state_dict(SEED, **cfg)
        model.load_state_dict(sd, strict=True)
        x = synth.tensor(SEED, name + ".x", (B, N, cfg["in_channels"]))
        y = synth.tensor(SEED, name + ".y", (B, L, cfg["condition_channels"]))
        t = torch.tensor([960, 40][:B], dtype=torch.int64)
        out = {}
        with torch.no_grad():
            out["t_emb_freq"] = model.t_embedder.timestep_embedding(torch.tensor([0, 1, 40, 500, 960, 999]), 256).numpy()
            out["forward"] = model(x, t, y).numpy()
            out["forward_cfg"] = model.forward_with_cfg(x, t, y, cfg_scale=6.0).numpy()
            blk = model.blocks[0]
            te = model.t_embedder(t)
            out["block0"] = blk(model.x_embedder(x), y, te).numpy()
            d = diffusion_pkg.create_diffusion(timestep_respacing="ddim5", noise_schedule="squaredcos_cap_v2",
                                               parameterization="v", diffusion_steps=1000)
            traj = []
            for s in d.ddim_sample_loop_progressive(model.forward_with_cfg, x.shape, noise=x, clip_denoised=False,
                                                    model_kwargs=dict(y=y, cfg_scale=6.0), device="cpu"):
                traj.append(s["sample"].numpy())
            out["ddim5_samples"] = np.stack(traj)
            last = s["pred_xstart"].numpy()
            out["ddim5_last_pred_xstart"] = last
            # one ancestral step (p_sample) with fixed noise via a seeded generator is RNG-dependent;
            # store the pieces that are not: mean / log-variance of p_mean_variance at spaced t = 3
            tt = torch.full((B,), 3, dtype=torch.int64)
            pmv = d.p_mean_variance(model.forward_with_cfg, x, tt, clip_denoised=False,
                                    model_kwargs=dict(y=y, cfg_scale=6.0))
            out["pmv_mean"] = pmv["mean"].numpy()
            out["pmv_log_variance"] = pmv["log_variance"].numpy()
        out["seed"] = np.int64(SEED)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def gen_dit_addpos(dit_mod):
    """The second DiT class of the reference file (DiTAdditivePosEmb, dit_crossattn.py:215-301)."""
    cfg = dict(in_channels=68, condition_channels=64, hidden_size=288, depth=2)
    N, L, B, heads = 96, 37, 2, 4
    model = dit_mod.DiTAdditivePosEmb(seq_length=N, num_heads=heads, attn_proj_bias=True, **cfg).eval()
    sd = synth.state_dict_like(SEED, model.state_dict())
    sd["point_emb.basis"] = model.point_emb.basis.clone()            # a constant buffer, not a weight
    model.load_state_dict(sd, strict=True)
    x = synth.tensor(SEED, "addpos.x", (B, N, cfg["in_channels"]))
    y = synth.tensor(SEED, "addpos.y", (B, L, cfg["condition_channels"]))
    t = torch.tensor([960, 40], dtype=torch.int64)
    with torch.no_grad():
        out = model(x, t, y).numpy()
        pe = model.point_emb(x[:, :, 1:4]).numpy()
    np.savez_compressed(os.path.join(HERE, "dit_addpos.npz"), forward=out, point_emb=pe, seed=np.int64(SEED),
                        keys=np.array(sorted(sd.keys())))


DINO_CFG = dict(img_size=56, patch_size=14, embed_dim=96, depth=2, num_heads=3, mlp_ratio=4, init_values=1.0,
                num_register_tokens=4, interpolate_antialias=True, interpolate_offset=0.0, block_chunks=0)


def dino_state_dict(reference_sd):
    """Synthetic DINOv2 weights: LayerScale gammas O(1) like the released checkpoints (init 1.0, trained), tokens and the
    positional table ~ N(0, 0.5), everything else as synth.state_dict_like."""
    sd = synth.state_dict_like(SEED, reference_sd)
    for k in reference_sd:
        if k.endswith(".gamma"):
            sd[k] = synth.tensor(SEED, k, tuple(reference_sd[k].shape), 0.3, 1.0)
        elif k in ("cls_token", "pos_embed", "register_tokens"):
            sd[k] = synth.tensor(SEED, k, tuple(reference_sd[k].shape), 0.5)
    return sd


def gen_dinov2():
    """The vendored DINOv2 of the reference (models/conditioner/dinov2), small configuration, two image sizes: the
    training resolution (positional table used as is) and a larger one (bicubic resampling of the table)."""
    import importlib
    vt = importlib.import_module("models.conditioner.dinov2.models.vision_transformer")
    model = vt.DinoVisionTransformer(**DINO_CFG).eval()
    sd = dino_state_dict(model.state_dict())
    model.load_state_dict(sd, strict=True)
    out = {"keys": np.array(sorted(sd.keys())), "seed": np.int64(SEED)}
    with torch.no_grad():
        for tag, size in (("native", 56), ("resampled", 84)):
            x = synth.tensor(SEED, f"dino.x.{size}", (2, 3, size, size))
            ret = model(x, is_training=True)
            out[f"{tag}_cls"] = ret["x_norm_clstoken"].numpy()
            out[f"{tag}_reg"] = ret["x_norm_regtokens"].numpy()
            out[f"{tag}_patch"] = ret["x_norm_patchtokens"].numpy()
            out[f"{tag}_prenorm"] = ret["x_prenorm"].numpy()
    np.savez_compressed(os.path.join(HERE, "dinov2.npz"), **out)


PRIMSDF_CFG = dict(num_prims=48, dim_feat=6, prim_shape=8)


def primsdf_params():
    """Fitted-primitive-like parameters: centres in [-0.7, 0.7]^3, half-extents 0.15..0.4 (overlapping, yet leaving part
    of [-1, 1]^3 uncovered so that the nearest-voxel fill is exercised), SDF payload of both signs, tex / mat partly
    outside [0, 1] (clipping)."""
    P, S = PRIMSDF_CFG["num_prims"], PRIMSDF_CFG["prim_shape"]
    srt = torch.cat([0.15 + 0.25 * synth.tensor(SEED, "psdf.scale", (P, 1)).abs().clamp(max=1.0),
                     0.7 * torch.tanh(synth.tensor(SEED, "psdf.pos", (P, 3)))], dim=1)
    feat = synth.tensor(SEED, "psdf.feat", (P, 6 * S ** 3), 0.6, 0.3)
    pts = 2.0 * torch.rand(4000, 3, generator=torch.Generator().manual_seed(SEED)) - 1.0
    return srt, feat, pts


def gen_primsdf():
    """models/primsdf.py:PrimSDF.forward in eval and train mode (trimesh, imported at module top and unused by forward,
    is stubbed)."""
    import importlib
    import types
    sys.modules.setdefault("trimesh", types.ModuleType("trimesh"))
    mod = importlib.import_module("models.primsdf")
    m = mod.PrimSDF(**PRIMSDF_CFG)
    srt, feat, pts = primsdf_params()
    m.srt_param.data, m.feat_param.data = srt.clone(), feat.clone()
    out = {"seed": np.int64(SEED)}
    with torch.no_grad():
        for tag in ("eval", "train"):
            m.train(tag == "train")
            pr = m(pts)
            out[f"{tag}_sdf"], out[f"{tag}_tex"], out[f"{tag}_mat"] = pr["sdf"].numpy(), pr["tex"].numpy(), pr["mat"].numpy()
        out["covered"] = (m.prim_weight(pts).sum(1) > 0).numpy()
    np.savez_compressed(os.path.join(HERE, "primsdf.npz"), **out)


def gen_attention(att_mod):
    out = {}
    with torch.no_grad():
        m = att_mod.MemEffAttention(dim=256, num_heads=8, qkv_bias=False, proj_bias=True).eval()
        sd = synth.state_dict_like(SEED, m.state_dict())
        m.load_state_dict(sd)
        x = synth.tensor(SEED, "att.x", (3, 64, 256))
        out["self_dh32"] = m(x).numpy()
        c = att_mod.MemEffCrossAttention(dim=144, dim_q=144, dim_k=40, dim_v=40, num_heads=2, qkv_bias=True,
                                         proj_bias=True).eval()
        c.load_state_dict(synth.state_dict_like(SEED, c.state_dict()))
        q = synth.tensor(SEED, "catt.q", (2, 96, 144))
        kv = synth.tensor(SEED, "catt.kv", (2, 37, 40))
        out["cross_dh72"] = c(q, kv, kv).numpy()
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **out)


def gen_vae(vae_mod):
    vae = vae_mod.VAE(**VAE_CFG).eval()
    sd = synth.state_dict_like(SEED, vae.state_dict())
    vae.load_state_dict(sd, strict=True)
    z = synth.tensor(SEED, "vae.z", (3, 1, 4, 4, 4))
    with torch.no_grad():
        dec = vae.decode(z).numpy()
    np.savez_compressed(os.path.join(HERE, "vae_decode.npz"), decoded=dec, seed=np.int64(SEED),
                        keys=np.array(sorted(sd.keys())),
                        shapes=np.array([str(tuple(sd[k].shape)) for k in sorted(sd.keys())]))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dit_mod, vae_mod, diffusion_pkg, att_mod = ref_import.load()
    gen_schedule(diffusion_pkg)
    gen_dit(dit_mod, diffusion_pkg)
    gen_attention(att_mod)
    gen_dit_addpos(dit_mod)
    gen_dinov2()
    gen_primsdf()
    gen_vae(vae_mod)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
