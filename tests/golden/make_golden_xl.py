"""Golden vectors of the REAL reference at the BASELINE shapes (run in the build container only; takes a few minutes):

    python tests/golden/make_golden_xl.py

Same mechanism as make_golden.py (unmodified reference modules through oracle/ref_import.py, deterministic synthetic
weights from oracle/synth.py, fp32 on the CPU), at the sizes BASELINE.json names, so that the `-m gpu` parity tests of
the full configurations need no CPU oracle at test time:

  xl_c2     configs[1]: DiT-XL (d=1152, 28 blocks, 16 heads x 72), N_prim=2048, 1370 x 768 condition tokens, batch 1,
            CFG 6: `forward_with_cfg` at the first ddim5 timestep and the whole 5-step DDIM trajectory.
  xl_c3blk  configs[2] per-GPU shape on ONE block: batch 8 (effective 16 with CFG), N_prim=2048.
  xl_c5blk  configs[4] per-GPU shape on ONE block: batch 4 (effective 8), N_prim=4096.
  xl_c2_ddim25  configs[1] again (same weights and inputs as xl_c2): the 25-step DDIM trajectory the headline metric
            is quoted on - every 5th sample, the final sample and the final pred_xstart (round 3; ~30 min on 8 cores).
  xl_c3     configs[2] per-GPU shape at FULL depth (28 blocks): batch 8, one `forward_with_cfg` (round 3; ~10 min).
  xl_c5     configs[4] per-GPU shape at FULL depth (28 blocks): batch 4, N_prim=4096 (round 3; ~12 min).
  xl_c4_e2e configs[3] END TO END at batch 1 (round 4; ~105 min): the 100-step DDIM trajectory (every 20th sample, the
            final sample in full) FOLLOWED BY what inference.py:326-348 does with the final sample - per-channel
            de-normalisation with configs/inference_dit.yml:63-65, split, the reference `VAE.decode` of all 2048
            primitives, the inverse normalisation, concatenation to `recon_param` (stored for every 8th primitive).
  xl_c3_ddim5 configs[2] per-GPU shape (batch 8, full depth): the whole 5-step DDIM trajectory (round 4; ~50 min) -
            error accumulation along a loop at the effective batch of 16.
  c1_named  configs[0] at its NAMED shape: depth 12, d=384, 6 heads (dh 64), N_prim=256, ONE condition token, batch 1,
            5 DDIM steps with CFG (round 4; seconds).  Stored in full.

    python tests/golden/make_golden_xl.py [case ...]        # default: every case

Outputs are stored for every TOKEN_STRIDE-th token (fixtures stay ~1-2 MB each; the tests compare the same subset).
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
XL_SEED = 4321
XL = dict(in_channels=68, condition_channels=768, hidden_size=1152)
HEADS, L_COND = 16, 1370
# name -> (depth, N_prim, batch, token stride of the stored outputs)
XL_CASES = {"xl_c2": (28, 2048, 1, 2), "xl_c3blk": (1, 2048, 8, 8), "xl_c5blk": (1, 4096, 4, 8),
            "xl_c2_ddim25": (28, 2048, 1, 2), "xl_c3": (28, 2048, 8, 16), "xl_c5": (28, 4096, 4, 16),
            "xl_c4_e2e": (28, 2048, 1, 2), "xl_c3_ddim5": (28, 2048, 8, 16), "c1_named": (12, 256, 1, 1)}
# cases that share another case's inputs (and, at equal depth, its weights)
XL_INPUTS_OF = {"xl_c2_ddim25": "xl_c2", "xl_c4_e2e": "xl_c2", "xl_c3_ddim5": "xl_c3"}
# configs[0] (BASELINE.json): "DiT-S/2 (depth=12, d=384), N_prim=256, 1 cond token"
C1 = dict(in_channels=68, condition_channels=768, hidden_size=384)
C1_HEADS, C1_L = 6, 1
VAE_CFG = dict(in_channels=6, latent_channels=1, out_channels=6, down_channels=[32, 256], mid_attention=True,
               up_channels=[256, 32], layers_per_block=2, gradient_checkpointing=False)
PRIM_STRIDE = 8


def model_cfg(name: str):
    """(DiT kwargs, heads, condition tokens) of a case."""
    return (C1, C1_HEADS, C1_L) if name == "c1_named" else (XL, HEADS, L_COND)


def latent_stats():
    """latent_nf / latent_mean / latent_std of the shipped configuration (configs/inference_dit.yml:63-65)."""
    import yaml
    with open(os.path.join(ref_import.REFERENCE_ROOT, "configs", "inference_dit.yml")) as f:
        m = yaml.safe_load(f)["model"]
    return float(m["latent_nf"]), [float(v) for v in m["latent_mean"]], [float(v) for v in m["latent_std"]]


def reference_postprocess(vae, sample, nf, mean, std):
    """inference.py:326-348 on the reference's own modules, statement by statement (the CLI itself cannot be imported
    here: rembg, the CUDA ray marcher).  sample (1, N, 68) -> recon_param (1, N, 4 + 6 * 8^3)."""
    inf_bs, num_prims = sample.shape[0], sample.shape[1]
    latent_mean = torch.Tensor(mean)[None, None, :]
    latent_std = torch.Tensor(std)[None, None, :]
    recon_param = sample.reshape(inf_bs, num_prims, -1)
    recon_param = recon_param / nf * latent_std + latent_mean
    recon_srt_param = recon_param[:, :, 0:4]
    recon_feat_param = recon_param[:, :, 4:]
    lst = []
    for b in range(inf_bs):
        decoded = vae.decode(recon_feat_param[b, ...].reshape(1 * num_prims, 1, 4, 4, 4))
        lst.append(decoded.detach())
    recon_feat_param = torch.concat(lst, dim=0)
    recon_feat_param[:, 0:1, ...] /= 5.
    recon_feat_param[:, 1:, ...] = (recon_feat_param[:, 1:, ...] + 1) / 2.
    recon_feat_param = recon_feat_param.reshape(inf_bs, num_prims, -1)
    return torch.concat([recon_srt_param, recon_feat_param], dim=-1)


def xl_inputs(name: str):
    depth, N, B, stride = XL_CASES[name]
    src = XL_INPUTS_OF.get(name, name)
    L = model_cfg(name)[2]
    x = synth.tensor(XL_SEED, src + ".x", (B, N, 68))
    y = synth.tensor(XL_SEED, src + ".y", (B, L, 768))
    return depth, N, B, stride, x, y


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    dit_mod, vae_mod, diffusion_pkg, _ = ref_import.load()
    for name in (sys.argv[1:] or list(XL_CASES)):
        depth, N, B, stride, x, y = xl_inputs(name)
        t0 = time.time()
        base, heads, _ = model_cfg(name)
        cfg = dict(depth=depth, **base)
        model = dit_mod.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
        model.load_state_dict(synth.dit_state_dict(XL_SEED, **cfg), strict=True)
        out = {"seed": np.int64(XL_SEED), "token_stride": np.int64(stride)}
        d = diffusion_pkg.create_diffusion(timestep_respacing="ddim5", noise_schedule="squaredcos_cap_v2",
                                           parameterization="v", diffusion_steps=1000)
        t = torch.full((B,), int(d.timestep_map[-1]), dtype=torch.int64)      # the first timestep the loop visits
        with torch.no_grad():
            if name == "xl_c2":
                traj, x0 = [], []
                for s in d.ddim_sample_loop_progressive(model.forward_with_cfg, x.shape, noise=x, clip_denoised=False,
                                                        model_kwargs=dict(y=y, cfg_scale=6.0), device="cpu"):
                    traj.append(s["sample"][:, ::stride].numpy())
                    x0.append(s["pred_xstart"][:, ::stride].numpy())
                out["ddim5_samples"] = np.stack(traj)
                out["ddim5_pred_xstart"] = np.stack(x0)
            if name == "xl_c2_ddim25":
                d25 = diffusion_pkg.create_diffusion(timestep_respacing="ddim25", noise_schedule="squaredcos_cap_v2",
                                                     parameterization="v", diffusion_steps=1000)
                keep, traj = [], []
                for i, s in enumerate(d25.ddim_sample_loop_progressive(
                        model.forward_with_cfg, x.shape, noise=x, clip_denoised=False,
                        model_kwargs=dict(y=y, cfg_scale=6.0), device="cpu")):
                    if i % 5 == 4:
                        keep.append(i)
                        traj.append(s["sample"][:, ::stride].numpy())
                    print(name, "step", i, f"{time.time() - t0:.0f} s", flush=True)
                out["ddim25_steps"] = np.asarray(keep, dtype=np.int64)
                out["ddim25_samples"] = np.stack(traj)
                out["ddim25_final_pred_xstart"] = s["pred_xstart"][:, ::stride].numpy()
            if name in ("xl_c3_ddim5", "c1_named"):
                traj = []
                for i, s in enumerate(d.ddim_sample_loop_progressive(
                        model.forward_with_cfg, x.shape, noise=x, clip_denoised=False,
                        model_kwargs=dict(y=y, cfg_scale=6.0), device="cpu")):
                    traj.append(s["sample"][:, ::stride].numpy())
                    print(name, "step", i, f"{time.time() - t0:.0f} s", flush=True)
                out["ddim5_samples"] = np.stack(traj)
                out["ddim5_final_pred_xstart"] = s["pred_xstart"][:, ::stride].numpy()
            if name == "xl_c4_e2e":
                d100 = diffusion_pkg.create_diffusion(timestep_respacing="ddim100", noise_schedule="squaredcos_cap_v2",
                                                      parameterization="v", diffusion_steps=1000)
                keep, traj = [], []
                for i, s in enumerate(d100.ddim_sample_loop_progressive(
                        model.forward_with_cfg, x.shape, noise=x, clip_denoised=False,
                        model_kwargs=dict(y=y, cfg_scale=6.0), device="cpu")):
                    if i % 20 == 19:
                        keep.append(i)
                        traj.append(s["sample"][:, ::stride].numpy())
                    print(name, "step", i, f"{time.time() - t0:.0f} s", flush=True)
                out["ddim100_steps"] = np.asarray(keep, dtype=np.int64)
                out["ddim100_samples"] = np.stack(traj)
                out["ddim100_final_sample"] = s["sample"].numpy()                  # in full: the decoder's input
                vae = vae_mod.VAE(**VAE_CFG).eval()
                vae.load_state_dict(synth.state_dict_like(XL_SEED, vae.state_dict()), strict=True)
                nf, mean, std = latent_stats()
                recon = reference_postprocess(vae, s["sample"], nf, mean, std)
                out["latent_nf"], out["latent_mean"], out["latent_std"] = np.float64(nf), np.asarray(mean), np.asarray(std)
                out["prim_stride"] = np.int64(PRIM_STRIDE)
                out["recon_param"] = recon[:, ::PRIM_STRIDE].numpy()
                print(name, "decoded", f"{time.time() - t0:.0f} s", flush=True)
            if name not in ("xl_c2_ddim25", "xl_c4_e2e", "xl_c3_ddim5", "c1_named"):
                out["forward_cfg"] = model.forward_with_cfg(x, t, y, cfg_scale=6.0)[:, ::stride].numpy()
                out["t"] = t.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, f"{time.time() - t0:.1f} s", os.path.getsize(os.path.join(HERE, name + ".npz")), flush=True)
        del model


if __name__ == "__main__":
    main()
