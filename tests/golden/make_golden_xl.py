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
            "xl_c2_ddim25": (28, 2048, 1, 2), "xl_c3": (28, 2048, 8, 16), "xl_c5": (28, 4096, 4, 16)}
# cases that share another case's inputs (and, at equal depth, its weights)
XL_INPUTS_OF = {"xl_c2_ddim25": "xl_c2"}


def xl_inputs(name: str):
    depth, N, B, stride = XL_CASES[name]
    src = XL_INPUTS_OF.get(name, name)
    x = synth.tensor(XL_SEED, src + ".x", (B, N, 68))
    y = synth.tensor(XL_SEED, src + ".y", (B, L_COND, 768))
    return depth, N, B, stride, x, y


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    dit_mod, _, diffusion_pkg, _ = ref_import.load()
    for name in (sys.argv[1:] or list(XL_CASES)):
        depth, N, B, stride, x, y = xl_inputs(name)
        t0 = time.time()
        cfg = dict(depth=depth, **XL)
        model = dit_mod.DiT(seq_length=N, num_heads=HEADS, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
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
            else:
                out["forward_cfg"] = model.forward_with_cfg(x, t, y, cfg_scale=6.0)[:, ::stride].numpy()
                out["t"] = t.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, f"{time.time() - t0:.1f} s", os.path.getsize(os.path.join(HERE, name + ".npz")), flush=True)
        del model


if __name__ == "__main__":
    main()
