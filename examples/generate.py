"""End-to-end run of the MI355X path with synthetic weights and a synthetic image, stage by stage as
inference.py:312-352 does it:

  image -> DINOv2 tokens -> 25-step DDIM with CFG (DiT) -> de-normalise + VAE decode -> denoised.pt
        -> PrimSDF lattice query (mesh-extraction input) + one ray-marched preview

There are no checkpoints offline: every network carries random weights, so the outputs are noise - the point is the
data flow, the shapes and the per-stage timing.   python examples/generate.py [--steps 25] [--res 128] [--lattice 96]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--res", type=int, default=128, help="preview resolution")
    ap.add_argument("--lattice", type=int, default=96, help="SDF lattice resolution (the CLI uses 256)")
    ap.add_argument("--small", action="store_true", help="tiny networks (smoke run)")
    a = ap.parse_args()
    __graft_entry__.build()
    import topia_xl_amd as pkg
    from topia_xl_amd import dinov2, pipeline, raymarch

    dev = "cuda:0"
    torch.manual_seed(42)
    if a.small:
        cond = dinov2.DinoVisionTransformer(img_size=56, embed_dim=96, depth=2, num_heads=3)
        dit = pkg.DiT(seq_length=64, in_channels=68, condition_channels=96, hidden_size=288, depth=2, num_heads=4,
                      attn_proj_bias=True, cond_drop_prob=0.1)
        n_prims, img = 64, 56
    else:
        cond = dinov2.vit_base(img_size=518, init_values=1.0)
        dit = pkg.DiT(seq_length=2048, in_channels=68, condition_channels=768, hidden_size=1152, depth=28, num_heads=16,
                      attn_proj_bias=True, cond_drop_prob=0.1)
        n_prims, img = 2048, 518
    vae = pkg.VAE(in_channels=6, latent_channels=1, out_channels=6, down_channels=[32, 256], mid_attention=True,
                  up_channels=[256, 32], layers_per_block=2)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():   # trained-network-like magnitudes: activations O(1) through depth, small non-zero gates
        for m in (cond, dit, vae):
            for name, p in m.named_parameters():
                if p.dim() > 1 and "token" not in name and "pos_embed" not in name:
                    std = (0.6 if "adaLN" in name else 1.0) * p[0].numel() ** -0.5
                    p.copy_(torch.randn(p.shape, generator=g) * std)
                elif "norm" in name and name.endswith("weight") or name.endswith("gamma"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=g))
            m.eval().to(dev)
    diffusion = pkg.create_diffusion(f"ddim{a.steps}", noise_schedule="squaredcos_cap_v2", parameterization="v")

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        print(f"{name:34s} {1e3 * (time.perf_counter() - t0):9.2f} ms", flush=True)
        return r

    image = torch.rand(1, 3, img, img, device=dev)
    y = timed("DINOv2 conditioner tokens", lambda: cond.conditioner_tokens(image))
    x = torch.randn(1, n_prims, 68).to(dev)
    kw = dict(y=y, cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
    timed("DDIM loop (warm-up, 1st call)", lambda: diffusion.ddim_sample_loop(dit.forward_with_cfg, x.shape, noise=x, clip_denoised=False, model_kwargs=kw))
    samples = timed(f"DDIM loop, {a.steps} steps, CFG 6", lambda: diffusion.ddim_sample_loop(dit.forward_with_cfg, x.shape, noise=x, clip_denoised=False, model_kwargs=kw))
    mean, std = [0.0] * 68, [1.0] * 68
    recon = timed("de-normalise + VAE decode", lambda: pipeline.latents_to_primitives(samples, vae, mean, std))
    # random networks produce arbitrary scales: give the primitives a plausible pose so that the last two stages have work
    recon[:, :, 0] = 0.05 + 0.03 * torch.rand_like(recon[:, :, 0])
    recon[:, :, 1:4] = 1.2 * torch.rand_like(recon[:, :, 1:4]) - 0.6
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "denoised.pt")
        pipeline.save_denoised(path, recon)
        field = pipeline.primsdf_from_denoised(path, dev)
    xx = torch.linspace(-1, 1, a.lattice, device=dev)
    pts = torch.stack(torch.meshgrid(xx, xx, xx, indexing="ij"), dim=-1).reshape(-1, 3)
    sdf = timed(f"PrimSDF query, {a.lattice}^3 lattice", lambda: field(pts)["sdf"])
    rm = raymarch.RayMarcher(a.res, a.res, volradius=10000.0, dt=1.0).eval()
    K, Rt = pipeline.preview_camera(rm.volradius, a.res, a.res, dev)
    rgba, pos, rot, scale = pipeline.primitives_to_marcher_inputs(recon, rm.volradius)
    view = timed(f"ray-marched preview {a.res}x{a.res}", lambda: rm(rgba, pos, rot, scale, K, Rt)["rgba_image"])
    print("tokens", tuple(y.shape), "samples", tuple(samples.shape), "recon_param", tuple(recon.shape), "sdf grid",
          tuple(sdf.reshape(a.lattice, a.lattice, a.lattice).shape), "preview", tuple(view.shape),
          "coverage %.2f" % float((view[0, 3] > 0).float().mean()))


if __name__ == "__main__":
    main()
