"""Drop-in surface of the modules (no GPU): same state_dict keys/shapes as the reference, strict
load works, CPU tensors are refused loudly (there is no fallback path)."""
import subprocess
import sys
import os

import pytest
import torch

from oracle import synth
from tests.golden.make_golden import SEED, VAE_CFG

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XL = dict(seq_length=2048, in_channels=68, condition_channels=768, hidden_size=1152, depth=28, num_heads=16,
          attn_proj_bias=True, cond_drop_prob=0.1, gradient_checkpointing=False)  # configs/inference_dit.yml:52-62


def test_dit_state_dict_keys_match_reference_layout(pkg):
    cfg = dict(in_channels=68, condition_channels=96, hidden_size=384, depth=3)
    m = pkg.DiT(seq_length=64, num_heads=6, attn_proj_bias=True, cond_drop_prob=0.1, **cfg)
    want = synth.dit_shapes(**cfg)
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    m.load_state_dict(synth.dit_state_dict(SEED, **cfg), strict=True)
    # zero-init of adaLN / final layers as in the reference (dit_crossattn.py:173-182)
    fresh = pkg.DiT(seq_length=64, num_heads=6, attn_proj_bias=True, cond_drop_prob=0.1, **cfg)
    assert float(fresh.final_layer.linear.weight.detach().abs().max()) == 0.0
    assert float(fresh.blocks[0].adaLN_modulation[1].weight.detach().abs().max()) == 0.0


def test_dit_xl_has_515_tensors_909M_params(pkg):
    with torch.device("meta"):
        m = pkg.DiT(**XL)
    sd = m.state_dict()
    assert len(sd) == 515
    assert abs(sum(v.numel() for v in sd.values()) / 1e6 - 909.43) < 0.01
    assert tuple(sd["blocks.0.attn.qkv.weight"].shape) == (3456, 1152)
    assert tuple(sd["blocks.27.adaLN_modulation.1.weight"].shape) == (10368, 1152)
    assert tuple(sd["null_cond_embedding"].shape) == (768,)


def test_vae_state_dict_keys_match_reference(pkg, golden):
    g = golden("vae_decode")
    want = {str(k): eval(str(s)) for k, s in zip(g["keys"], g["shapes"])}
    vae = pkg.VAE(**VAE_CFG)
    got = {k: tuple(v.shape) for k, v in vae.state_dict().items()}
    assert got == want
    vae.load_state_dict(synth.state_dict_like(SEED, vae.state_dict()), strict=True)


def test_cpu_tensors_are_refused(pkg):
    cfg = dict(in_channels=68, condition_channels=96, hidden_size=384, depth=1)
    m = pkg.DiT(seq_length=64, num_heads=6, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    x, t, y = torch.zeros(1, 64, 68), torch.zeros(1, dtype=torch.long), torch.zeros(1, 3, 96)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(x, t, y, torch.float16, True)
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(x, t, y)                                 # the fp32 / no-amp signature default also runs on the HIP path only
    with pytest.raises(RuntimeError, match="no CPU path"):
        m.forward_with_cfg(x, t, y, 6.0)
    vae = pkg.VAE(**VAE_CFG)
    with pytest.raises(RuntimeError, match="no CPU path"):
        vae.decode(torch.zeros(2, 1, 4, 4, 4))
    with pytest.raises(RuntimeError):
        pkg.memory_efficient_attention(torch.zeros(1, 4, 2, 32, dtype=torch.float16),
                                       torch.zeros(1, 4, 2, 32, dtype=torch.float16),
                                       torch.zeros(1, 4, 2, 32, dtype=torch.float16))


def test_denoised_pt_round_trip(tmp_path):
    """The on-disk format between sampling and mesh extraction (inference.py:351-352) and the strict checkpoint loaders."""
    import torch
    from topia_xl_amd import pipeline
    recon = torch.randn(2, 16, 4 + 6 * 512)
    p = tmp_path / "denoised.pt"
    pipeline.save_denoised(str(p), recon, index=1)
    blob = torch.load(str(p))
    assert set(blob) == {"model_state_dict"} and set(blob["model_state_dict"]) == {"srt_param", "feat_param"}
    assert torch.equal(blob["model_state_dict"]["srt_param"], recon[1, :, :4])
    m = pipeline.primsdf_from_denoised(str(p))
    assert m.num_prims == 16 and m.prim_shape == 8 and torch.equal(m.feat_param.data, recon[1, :, 4:])
    import topia_xl_amd as pkg
    dit = pkg.DiT(seq_length=8, in_channels=4, condition_channels=8, hidden_size=64, depth=1, num_heads=2,
                  cond_drop_prob=0.1, attn_proj_bias=True)
    ck = tmp_path / "dit.pt"
    torch.save({"ema": {k: v.half() for k, v in dit.state_dict().items()}}, str(ck))     # fp16 file, like the release
    dit2 = pkg.DiT(seq_length=8, in_channels=4, condition_channels=8, hidden_size=64, depth=1, num_heads=2,
                   cond_drop_prob=0.1, attn_proj_bias=True)
    pipeline.load_checkpoints(model=dit2, dit_checkpoint_path=str(ck))
    assert all(torch.equal(a.half(), b.half()) for a, b in zip(dit.state_dict().values(), dit2.state_dict().values()))
