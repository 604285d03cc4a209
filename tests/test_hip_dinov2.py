"""GPU parity of the DINOv2 conditioner forward: HIP path vs the reference's vendored
implementation (fp32 golden) on the small configuration, and vs the oracle at the shipped ViT-B/14-reg shape."""
import numpy as np
import pytest
import torch

from oracle import dinov2_ref, synth
from tests.golden.make_golden import DINO_CFG, SEED, dino_state_dict
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: 5e-3, torch.bfloat16: 2e-2}   # vs the fp32 reference, as for the DiT forward


@pytest.fixture(scope="module")
def dino():
    import __graft_entry__
    __graft_entry__.build()
    from topia_xl_amd import dinov2
    return dinov2


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tag,size", [("native", 56), ("resampled", 84)])
def test_small_config_against_reference(dino, golden, tag, size, dtype):
    g = golden("dinov2")
    m = dino.DinoVisionTransformer(**DINO_CFG).eval()
    m.load_state_dict(dino_state_dict(m.state_dict()), strict=True)
    m.to(DEV)
    x = synth.tensor(SEED, f"dino.x.{size}", (2, 3, size, size)).to(DEV)
    out = m(x, is_training=True, precision_dtype=dtype)
    assert set(out) == {"x_norm_clstoken", "x_norm_regtokens", "x_norm_patchtokens", "x_prenorm", "masks"}
    for key, name in (("x_norm_clstoken", "cls"), ("x_norm_regtokens", "reg"), ("x_norm_patchtokens", "patch"),
                      ("x_prenorm", "prenorm")):
        assert out[key].dtype == torch.float32
        assert rel_l2(out[key], g[f"{tag}_{name}"]) < TOL[dtype], (name, rel_l2(out[key], g[f"{tag}_{name}"]))
    tok = m.conditioner_tokens(x, precision_dtype=dtype)
    assert tok.shape == (2, 1 + (size // 14) ** 2, 96)
    assert torch.equal(tok[:, 0], out["x_norm_clstoken"]) and torch.equal(tok[:, 1:], out["x_norm_patchtokens"])


def test_vitb14_reg_shape_against_oracle(dino):
    """The shipped conditioner (ViT-B/14 with 4 registers, 518 x 518 -> 1370 tokens x 768, the L x Dc of the DiT's
    cross-attention): synthetic weights, one image, HIP vs the fp32 oracle."""
    m = dino.vit_base(img_size=518, init_values=1.0, interpolate_antialias=True, interpolate_offset=0.0).eval()
    sd = dino_state_dict(m.state_dict())
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    x = synth.tensor(SEED, "dino.x.518", (1, 3, 518, 518))
    tok = m.conditioner_tokens(x.to(DEV))
    assert tok.shape == (1, 1370, 768)
    with torch.no_grad():
        torch.set_num_threads(16)
        ref = dinov2_ref.conditioner_tokens(sd, x, 14, 12)
    assert rel_l2(tok, ref) < TOL[torch.float16], rel_l2(tok, ref)
