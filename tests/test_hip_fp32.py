"""The reference's fp32 call path on the HIP device (csrc/fp32.hip): `DiT.forward(x, t, y)` with the signature defaults
(precision_dtype=float32, enable_amp=False - models/dit_crossattn.py:184) and `precision: tf32` of the CLI
(inference.py:239-247).  Exact fp32 on v_mfma_f32_32x32x2_f32; stated tolerance vs the fp32 goldens of the REAL
reference: rel-L2 <= 1e-4."""
import pytest
import torch

from oracle import dit_ref, synth
from tests.golden.make_golden import DIT_CASES, SEED
from tests.util import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL32 = 1e-4


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__
    __graft_entry__.build()
    from topia_xl_amd import ops
    return ops


@pytest.fixture(scope="module")
def pkg(ops):
    import topia_xl_amd
    return topia_xl_amd


@pytest.mark.parametrize("M,N,K", [(200, 136, 68), (257, 384, 1152), (2, 300, 256), (512, 128, 4608)])
def test_gemm_f32_plain_act_scale(ops, M, N, K):
    A, W, b = synth.tensor(1, "A", (M, K)), synth.tensor(1, "W", (N, K), K ** -0.5), synth.tensor(1, "b", (N,), 0.1)
    ref = A.double() @ W.double().T + b.double()
    out = ops.gemm_f32(A.to(DEV), W.to(DEV), b.to(DEV))
    assert rel_l2(out, ref) < 2e-6
    out = ops.gemm_f32(A.to(DEV), W.to(DEV), None, out_scale=0.25)
    assert rel_l2(out, 0.25 * (A.double() @ W.double().T)) < 2e-6
    out = ops.gemm_f32(A.to(DEV), W.to(DEV), b.to(DEV), act=1)
    assert rel_l2(out, torch.nn.functional.gelu(ref, approximate="tanh")) < 2e-6


def test_gemm_f32_gate_residual_in_place(ops):
    B, n, N, K = 3, 70, 288, 96
    M = B * n
    A, W, b = synth.tensor(2, "A", (M, K)), synth.tensor(2, "W", (N, K), K ** -0.5), synth.tensor(2, "b", (N,), 0.1)
    mod = synth.tensor(2, "mod", (B, 3 * N))
    x = synth.tensor(2, "x", (M, N))
    gate = mod[:, N:2 * N]
    ref = x.double() + gate.double().repeat_interleave(n, 0) * (A.double() @ W.double().T + b.double())
    xd = x.to(DEV).clone()
    ops.gemm_f32(A.to(DEV), W.to(DEV), b.to(DEV), out=xd, gate=mod.to(DEV)[:, N:2 * N], rows_per_batch=n)
    assert rel_l2(xd, ref) < 2e-6


@pytest.mark.parametrize("dh,H,Nq,Nk", [(72, 4, 200, 200), (72, 2, 130, 45), (64, 3, 96, 1), (32, 8, 64, 64), (128, 1, 33, 70)])
def test_attention_f32_strided_views(ops, dh, H, Nq, Nk):
    """self-attention on the unbind() views of a fused qkv buffer (attention.py:50-54) and cross-attention on separate
    buffers, ragged query / key counts, vs float64."""
    B = 2
    if Nq == Nk:
        qkv = synth.tensor(3, f"qkv{dh}", (B, Nq, 3, H, dh)).to(DEV)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q = synth.tensor(3, "q", (B, Nq, H, dh)).to(DEV)
        k = synth.tensor(3, "k", (B, Nk, H, dh)).to(DEV)
        v = synth.tensor(3, "v", (B, Nk, H, dh)).to(DEV)
    for scale in (None, 1.0 / dh):
        s = dh ** -0.5 if scale is None else scale
        ref = dit_ref.attention_core(q.cpu(), k.cpu(), v.cpu(), s)
        out = ops.attention_f32(q, k, v, scale)
        assert out.shape == (B, Nq, H, dh) and out.is_contiguous()
        assert max_abs(out, ref) < 5e-6 * max(1.0, float(ref.abs().max())), (dh, Nq, Nk)


def test_attention_f32_large_logits(ops):
    """online softmax across key tiles with a late, much larger score (rescale path) and a first-tile maximum."""
    B, H, N, dh = 1, 2, 160, 72
    q, k, v = (synth.tensor(4, n, (B, N, H, dh)) for n in "qkv")
    k[:, 150] = 6.0 * q[:, 3]
    k[:, 2] = 4.0 * q[:, 77]
    ref = dit_ref.attention_core(q, k, v, dh ** -0.5)
    out = ops.attention_f32(q.to(DEV), k.to(DEV), v.to(DEV))
    assert max_abs(out, ref) < 1e-5


def test_layernorm_modulate_f32_and_silu(ops):
    B, n, D = 2, 37, 1152
    x = synth.tensor(5, "x", (B * n, D), 3.0, 0.5)
    mod = synth.tensor(5, "mod", (B, 2 * D), 0.3)
    ref = torch.nn.functional.layer_norm(x.double(), (D,), None, None, 1e-6).view(B, n, D) * (1 + mod[:, D:].double()[:, None]) \
        + mod[:, :D].double()[:, None]
    md = mod.to(DEV)
    out = ops.layernorm_modulate_f32(x.to(DEV), md[:, :D], md[:, D:], n, 1e-6)
    assert rel_l2(out, ref.view(B * n, D)) < 1e-6
    t = synth.tensor(5, "t", (3, 1152), 2.0)
    assert max_abs(ops.silu_f32(t.to(DEV)), torch.nn.functional.silu(t.double())) < 1e-6


@pytest.mark.parametrize("case", [0, 1])
def test_dit_fp32_default_signature_against_reference_golden(pkg, golden, case):
    """`model(x, t, y)` - no precision arguments, exactly as the reference's signature defaults - and
    `forward_with_cfg(x, t, y, cfg_scale)` vs the REAL reference's fp32 outputs."""
    name, cfg, heads, N, L, B = DIT_CASES[case]
    sd = synth.dit_state_dict(SEED, **cfg)
    m = pkg.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    x = synth.tensor(SEED, name + ".x", (B, N, cfg["in_channels"])).to(DEV)
    y = synth.tensor(SEED, name + ".y", (B, L, cfg["condition_channels"])).to(DEV)
    t = torch.tensor([960, 40][:B], dtype=torch.int64, device=DEV)
    g = golden(name)
    out = m(x, t, y)
    assert out.dtype == torch.float32 and out.shape == g["forward"].shape
    assert rel_l2(out, g["forward"]) < TOL32, rel_l2(out, g["forward"])
    cfg_out = m.forward_with_cfg(x, t, y, cfg_scale=6.0)
    assert cfg_out.dtype == torch.float32 and cfg_out.shape == g["forward_cfg"].shape
    assert rel_l2(cfg_out, g["forward_cfg"]) < TOL32, rel_l2(cfg_out, g["forward_cfg"])
    # `precision: tf32` of the CLI = (float32, enable_amp=False); autocast to float32 is the same arithmetic
    assert torch.equal(m(x, t, y, torch.float32, False), out) and torch.equal(m(x, t, y, torch.float32, True), out)
    # the sampler drives it like any other model callable
    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    traj = [s["sample"] for s in d.ddim_sample_loop_progressive(m.forward_with_cfg, x.shape, noise=x, clip_denoised=False,
                                                                model_kwargs=dict(y=y, cfg_scale=6.0), device=DEV)]
    for i in range(5):
        assert rel_l2(traj[i], g["ddim5_samples"][i]) < 2 * TOL32, (i, rel_l2(traj[i], g["ddim5_samples"][i]))


def test_dit_fp32_full_width_block(pkg, golden):
    """One DiT-XL block at the BASELINE width (d = 1152, 16 x 72 heads, 2048 tokens, 1370 condition tokens, CFG) in
    fp32 vs the REAL reference's fp32 output (tests/golden/xl_c3blk.npz holds batch 8; the first two entries are used)."""
    from tests.golden.make_golden_xl import HEADS, XL, XL_SEED, xl_inputs
    depth, N, B, stride, x, y = xl_inputs("xl_c3blk")
    cfg = dict(depth=depth, **XL)
    with torch.device(DEV):
        m = pkg.DiT(seq_length=N, num_heads=HEADS, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(synth.dit_state_dict(XL_SEED, **cfg), strict=True)
    g = golden("xl_c3blk")
    t = torch.as_tensor(g["t"])[:2].to(DEV)
    out = m.forward_with_cfg(x[:2].to(DEV), t, y[:2].to(DEV), cfg_scale=6.0)
    err = rel_l2(out[:, ::stride], g["forward_cfg"][:2])
    print(f"fp32 full-width block vs reference fp32: rel-L2 = {err:.3e}")
    assert err < TOL32, err
