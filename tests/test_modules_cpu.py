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


# ---- checkpoint -> packed 16-bit blob: host-side logic, runs on CPU tensors
_SMALL_DIT = dict(seq_length=16, in_channels=8, condition_channels=24, hidden_size=64, depth=2, num_heads=4, cond_drop_prob=0.1,
                  attn_proj_bias=True)


def _fp16_checkpoint(pkg):
    torch.manual_seed(SEED)
    m = pkg.DiT(**_SMALL_DIT)
    for prm in m.parameters():
        prm.data.normal_()
    return {k: v.half() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_packed_blob_straight_from_the_checkpoint_equals_the_repacked_parameters(pkg, dtype):
    sd = _fp16_checkpoint(pkg)
    a = pkg.DiT(**_SMALL_DIT)
    a.load_state_dict(sd, strict=True)                       # the reference's route: fp16 file -> fp32 parameters -> 16-bit
    b = pkg.DiT(**_SMALL_DIT)
    pb = b.pack_from_state_dict(sd, dtype)
    assert torch.equal(a.packed(dtype)["_flat"].view(torch.int16), pb["_flat"].view(torch.int16))
    for x, y in zip(a.small_fp32_tensors(), b.small_fp32_tensors()):
        assert torch.equal(x, y)
    assert b._packed_only
    with pytest.raises(RuntimeError, match="packed"):       # another dtype would need the fp32 parameters it never loaded
        b.packed(torch.bfloat16 if dtype == torch.float16 else torch.float16)


def test_pack_from_state_dict_is_strict(pkg):
    sd = _fp16_checkpoint(pkg)
    short = dict(sd)
    short.pop("blocks.0.mlp.fc1.bias")
    with pytest.raises(RuntimeError, match="missing keys"):
        pkg.DiT(**_SMALL_DIT).pack_from_state_dict(short, torch.float16)
    with pytest.raises(RuntimeError, match="unexpected keys"):
        pkg.DiT(**_SMALL_DIT).pack_from_state_dict({**sd, "pos_embed": torch.zeros(1)}, torch.float16)
    bad = dict(sd)
    bad["blocks.1.attn.qkv.weight"] = bad["blocks.1.attn.qkv.weight"][:-1]
    with pytest.raises(RuntimeError, match="shape mismatch"):
        pkg.DiT(**_SMALL_DIT).pack_from_state_dict(bad, torch.float16)


def test_packed_file_round_trip_and_layout_check(pkg, tmp_path):
    sd = _fp16_checkpoint(pkg)
    a = pkg.DiT(**_SMALL_DIT)
    a.load_state_dict(sd)
    path = str(tmp_path / "dit.primxpk")
    size = a.save_packed(path, torch.float16)
    assert size == os.path.getsize(path)
    with open(path, "rb") as f:
        assert f.read(8) == b"PRIMXPK1"
    c = pkg.DiT(**_SMALL_DIT)
    pc = c.load_packed(path)
    assert torch.equal(a.packed(torch.float16)["_flat"].view(torch.int16), pc["_flat"].view(torch.int16))
    for x, y in zip(a.small_fp32_tensors(), c.small_fp32_tensors()):
        assert torch.equal(x, y)
    with pytest.raises(RuntimeError, match="different model"):
        pkg.DiT(**{**_SMALL_DIT, "depth": 3}).load_packed(path)
    with pytest.raises(RuntimeError, match="not a packed"):
        bogus = str(tmp_path / "x.primxpk")
        open(bogus, "wb").write(b"\0" * 64)
        pkg.DiT(**_SMALL_DIT).load_packed(bogus)
    # a file of an older header format is refused with a message that says what to do (round-3 advisor finding: format 1 had
    # two incompatible layouts and answered "packed for a different model")
    import json
    raw = bytearray(open(path, "rb").read())
    hlen = int.from_bytes(raw[8:16], "little")
    head = json.loads(bytes(raw[16:16 + hlen]))
    assert head["format"] == pkg.DiT.PACKED_FORMAT == 2
    old_head = json.dumps({**head, "format": 1}, sort_keys=True).encode()
    assert len(old_head) == hlen
    raw[16:16 + hlen] = old_head
    legacy = str(tmp_path / "legacy.primxpk")
    open(legacy, "wb").write(bytes(raw))
    with pytest.raises(RuntimeError, match="repack required"):
        pkg.DiT(**_SMALL_DIT).load_packed(legacy)


def test_attention_workspaces_shared_between_shape_groups_survive_the_first_owner(pkg, monkeypatch):
    """Round-3 advisor finding: a workspace key used by several shape groups (the broadcast entries do not depend on the batch
    size) was owned by the group that allocated it first and freed with it.  Every using group now holds it."""
    from importlib import import_module
    ops = import_module(pkg.__name__ + ".ops")
    m = pkg.DiT(**_SMALL_DIT)
    made = []
    monkeypatch.setattr(ops, "alloc_heads", lambda *a, **k: made.append(a) or torch.zeros(1))
    dt, dev = torch.float16, "cpu"
    m._heads_begin("g1")
    shared = m._heads("Kn", 2, 90, 2, dt, dev, 64)
    m._heads("Qs", 2, 16, 0, dt, dev, 128)
    m._heads_begin("g2")
    assert m._heads("Kn", 2, 90, 2, dt, dev, 64) is shared and len(made) == 2          # a hit: g2 now holds it too
    m._heads("Qs", 4, 16, 0, dt, dev, 128)
    m._heads_begin("g3")
    m._heads_begin("g4")                                                                # evicts g1 (three groups stay)
    assert "g1" not in m._heads_lru
    m._heads_group = "g2"
    assert m._heads("Kn", 2, 90, 2, dt, dev, 64) is shared and len(made) == 3          # still there: g2 uses it
    n = len(made)
    m._heads("Qs", 2, 16, 0, dt, dev, 128)                                              # g1's private workspace is gone
    assert len(made) == n + 1
    m._heads_begin("g5")                                                                # evicts g2: now the last user is gone
    m._heads_group = "g5"
    m._heads("Kn", 2, 90, 2, dt, dev, 64)
    assert len(made) == n + 2


def test_load_checkpoints_takes_the_packed_routes(pkg, tmp_path):
    from importlib import import_module
    pipeline = import_module(pkg.__name__ + ".pipeline")
    sd = _fp16_checkpoint(pkg)
    pt = str(tmp_path / "model_sview_dit_fp16.pt")
    torch.save({"ema": sd}, pt)
    ref = pkg.DiT(**_SMALL_DIT)
    pipeline.load_checkpoints(model=ref, dit_checkpoint_path=pt)                                  # inference.py:257-259
    direct = pkg.DiT(**_SMALL_DIT)
    pipeline.load_checkpoints(model=direct, dit_checkpoint_path=pt, packed_dtype=torch.float16)
    want = ref.packed(torch.float16)["_flat"].view(torch.int16)
    assert torch.equal(want, direct._pack[(torch.float16, torch.device("cpu"))]["_flat"].view(torch.int16))
    pk_path = str(tmp_path / "dit.primxpk")
    ref.save_packed(pk_path, torch.float16)
    mapped = pkg.DiT(**_SMALL_DIT)
    pipeline.load_checkpoints(model=mapped, dit_checkpoint_path=pk_path)
    assert torch.equal(want, mapped._pack[(torch.float16, torch.device("cpu"))]["_flat"].view(torch.int16))


def test_additive_pos_emb_variant_travels_on_the_packed_routes(pkg, tmp_path):
    """DiTAdditivePosEmb has parameters the base class lacks (point_emb.mlp, fp32 next to x_embedder) and a buffer
    (point_emb.basis): the packed routes must carry the former, tolerate the latter, and refuse a base-class file."""
    cfg = dict(seq_length=8, in_channels=4, condition_channels=8, hidden_size=64, depth=2, num_heads=2, attn_proj_bias=True)
    a = pkg.DiTAdditivePosEmb(**cfg)
    for prm in a.parameters():
        prm.data.normal_()
    sd = {k: v.clone() for k, v in a.state_dict().items()}
    assert "point_emb.basis" in sd and "point_emb.mlp.weight" in sd
    names = {id(p): n for n, p in a.named_parameters()}
    small = [names[id(p)] for p in a._small_fp32_params()]
    assert "point_emb.mlp.weight" in small and "point_emb.mlp.bias" in small
    assert a._hyper()["class"] == "DiTAdditivePosEmb"
    b = pkg.DiTAdditivePosEmb(**cfg)
    b.pack_from_state_dict(sd, torch.float16)                  # the buffer key is not "unexpected"
    assert torch.equal(b.point_emb.mlp.weight, a.point_emb.mlp.weight) and torch.equal(b.point_emb.mlp.bias, a.point_emb.mlp.bias)
    path = str(tmp_path / "addpos.primxpk")
    a.save_packed(path, torch.float16)
    c = pkg.DiTAdditivePosEmb(**cfg)
    c.load_packed(path)
    assert torch.equal(c.point_emb.mlp.weight, a.point_emb.mlp.weight)
    assert torch.equal(c.packed(torch.float16)["_flat"].view(torch.int16), a.packed(torch.float16)["_flat"].view(torch.int16))
    with pytest.raises(RuntimeError, match="different model"):   # same hyper-parameters, other class: other parameter set
        pkg.DiT(cond_drop_prob=0.0, **cfg).load_packed(path)


def test_layernorm_fold_host_logic(pkg, monkeypatch):
    """The fold's host side without a device: (a) the A operands of its u / v GEMMs pick shift / scale of (block, site) out of the
    loop's modulation table exactly as DiTBlock chunks its adaLN output (dit_crossattn.py:54: shift, scale, gate x cross-attention,
    self-attention, MLP) and form (1 + scale) in the 16-bit type; (b) the shape rule that decides whether a forward folds follows
    the kernel-selection switches; (c) the folded algebra itself - (rho / rho_p) cast16((x - c) rho_p m) W^T - rho (mu - c) u + v - against
    the reference's LayerNorm -> modulate -> Linear in float64, with (centre, scale) one gated branch away from (mean, rstd), at
    row spreads of 2, 3e4 and 1e-5 on a magnitude of 1e4 (fp16: the ABI-22 operand cast16((x - c) m) fails the last two)."""
    from importlib import import_module
    import __graft_entry__
    __graft_entry__.build()                      # (b) asks the library whether it carries the fold entry points of this ABI
    ops = import_module(pkg.__name__ + ".ops")
    depth, D, n = 3, 16, 5
    tab = synth.tensor(7, "fold.tab", (n, depth * 9 * D + 2 * D), 0.5).to(torch.float16)
    A = pkg.DiT._fold_rows(tab, depth, D)
    assert A.shape == (depth, 3, 2, n, D) and A.dtype == torch.float16
    for i in range(depth):
        for s in range(3):
            shift = tab[:, (i * 9 + 3 * s) * D:(i * 9 + 3 * s + 1) * D]
            scale = tab[:, (i * 9 + 3 * s + 1) * D:(i * 9 + 3 * s + 2) * D]
            assert torch.equal(A[i, s, 1], shift)
            assert torch.equal(A[i, s, 0], (1.0 + scale.float()).to(torch.float16))
    # (b)
    for var in ("PRIMX_GEMM_NOBIG", "PRIMX_GEMM_LOADER", "PRIMX_GEMM_BIGHEADS_MIN"):
        monkeypatch.delenv(var, raising=False)
    assert ops.fold_supported(1152, 16) and not ops.fold_supported(384, 6) and not ops.fold_supported(1152 * 2, 32)
    assert ops.fold_shapes_ok(4096, 2048, 1152, 16) and ops.fold_shapes_ok(32768, 2048, 1152, 16)
    assert not ops.fold_shapes_ok(2048, 2048, 1152, 16)          # qkv would not reach the 256 x 288 heads tile
    assert not ops.fold_shapes_ok(4096 + 300, 2198, 1152, 16)    # token count per batch entry not a multiple of 256
    monkeypatch.setenv("PRIMX_GEMM_NOBIG", "1")
    assert not ops.fold_shapes_ok(4096, 2048, 1152, 16)
    monkeypatch.delenv("PRIMX_GEMM_NOBIG")
    monkeypatch.setenv("PRIMX_GEMM_BIGHEADS_MIN", "0")
    assert not ops.fold_shapes_ok(4096, 2048, 1152, 16)
    monkeypatch.delenv("PRIMX_GEMM_BIGHEADS_MIN")
    # (c) float64 algebra with the kernel's rounding points (fp16)
    r16 = lambda t: t.to(torch.float16).double()
    M, Dm, O = 64, 1152, 96
    shift, scale = r16(synth.tensor(8, "fold.sh", (Dm,), 0.3)), r16(synth.tensor(8, "fold.sc", (Dm,), 0.3))
    W, b = r16(synth.tensor(8, "fold.W", (O, Dm), Dm ** -0.5)), r16(synth.tensor(8, "fold.b", (O,), 0.2))
    m = r16(1 + scale)
    u, v = (m @ W.t()).float().double(), (shift @ W.t() + b).float().double()
    for spread, offset in ((2.0, 0.8), (3e4, 1e4), (1e-5, 0.0)):
        x = (synth.tensor(8, "fold.x", (M, Dm)).double() * spread + offset).float().double()
        mu = x.mean(-1, keepdim=True)
        rho = 1.0 / torch.sqrt(((x - mu) ** 2).mean(-1, keepdim=True) + 1e-6)
        ref = ((x - mu) * rho * m + shift) @ W.t() + b
        autocast = r16(r16((x - mu) * rho * m + shift) @ W.t() + b)
        c = mu + 0.1 * x.std(-1, keepdim=True)                    # the centre: the mean one branch ago
        rho_p = (1.1 * rho).float().double()                      # the scale: the rstd one branch ago
        d = (x - c).float().double()
        part = torch.stack([d.view(M, 8, 144).sum(-1), (d * d).view(M, 8, 144).sum(-1)], -1).float().double()   # the producer's fp32 partials
        mu_p = part[..., 0].sum(-1, keepdim=True) / Dm
        rho_f = 1.0 / torch.sqrt(part[..., 1].sum(-1, keepdim=True) / Dm - mu_p ** 2 + 1e-6)
        a16 = r16((d * rho_p).float().double() * m)
        assert bool(torch.isfinite(a16).all()) and float(a16.abs().max()) < 64
        folded = r16((rho_f / rho_p) * (a16 @ W.t() - rho_p * mu_p * u) + v)
        e_ref = lambda t: float((t - ref).norm() / ref.norm())
        assert e_ref(folded) < 1.2 * e_ref(autocast) + 1e-5 and e_ref(folded) < 6e-4, (spread, e_ref(folded), e_ref(autocast))
        if spread == 3e4:                                         # what ABI 22 stored
            assert not bool(torch.isfinite(r16(d * m)).all())


def test_fold_overflowed_answers_once_per_loop(pkg):
    """The sampling loop asks `DiT.fold_overflowed` about its final sample: True if - and only if - folded fp16 forwards ran since
    the last question and the sample is non-finite (the sampler then repeats the loop with LayerNorm launches)."""
    m = pkg.DiT(seq_length=64, in_channels=8, condition_channels=16, hidden_size=48, depth=1, num_heads=2).eval()
    bad, good = torch.tensor([1.0, float("nan")]), torch.ones(4)
    assert not m.fold_overflowed(bad)                # nothing folded: nothing to answer for
    m._fold_fp16_used = True
    assert not m.fold_overflowed(good)
    assert m._fold_fp16_used is False
    m._fold_fp16_used = True
    assert m.fold_overflowed(bad)
    assert m._fold_fp16_used is False
    assert not m.fold_overflowed(bad)
    m._fold_fp16_used = True
    m.clear_timestep_plan()                          # a loop that ended early leaves nothing for the next one
    assert m._fold_fp16_used is False
