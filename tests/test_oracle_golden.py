"""The oracle is pinned against outputs of the REAL reference (tests/golden/*.npz, produced by
tests/golden/make_golden.py from /root/reference).  fp32 everywhere; tolerances only absorb
summation-order differences (the oracle evaluates attention in float64)."""
import numpy as np
import torch

from oracle import diffusion_ref as dref
from oracle import dit_ref, synth, vae_ref
from tests.golden.make_golden import DIT_CASES, SEED, VAE_CFG


def _close(a, b, atol, rtol=1e-4):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max()
    assert np.allclose(a, b, atol=atol, rtol=rtol), f"max abs err {err:.3e} (ref scale {np.abs(b).max():.3e})"


def test_schedule_tables_bit_exact(golden):
    g = golden("schedule")
    for n in (5, 25, 50, 100, 200):
        tab, tmap = dref.make("squaredcos_cap_v2", 1000, f"ddim{n}")
        assert tmap == g[f"ddim{n}_map"].tolist()
        for mine, ref in ((tab.betas, "betas"), (tab.acp, "alphas_cumprod"), (tab.acp_prev, "alphas_cumprod_prev"),
                          (tab.sqrt_acp, "sqrt_alphas_cumprod"), (tab.sqrt_1m_acp, "sqrt_one_minus_alphas_cumprod"),
                          (tab.sqrt_recip_acp, "sqrt_recip_alphas_cumprod"),
                          (tab.sqrt_recipm1_acp, "sqrt_recipm1_alphas_cumprod"), (tab.post_var, "posterior_variance"),
                          (tab.post_logvar_clipped, "posterior_log_variance_clipped"),
                          (tab.post_c1, "posterior_mean_coef1"), (tab.post_c2, "posterior_mean_coef2")):
            assert np.array_equal(mine, g[f"ddim{n}_{ref}"]), (n, ref)
    assert np.array_equal(dref.cosine_betas(1000), g["cos1000_betas"])
    assert np.array_equal(dref.linear_betas(1000), g["lin1000_betas"])
    assert np.array_equal(dref.linear_betas(250), g["lin250_betas"])
    # survey anchors
    tab, _ = dref.make("squaredcos_cap_v2", 1000, "ddim25")
    assert abs(tab.acp[0] - 0.9999587158) < 1e-10 and abs(tab.acp[-1] - 3.689611e-3) < 1e-9


def test_timestep_embedding(golden):
    g = golden("dit_dh64")
    mine = dit_ref.timestep_embedding(torch.tensor([0, 1, 40, 500, 960, 999]))
    assert np.array_equal(mine.numpy(), g["t_emb_freq"])


def _case(i):
    name, cfg, heads, N, L, B = DIT_CASES[i]
    sd = synth.dit_state_dict(SEED, **cfg)
    x = synth.tensor(SEED, name + ".x", (B, N, cfg["in_channels"]))
    y = synth.tensor(SEED, name + ".y", (B, L, cfg["condition_channels"]))
    t = torch.tensor([960, 40][:B], dtype=torch.int64)
    return name, sd, heads, x, y, t


def test_dit_forward_matches_reference(golden):
    for i in range(len(DIT_CASES)):
        name, sd, heads, x, y, t = _case(i)
        g = golden(name)
        out = dit_ref.dit_forward(sd, x, t, y, heads)
        assert np.abs(g["forward"]).max() > 0.1, "vacuous golden"
        _close(out, g["forward"], atol=2e-4)
        _close(dit_ref.dit_forward_with_cfg(sd, x, t, y, heads, 6.0), g["forward_cfg"], atol=1e-3)
        te = dit_ref.timestep_embedding(t)
        import torch.nn.functional as F
        te = F.linear(F.silu(F.linear(te, sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])),
                      sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])
        h0 = F.linear(x, sd["x_embedder.weight"], sd["x_embedder.bias"])
        _close(dit_ref.dit_block(sd, 0, h0, y, te, heads, None), g["block0"], atol=2e-4)


def test_ddim_trajectory_matches_reference(golden):
    name, sd, heads, x, y, t = _case(1)
    g = golden(name)
    tab, tmap = dref.make("squaredcos_cap_v2", 1000, "ddim5")
    assert tmap == [0, 200, 400, 600, 800]
    model = lambda xx, tt, **kw: dit_ref.dit_forward_with_cfg(sd, xx, tt, y, heads, 6.0)
    outs = dref.ddim_loop(model, x, tab, tmap, "v", 0.0, False)
    traj = np.stack([o["sample"].numpy() for o in outs])
    _close(traj, g["ddim5_samples"], atol=2e-3)
    _close(outs[-1]["pred_xstart"], g["ddim5_last_pred_xstart"], atol=2e-3)
    # p_mean_variance pieces at spaced t = 3 (ancestral path: posterior mean + learned-range log-variance)
    out = model(x, torch.full((x.shape[0],), tmap[3]))
    C = x.shape[-1]
    step = dref.ancestral_step(tab, 3, x, out, torch.zeros_like(x))
    _close(step["sample"], g["pmv_mean"], atol=1e-3)  # zero noise -> sample == mean
    frac = (out[..., C:] + 1) / 2
    logvar = frac * float(np.float32(np.log(tab.betas)[3])) + (1 - frac) * float(np.float32(tab.post_logvar_clipped[3]))
    _close(logvar, g["pmv_log_variance"], atol=1e-3)


def test_configs0_named_shape_trajectory_matches_reference(golden):
    """BASELINE configs[0] at its named shape (depth 12, d = 384, 6 heads, N_prim = 256, one condition token, 5 DDIM steps with
    CFG; tests/golden/make_golden_xl.py `c1_named`): the oracle's loop against the unmodified reference's."""
    from tests.golden.make_golden_xl import C1, C1_HEADS, XL_SEED, xl_inputs
    depth, N, B, stride, x, y = xl_inputs("c1_named")
    g = golden("c1_named")
    sd = synth.dit_state_dict(XL_SEED, depth=depth, **C1)
    tab, tmap = dref.make("squaredcos_cap_v2", 1000, "ddim5")
    model = lambda xx, tt, **kw: dit_ref.dit_forward_with_cfg(sd, xx, tt, y, C1_HEADS, 6.0)
    outs = dref.ddim_loop(model, x, tab, tmap, "v", 0.0, False)
    assert (depth, N, tuple(y.shape)) == (12, 256, (1, 1, 768)) and np.abs(g["ddim5_samples"][-1]).max() > 0.1
    _close(np.stack([o["sample"].numpy() for o in outs]), g["ddim5_samples"], atol=2e-3)
    _close(outs[-1]["pred_xstart"], g["ddim5_final_pred_xstart"], atol=2e-3)


def test_attention_modules_match_reference(golden):
    g = golden("attention")
    # MemEffAttention(dim=256, heads=8, qkv_bias=False)  /  MemEffCrossAttention(dim=144, heads=2)
    sd = synth.state_dict_like(SEED, {"qkv.weight": torch.empty(768, 256), "proj.weight": torch.empty(256, 256),
                                      "proj.bias": torch.empty(256)})
    x = synth.tensor(SEED, "att.x", (3, 64, 256))
    _close(dit_ref.self_attention(sd, "", x, 8, None), g["self_dh32"], atol=1e-5)
    shapes = {"to_q.weight": (144, 144), "to_q.bias": (144,), "to_k.weight": (144, 40), "to_k.bias": (144,),
              "to_v.weight": (144, 40), "to_v.bias": (144,), "proj.weight": (144, 144), "proj.bias": (144,)}
    sdc = synth.state_dict_like(SEED, {k: torch.empty(v) for k, v in shapes.items()})
    q = synth.tensor(SEED, "catt.q", (2, 96, 144))
    kv = synth.tensor(SEED, "catt.kv", (2, 37, 40))
    _close(dit_ref.cross_attention(sdc, "", q, kv, 2, None), g["cross_dh72"], atol=1e-5)


def vae_synth_state_dict(golden):
    g = golden("vae_decode")
    keys = [str(k) for k in g["keys"]]
    shapes = [eval(str(s)) for s in g["shapes"]]
    return synth.state_dict_like(SEED, {k: torch.empty(s) for k, s in zip(keys, shapes)})


def test_vae_decode_matches_reference(golden):
    g = golden("vae_decode")
    sd = vae_synth_state_dict(golden)
    z = synth.tensor(SEED, "vae.z", (3, 1, 4, 4, 4))
    out = vae_ref.vae_decode(sd, z, VAE_CFG["up_channels"], VAE_CFG["layers_per_block"])
    assert out.shape == (3, 6, 8, 8, 8) and np.abs(g["decoded"]).max() > 0.05
    _close(out, g["decoded"], atol=2e-5)


def _addpos_case(module_cls):
    """State dict exactly as tests/golden/make_golden.py:gen_dit_addpos built it (key set taken from the module)."""
    cfg = dict(in_channels=68, condition_channels=64, hidden_size=288, depth=2)
    N, L, B, heads = 96, 37, 2, 4
    m = module_cls(seq_length=N, num_heads=heads, attn_proj_bias=True, **cfg).eval()
    sd = synth.state_dict_like(SEED, m.state_dict())
    sd["point_emb.basis"] = m.point_emb.basis.clone()
    x = synth.tensor(SEED, "addpos.x", (B, N, cfg["in_channels"]))
    y = synth.tensor(SEED, "addpos.y", (B, L, cfg["condition_channels"]))
    t = torch.tensor([960, 40], dtype=torch.int64)
    return m, sd, heads, x, y, t


def test_dit_additive_pos_emb_matches_reference(golden):
    """DiTAdditivePosEmb (dit_crossattn.py:215-301): the module mirror has the reference's key set and the oracle
    restatement (point embedding + the shared block stack) reproduces the real reference's output."""
    from topia_xl_amd.dit import DiTAdditivePosEmb
    g = golden("dit_addpos")
    m, sd, heads, x, y, t = _addpos_case(DiTAdditivePosEmb)
    assert sorted(sd.keys()) == list(g["keys"])                      # strict checkpoint compatibility
    m.load_state_dict(sd, strict=True)
    assert not hasattr(m, "null_cond_embedding")
    with torch.no_grad():
        pe = dit_ref.point_embed(sd, x[:, :, 1:4])
        out = dit_ref.dit_forward(sd, x, t, y, heads)
    _close(pe.numpy(), g["point_emb"], 2e-5)
    _close(out.numpy(), g["forward"], 2e-4)
