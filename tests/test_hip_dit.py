"""GPU parity of the DiT forward, CFG and the DDIM loop (through the module mirrors and the C ABI):
 * against the oracle with the reference's autocast rounding points emulated (tight),
 * against the golden outputs of the REAL reference in fp32 (the stated fp16 / bf16 tolerance)."""
import numpy as np
import pytest
import torch

from oracle import diffusion_ref as dref
from oracle import dit_ref, synth
from tests.golden.make_golden import DIT_CASES, SEED
from tests.util import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# stated tolerances: single forward vs fp32 reference rel-L2 <= 5e-3 (fp16) / 2e-2 (bf16)
TOL_FWD = {torch.float16: 5e-3, torch.bfloat16: 2e-2}
TOL_EMU = {torch.float16: 2.5e-3, torch.bfloat16: 1.5e-2}   # vs oracle with the same rounding points


@pytest.fixture(scope="module")
def pkg():
    import __graft_entry__
    __graft_entry__.build()
    import topia_xl_amd
    return topia_xl_amd


def _case(pkg, i):
    name, cfg, heads, N, L, B = DIT_CASES[i]
    sd = synth.dit_state_dict(SEED, **cfg)
    m = pkg.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    x = synth.tensor(SEED, name + ".x", (B, N, cfg["in_channels"]))
    y = synth.tensor(SEED, name + ".y", (B, L, cfg["condition_channels"]))
    t = torch.tensor([960, 40][:B], dtype=torch.int64)
    return name, sd, heads, m, x, y, t


@pytest.mark.parametrize("case", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_forward_and_cfg(pkg, golden, case, dtype):
    name, sd, heads, m, x, y, t = _case(pkg, case)
    g = golden(name)
    out = m(x.to(DEV), t.to(DEV), y.to(DEV), dtype, True)
    assert out.dtype == dtype and out.shape == g["forward"].shape
    emu = dit_ref.dit_forward(sd, x, t, y, heads, dtype)
    assert rel_l2(out, emu) < TOL_EMU[dtype], ("emulated", rel_l2(out, emu))
    assert rel_l2(out, g["forward"]) < TOL_FWD[dtype], ("reference fp32", rel_l2(out, g["forward"]))
    cfg = m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, dtype, True)
    assert cfg.shape == g["forward_cfg"].shape                      # B, not 2B (dit_crossattn.py:213)
    assert rel_l2(cfg, g["forward_cfg"]) < 3 * TOL_FWD[dtype]         # guidance amplifies the difference 6x-ish
    assert rel_l2(cfg, dit_ref.dit_forward_with_cfg(sd, x, t, y, heads, 6.0, dtype)) < 3 * TOL_EMU[dtype]


def test_ddim_trajectory_against_reference(pkg, golden):
    """5-step DDIM with CFG 6 from the same noise: every step's sample vs the REAL reference (fp32)."""
    name, sd, heads, m, x, y, t = _case(pkg, 1)
    g = golden(name)
    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
    traj = []
    for i, s in enumerate(d.ddim_sample_loop_progressive(m.forward_with_cfg, x.shape, noise=x.to(DEV),
                                                         clip_denoised=False, model_kwargs=kw, device=DEV)):
        assert set(s) == {"sample", "pred_xstart"} and s["sample"].dtype == torch.float32
        traj.append(s["sample"].cpu().numpy())
    assert len(traj) == 5 == d.num_timesteps
    for i in range(5):
        assert rel_l2(traj[i], g["ddim5_samples"][i]) < 2e-2, (i, rel_l2(traj[i], g["ddim5_samples"][i]))
    final = d.ddim_sample_loop(m.forward_with_cfg, x.shape, noise=x.to(DEV), clip_denoised=False, model_kwargs=kw)
    assert np.array_equal(final.cpu().numpy(), traj[-1])             # deterministic (eta = 0)
    assert rel_l2(s["pred_xstart"], g["ddim5_last_pred_xstart"]) < 2e-2


def test_single_step_apis_and_ancestral(pkg, golden):
    name, sd, heads, m, x, y, t = _case(pkg, 1)
    g = golden(name)
    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
    tt = torch.full((x.shape[0],), 4, dtype=torch.int64, device=DEV)
    one = d.ddim_sample(m.forward_with_cfg, x.to(DEV), tt, clip_denoised=False, model_kwargs=kw)
    assert rel_l2(one["sample"], g["ddim5_samples"][0]) < 2e-2
    # ancestral (p_sample) path: finite, right shapes, mean part matches the reference's p_mean_variance at t = 3
    torch.manual_seed(0)
    t3 = torch.full((x.shape[0],), 3, dtype=torch.int64, device=DEV)
    out = d.p_sample(m.forward_with_cfg, x.to(DEV), t3, clip_denoised=False, model_kwargs=kw)
    assert torch.isfinite(out["sample"]).all()
    tab, tmap = dref.make("squaredcos_cap_v2", 1000, "ddim5")
    mo = m.forward_with_cfg(x.to(DEV), torch.full((x.shape[0],), tmap[3], device=DEV), **kw).cpu()
    ref = dref.ancestral_step(tab, 3, x, mo, torch.zeros_like(x))
    noise_free = out["sample"].cpu() - ref["sample"]
    sd_ref = torch.exp(0.5 * torch.as_tensor(g["pmv_log_variance"]))
    assert 0.5 < float((noise_free / sd_ref).std()) < 1.5            # sample = mean + exp(.5 logvar) * N(0,1)


def test_seq_length_is_not_baked_in(pkg):
    """seq_length is stored but unused (dit_crossattn.py:134): same weights, another token count, ragged L."""
    name, sd, heads, m, x, y, t = _case(pkg, 1)
    x2 = synth.tensor(3, "x2", (1, 96, 68))
    y2 = synth.tensor(3, "y2", (1, 1, 64))
    out = m(x2.to(DEV), t[:1].to(DEV), y2.to(DEV), torch.float16, True)
    assert rel_l2(out, dit_ref.dit_forward(sd, x2, t[:1], y2, heads, torch.float16)) < TOL_EMU[torch.float16]


def test_repack_after_weight_update(pkg):
    name, sd, heads, m, x, y, t = _case(pkg, 0)
    a = m(x.to(DEV), t.to(DEV), y.to(DEV), torch.float16, True).clone()
    sd2 = {k: v * 1.01 for k, v in sd.items()}
    m.load_state_dict(sd2)
    b = m(x.to(DEV), t.to(DEV), y.to(DEV), torch.float16, True)
    assert rel_l2(b, dit_ref.dit_forward(sd2, x, t, y, heads, torch.float16)) < TOL_EMU[torch.float16]
    assert rel_l2(a, b) > 1e-3


def test_packed_only_model_from_checkpoint_and_from_packed_file(pkg, tmp_path):
    """N4: fp16 checkpoint -> packed blob (no fp32 parameters) and the mapped `.primxpk` file run the same kernels on the same
    bytes as the reference's load_state_dict route: outputs are bit-identical; the fp32 route refuses on such a model."""
    name, sd, heads, m, x, y, t = _case(pkg, 0)
    cfg = DIT_CASES[0][1]
    N = DIT_CASES[0][3]
    sd16 = {k: v.half() for k, v in sd.items()}
    m.load_state_dict(sd16)
    want = m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, torch.float16, True).clone()
    direct = pkg.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval().to(DEV)
    direct.pack_from_state_dict(sd16, torch.float16)
    assert torch.equal(direct.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, torch.float16, True), want)
    with pytest.raises(RuntimeError, match="packed"):
        direct(x.to(DEV), t.to(DEV), y.to(DEV), torch.float32, False)
    path = str(tmp_path / "dit.primxpk")
    m.save_packed(path, torch.float16)
    mapped = pkg.DiT(seq_length=N, num_heads=heads, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval().to(DEV)
    mapped.load_packed(path)
    assert mapped.packed(torch.float16)["_flat"].is_cuda
    assert torch.equal(mapped.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, torch.float16, True), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_planned_and_unplanned_loops_give_identical_samples(pkg, dtype, monkeypatch):
    """The sampler announces its timesteps (DiT.plan_timesteps: the timestep-only adaLN modulation of the whole loop goes through
    the few-row kernel eight rows per pass instead of one pass per step).  Same kernels, same per-row arithmetic: every
    step's sample is bit-identical to the loop that computes the modulation per step, and the plan does not outlive the loop."""
    from importlib import import_module
    sampler = import_module(pkg.__name__ + ".diffusion.sampler")
    monkeypatch.setattr(sampler, "PLAN_TIMESTEPS", True)        # (the test is about planning: independent of PRIMX_PLAN_TIMESTEPS=0 in the environment)
    name, sd, heads, m, x, y, t = _case(pkg, 1)
    d = pkg.create_diffusion("ddim10", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=dtype, enable_amp=True)
    noise = synth.tensor(5, "plan.noise", tuple(x.shape)).to(DEV)

    def run():
        return [o["sample"].clone() for o in d.ddim_sample_loop_progressive(m.forward_with_cfg, tuple(x.shape), noise=noise,
                                                                            clip_denoised=False, model_kwargs=kw)]
    calls = []
    real = m.plan_timesteps
    monkeypatch.setattr(m, "plan_timesteps", lambda ts: (calls.append(ts.clone()), real(ts))[1])
    planned = run()
    assert len(calls) == 1 and calls[0].tolist() == list(d.timestep_map) and m._t_plan is None
    monkeypatch.setattr(sampler, "PLAN_TIMESTEPS", False)
    unplanned = run()
    assert len(calls) == 1 and len(planned) == len(unplanned) == 10
    for a, b in zip(planned, unplanned):
        assert torch.equal(a, b)
    # a direct call between steps (no selected row) computes its modulation from t, whatever plan is pending
    m.plan_timesteps(calls[0])
    direct = m.forward_with_cfg(x.to(DEV), t[:x.shape[0]].to(DEV), y.to(DEV), 6.0, dtype, True)
    m.clear_timestep_plan()
    assert torch.equal(direct, m.forward_with_cfg(x.to(DEV), t[:x.shape[0]].to(DEV), y.to(DEV), 6.0, dtype, True))


def test_full_width_block_at_baseline_shape(pkg):
    """BASELINE configs[1] shapes on ONE block: d=1152, 16 heads x 72, N_prim=2048, 1370 x 768 condition tokens, CFG 6
    (effective batch 2) - the exact GEMM tiles (128x144 LDS-DMA kernel, K = 1152 / 4608 / 768), the 2048 x 2048 and
    2048 x 1370 attention problems and the batched to_k/to_v projection - against the oracle with emulated rounding."""
    cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=1)
    sd = synth.dit_state_dict(77, **cfg)
    m = pkg.DiT(seq_length=2048, num_heads=16, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(sd)
    m.to(DEV)
    x, y = synth.tensor(77, "x", (1, 2048, 68)), synth.tensor(77, "y", (1, 1370, 768))
    t = torch.tensor([520])
    got = m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, torch.float16, True)
    ref = dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0, torch.float16)
    assert got.shape == (1, 2048, 136)
    assert rel_l2(got, ref) < 3 * TOL_EMU[torch.float16], rel_l2(got, ref)
    ref32 = dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0, None)
    assert rel_l2(got, ref32) < 3 * TOL_FWD[torch.float16], rel_l2(got, ref32)


def test_layernorm_in_the_gemm_tail_changes_no_bit(pkg):
    """`DiT.fuse_ln` (round 4): the LayerNorm + modulate behind every gated residual add runs in the tail of that GEMM's kernel
    (ops.linear_gate_residual(ln=...)).  At the configs[1] width (two blocks, so that the hand-over between blocks and to the
    final layer is exercised) the forward, the CFG forward and the two-stream CFG forward are BIT-IDENTICAL with and without
    it; the fused kernel is really what ran; the small configurations (N != 1152: two-launch route inside the same entry point)
    are identical as well; no in-kernel wait timed out."""
    from importlib import import_module
    ops = import_module(pkg.__name__ + ".ops")
    cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=2)
    m = pkg.DiT(seq_length=2048, num_heads=16, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(synth.dit_state_dict(79, **cfg))
    m.to(DEV)
    x, y = synth.tensor(79, "x", (1, 2048, 68)).to(DEV), synth.tensor(79, "y", (1, 1370, 768)).to(DEV)
    t = torch.tensor([520], device=DEV)
    t0 = ops.ln_sync_timeouts()
    outs = {}
    import os
    switched = any(os.environ.get(v) for v in ("PRIMX_DIT_FUSE_LN", "PRIMX_DIT_LN_TAIL", "PRIMX_LN_FUSE", "PRIMX_GEMM_LOADER", "PRIMX_GEMM_PROF", "PRIMX_WPREFETCH"))
    if not switched:
        assert m.fuse_ln and not m.ln_in_kernel      # defaults: one entry point, the LayerNorm as the library's second launch
    for fuse in (True, False):
        m.fuse_ln = m.ln_in_kernel = fuse                # True: the LayerNorm in the GEMM kernel's tail; False: separate calls
        tags = []
        ops.PROFILE = tags
        try:
            a = m.forward_with_cfg(x, t, y, 6.0, torch.float16, True)
        finally:
            ops.PROFILE = None
        n_fused = sum(1 for tg in tags if tg[0].startswith("gemm144l_dma_kernel<1, 5>"))
        if not switched:
            assert n_fused == (6 if fuse else 0), [tg[0] for tg in tags]   # 3 gated residual adds per block
        b = m(x, t, y, torch.bfloat16, True)
        m.cfg_streams = True
        c = m.forward_with_cfg(x, t, y, 6.0, torch.float16, True)
        m.cfg_streams = False
        outs[fuse] = (a, b, c)
    for u, v in zip(outs[True], outs[False]):
        assert torch.equal(u, v)
    # (the two-stream form launches half-size GEMMs, which take other tiles and sum in another order: rounding-level differences)
    assert rel_l2(outs[True][2], outs[True][0]) < 2e-3
    name, sd, heads, ms, xs, ys, ts = _case(pkg, 0)
    ms.fuse_ln = ms.ln_in_kernel = True
    a = ms.forward_with_cfg(xs.to(DEV), ts[:xs.shape[0]].to(DEV), ys.to(DEV), 6.0, torch.float16, True)
    ms.fuse_ln = False
    assert torch.equal(a, ms.forward_with_cfg(xs.to(DEV), ts[:xs.shape[0]].to(DEV), ys.to(DEV), 6.0, torch.float16, True))
    assert ops.ln_sync_timeouts() == t0


def test_long_token_bf16_stress(pkg):
    """BASELINE configs[4] flavour: bf16, N_prim = 4096 (64 KV tiles per head) on one block; checked against the oracle
    on a strided subset of tokens would need the full attention anyway, so compare the whole output (one block)."""
    cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=1)
    sd = synth.dit_state_dict(78, **cfg)
    m = pkg.DiT(seq_length=4096, num_heads=16, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(sd)
    m.to(DEV)
    x, y = synth.tensor(78, "x", (1, 4096, 68)), synth.tensor(78, "y", (1, 1370, 768))
    t = torch.tensor([40])
    got = m(x.to(DEV), t.to(DEV), y.to(DEV), torch.bfloat16, True)
    assert got.dtype == torch.bfloat16 and torch.isfinite(got.float()).all()
    ref = dit_ref.dit_forward(sd, x, t, y, 16, torch.bfloat16)
    assert rel_l2(got, ref) < TOL_EMU[torch.bfloat16], rel_l2(got, ref)


def test_sharded_sampler_single_rank_matches_plain_loop(pkg):
    """world_size 1 (no process group): ShardedSampler = seeded CPU noise -> device -> the same DDIM loop."""
    from topia_xl_amd.sharding import ShardedSampler
    name, sd, heads, m, x, y, t = _case(pkg, 1)
    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
    B, N, Cc = x.shape
    out = ShardedSampler(m, d, DEV).sample(B, N, Cc, y, seed=7, **kw)
    noise = torch.randn(B, N, Cc, generator=torch.Generator().manual_seed(7))
    want = d.ddim_sample_loop(m.forward_with_cfg, (B, N, Cc), noise=noise.to(DEV), clip_denoised=False,
                              model_kwargs=dict(y=y.to(DEV), **kw))
    assert out.shape == (B, N, Cc) and torch.equal(out.cpu(), want.cpu())


def test_additive_pos_emb_variant(pkg, golden):
    """DiTAdditivePosEmb (the second class of models/dit_crossattn.py): HIP forward vs the REAL reference (fp32 golden)
    and vs the oracle with emulated rounding; no forward_with_cfg, as in the reference."""
    from tests.test_oracle_golden import _addpos_case
    g = golden("dit_addpos")
    m, sd, heads, x, y, t = _addpos_case(pkg.DiTAdditivePosEmb)
    m.load_state_dict(sd, strict=True)
    m.to(DEV)
    for dtype in (torch.float16, torch.bfloat16):
        out = m(x.to(DEV), t.to(DEV), y.to(DEV), precision_dtype=dtype, enable_amp=True)
        assert out.shape == (2, 96, 136) and out.dtype == dtype
        assert rel_l2(out, g["forward"]) < TOL_FWD[dtype], rel_l2(out, g["forward"])
        emu = dit_ref.dit_forward(sd, x, t, y, heads, emulate=dtype)
        assert rel_l2(out, emu) < TOL_EMU[dtype], rel_l2(out, emu)
    with pytest.raises(AttributeError):
        m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), cfg_scale=6.0)


def test_inference_mode_conditioning_tensor(pkg):
    """A conditioning tensor created under torch.inference_mode() tracks no version counter: the conditioning cache must not
    read `_version` of it (round-2 advisor finding: RuntimeError), and must not serve a stale image after an in-place edit."""
    from oracle import synth
    cfg = dict(in_channels=68, condition_channels=64, hidden_size=288, depth=1)
    m = pkg.DiT(seq_length=128, num_heads=4, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(synth.dit_state_dict(7, **cfg))
    m.to(DEV)
    x = synth.tensor(7, "x", (1, 128, 68)).to(DEV)
    t = torch.tensor([500], device=DEV)
    with torch.inference_mode():
        y = synth.tensor(7, "y", (1, 70, 64)).to(DEV) * 1.0
        assert y.is_inference()
        a = m.forward_with_cfg(x, t, y, 6.0, torch.float16, True).clone()
        y.mul_(0.5)                                    # in place, undetectable through a version counter
        b = m.forward_with_cfg(x, t, y, 6.0, torch.float16, True).clone()
    y2 = (synth.tensor(7, "y", (1, 70, 64)).to(DEV) * 1.0) * 0.5
    c = m.forward_with_cfg(x, t, y2, 6.0, torch.float16, True)
    assert not torch.equal(a, b) and torch.equal(b, c)


def test_null_cross_attention_collapse_is_exact_algebra(pkg):
    """`DiT.collapse_null_cross_attention`: the unconditional CFG half attends to ONE row repeated L times, so its cross-attention
    output is that row's value projection whatever the query (dit_crossattn.py:207, attention.py:96-114).  The shortcut must give
    the same guided output as the full computation up to the rounding of a uniform softmax."""
    from oracle import synth
    cfg = dict(in_channels=68, condition_channels=64, hidden_size=288, depth=2)
    m = pkg.DiT(seq_length=128, num_heads=4, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(synth.dit_state_dict(7, **cfg))
    m.to(DEV)
    x = synth.tensor(7, "x", (2, 128, 68)).to(DEV)
    y = synth.tensor(7, "y", (2, 70, 64)).to(DEV)
    t = torch.tensor([500, 500], device=DEV)
    for dt, tol in ((torch.float16, 1e-3), (torch.bfloat16, 8e-3)):
        full = m.forward_with_cfg(x, t, y, 6.0, dt, True).float()
        m.collapse_null_cross_attention = True
        try:
            short = m.forward_with_cfg(x, t, y, 6.0, dt, True).float()
        finally:
            m.collapse_null_cross_attention = False
        assert rel_l2(short, full) < tol, (dt, rel_l2(short, full))
