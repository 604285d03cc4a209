"""GPU parity of the row / elementwise kernels against the oracle's formulas (through the C ABI)."""
import numpy as np
import pytest
import torch

from oracle import diffusion_ref as dref
from oracle import dit_ref, synth
from tests.util import max_abs, rel_l2, unpack_rows, unpack_vt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__
    __graft_entry__.build()
    from topia_xl_amd import ops
    return ops


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("D,rows,rpb", [(1152, 4096, 2048), (384, 512, 256), (256, 70, 35)])
def test_layernorm_modulate(ops, dtype, D, rows, rpb):
    B = rows // rpb
    x = synth.tensor(1, "ln.x", (rows, D), 1.5, 0.3)
    mod = synth.tensor(1, "ln.mod", (B, 3 * D), 0.3).to(dtype)   # shift | junk | scale chunks of a wider row
    shift, scale = mod[:, :D], mod[:, 2 * D:]
    ref = dit_ref._modulate(dit_ref.layer_norm(x).view(B, rpb, D), shift.float(), scale.float(), dtype).view(rows, D)
    ref = ref.to(dtype).float()
    md = mod.to(DEV)
    out = torch.empty(rows, D, dtype=dtype, device=DEV)
    ops.layernorm_modulate(x.to(DEV), md[:, :D], md[:, 2 * D:], rpb, out, 1e-6)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2   # <= 1-2 ulp of the 16-bit type on O(1..8) values
    assert max_abs(out, ref) <= tol * max(1.0, float(ref.abs().max())), max_abs(out, ref)
    assert rel_l2(out, ref) < (3e-4 if dtype == torch.float16 else 2.5e-3)


def test_timestep_embedding_and_mlp(ops):
    t = torch.tensor([0, 1, 40, 500, 960, 999])
    got = ops.timestep_embedding(t.to(DEV), 256)
    ref = dit_ref.timestep_embedding(t)
    assert max_abs(got, ref) < 1e-6        # same host frequency table, fp32 product; device sin/cos within ~2 ulp
    W = synth.tensor(2, "w", (1152, 256), 0.05)
    b = synth.tensor(2, "b", (1152,), 0.05)
    import torch.nn.functional as F
    got = ops.linear_f32(got, W.to(DEV), b.to(DEV), act_out=1)
    assert rel_l2(got, F.silu(F.linear(ref, W, b))) < 1e-5


@pytest.mark.parametrize("M,N,K", [(4096, 1152, 68), (2, 384, 256), (130, 70, 12),
                                   (8, 1152, 1152), (1, 70, 12), (5, 333, 260), (9, 70, 260)])   # M <= 8: wave-per-column path
def test_linear_f32(ops, M, N, K):
    import torch.nn.functional as F
    x, W, b = synth.tensor(3, "x", (M, K)), synth.tensor(3, "W", (N, K), K ** -0.5), synth.tensor(3, "b", (N,))
    got = ops.linear_f32(x.to(DEV), W.to(DEV), b.to(DEV))
    assert rel_l2(got, F.linear(x.double(), W.double(), b.double())) < 1e-6
    if M > 8:   # the second destination of the tiled kernel (forward_with_cfg embeds the same tokens into both halves of the stream)
        two = torch.full((2 * M + 1, N), 7.0, device=DEV)
        ops.linear_f32(x.to(DEV), W.to(DEV), b.to(DEV), out=two[:M], out2=two[M:2 * M])
        assert torch.equal(two[:M], got) and torch.equal(two[M:2 * M], got) and float(two[2 * M].min()) == 7.0 == float(two[2 * M].max())
    else:
        with pytest.raises(Exception):
            ops.linear_f32(x.to(DEV), W.to(DEV), b.to(DEV), out2=torch.empty(M, N, device=DEV))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_casts_and_cfg(ops, dtype):
    import torch.nn.functional as F
    x = synth.tensor(4, "x", (3, 1000), 2.0)
    assert torch.equal(ops.cast16(x.to(DEV), dtype).cpu(), x.to(dtype))
    ref = F.silu(x).to(dtype)
    got = ops.silu_cast(x.to(DEV), dtype).cpu()
    assert max_abs(got, ref) <= (2e-3 if dtype == torch.float16 else 1.6e-2)
    mo = synth.tensor(4, "mo", (4, 64, 136)).to(dtype)
    cond, unc = mo[:2], mo[2:]
    ref = unc + 6.0 * (cond - unc)                   # torch CPU 16-bit arithmetic rounds after every op
    got = ops.cfg_combine(mo.to(DEV), 6.0).cpu()
    assert torch.equal(got, ref)
    mo32 = synth.tensor(4, "mo32", (4, 64, 136))
    ref32 = mo32[2:] + 6.0 * (mo32[:2] - mo32[2:])
    assert torch.equal(ops.cfg_combine(mo32.to(DEV), 6.0).cpu(), ref32)


@pytest.mark.parametrize("n,eta", [(5, 0.0), (25, 0.0), (25, 0.5)])
@pytest.mark.parametrize("out_dtype", [torch.float16, torch.float32])
def test_diffusion_step_bit_exact(ops, n, eta, out_dtype):
    """Given the same model output the fused update is BIT-IDENTICAL to the reference formulas
    (gaussian_diffusion.py:340-356,531-578) - every spaced step, v / eps / x0 parameterisations."""
    import topia_xl_amd as pkg
    for par in ("v", "eps", "xstart"):
        d = pkg.create_diffusion(f"ddim{n}", noise_schedule="squaredcos_cap_v2", parameterization=par)
        tab, _ = dref.make("squaredcos_cap_v2", 1000, f"ddim{n}")
        coef = torch.from_numpy(d.step_coefficients(eta)).to(DEV)
        x = synth.tensor(5, "x", (2, 96, 68))
        mo = synth.tensor(5, "mo", (2, 96, 136)).to(out_dtype)
        noise = synth.tensor(5, "noise", (2, 96, 68))
        for i in range(n):
            ref = dref.ddim_step(tab, i, x, mo, par, eta, False, noise)
            s, x0 = ops.diffusion_step(x.to(DEV), mo.to(DEV), coef, i, mean_type={"eps": 0, "xstart": 1, "v": 2}[par],
                                       var_type=3, ancestral=False, clip_denoised=False,
                                       noise=noise.to(DEV) if eta else None)
            assert torch.equal(x0.cpu(), ref["pred_xstart"].float()), (par, i)
            assert torch.equal(s.cpu(), ref["sample"]), (par, i, max_abs(s, ref["sample"]))
    # clip_denoised clamps pred_xstart
    s, x0 = ops.diffusion_step(x.to(DEV) * 3, mo.to(DEV), coef, n - 1, mean_type=1, var_type=3, ancestral=False,
                               clip_denoised=True, noise=None)
    assert float(x0.abs().max()) <= 1.0


@pytest.mark.parametrize("ancestral", [False, True])
def test_clip_denoised_propagates_nan_like_torch_clamp(ops, ancestral):
    """`x.clamp(-1, 1)` of the reference (gaussian_diffusion.py:287-291) PROPAGATES NaN and maps +-inf to +-1: a non-finite
    model output must not come out of a clipped step as a finite -1 (fminf / fmaxf would do that) - the sampling loop's
    overflow guard (diffusion/sampler.py `_fold_guard`) reads the final sample, clipped or not."""
    import topia_xl_amd as pkg
    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    tab, _ = dref.make("squaredcos_cap_v2", 1000, "ddim5")
    coef = torch.from_numpy(d.step_coefficients(0.0)).to(DEV)
    x = synth.tensor(9, "x", (1, 32, 68))
    mo = synth.tensor(9, "mo", (1, 32, 136)).to(torch.float16)
    mo[0, 3, 5] = float("nan")
    mo[0, 4, 6] = float("inf")
    mo[0, 5, 7] = float("-inf")
    noise = synth.tensor(9, "n", (1, 32, 68))
    for i in (4, 0):
        if ancestral:
            ref = dref.ancestral_step(tab, i, x, mo, noise, "v", True)
        else:
            ref = dref.ddim_step(tab, i, x, mo, "v", 0.0, True, None)
        s, x0 = ops.diffusion_step(x.to(DEV), mo.to(DEV), coef, i, mean_type=2, var_type=3, ancestral=ancestral, clip_denoised=True,
                                   noise=noise.to(DEV) if ancestral else None)
        x0 = x0.cpu()
        assert torch.isnan(x0[0, 3, 5]) and torch.isnan(s.cpu()[0, 3, 5])
        assert float(x0[0, 4, 6]) == -1.0 and float(x0[0, 5, 7]) == 1.0          # v-prediction: x0 = a x - b v
        fin = torch.ones_like(x0, dtype=torch.bool)
        fin[0, 3, 5] = False
        assert bool(torch.isfinite(x0[fin]).all()) and float(x0[fin].abs().max()) <= 1.0
        assert torch.equal(torch.isnan(x0), torch.isnan(ref["pred_xstart"]))
        assert torch.equal(x0[fin], ref["pred_xstart"].float()[fin])


def test_sampling_loop_repeats_an_overflowed_folded_fp16_loop(ops):
    """diffusion/sampler.py `_fold_guard`: a planner that reports folded fp16 forwards and a non-finite FINAL sample gets its loop
    repeated with `fold_ln = False`, before the final item is yielded (consumers stop at the last item) and with the default
    `clip_denoised=True` (the clip propagates NaN).  The stand-in model overflows exactly when it "folds"."""
    import warnings

    import topia_xl_amd as pkg

    class Model:
        def __init__(self):
            self.fold_ln, self.calls, self._used, self.plans = True, [], False, 0

        def plan_timesteps(self, ts):
            self.plans += 1

        def select_planned_timestep(self, row):
            pass

        def clear_timestep_plan(self):
            self._used = False

        def fold_overflowed(self, sample):
            used, self._used = self._used, False
            return used and not bool(torch.isfinite(sample).all())

        def __call__(self, x, t, **kw):
            self.calls.append(bool(self.fold_ln))
            out = torch.cat([0.1 * x, torch.zeros_like(x)], -1).half()
            if self.fold_ln:
                self._used = True
                if int(t[0]) < 300:
                    out[0, 0, 0] = float("nan")
            return out

    d = pkg.create_diffusion("ddim5", noise_schedule="squaredcos_cap_v2", parameterization="v")
    x = synth.tensor(10, "x", (1, 16, 68)).to(DEV)
    m = Model()
    with pytest.warns(RuntimeWarning, match="LayerNorm fold"):
        items = list(d.ddim_sample_loop_progressive(m, tuple(x.shape), noise=x))       # default clip_denoised=True
    assert len(items) == 5 and m.calls == [True] * 5 + [False] * 5 and m.fold_ln is True
    assert bool(torch.isfinite(items[-1]["sample"]).all())
    m2 = Model()
    m2.fold_ln = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        want = d.ddim_sample_loop(m2, tuple(x.shape), noise=x)
    assert torch.equal(items[-1]["sample"], want) and m2.calls == [False] * 5


def test_ancestral_step(ops):
    import topia_xl_amd as pkg
    d = pkg.create_diffusion("ddim25", noise_schedule="squaredcos_cap_v2", parameterization="v")
    tab, _ = dref.make("squaredcos_cap_v2", 1000, "ddim25")
    coef = torch.from_numpy(d.step_coefficients(0.0)).to(DEV)
    x, noise = synth.tensor(6, "x", (2, 64, 68)), synth.tensor(6, "n", (2, 64, 68))
    for dt in (torch.float16, torch.float32):
        mo = (synth.tensor(6, "mo", (2, 64, 136)) * 0.7).to(dt)
        for i in (24, 7, 0):
            ref = dref.ancestral_step(tab, i, x, mo, noise)
            # ModelVarType.LEARNED takes the same log-variance interpolation as LEARNED_RANGE (gaussian_diffusion.py:285-293)
            for var_type in (3, 2):
                s, x0 = ops.diffusion_step(x.to(DEV), mo.to(DEV), coef, i, mean_type=2, var_type=var_type, ancestral=True,
                                           clip_denoised=False, noise=noise.to(DEV))
                assert torch.equal(x0.cpu(), ref["pred_xstart"])
                assert max_abs(s, ref["sample"]) < 2e-6 * max(1.0, float(ref["sample"].abs().max()))  # expf ulp


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_pack_heads_layouts(ops, dtype):
    from topia_xl_amd._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT
    B, M, H, dh = 2, 70, 3, 72
    qkv = synth.tensor(7, "qkv", (B, M, 3, H, dh)).to(dtype).to(DEV)
    for which, kind in ((0, HEADS_ROWS), (1, HEADS_KROWS), (2, HEADS_VT)):
        src = qkv[:, :, which]                       # strided BMHK view like the reference's unbind()
        buf = ops.pack_heads(src, kind, 64)
        back = unpack_vt(buf, M, dh) if kind == HEADS_VT else unpack_rows(buf, M, dh)
        assert torch.equal(back, src)
        data = buf[:, :, :dh, :] if kind == HEADS_VT else buf[:, :, :, :dh]
        assert buf.shape[-1] == {HEADS_ROWS: 80, HEADS_KROWS: 88, HEADS_VT: 128}[kind]
        assert float(data.float().abs().sum()) == pytest.approx(float(src.float().abs().sum()), rel=1e-3)  # pads stay 0
        if kind == HEADS_VT:   # spare rows: row dh = ones at the valid keys (softmax denominator row), the rest zero
            assert float(buf[:, :, dh].float().sum()) == B * H * M and float(buf[:, :, dh + 1:].float().abs().sum()) == 0.0


def test_prefetch_entry_points_change_nothing(ops):
    """primx_prefetch and the pf0 / pf1 ranges of primx_layernorm_modulate only pull bytes into the caches: the LayerNorm launch
    that carries ranges gives the same bits as one without, a third range is refused by the wrapper, and nothing is kept
    between calls (explicit arguments since ABI 21)."""
    x = synth.tensor(5, "x", (300, 1152)).to(DEV)
    sh = synth.tensor(5, "sh", (3, 1152), 0.3).half().to(DEV)
    sc = synth.tensor(5, "sc", (3, 1152), 0.3).half().to(DEV)
    w1 = torch.randn(1152, 1152, device=DEV).half()
    w2 = torch.randn(777, 64, device=DEV).half()            # a byte count that is not a multiple of a line
    new = lambda: torch.empty(300, 1152, dtype=torch.float16, device=DEV)
    plain = ops.layernorm_modulate(x, sh, sc, 100, new())
    assert torch.equal(plain, ops.layernorm_modulate(x, sh, sc, 100, new(), prefetch=(w1, w2)))
    assert torch.equal(plain, ops.layernorm_modulate(x, sh, sc, 100, new(), prefetch=(w2,)))
    with pytest.raises(RuntimeError):
        ops.layernorm_modulate(x, sh, sc, 100, new(), prefetch=(w1, w2, w1))
    del w1
    torch.cuda.empty_cache()
    assert torch.equal(plain, ops.layernorm_modulate(x, sh, sc, 100, new()))
    ops.prefetch(w2, torch.cuda.current_stream())
    torch.cuda.synchronize()
