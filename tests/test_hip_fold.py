"""GPU parity of the LayerNorm fold (include/primx_hip.h, ABI 23; csrc/gemm.hip "LayerNorm fold"): the gate-residual GEMM as the
PRODUCER of a LayerNorm site (16-bit operand centred and scaled with the previous site's (mean, rstd) + partial row sums next to
the residual update), the three CONSUMER GEMMs (to_q / qkv / fc1 forms: statistics from the partials,
y = (rho / rho_p) acc - rho mu' u + v in the epilogue), the fp32-row GEMM of u / v, the dynamic range of the operand in fp16
(row spreads 3e4 and 1e-5, magnitudes 1e4: `test_fold_is_range_safe`), and the DiT with `fold_ln` against the unfolded path
and the fp32 oracle.

What is compared with what: a float64 LayerNorm -> modulate -> Linear of the UPDATED residual rows (the reference's arithmetic,
models/dit_crossattn.py:32-36,55-57) is the truth; the folded chain must meet it with the tolerance of one 16-bit rounding of the
operand and one of the output - the same bar as the unfolded HIP path, whose error is measured next to it."""
import os

import pytest
import torch

from oracle import dit_ref, synth
from tests.util import max_abs, rel_l2, unpack_rows, unpack_vt

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}   # rel-L2 of one 16-bit rounding (tests/test_hip_gemm.py)
EPS = 1e-6


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__
    __graft_entry__.build()
    from topia_xl_amd import ops
    return ops


def _default_dispatch() -> bool:
    return not any(os.environ.get(v) for v in ("PRIMX_GEMM_LOADER", "PRIMX_GEMM_NOBIG", "PRIMX_GEMM_BIG_MIN",
                                               "PRIMX_GEMM_BIGHEADS_MIN", "PRIMX_GEMM_PROF", "PRIMX_LIB"))


def _fold_kernels_selectable(ops) -> bool:
    """False under the kernel-selection switches that take away a tile shape the fold kernels need (PRIMX_GEMM_NOBIG,
    PRIMX_GEMM_LOADER=0, PRIMX_GEMM_BIGHEADS_MIN=0): the DiT then keeps its LayerNorm launches (ops.fold_shapes_ok) and the
    operator tests of those kernels do not apply."""
    return ops.fold_shapes_ok(4096, 2048, 1152, 16)


# the ring the 256 x 288 kernel's heads epilogues run on: 128-byte row segments (two stages of 64-wide k-tiles) unless switched back
_HEADS_RING = ", 32>" if (os.environ.get("PRIMX_GEMM_HEADS_KT32", "0") not in ("", "0") or os.environ.get("PRIMX_GEMM_KT32", "0") not in ("", "0")) else ", 64>"


def _folding_mode() -> bool:
    """Do planned forwards fold under the environment's DiT switches?  (unplanned loops, the two-stream mode, LayerNorm launches that are not fused into the
    gate-residual call and the LayerNorm-carried weight prefetch keep the LayerNorm launches: DiT._forward16)"""
    return (os.environ.get("PRIMX_PLAN_TIMESTEPS", "1") != "0" and not os.environ.get("PRIMX_CFG_STREAMS") and os.environ.get("PRIMX_DIT_FUSE_LN") != "0"
            and os.environ.get("PRIMX_WPREFETCH", "2") != "1" and os.environ.get("PRIMX_DIT_FOLD") != "0")


def _last_kernel(ops):
    from topia_xl_amd import _lib
    return _lib.load().primx_last_gemm_kernel().decode()


@pytest.mark.parametrize("spread,offset", [(3.0, 0.7), (3e4, -1e4), (1e-5, 1e4), (1e-5, 0.0)])
def test_row_stats(ops, spread, offset):
    """(mean, rstd) pairs of fp32 rows - the statistics of primx_layernorm_modulate - against float64."""
    x = (synth.tensor(3, "x", (300, 1152)) * spread + offset).to(DEV)
    out = torch.full((300, 2), float("nan"), device=DEV)
    ops.row_stats(x, EPS, out)
    xd = x.double().cpu()
    mean = xd.mean(-1)
    rstd = 1 / torch.sqrt(((xd - mean[:, None]) ** 2).mean(-1) + EPS)
    assert max_abs(out[:, 0], mean) < 2e-6 * max(1.0, abs(offset), spread)
    assert float(((out[:, 1].cpu().double() - rstd) / rstd).abs().max()) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,frm", [(50, 3456, 1152, 25), (6, 1152, 1152, 3), (130, 4608, 1152, 65), (50, 200, 72, 0)])
def test_linear_f32out(ops, dtype, M, N, K, frm):
    """fp32 rows out of 16-bit operands; the bias joins the rows from `frm` on (the fold's v rows, not its u rows)."""
    A = synth.tensor(4, "A", (M, K)).to(dtype)
    W = synth.tensor(4, "W", (N, K), K ** -0.5).to(dtype)
    b = synth.tensor(4, "b", (N,), 0.3).to(dtype)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_f32out(A.to(DEV), W.to(DEV), b.to(DEV), out, frm)
    ref = A.double() @ W.double().t()
    ref[frm:] += b.double()
    assert rel_l2(out, ref) < 2e-6 and max_abs(out, ref) < 2e-5 * float(ref.abs().max())
    out2 = torch.empty_like(out)
    ops.linear_f32out(A.to(DEV), W.to(DEV), None, out2, 0)
    assert rel_l2(out2, A.double() @ W.double().t()) < 2e-6


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,frm", [(50, 25), (8, 4), (130, 65), (64, 0)])
def test_linear_f32out_group(ops, dtype, M, frm):
    """primx_linear_f32out_group (ABI 26): many few-row fp32-out problems from one launch - every problem against float64 and against its own
    primx_linear_f32out call; ragged last column group (N = 160: a workgroup with idle waves), a problem without a bias, more than 64 rows
    (two or three row blocks); the same problem alone or inside a group gives the same bits."""
    from topia_xl_amd import _lib
    if not _lib.f32out_group_available() or os.environ.get("PRIMX_UV_GROUP") == "0":
        pytest.skip("no grouped f32out kernel in this library / switched off")
    K = 1152
    probs, refs = [], []
    for i, N in enumerate((1152, 3456, 160, 4608, 32)):
        A = synth.tensor(40 + i, "A", (M, K)).to(dtype)
        W = synth.tensor(40 + i, "W", (N, K), K ** -0.5).to(dtype)
        b = None if i == 2 else synth.tensor(40 + i, "b", (N,), 0.3).to(dtype)
        out = torch.full((M, N), float("nan"), device=DEV)
        probs.append((A.to(DEV), W.to(DEV), None if b is None else b.to(DEV), out))
        ref = A.double() @ W.double().t()
        if b is not None:
            ref[frm:] += b.double()
        refs.append(ref)
    assert ops.linear_f32out_group(probs, frm)
    assert _last_kernel(ops) == f"f32out_group_kernel<{1 if dtype == torch.float16 else 2}>"
    for (A, W, b, out), ref in zip(probs, refs):
        assert rel_l2(out, ref) < 2e-6 and max_abs(out, ref) < 2e-5 * float(ref.abs().max())
        one = torch.empty_like(out)
        ops.linear_f32out(A, W, b, one, frm)
        assert rel_l2(one, out.double().cpu()) < 1e-6
    alone = torch.full_like(probs[1][3], float("nan"))
    assert ops.linear_f32out_group([probs[1][:3] + (alone,)], frm)
    assert torch.equal(alone, probs[1][3])
    assert not ops.linear_f32out_group([(probs[0][0], probs[0][1][:40], None, torch.empty(M, 40, device=DEV))], frm)   # N % 32 != 0: the caller's loop


def _site(seed, B, n, D, K, dtype, mean_ratio=0.3, spread=2.0, offset=0.0, branch=1.0):
    """Inputs of one folded LayerNorm site: the branch operand A [B n, K] and weights of the PRODUCER (N = D), the residual
    stream x - row spread `spread`, a per-row mean of `mean_ratio` x the spread (+ `offset` with a per-row sign), gate (scaled
    by `branch`) / shift / scale vectors per batch entry - and the (centre, scale) pairs of the rows one branch ago: the true
    (mean, rstd) of x, the mean perturbed by a tenth of the spread and the rstd by 10 %."""
    M = B * n
    A = synth.tensor(seed, "A", (M, K)).to(dtype)
    W = synth.tensor(seed, "W", (D, K), K ** -0.5).to(dtype)
    b = synth.tensor(seed, "b", (D,), 0.3).to(dtype)
    mod = synth.tensor(seed, "mod", (B, 3 * D), 0.4)
    mod[:, :D] *= branch
    mod = mod.to(dtype)
    x = synth.tensor(seed, "x", (M, D)) * spread
    x = x + mean_ratio * x.std(-1, keepdim=True) * synth.tensor(seed, "mr", (M, 1))
    x = (x + offset * torch.sign(synth.tensor(seed, "sg", (M, 1)))).float()
    xd = x.double()
    c = xd.mean(-1) + 0.1 * xd.std(-1) * synth.tensor(seed, "cn", (M,)).double()
    rp = (1 + 0.1 * synth.tensor(seed, "rn", (M,)).clamp(-2, 2).double()) / torch.sqrt(xd.var(-1, unbiased=False) + EPS)
    return A, W, b, mod, x, torch.stack([c, rp], -1).float().contiguous()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,n,K", [(2, 2048, 1152), (2, 2048, 4608), (2, 487, 192), (1, 130, 64), (8, 2048, 192), (7, 2100, 64)])
def test_fold_producer(ops, dtype, B, n, K):
    """x: the same bits as the plain gate-residual GEMM.  a16: EXACTLY cast16(((x_new - c) rho_p) cast16(1 + scale)) of the stored rows.
    Partial sums: the 144-column sums of (x_new - c) and its square (fp32, fixed order) against float64.  The last two shapes are
    a large batch (>= 14336 rows): the 256 x 288 tile, whose register epilogue forms the partial sums by lane shuffles."""
    D = 1152
    M = B * n
    A, W, b, mod, x, c = _site(21, B, n, D, K, dtype)
    gate, scale = mod[:, :D], mod[:, D:2 * D]
    modd = mod.to(DEV)
    x_plain = x.to(DEV)
    ops.linear_gate_residual(A.to(DEV), W.to(DEV), b.to(DEV), modd[:, :D], x_plain, n)
    x_fold = x.to(DEV)
    a16 = torch.full((M, D), float("nan"), dtype=dtype, device=DEV)
    part = torch.full((M, D // 144, 2), float("nan"), device=DEV)
    ops.linear_gate_residual_fold(A.to(DEV), W.to(DEV), b.to(DEV), modd[:, :D], x_fold, n, modd[:, D:2 * D], c.to(DEV), a16, part)
    if _default_dispatch():
        want_kernel = "gemm288q_dma_kernel<" if M >= 14336 else "gemm144l_dma_kernel<"
        assert _last_kernel(ops).startswith(want_kernel) and _last_kernel(ops).endswith((", 6>", ", 6, 32>", ", 6, 64>")), _last_kernel(ops)
        assert torch.equal(x_fold, x_plain)
    else:
        assert rel_l2(x_fold, x_plain) < 1e-5
    d = x_fold.cpu() - c[:, :1]                                                     # fp32, as the kernel forms it
    m16 = (1 + scale).float().repeat_interleave(n, 0)                               # (1 + scale) formed in the 16-bit type
    assert torch.equal(a16.cpu(), ((d * c[:, 1:]) * m16).to(dtype))
    dd = d.double().view(M, D // 144, 144)
    want = torch.stack([dd.sum(-1), (dd * dd).sum(-1)], -1)
    assert max_abs(part[..., 0], want[..., 0]) < 1e-5 * float(dd.abs().sum(-1).max())
    assert rel_l2(part[..., 1], want[..., 1]) < 1e-6


def _ln_linear_ref(x_new, shift, scale, n, W, b, dtype):
    """float64 LayerNorm -> modulate -> Linear of the reference, with autocast's (1 + scale) in the 16-bit type."""
    xd = x_new.double()
    mu = xd.mean(-1, keepdim=True)
    ln = (xd - mu) / torch.sqrt(((xd - mu) ** 2).mean(-1, keepdim=True) + EPS)
    m = (1 + scale).double().repeat_interleave(n, 0)
    a = ln * m + shift.double().repeat_interleave(n, 0)
    return a @ W.double().t() + (0 if b is None else b.double())


def _chain(ops, dtype, B, n, Wc, bc, seed, consumer, **site):
    """Producer (N = 1152, K = 1152) -> consumer `consumer(a16, part, u, v, center, center_out)` against the float64 reference and
    the unfolded HIP path.  Returns (reference [M, Nc], LayerNorm output of the unfolded path)."""
    D = 1152
    M = B * n
    A, W, b, mod, x, c = _site(seed, B, n, D, D, dtype, **site)
    shift, scale = mod[:, D:2 * D], mod[:, 2 * D:]
    modd = mod.to(DEV)
    xd = x.to(DEV)
    a16 = torch.empty(M, D, dtype=dtype, device=DEV)
    part = torch.empty(M, D // 144, 2, device=DEV)
    cdev = c.to(DEV)
    # (the modulation of a planned loop is shared by the batch: entry 0's scale / shift vectors serve all rows, row stride 0)
    ops.linear_gate_residual_fold(A.to(DEV), W.to(DEV), b.to(DEV), modd[:, :D], xd, n, modd[:1, 2 * D:].expand(B, -1), cdev, a16, part)
    # u, v of the site: rows [cast16(1 + scale); shift] through the fp32-row GEMM (one "timestep")
    rows = torch.stack([(1 + scale[0]), shift[0]]).to(dtype).to(DEV)
    uv = torch.empty(2, Wc.shape[0], device=DEV)
    ops.linear_f32out(rows, Wc.to(DEV), None if bc is None else bc.to(DEV), uv, 1)
    assert bool(torch.isfinite(a16.float()).all())
    cnext = torch.full((M, 2), float("nan"), device=DEV)
    consumer(a16, part, uv[0], uv[1], cdev, cnext)
    assert torch.equal(cdev.cpu(), c)                                           # a consumer never touches the pairs it reads
    x_new = xd.cpu()
    ref = _ln_linear_ref(x_new, shift[:1].expand(B, -1), scale[:1].expand(B, -1), n, Wc, bc, dtype)
    # the next site's pairs: (mean, rstd) of the updated stream
    xn64 = x_new.double()
    mu = xn64.mean(-1)
    rstd = 1 / torch.sqrt(xn64.var(-1, unbiased=False) + EPS)
    assert max_abs(cnext[:, 0], mu) < 2e-5 * float(x_new.abs().max())
    assert float(((cnext[:, 1].cpu().double() - rstd) / rstd).abs().max()) < 2e-3
    xn = torch.empty(M, D, dtype=dtype, device=DEV)
    m0 = modd[:1].expand(B, -1)
    ops.layernorm_modulate(xd, m0[:, D:2 * D], m0[:, 2 * D:], n, xn, EPS)
    return ref, xn


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,n", [(2, 2048), (2, 300), (5, 2048)])
def test_fold_consumer_to_q(ops, dtype, B, n):
    """The to_q form: one ROWS segment, scale0, the loader-wave 128 x 144 kernel - the 256 x 288 tile from 10240 rows on."""
    from topia_xl_amd._lib import HEADS_ROWS
    if not _fold_kernels_selectable(ops):
        pytest.skip("a kernel-selection switch removes the loader-wave kernel")
    D, H, dh = 1152, 16, 72
    Wc = synth.tensor(31, "Wq", (D, D), D ** -0.5).to(dtype)
    bq = synth.tensor(31, "bq", (D,), 0.3).to(dtype)
    s0 = dh ** -0.5
    Q = ops.alloc_heads(B, H, n, dh, HEADS_ROWS, dtype, DEV, 128)
    names = []

    def consumer(a16, part, u, v, c, cn):
        ops.linear_heads_fold(a16, Wc.to(DEV), n, H, dh, [HEADS_ROWS], [Q], Q.shape[2], part, u, v, c, cn, EPS, scale0=s0)
        names.append(_last_kernel(ops))
    ref, xn = _chain(ops, dtype, B, n, Wc, bq, 31, consumer)
    want = (s0 * ref.to(dtype).float()).to(dtype).view(B, n, H, dh)
    err = rel_l2(unpack_rows(Q, n, dh), want)
    Q2 = ops.alloc_heads(B, H, n, dh, HEADS_ROWS, dtype, DEV, 128)
    ops.linear_heads(xn, Wc.to(DEV), bq.to(DEV), n, H, dh, [HEADS_ROWS], [Q2], Q2.shape[2], scale0=s0)
    err_unfolded = rel_l2(unpack_rows(Q2, n, dh), want)
    print(f"to_q {dtype} B={B} n={n}: folded {err:.2e}, unfolded {err_unfolded:.2e}")
    assert err < 2 * TOL[dtype] and err < 1.5 * err_unfolded + 1e-4
    if _default_dispatch():
        assert names[0].startswith("gemm288q_dma_kernel<" if B * n >= 10240 else "gemm144l_dma_kernel<") and names[0].endswith((", 7>", ", 7, 32>", ", 7, 64>")), names


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_fold_consumer_qkv(ops, dtype):
    """The qkv form at the configs[1] shape (T = 4096): ROWS / KROWS / V^T segments on the 256 x 288 tile."""
    from topia_xl_amd._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT
    if not _fold_kernels_selectable(ops):
        pytest.skip("a kernel-selection switch removes the 256 x 288 heads tile")
    B, n, D, H, dh = 2, 2048, 1152, 16, 72
    Wc = synth.tensor(32, "Wqkv", (3 * D, D), D ** -0.5).to(dtype)
    bc = synth.tensor(32, "bqkv", (3 * D,), 0.3).to(dtype)
    bufs = [ops.alloc_heads(B, H, n, dh, k, dtype, DEV, 128) for k in (HEADS_ROWS, HEADS_KROWS, HEADS_VT)]
    names = []

    def consumer(a16, part, u, v, c, cn):
        ops.linear_heads_fold(a16, Wc.to(DEV), n, H, dh, [HEADS_ROWS, HEADS_KROWS, HEADS_VT], bufs, bufs[0].shape[2], part, u, v, c, cn, EPS)
        names.append(_last_kernel(ops))
    ref, xn = _chain(ops, dtype, B, n, Wc, bc, 32, consumer)
    want = ref.to(dtype).view(B, n, 3, H, dh)
    got = [unpack_rows(bufs[0], n, dh), unpack_rows(bufs[1], n, dh), unpack_vt(bufs[2], n, dh)]
    ref_bufs = [ops.alloc_heads(B, H, n, dh, k, dtype, DEV, 128) for k in (HEADS_ROWS, HEADS_KROWS, HEADS_VT)]
    ops.linear_heads(xn, Wc.to(DEV), bc.to(DEV), n, H, dh, [HEADS_ROWS, HEADS_KROWS, HEADS_VT], ref_bufs, ref_bufs[0].shape[2])
    unf = [unpack_rows(ref_bufs[0], n, dh), unpack_rows(ref_bufs[1], n, dh), unpack_vt(ref_bufs[2], n, dh)]
    for s in range(3):
        err, err_u = rel_l2(got[s], want[:, :, s]), rel_l2(unf[s], want[:, :, s])
        print(f"qkv {dtype} segment {s}: folded {err:.2e}, unfolded {err_u:.2e}")
        assert err < 2 * TOL[dtype] and err < 1.5 * err_u + 1e-4
    # the operand-level mask / denominator markers of the layouts survive (include/primx_hip.h)
    assert float(bufs[2][:, :, dh].float().sum()) == B * H * n
    if _default_dispatch():
        assert names[0].startswith("gemm288q_dma_kernel<") and names[0].endswith(_HEADS_RING)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,n,paired", [(2, 2048, True), (4, 2048, False)])
def test_fold_consumer_qkv_with_a_rider(ops, dtype, B, n, paired):
    """primx_linear_heads_fold_pair (ABI 25): the qkv consumer + the to_k / to_v projection of 1536 conditioning rows (K = 768, KROWS / V^T)
    from one launch - every output the same BITS as the two calls (same tile kernel, same tiles); the problem-1-alone form (A = NULL) too.
    T = 4096: 192 + 48 workgroups in one round (the pair kernel); T = 8192 (384 tiles: more than a round): two launches."""
    from topia_xl_amd._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT
    if not _fold_kernels_selectable(ops):
        pytest.skip("a kernel-selection switch removes the 256 x 288 heads tile")
    D, H, dh, Lk, L, Dc = 1152, 16, 72, 1536, 1370, 768
    dt = 1 if dtype == torch.float16 else 2
    Wc = synth.tensor(34, "Wqkv", (3 * D, D), D ** -0.5).to(dtype)
    bc = synth.tensor(34, "bqkv", (3 * D,), 0.3).to(dtype)
    y16 = torch.zeros(Lk, Dc, dtype=dtype)
    y16[:L] = synth.tensor(34, "y", (L, Dc)).to(dtype)
    Wkv = synth.tensor(34, "Wkv", (2 * D, Dc), Dc ** -0.5).to(dtype).to(DEV)
    bkv = synth.tensor(34, "bkv", (2 * D,), 0.3).to(dtype).to(DEV)
    y16 = y16.to(DEV)
    kinds = [HEADS_ROWS, HEADS_KROWS, HEADS_VT]
    mk = lambda: [ops.alloc_heads(B, H, n, dh, k, dtype, DEV, 128) for k in kinds]
    mkkv = lambda: [ops.alloc_heads(1, H, L, dh, k, dtype, DEV, 256) for k in (HEADS_KROWS, HEADS_VT)]
    got, ref, kv_got, kv_ref, kv_alone = mk(), mk(), mkkv(), mkkv(), mkkv()
    cn_ref = []
    names = []

    def consumer(a16, part, u, v, c, cn):
        ops.linear_heads_fold(a16, Wc.to(DEV), n, H, dh, kinds, ref, ref[0].shape[2], part, u, v, c, cn, EPS)
        cn_ref.append(cn.clone())
        cn.fill_(float("nan"))
        ops.linear_heads_fold_pair(dict(A=a16, W=Wc.to(DEV), rows_per_batch=n, heads=H, dh=dh, kinds=kinds, dsts=got, n_pad=got[0].shape[2],
                                        part=part, u=u, v=v, center=c, center_out=cn, eps=EPS),
                                   y16, Wkv, bkv, Lk, H, dh, [HEADS_KROWS, HEADS_VT], kv_got, kv_got[0].shape[2])
        names.append(_last_kernel(ops))
        assert torch.equal(cn, cn_ref[0])
    _chain(ops, dtype, B, n, Wc, bc, 34, consumer)
    ops.linear_heads_fold_pair(None, y16, Wkv, bkv, Lk, H, dh, [HEADS_KROWS, HEADS_VT], kv_alone, kv_alone[0].shape[2])
    names.append(_last_kernel(ops))
    # the batched projection of DiT._forward16 (n_rep = 2 repetitions of the same weights: 96 tiles; the rule of launch<> gives it the
    # same tile kernel only from PRIMX_GEMM_BIGHEADS_MIN = 160 workgroups on, so the reference here is the explicit big-tile form above)
    ops.linear_heads(y16, Wkv, bkv, Lk, H, dh, [HEADS_KROWS, HEADS_VT], kv_ref, kv_ref[0].shape[2])
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    for a, b in zip(kv_got, kv_alone):
        assert torch.equal(a, b)
    want = (y16.double().cpu() @ Wkv.double().cpu().t() + bkv.double().cpu()).to(dtype).view(1, Lk, 2, H, dh)[:, :L]
    for s, unpack in ((0, unpack_rows), (1, unpack_vt)):
        assert rel_l2(unpack(kv_got[s], L, dh), want[:, :, s]) < TOL[dtype]
        assert rel_l2(unpack(kv_ref[s], L, dh), want[:, :, s]) < TOL[dtype]
    if _default_dispatch():
        assert names[0] == (f"gemm288q_pair_kernel<{dt}{_HEADS_RING}" if paired else f"gemm288q_dma_kernel<{dt}, 2{_HEADS_RING}"), names
        assert names[1] == f"gemm288q_dma_kernel<{dt}, 2{_HEADS_RING}", names


def test_fold_pair_rejects_bad_arguments(ops):
    """Argument validation of primx_linear_heads_fold_pair happens before anything is launched."""
    import ctypes as C
    from topia_xl_amd import _lib
    lib = _lib.load()
    k2, d2 = (C.c_int * 2)(2, 1), (C.c_void_p * 2)(16, 16)
    nul = (None, None, 0, 0, 0, 0, 0, 0, 0, None, None, 0, 0.0, None, None, None, None, None, 0.0)
    assert lib.primx_linear_heads_fold_pair(*nul, None, 16, None, 1536, 2304, 768, 1536, 16, 72, 2, k2, d2, 1536, 1.0, 1, None) == -1
    assert b"problem 1" in lib.primx_last_error()
    assert lib.primx_linear_heads_fold_pair(*nul, 16, 16, None, 1536, 2300, 768, 1536, 16, 72, 2, k2, d2, 1536, 1.0, 1, None) == -1
    assert b"n_seg*heads*dh" in lib.primx_last_error()
    bad0 = (16, 16, 4096, 3456, 1152, 2048, 16, 72, 3, None, None, 2048, 1.0, None, None, None, None, None, 1e-6)
    assert lib.primx_linear_heads_fold_pair(*bad0, 16, 16, None, 1536, 2304, 768, 1536, 16, 72, 2, k2, d2, 1536, 1.0, 1, None) == -1
    assert b"problem 0" in lib.primx_last_error()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,n,kernel", [(2, 2048, "gemm288q_dma_kernel"), (2, 1950, "gemm288q_dma_kernel"), (1, 1024, "gemm144l_dma_kernel"),
                                        (1, 333, "gemm144l_dma_kernel"), (4, 2048, "gemm288q_dma_kernel"), (3, 1500, "gemm288q_dma_kernel")])
def test_fold_consumer_fc1(ops, dtype, B, n, kernel):
    """The fc1 form: GELU(tanh) behind the fold, on the 256 x 288 kernel (T = 4096: ONE round of 256 workgroups - the two-pass kernel's
    launches until round 6 - and more) and on the loader-wave 128 x 144 kernel (smaller launches); ragged last tiles."""
    from topia_xl_amd._lib import ACT_GELU_TANH
    D, Hm = 1152, 4608
    Wc = synth.tensor(33, "Wfc1", (Hm, D), D ** -0.5).to(dtype)
    bc = synth.tensor(33, "bfc1", (Hm,), 0.3).to(dtype)
    out = torch.empty(B * n, Hm, dtype=dtype, device=DEV)
    names = []

    def consumer(a16, part, u, v, c, cn):
        ops.linear_fold(a16, Wc.to(DEV), out, part, u, v, c, cn, EPS, act=ACT_GELU_TANH)
        names.append(_last_kernel(ops))
    ref, xn = _chain(ops, dtype, B, n, Wc, bc, 33, consumer)
    want = torch.nn.functional.gelu(ref.to(dtype).double(), approximate="tanh")
    unf = ops.linear(xn, Wc.to(DEV), bc.to(DEV), act=ACT_GELU_TANH)
    err, err_u = rel_l2(out, want), rel_l2(unf, want)
    print(f"fc1 {dtype} B={B} n={n}: folded {err:.2e}, unfolded {err_u:.2e}  ({names[0]})")
    assert err < 2 * TOL[dtype] and err < 1.5 * err_u + 1e-4
    if _default_dispatch():
        assert names[0].startswith(kernel), names


def test_fold_large_row_mean_is_what_the_centre_is_for(ops):
    """Rows whose mean is 20 x their spread: with the centre at the previous site's mean (within a tenth of the spread of the
    true one) the folded result keeps the unfolded path's accuracy - without a centre it would lose a factor ~10
    (tools/ln_fold_study.py)."""
    from topia_xl_amd._lib import ACT_NONE
    dtype, B, n, D = torch.float16, 2, 512, 1152
    Wc = synth.tensor(34, "W", (D, D), D ** -0.5).to(dtype)
    out = torch.empty(B * n, D, dtype=dtype, device=DEV)

    def consumer(a16, part, u, v, c, cn):
        ops.linear_fold(a16, Wc.to(DEV), out, part, u, v, c, cn, EPS, act=ACT_NONE)
    ref, xn = _chain(ops, dtype, B, n, Wc, None, 34, consumer, mean_ratio=20.0)
    unf = ops.linear(xn, Wc.to(DEV), None)
    err, err_u = rel_l2(out, ref), rel_l2(unf, ref)
    print(f"row mean = 20 sigma: folded {err:.2e}, unfolded {err_u:.2e}")
    assert err < 1.5 * err_u + 1e-4


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("spread,offset,branch", [(3e4, 0.0, 1.0), (3e4, 1e4, 1.0), (2.0, 1e4, 1.0), (1e-5, 0.0, 1e-5), (1e-5, 0.0, 1.0),
                                                  (1e-5, 1e4, 1e-5)])
def test_fold_is_range_safe(ops, dtype, spread, offset, branch):
    """The dynamic range of the folded operand (the reason for ABI 23): the reference normalises BEFORE it rounds to 16 bits
    (models/dit_crossattn.py:32-36,55-57; models/utils.py:19-20), so neither the magnitude nor the spread of the residual stream
    can overflow or flush a Linear input.  The fold's operand is scaled by the previous site's rstd, so the same holds: row
    spreads of 3e4 (cast16((x - c) m) of ABI 22 overflows fp16 there) and 1e-5 (fp16 subnormals there), residual magnitudes of
    1e4, a branch that multiplies the spread by 4e4 in one step ((1e-5, 0, 1): rho_p / rho = 400) - the folded result keeps the
    unfolded path's accuracy against the float64 LayerNorm -> modulate -> Linear, and the operand is finite.  Through the
    128 x 144 producer / consumer (B n = 1024) and, for one case, the configs[1] kernels."""
    from topia_xl_amd._lib import ACT_GELU_TANH
    D, Hm = 1152, 4608
    Wc = synth.tensor(35, "Wfc1", (Hm, D), D ** -0.5).to(dtype)
    bc = synth.tensor(35, "bfc1", (Hm,), 0.3).to(dtype)
    for B, n in ([(1, 1024), (2, 2048)] if (spread, offset) == (3e4, 1e4) else [(1, 1024)]):
        out = torch.full((B * n, Hm), float("nan"), dtype=dtype, device=DEV)

        def consumer(a16, part, u, v, c, cn):
            ops.linear_fold(a16, Wc.to(DEV), out, part, u, v, c, cn, EPS, act=ACT_GELU_TANH)
        ref, xn = _chain(ops, dtype, B, n, Wc, bc, 35, consumer, spread=spread, offset=offset, branch=branch)
        want = torch.nn.functional.gelu(ref.to(dtype).double(), approximate="tanh")
        unf = ops.linear(xn, Wc.to(DEV), bc.to(DEV), act=ACT_GELU_TANH)
        err, err_u = rel_l2(out, want), rel_l2(unf, want)
        print(f"range {dtype} spread {spread:g} offset {offset:g} branch {branch:g} rows {B * n}: folded {err:.2e}, unfolded {err_u:.2e}")
        assert bool(torch.isfinite(out.float()).all())
        assert err < 2 * TOL[dtype] and err < 1.5 * err_u + 1e-4


def test_fold_shape_errors(ops):
    """Shapes outside the fold kernels are errors, not silent fall-backs."""
    from topia_xl_amd._lib import PrimxError
    dtype = torch.float16
    A = torch.zeros(256, 1152, dtype=dtype, device=DEV)
    W = torch.zeros(1280, 1152, dtype=dtype, device=DEV)          # N % 144 != 0
    out = torch.zeros(256, 1280, dtype=dtype, device=DEV)
    part = torch.zeros(256, 8, 2, device=DEV)
    c = torch.ones(256, 2, device=DEV)
    u = torch.zeros(1280, device=DEV)
    with pytest.raises(PrimxError):
        ops.linear_fold(A, W, out, part, u, u, c, torch.ones_like(c), EPS)
    W2 = torch.zeros(1152, 1152, dtype=dtype, device=DEV)
    u2 = torch.zeros(1152, device=DEV)
    with pytest.raises(RuntimeError):                              # a consumer must not write the pairs it reads
        ops.linear_fold(A, W2, torch.zeros(256, 1152, dtype=dtype, device=DEV), part, u2, u2, c, c, EPS)
    assert not ops.fold_supported(384, 6) and ops.fold_supported(1152, 16)
    assert ops.fold_shapes_ok(4096, 2048, 1152, 16) == _default_dispatch() or not _default_dispatch()
    assert not ops.fold_shapes_ok(2048, 2048, 1152, 16)


def _fold_model(pkg, depth, seed):
    cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=depth)
    sd = synth.dit_state_dict(seed, **cfg)
    m = pkg.DiT(seq_length=2048, num_heads=16, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(sd)
    m.to(DEV)
    return sd, m


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_dit_with_the_fold_against_the_unfolded_path_and_the_oracle(ops, dtype, monkeypatch):
    """Three full-width blocks at the configs[1] shape (N_prim = 2048, CFG -> 4096 token rows), a planned 4-step DDIM loop: every
    step's sample with `fold_ln` within 2e-3 (fp16) of the unfolded loop; one planned forward against the fp32 oracle with the
    error of the unfolded path next to it; the launch list really changes (no LayerNorm launches between the first and the last,
    the fold kernels in their place); unplanned calls are untouched."""
    import topia_xl_amd as pkg
    sd, m = _fold_model(pkg, 3, 81)
    x, y = synth.tensor(81, "x", (1, 2048, 68)), synth.tensor(81, "y", (1, 1370, 768))
    d = pkg.create_diffusion("ddim4", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=dtype, enable_amp=True)

    def loop():
        return [o["sample"].clone() for o in d.ddim_sample_loop_progressive(m.forward_with_cfg, tuple(x.shape), noise=x.to(DEV),
                                                                            clip_denoised=False, model_kwargs=kw)]
    m.fold_ln = False
    base = loop()
    m.fold_ln = True
    tags, ln_calls = [], []
    real_ln = ops.layernorm_modulate
    monkeypatch.setattr(ops, "layernorm_modulate", lambda *a, **k: (ln_calls.append(1), real_ln(*a, **k))[1])
    ops.PROFILE = tags       # (with per-launch timing on, every LayerNorm launch is issued from Python: countable)
    try:
        folded = loop()
    finally:
        ops.PROFILE = None
        monkeypatch.setattr(ops, "layernorm_modulate", real_ln)
    tol = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}[dtype]
    for i, (a, b) in enumerate(zip(folded, base)):
        e = rel_l2(a, b)
        print(f"{dtype} step {i}: folded vs unfolded sample {e:.2e}")
        assert e < tol
    names = [tg[0] for tg in tags]
    launch_list_applies = (_default_dispatch() and os.environ.get("PRIMX_DIT_FOLD") != "0" and os.environ.get("PRIMX_DIT_FUSE_LN") != "0"
                           and not os.environ.get("PRIMX_CFG_STREAMS") and os.environ.get("PRIMX_WPREFETCH", "2") != "1"
                           and os.environ.get("PRIMX_PLAN_TIMESTEPS", "1") != "0"       # unplanned loops never fold
                           and os.environ.get("PRIMX_DIT_LN_TAIL") != "1")               # the final layer's LayerNorm runs inside the last GEMM
    if launch_list_applies:
        assert len(ln_calls) == 2 * 4, len(ln_calls)                       # the first LayerNorm and the final layer's, per forward
        def epi(nm):                                                               # the epilogue template argument of a GEMM tag
            args = nm.split("<")[1].split(">")[0].split(", ") if "<" in nm else []
            return args[1] if len(args) > 1 else ""
        # qkv of blocks 0, 1 carries the next block's to_k / to_v - where the pair fits one round (not with expanded null K / V: 192 + 96 tiles)
        riders = 2 if os.environ.get("PRIMX_DIT_KV_RIDE", "1") != "0" and os.environ.get("PRIMX_NULL_KV_DEDUP", "1") != "0" else 0
        assert sum(1 for nm in names if epi(nm) == "6") == 4 * (3 * 3 - 1), names   # producers: every gated add but the last
        assert sum(1 for nm in names if epi(nm) == "7") == 4 * (2 * 3 - 1 - riders), names      # to_q (blocks 1, 2) + qkv
        assert sum(1 for nm in names if nm.startswith("gemm288q_pair_kernel<")) == 4 * riders, names
        fc1_ring = ", 8, 32>" if (os.environ.get("PRIMX_GEMM_KT32", "0") not in ("", "0") or int(os.environ.get("PRIMX_GEMM_KT64_MIN", "1")) > 256) else ", 8, 64>"
        assert sum(1 for nm in names if nm.startswith("gemm288q_dma_kernel") and fc1_ring in nm) == 4 * 3
    # one planned forward against the fp32 oracle
    t = torch.tensor([520])
    ref32 = dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0, None)
    errs = {}
    for fold in (False, True):
        m.fold_ln = fold
        m.plan_timesteps(t.to(DEV))
        m.select_planned_timestep(0)
        errs[fold] = rel_l2(m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, dtype, True), ref32)
        m.clear_timestep_plan()
    print(f"{dtype} forward_with_cfg vs fp32 oracle: unfolded {errs[False]:.3e}, folded {errs[True]:.3e}")
    assert errs[True] < 1.25 * errs[False] + 1e-4
    # an unplanned call never folds: bit-identical with the flag on and off
    m.fold_ln = True
    a = m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, dtype, True)
    m.fold_ln = False
    assert torch.equal(a, m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, dtype, True))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_dit_fold_with_massive_activation_channels(ops, dtype):
    """Trained transformers carry a few "massive activation" channels in the residual stream (round-4 advisor note); the synthetic
    N(0, sigma) models of this suite do not.  Here four channels of the token embedding sit at 4e4 - the fp32 residual stream carries
    them through every block, rows have a spread of ~2e3 - where the ABI-22 operand cast16((x - c)(1 + scale)) leaves the fp16 range.
    Two full-width blocks, planned 3-step DDIM loop: the folded loop is finite, raises no overflow warning (the sampler would
    repeat the loop), stays within the 16-bit rounding level of the unfolded loop, and one planned forward meets the fp32 oracle
    as well as the unfolded path does."""
    import warnings

    import topia_xl_amd as pkg
    if os.environ.get("PRIMX_CFG_STREAMS"):
        pytest.skip("the two-stream route does not fill `block_probe`")
    cfg = dict(in_channels=68, condition_channels=768, hidden_size=1152, depth=2)
    sd = synth.dit_state_dict(83, **cfg)
    sd["x_embedder.bias"] = sd["x_embedder.bias"].clone()
    sd["x_embedder.bias"][[5, 300, 777, 1100]] += torch.tensor([4e4, -4e4, 3e4, 4e4])
    m = pkg.DiT(seq_length=2048, num_heads=16, attn_proj_bias=True, cond_drop_prob=0.1, **cfg).eval()
    m.load_state_dict(sd)
    m.to(DEV)
    x, y = synth.tensor(83, "x", (1, 2048, 68)), synth.tensor(83, "y", (1, 1370, 768))
    d = pkg.create_diffusion("ddim3", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=dtype, enable_amp=True)

    def loop():
        with warnings.catch_warnings():
            warnings.simplefilter("error")                         # the sampler's overflow guard would warn and repeat
            return [o["sample"].clone() for o in d.ddim_sample_loop_progressive(m.forward_with_cfg, tuple(x.shape), noise=x.to(DEV),
                                                                                model_kwargs=kw)]   # (default clip_denoised=True)
    m.fold_ln = False
    base = loop()
    m.fold_ln = True
    m.block_probe = []
    folded = loop()
    probe, m.block_probe = m.block_probe, None
    assert any(b["folded"] for b in probe) == (_fold_kernels_selectable(ops) and _folding_mode())
    assert max(b["residual_abs_max"] for b in probe) > 3e4 and all(b["next_operand_finite"] for b in probe)
    assert max(b["next_operand_abs_max"] for b in probe) < 200, max(b["next_operand_abs_max"] for b in probe)   # normalised, not 4e4
    tol = {torch.float16: 3e-3, torch.bfloat16: 2.5e-2}[dtype]
    for i, (a, b) in enumerate(zip(folded, base)):
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all())
        e = rel_l2(a, b)
        print(f"massive channels {dtype} step {i}: folded vs unfolded sample {e:.2e}")
        assert e < tol
    t = torch.tensor([520])
    ref32 = dit_ref.dit_forward_with_cfg(sd, x, t, y, 16, 6.0, None)
    errs = {}
    for fold in (False, True):
        m.fold_ln = fold
        m.plan_timesteps(t.to(DEV))
        m.select_planned_timestep(0)
        errs[fold] = rel_l2(m.forward_with_cfg(x.to(DEV), t.to(DEV), y.to(DEV), 6.0, dtype, True), ref32)
        m.clear_timestep_plan()
    print(f"massive channels {dtype} forward_with_cfg vs fp32 oracle: unfolded {errs[False]:.3e}, folded {errs[True]:.3e}")
    assert errs[True] < 1.25 * errs[False] + 1e-4


@pytest.mark.parametrize("dtype,batch,dedup", [(torch.float16, 1, True), (torch.bfloat16, 1, True), (torch.float16, 2, False)])
def test_blocks_call_is_bit_identical_to_the_python_block_loop(ops, dtype, batch, dedup):
    """primx_dit_blocks_fold (ABI 24): a folded forward's blocks issued by the library from ONE foreign call - the same entry points with
    the same arguments as DiT._forward16's Python loop (`blocks_call = False`), so every sample of a planned DDIM loop is the same to the
    last bit, with the broadcast null K / V entry and with expanded ones, at batch 1 and 2; the host really makes one call per forward."""
    import topia_xl_amd as pkg
    from topia_xl_amd import _lib
    if not _lib.blocks_call_available():
        pytest.skip("the loaded library has no primx_dit_blocks_fold")
    # (depth 4: the Python loop's batched K / V projection - 6 x 8 x depth tiles at batch 1 - takes the 256 x 288 tile from 160 tiles on, the tile
    #  the one-call route's riders always use; below that it runs on the 128 x 144 kernel and the operands differ at rounding level)
    sd, m = _fold_model(pkg, 4, 83)
    m.dedup_null_kv = dedup
    x, y = synth.tensor(83, "x", (batch, 2048, 68)), synth.tensor(83, "y", (batch, 1370, 768))
    d = pkg.create_diffusion("ddim4", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=dtype, enable_amp=True)

    def loop():
        return [o["sample"].clone() for o in d.ddim_sample_loop_progressive(m.forward_with_cfg, tuple(x.shape), noise=x.to(DEV),
                                                                            clip_denoised=False, model_kwargs=kw)]
    lib = _lib.load()
    calls = []
    real = lib.primx_dit_blocks_fold

    class Spy:          # (ctypes function objects cannot be monkeypatched in place: wrap the attribute)
        def __call__(self, *a):
            calls.append(1)
            return real(*a)
    m.blocks_call = False
    base = loop()
    m.blocks_call = True
    lib.primx_dit_blocks_fold = Spy()
    try:
        got = loop()
    finally:
        lib.primx_dit_blocks_fold = real
    folds = m.fold_ln and m._fold_ok(2 * batch * 2048, 2048) and os.environ.get("PRIMX_PLAN_TIMESTEPS", "1") != "0" \
        and not os.environ.get("PRIMX_CFG_STREAMS") and os.environ.get("PRIMX_DIT_LN_TAIL") != "1" and os.environ.get("PRIMX_DIT_FUSE_LN") != "0" \
        and os.environ.get("PRIMX_WPREFETCH", "2") != "1"          # (`blocks_call` is set on the model above, whatever PRIMX_DIT_BLOCKS_CALL says)
    assert len(calls) == (4 if folds else 0), len(calls)
    for a, b in zip(got, base):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dtype,batch,dedup", [(torch.float16, 1, True), (torch.bfloat16, 1, False)])
def test_kv_ride_is_bit_identical_to_the_batched_projection(ops, dtype, batch, dedup):
    """`DiT.kv_ride` (ABI 25): on the one-call route the library projects the conditioning K / V itself - block 0's as a launch of its own,
    block i + 1's riding on block i's qkv launch - instead of one batched GEMM per forward from Python: every sample of a planned loop is
    the same to the last bit, with `reuse_cond_kv` (nothing to project after the first forward) as well."""
    import topia_xl_amd as pkg
    from topia_xl_amd import _lib
    if not (_lib.blocks_call_available() and _lib.kv_ride_available()):
        pytest.skip("the loaded library has no K / V riders")
    sd, m = _fold_model(pkg, 4, 87)        # (depth 4: the batched projection on the riders' tile kernel - see the test above)
    m.dedup_null_kv = dedup
    x, y = synth.tensor(87, "x", (batch, 2048, 68)), synth.tensor(87, "y", (batch, 1370, 768))
    d = pkg.create_diffusion("ddim3", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=dtype, enable_amp=True)
    loop = lambda: [o["sample"].clone() for o in d.ddim_sample_loop_progressive(m.forward_with_cfg, tuple(x.shape), noise=x.to(DEV),
                                                                                clip_denoised=False, model_kwargs=kw)]
    m.kv_ride = False
    base = loop()
    m.kv_ride = True
    got = loop()
    m.reuse_cond_kv = True
    m._cond = None
    again = loop()
    m.reuse_cond_kv = False
    for a, b, c in zip(got, base, again):
        assert torch.equal(a, b) and torch.equal(c, b)


def test_blocks_call_rejects_bad_descriptors(ops):
    """Argument validation of primx_dit_blocks_fold happens before anything is launched (no compute needed to see the error codes)."""
    import ctypes as C
    from topia_xl_amd import _lib
    if not _lib.blocks_call_available():
        pytest.skip("the loaded library has no primx_dit_blocks_fold")
    lib = _lib.load()
    blk = (_lib.DitBlockFold * 1)()
    f = _lib.DitForwardFold(dtype=_lib.F16, Be=2, N=2048, D=1152, H=16, dh=72, hidden=4608, depth=1, L=1370, nq_pad=2048, nkv_pad_c=1536,
                            nkv_pad_b=128, b_from=1, step=0, n_steps=1, ln_eps=1e-6, scale=72 ** -0.5)
    assert lib.primx_dit_blocks_fold(C.byref(f), blk, None) == -1 and b"null workspace" in lib.primx_last_error()
    f.dh = 64
    assert lib.primx_dit_blocks_fold(C.byref(f), blk, None) == -1 and b"bad shape" in lib.primx_last_error()
    f.dh, f.step = 72, 3
    assert lib.primx_dit_blocks_fold(C.byref(f), blk, None) == -1 and b"outside the u / v tables" in lib.primx_last_error()
    f.step, f.dtype = 0, 0
    assert lib.primx_dit_blocks_fold(C.byref(f), blk, None) == -1 and b"dtype" in lib.primx_last_error()
    assert lib.primx_dit_blocks_fold(None, blk, None) == -1


def test_fold_guard_rerun_on_a_real_dit(ops):
    """The repeat-unfolded path of `diffusion/sampler.py::_fold_guard` with a real DiT: `fold_overflowed` is forced to answer True once,
    the loop is run again with LayerNorm launches and THAT sample is returned (bit-identical to a `fold_ln = False` loop), `fold_ln` is
    restored, the plan is cleared, a warning is raised; a progressive consumer sees the first loop's intermediates and the second loop's
    final item (documented in the guard's docstring)."""
    import topia_xl_amd as pkg
    if os.environ.get("PRIMX_PLAN_TIMESTEPS", "1") == "0":
        pytest.skip("unplanned loops never fold: the sampler has nothing to guard")
    sd, m = _fold_model(pkg, 2, 85)
    x, y = synth.tensor(85, "x", (1, 2048, 68)), synth.tensor(85, "y", (1, 1370, 768))
    d = pkg.create_diffusion("ddim3", noise_schedule="squaredcos_cap_v2", parameterization="v")
    kw = dict(y=y.to(DEV), cfg_scale=6.0, precision_dtype=torch.float16, enable_amp=True)
    run = lambda: list(d.ddim_sample_loop_progressive(m.forward_with_cfg, tuple(x.shape), noise=x.to(DEV), clip_denoised=False, model_kwargs=kw))
    m.fold_ln = False
    unfolded = run()
    m.fold_ln = True
    folded = run()
    answers = [True]
    real = m.fold_overflowed
    m.fold_overflowed = lambda sample: answers.pop() if answers else real(sample)
    try:
        with pytest.warns(RuntimeWarning, match="LayerNorm fold"):
            got = run()
    finally:
        del m.fold_overflowed
    assert m.fold_ln is True and m._t_plan is None
    assert torch.equal(got[-1]["sample"], unfolded[-1]["sample"])            # the final item is the second loop's ...
    if m._fold_ok(4096, 2048) and os.environ.get("PRIMX_PLAN_TIMESTEPS", "1") != "0":
        assert torch.equal(got[0]["sample"], folded[0]["sample"])            # ... the earlier ones were yielded by the first


def test_fold_is_capped_by_the_loop_length(ops, monkeypatch):
    """Loops longer than `DiT.fold_max_steps` are planned without the fold (its u / v tables are built for the whole loop at once)."""
    import topia_xl_amd as pkg
    sd, m = _fold_model(pkg, 1, 86)
    x, y = synth.tensor(86, "x", (1, 2048, 68)), synth.tensor(86, "y", (1, 1370, 768))
    t = torch.full((3,), 500, dtype=torch.int64, device=DEV)
    built = []
    real = m._fold_tables
    monkeypatch.setattr(m, "_fold_tables", lambda *a, **k: (built.append(1), real(*a, **k))[1])
    for cap, want in ((2, 0), (3, 1)):
        m.fold_max_steps = cap
        m.plan_timesteps(t)
        m.select_planned_timestep(1)
        m.forward_with_cfg(x.to(DEV), t[:1], y.to(DEV), 6.0, torch.float16, True)
        m.clear_timestep_plan()
        if m._fold_ok(4096, 2048) and _folding_mode():
            assert len(built) == want, (cap, built)
