"""GPU parity of the attention kernel against a float64 softmax(QK^T s)V, including the reference's
operator seam (memory_efficient_attention on strided BMHK views), ragged / tiny key counts, a forced
online-softmax rescale, and size-independent properties at the full N_prim = 2048 shape."""
import pytest
import torch

from oracle import dit_ref, synth
from tests.util import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: 2e-3, torch.bfloat16: 1.5e-2}


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__
    __graft_entry__.build()
    from topia_xl_amd import ops
    return ops


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Mq,Mk,H,dh", [(2, 256, 256, 4, 72), (1, 128, 1370, 2, 72), (2, 100, 1, 3, 72),
                                         (1, 300, 70, 2, 64), (5, 64, 64, 8, 32), (1, 2048, 2048, 2, 72)])
def test_memory_efficient_attention_matches_float64(ops, dtype, B, Mq, Mk, H, dh):
    q = synth.tensor(21, "q", (B, Mq, H, dh)).to(dtype)
    k = synth.tensor(21, "k", (B, Mk, H, dh)).to(dtype)
    v = synth.tensor(21, "v", (B, Mk, H, dh)).to(dtype)
    ref = dit_ref.attention_core(q, k, v, dh ** -0.5)
    got = ops.memory_efficient_attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert got.shape == (B, Mq, H, dh)
    assert rel_l2(got, ref) < TOL[dtype], rel_l2(got, ref)


def test_strided_qkv_views_like_the_reference(ops):
    """attention.py:50-54: q, k, v are non-contiguous views of one fused [B, N, 3, H, dh] buffer."""
    B, N, H, dh = 2, 192, 4, 72
    qkv = synth.tensor(22, "qkv", (B, N, 3, H, dh)).to(torch.float16)
    q, k, v = qkv.unbind(2)
    ref = dit_ref.attention_core(q, k, v, dh ** -0.5)
    qd = qkv.to(DEV)
    got = ops.memory_efficient_attention(*qd.unbind(2))
    assert rel_l2(got, ref) < TOL[torch.float16]


def test_forced_rescale_spike(ops):
    """One key whose score dwarfs the others arrives in a LATE tile: the running max jumps and
    every earlier contribution must be rescaled exactly once (guide T13 hazard)."""
    B, N, H, dh = 1, 512, 1, 72
    q = synth.tensor(23, "q", (B, N, H, dh)).to(torch.float16)
    k = synth.tensor(23, "k", (B, N, H, dh)).to(torch.float16)
    v = synth.tensor(23, "v", (B, N, H, dh)).to(torch.float16)
    k[0, 300, 0] = q[0, 17, 0] * 6.0          # q17 . k300 is huge; key 300 sits in tile 4
    k[0, 500, 0] = q[0, 200, 0] * 9.0
    ref = dit_ref.attention_core(q, k, v, dh ** -0.5)
    got = ops.memory_efficient_attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert max_abs(got[0, 17], ref[0, 17]) < 5e-3 and max_abs(got[0, 200], ref[0, 200]) < 5e-3
    assert rel_l2(got, ref) < TOL[torch.float16]


def test_deferred_rescale_ramp(ops):
    """The running max is only raised when a tile's max exceeds it by more than 2^8 (deferred rescale): a key ramp whose
    scores grow by ~3 (exp2 domain) per 64-key tile keeps the kernel for two or three tiles on a stale max - probabilities
    up to 2^8 - before each rescale; every query row sees the same ramp."""
    B, N, H, dh = 1, 1024, 2, 72
    q = synth.tensor(25, "q", (B, N, H, dh), 0.3).to(torch.float16)
    k = synth.tensor(25, "k", (B, N, H, dh), 0.3).to(torch.float16)
    v = synth.tensor(25, "v", (B, N, H, dh)).to(torch.float16)
    q[..., 0] = 4.0                                               # q . k picks up 4 * k[..., 0]
    ramp = (torch.arange(N) // 64).float() * 3.0 / (4.0 * dh ** -0.5 * 1.4427)   # +3 in the exp2 domain per tile
    k[0, :, :, 0] = ramp[:, None].to(torch.float16)
    ref = dit_ref.attention_core(q, k, v, dh ** -0.5)
    got = ops.memory_efficient_attention(q.to(DEV), k.to(DEV), v.to(DEV))
    assert rel_l2(got, ref) < TOL[torch.float16], rel_l2(got, ref)


def test_properties_at_full_size(ops):
    """N_prim = 2048, 16 heads x 72 (BASELINE config 2 shape), checked through properties that need
    no O(N^2) reference: (1) V = const -> output = const (softmax rows sum to 1); (2) linearity in V;
    (3) invariance under a permutation of the keys; (4) a spot-check of 64 rows against float64."""
    B, N, H, dh = 2, 2048, 16, 72
    q = synth.tensor(24, "q", (B, N, H, dh)).to(torch.float16).to(DEV)
    k = synth.tensor(24, "k", (B, N, H, dh)).to(torch.float16).to(DEV)
    v1 = synth.tensor(24, "v1", (B, N, H, dh)).to(torch.float16).to(DEV)
    v2 = synth.tensor(24, "v2", (B, N, H, dh)).to(torch.float16).to(DEV)
    ones = torch.full_like(v1, 0.5)
    o = ops.memory_efficient_attention(q, k, ones)
    assert max_abs(o, ones) < 1e-3
    o1 = ops.memory_efficient_attention(q, k, v1).float()
    o2 = ops.memory_efficient_attention(q, k, v2).float()
    o12 = ops.memory_efficient_attention(q, k, (v1.float() * 0.5 + v2.float() * 0.25).half()).float()
    assert rel_l2(o12, 0.5 * o1 + 0.25 * o2) < 4e-3
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0)).to(DEV)
    op = ops.memory_efficient_attention(q, k[:, perm].contiguous(), v1[:, perm].contiguous()).float()
    assert rel_l2(op, o1) < 2e-3
    rows = torch.arange(0, N, 32)
    ref = dit_ref.attention_core(q[:, rows].cpu(), k.cpu(), v1.cpu(), dh ** -0.5)
    assert rel_l2(o1[:, rows], ref) < TOL[torch.float16]


def test_attention_modules_against_golden(ops, golden):
    """The module mirrors reproduce the REAL reference modules' outputs (tests/golden/attention.npz)."""
    import topia_xl_amd.attention as A
    from tests.golden.make_golden import SEED
    g = golden("attention")
    m = A.MemEffAttention(dim=256, num_heads=8, qkv_bias=False, proj_bias=True).eval()
    m.load_state_dict(synth.state_dict_like(SEED, m.state_dict()))
    m.to(DEV)
    x = synth.tensor(SEED, "att.x", (3, 64, 256))
    assert rel_l2(m(x.to(DEV)), g["self_dh32"]) < 3e-3
    c = A.MemEffCrossAttention(dim=144, dim_q=144, dim_k=40, dim_v=40, num_heads=2, qkv_bias=True, proj_bias=True)
    c.load_state_dict(synth.state_dict_like(SEED, c.state_dict()))
    c.eval().to(DEV)
    q = synth.tensor(SEED, "catt.q", (2, 96, 144))
    kv = synth.tensor(SEED, "catt.kv", (2, 37, 40)).to(DEV)       # ragged: 37 keys, K = 40 (not a multiple of 64)
    assert rel_l2(c(q.to(DEV), kv, kv), g["cross_dh72"]) < 3e-3
    with pytest.raises(NotImplementedError):
        c(q.to(DEV), kv, kv.clone())                                # k and v must be the same conditioning tensor


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("nq,nkv", [(64, 64), (64, 50), (37, 64), (1, 1)])
def test_small_problem_kernel(ops, dtype, nq, nkv):
    """attn64_kernel (one wave per problem, compact operands: n_pad = 64, dh = 32 - the VAE mid-block attention over the 64 voxels
    of a primitive): against float64, ragged query / key counts included, many problems per launch (4 per workgroup + a tail)."""
    from topia_xl_amd._lib import HEADS_KROWS, HEADS_ROWS, HEADS_VT
    B, H, dh = 7, 3, 32
    q = synth.tensor(61, "q", (B, nq, H, dh)).to(dtype)
    k = synth.tensor(61, "k", (B, nkv, H, dh)).to(dtype)
    v = synth.tensor(61, "v", (B, nkv, H, dh)).to(dtype)
    Q = ops.pack_heads(q.to(DEV), HEADS_ROWS, 64, "q")
    K = ops.pack_heads(k.to(DEV), HEADS_KROWS, 64, "k")
    Vt = ops.pack_heads(v.to(DEV), HEADS_VT, 64)
    assert Q.shape[2] == 64 and K.shape[2] == 64 and Vt.shape[3] == 64
    got = ops.attention(Q, K, Vt, nq, nkv, dh, dh ** -0.5).view(B, nq, H, dh)
    qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * dh ** -0.5, -1) @ vd).permute(0, 2, 1, 3)
    assert rel_l2(got, ref) < TOL[dtype], rel_l2(got, ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,b_from,Mq,Mk,H,dh", [(2, 1, 300, 1370, 4, 72), (3, 0, 256, 1370, 2, 72), (2, 1, 256, 256, 2, 72),
                                                 (2, 1, 200, 64, 2, 72), (2, 1, 128, 200, 2, 64), (3, 2, 64, 129, 4, 32)])
def test_broadcast_key_value_entries(ops, dtype, B, b_from, Mq, Mk, H, dh):
    """primx_attention_bcast: the batch entries >= b_from attend to Mk copies of one key / value row, stored once (one full tile
    + the ragged last tile).  BIT-IDENTICAL to primx_attention on the expanded operands (the kernel visits the same tile
    contents in the same order) and right against float64; ragged (1370, 200, 129) and whole-tile (256, 64) key counts,
    every entry a broadcast one (b_from = 0), the three head dims."""
    from topia_xl_amd import _lib
    q = synth.tensor(25, "q", (B, Mq, H, dh)).to(dtype)
    k = synth.tensor(25, "k", (B, Mk, H, dh)).to(dtype)
    v = synth.tensor(25, "v", (B, Mk, H, dh)).to(dtype)
    krow, vrow = synth.tensor(25, "krow", (1, 1, H, dh)).to(dtype), synth.tensor(25, "vrow", (1, 1, H, dh)).to(dtype)
    k[b_from:] = krow                                                          # the expanded form: Mk identical rows
    v[b_from:] = vrow
    ref = dit_ref.attention_core(q, k, v, dh ** -0.5)
    Qp = ops.pack_heads(q.to(DEV), _lib.HEADS_ROWS, ops.BQ, "q")
    full = ops.attention(Qp, ops.pack_heads(k.to(DEV), _lib.HEADS_KROWS, ops.BKV, "k"), ops.pack_heads(v.to(DEV), _lib.HEADS_VT, ops.BKV),
                         Mq, Mk, dh, dh ** -0.5)
    nb = ops.bcast_keys(Mk)
    assert nb == 64 + (Mk % 64 if Mk > 64 else 0)
    Kb = ops.pack_heads(krow.expand(1, nb, H, dh).contiguous().to(DEV), _lib.HEADS_KROWS, ops.BKV, "k")
    Vb = ops.pack_heads(vrow.expand(1, nb, H, dh).contiguous().to(DEV), _lib.HEADS_VT, ops.BKV)
    Kp = ops.pack_heads(k[:b_from].to(DEV), _lib.HEADS_KROWS, ops.BKV, "k") if b_from else None
    Vt = ops.pack_heads(v[:b_from].to(DEV), _lib.HEADS_VT, ops.BKV) if b_from else None
    got = ops.attention(Qp, Kp, Vt, Mq, Mk, dh, dh ** -0.5, bcast=(Kb, Vb))
    assert torch.equal(got, full)
    assert rel_l2(got.view(B, Mq, H, dh), ref) < TOL[dtype]
    # identical keys: every broadcast entry's output rows are the value row itself (softmax weights 1 / Mk)
    assert max_abs(got.view(B, Mq, H, dh)[b_from:], vrow.double().expand(B - b_from, Mq, H, dh)) < (2e-3 if dtype == torch.float16 else 1.6e-2)
    with pytest.raises(RuntimeError):
        ops.attention(Qp, Kp, Vt, Mq, Mk, dh, dh ** -0.5, bcast=(Kb[:, :, :32], Vb))          # operand layout mismatch
