"""Tensor-level wrappers over the C ABI: pointer extraction, shape checks, current-stream plumbing.

PyTorch is used for device memory and streams only; every function here ends in exactly one (or a
short fixed sequence of) ``primx_*`` calls enqueued on ``torch.cuda.current_stream()`` - the
reference's extensions hard-code stream 0 (mvpraymarch.cpp:121), this does not.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_GELU_TANH, ACT_NONE, BF16, F16, F32, HEADS_KROWS, HEADS_ROWS, HEADS_VT, check

_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}
BQ, BKV = 128, 64  # attention query-tile / key-tile sizes (csrc/attention.hip)


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise TypeError(f"unsupported dtype {dt}") from None


_launch_dev: Optional[int] = None  # device index of the operands of the launch being assembled (set by _dev)


def _dev(t: torch.Tensor, name: str, dtype: Optional[torch.dtype] = None) -> int:
    global _launch_dev
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    _launch_dev = t.device.index
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.data_ptr()


def _stream() -> int:
    """Launch stream = torch's current stream of the CURRENT device.  The C ABI carries no device index, so the operands
    must live on the current device: a tensor on another GPU would be launched on the wrong device's stream (unordered
    against the torch ops that produced it).  The module entry points (DiT.forward, VAE.decode, the sampler, ...) switch
    to their input's device with ``device_of``; raw op calls on a non-current device fail loudly here."""
    cur = torch.cuda.current_device()
    if _launch_dev is not None and _launch_dev != cur:
        raise RuntimeError(f"operands live on cuda:{_launch_dev} but the current device is cuda:{cur}: call "
                           "torch.cuda.set_device(...) or wrap the call in `with torch.cuda.device(t.device)`")
    return torch.cuda.current_stream().cuda_stream


def device_of(t: torch.Tensor):
    """Context manager making ``t``'s HIP device current (no-op when it already is)."""
    if not t.is_cuda:
        raise RuntimeError(f"expected a HIP device tensor, got {t.device}; there is no CPU path")
    return torch.cuda.device(t.device)


def on_input_device(fn):
    """Decorator for module entry points: run ``fn`` with the device of its first tensor argument current."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor):
                if a.is_cuda and a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapped


# Optional per-launch timing of the MFMA kernels (bench.py's roofline leg): when PROFILE is a list, every
# GEMM / attention launch is bracketed by HIP events recorded on the launch stream (torch's current
# stream) and (kernel family, algorithmic FLOPs, start, end) is appended.  No host synchronisation.
PROFILE = None


def _timed(tag: str, flops: float, fn):
    if PROFILE is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    if tag.startswith("None "):      # a GEMM entry point ("None MxNxK"): ask the library which instantiation it launched -
        # the tag is the name the C side reports (primx_last_gemm_kernel), not a Python restatement of its dispatch rules
        lib = _lib.load()
        name = lib.primx_last_gemm_kernel().decode() if hasattr(lib, "primx_last_gemm_kernel") else "gemm(unreported)"
        tag = name + tag[4:]
    PROFILE.append((tag, flops, s, e))
    return r


def padded_head_dim(dh: int) -> int:
    return (dh + 15) // 16 * 16


def round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


# ----------------------------------------------------------------------------- row kernels
def _range(t: Optional[torch.Tensor]) -> Tuple[Optional[int], int]:
    """(pointer, bytes) of a cache-prefetch range argument; (None, 0) when there is none."""
    if t is None or not t.is_cuda or not t.is_contiguous() or t.numel() == 0:
        return None, 0
    return t.data_ptr(), t.numel() * t.element_size()


def _check_modulation(shift: torch.Tensor, scale: torch.Tensor, out: torch.Tensor) -> None:
    if shift.stride(-1) != 1 or scale.stride(-1) != 1 or shift.stride(0) != scale.stride(0):
        raise RuntimeError("shift/scale must be last-dim contiguous with equal row strides")
    if not (shift.is_cuda and scale.is_cuda and shift.dtype == out.dtype == scale.dtype):
        raise TypeError("shift/scale/out dtype or device mismatch")


def layernorm_modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, rows_per_batch: int,
                       out: torch.Tensor, eps: float = 1e-6, prefetch: Sequence[torch.Tensor] = ()) -> torch.Tensor:
    """x: [rows, D] fp32; shift/scale: [B, D] 16-bit views (last dim contiguous, row stride arbitrary).
    `prefetch`: up to two tensors whose bytes this launch also pulls into the caches (primx_layernorm_modulate pf0 / pf1)."""
    rows, D = x.shape
    _check_modulation(shift, scale, out)
    if len(prefetch) > 2:
        raise RuntimeError("layernorm_modulate carries at most two prefetch ranges")
    (p0, n0), (p1, n1) = (_range(prefetch[0]) if len(prefetch) > 0 else (None, 0)), (_range(prefetch[1]) if len(prefetch) > 1 else (None, 0))
    check(_lib.load().primx_layernorm_modulate(
        _dev(x, "x", torch.float32), shift.data_ptr(), scale.data_ptr(), shift.stride(0), _dev(out, "out"),
        dtype_code(out.dtype), rows, rows_per_batch, D, eps, p0, n0, p1, n1, _stream()), "primx_layernorm_modulate")
    return out


_FREQS = {}


def _freq_table(dim: int, max_period: float, device) -> torch.Tensor:
    """exp(-ln(P) * arange(dim/2, fp32) / (dim/2)) evaluated with torch CPU ops, exactly as the reference
    does before ``.to(device)`` (models/utils.py:51-54); cached per (dim, period, device)."""
    key = (dim, float(max_period), str(device))
    tab = _FREQS.get(key)
    if tab is None:
        import math
        half = dim // 2
        tab = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half).to(device)
        _FREQS[key] = tab
    return tab


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    t = t.to(torch.int64).contiguous()
    out = torch.empty(t.shape[0], dim, dtype=torch.float32, device=t.device)
    check(_lib.load().primx_timestep_embedding(_dev(t, "t"), _freq_table(dim, max_period, t.device).data_ptr(),
                                               out.data_ptr(), t.shape[0], dim, _stream()),
          "primx_timestep_embedding")
    return out


def point_features(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """x: [T, C>=4] fp32 token rows (point = channels 1..3), freqs: [F] fp32 -> PointEmbed features [T, 6F+3], returned
    zero-padded to a multiple of 4 columns (the K granularity of primx_linear_f32)."""
    T, F = x.shape[0], freqs.shape[0]
    if x.stride(1) != 1:
        raise RuntimeError("point_features: channels must be contiguous")
    out = torch.zeros(T, round_up(6 * F + 3, 4), dtype=torch.float32, device=x.device)
    check(_lib.load().primx_point_features(_dev(x, "x", torch.float32), x.stride(0), _dev(freqs, "freqs", torch.float32),
                                           out.data_ptr(), out.stride(0), T, F, _stream()), "primx_point_features")
    return out


def silu_cast(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    elif out.dtype != dtype or out.numel() != x.numel() or not out.is_contiguous():
        raise RuntimeError("silu_cast: out must be a contiguous tensor of the target dtype with x.numel() elements")
    check(_lib.load().primx_silu_cast(_dev(x, "x", torch.float32), out.data_ptr(), dtype_code(dtype), x.numel(),
                                      _stream()), "primx_silu_cast")
    return out


def cast16(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    elif out.dtype != dtype or out.numel() != x.numel() or not out.is_contiguous():
        raise RuntimeError("cast16: out must be a contiguous tensor of the target dtype with x.numel() elements")
    check(_lib.load().primx_cast16(_dev(x, "x", torch.float32), out.data_ptr(), dtype_code(dtype), x.numel(),
                                   _stream()), "primx_cast16")
    return out


def linear_f32(x: torch.Tensor, W: torch.Tensor, b: Optional[torch.Tensor], act_out: int = 0,
               out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`out2`: a second [M, N] fp32 destination that receives the same rows (M > 8)."""
    M, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    for o in (out, out2):
        if o is not None and (tuple(o.shape) != (M, N) or o.dtype != torch.float32 or not o.is_contiguous()):
            raise RuntimeError("linear_f32: out / out2 must be contiguous fp32 [M, N] tensors")
    check(_lib.load().primx_linear_f32(_dev(x, "x", torch.float32), _dev(W, "W", torch.float32),
                                       _dev(b, "b", torch.float32) if b is not None else None, out.data_ptr(),
                                       _dev(out2, "out2") if out2 is not None else None, M, N, K, act_out, _stream()), "primx_linear_f32")
    return out


def prefetch(t: torch.Tensor, stream: "torch.cuda.Stream") -> None:
    """Enqueue a cache prefetch of `t`'s bytes on `stream` (csrc/rowops.hip primx_prefetch); no result, no dependency."""
    check(_lib.load().primx_prefetch(_dev(t, "t"), t.numel() * t.element_size(), stream.cuda_stream), "primx_prefetch")


# ----------------------------------------------------------------------------- GEMMs
def linear(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
           act: int = ACT_NONE, out_scale: float = 1.0, carry: Optional[torch.Tensor] = None) -> torch.Tensor:
    M, K = A.shape
    N = W.shape[0]
    if W.shape[1] != K or W.dtype != A.dtype:
        raise RuntimeError("linear: operand mismatch")
    if out is None:
        out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    cp, cn = _range(carry)
    _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear(
        _dev(A, "A"), _dev(W, "W"), _dev(bias, "bias", A.dtype) if bias is not None else None,
        _dev(out, "out", A.dtype), M, N, K, dtype_code(A.dtype), act, out_scale, cp, cn, _stream()), "primx_linear"))
    return out


def ln_sync_words(rows: int) -> int:
    """32-bit words of the `sync` workspace primx_linear_gate_residual_ln wants for `rows` rows (two per 128-row block)."""
    return 2 * ((rows + 127) // 128)


def ln_sync_timeouts() -> int:
    """In-kernel waits of the fused gate-residual + LayerNorm route that gave up since the library was loaded (must be 0)."""
    return int(_lib.load().primx_ln_sync_timeouts())


def linear_gate_residual(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], gate: torch.Tensor,
                         x: torch.Tensor, rows_per_batch: int, carry: Optional[torch.Tensor] = None,
                         ln: Optional[tuple] = None) -> torch.Tensor:
    """x[M, N] (fp32, in place) += cast16(gate[b] * cast16(A W^T + bias)).  `carry`: a tensor (a later GEMM's weights) whose
    bytes this launch pulls towards the caches.
    `ln` = (shift, scale, out, eps, sync): ALSO out[m] = cast16(LN(x[m]) (1 + scale[b]) + shift[b]) of the updated rows - the
    LayerNorm + modulate that follows every gated residual add of a DiT block - in the tail of the same kernel where the shape
    allows, as a second launch otherwise (primx_linear_gate_residual_ln; `sync`: zeroed int32 workspace of ln_sync_words(M)
    words or None)."""
    M, K = A.shape
    N = W.shape[0]
    if gate.stride(-1) != 1 or gate.dtype != A.dtype or not gate.is_cuda:
        raise RuntimeError("gate must be a last-dim-contiguous 16-bit device view")
    cp, cn = _range(carry)
    if ln is None:
        _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_gate_residual(
            _dev(A, "A"), _dev(W, "W", A.dtype), _dev(bias, "bias", A.dtype) if bias is not None else None,
            gate.data_ptr(), gate.stride(0), _dev(x, "x", torch.float32), M, N, K, rows_per_batch,
            dtype_code(A.dtype), cp, cn, _stream()), "primx_linear_gate_residual"))
        return x
    shift, scale, out, eps, sync = ln
    _check_modulation(shift, scale, out)
    if PROFILE is not None and sync is None:
        # per-launch timing (bench.py's roofline leg): the two launches the library would issue are issued from here, so that the
        # GEMM's events do not include the LayerNorm kernel - same kernels, same bits
        linear_gate_residual(A, W, bias, gate, x, rows_per_batch, carry=carry)
        layernorm_modulate(x, shift, scale, rows_per_batch, out, eps)
        return x
    if out.dtype != A.dtype or tuple(out.shape) != (M, N):
        raise RuntimeError("linear_gate_residual: the LayerNorm output must be a 16-bit [M, N] tensor of A's dtype")
    if sync is not None and (sync.dtype != torch.int32 or not sync.is_cuda or sync.numel() < ln_sync_words(M)):
        raise RuntimeError("linear_gate_residual: sync must be an int32 device tensor of ln_sync_words(M) words")
    _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_gate_residual_ln(
        _dev(A, "A"), _dev(W, "W", A.dtype), _dev(bias, "bias", A.dtype) if bias is not None else None,
        gate.data_ptr(), gate.stride(0), _dev(x, "x", torch.float32), M, N, K, rows_per_batch,
        shift.data_ptr(), scale.data_ptr(), shift.stride(0), _dev(out, "ln_out"), eps,
        _dev(sync, "sync") if sync is not None else None, sync.numel() if sync is not None else 0,
        dtype_code(A.dtype), cp, cn, _stream()), "primx_linear_gate_residual_ln"))
    return x


def linear_heads(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], rows_per_batch: int, heads: int,
                 dh: int, kinds: Sequence[int], dsts: Sequence[torch.Tensor], n_pad: int,
                 scale0: float = 1.0, n_rep: int = 1, rep_batches: int = 0, real_rows: Optional[int] = None,
                 carry: Optional[torch.Tensor] = None) -> None:
    """n_rep > 1: the column groups repeat; repetition r fills batch entries [r*rep_batches, (r+1)*rep_batches) of dsts.
    `real_rows`: rows of A that are not zero padding (the conditioning tokens are padded 1370 -> 1536 per batch entry) -
    only the per-kernel FLOP credit of bench.py's roofline leg uses it; the launch covers all M rows."""
    M, K = A.shape
    N = W.shape[0]
    n_seg = len(kinds)
    kind_arr = (C.c_int * n_seg)(*kinds)
    dst_arr = (C.c_void_p * n_seg)(*[_dev(d, "dst", A.dtype) for d in dsts])
    cp, cn = _range(carry)
    _timed(f"None {M}x{N}x{K}", 2.0 * (real_rows if real_rows is not None else M) * N * K, lambda: check(_lib.load().primx_linear_heads(
        _dev(A, "A"), _dev(W, "W", A.dtype), _dev(bias, "bias", A.dtype) if bias is not None else None, M, N, K,
        rows_per_batch, heads, dh, n_seg, kind_arr, dst_arr, n_rep, rep_batches, n_pad, scale0, dtype_code(A.dtype),
        cp, cn, _stream()),
        "primx_linear_heads"))


# ----------------------------------------------------------------------------- the LayerNorm fold (include/primx_hip.h, ABI 23)
FOLD_TILE = 144     # columns per partial sum of the producer; the consumers read at most FOLD_MAX_PARTS of them per row
FOLD_MAX_PARTS = 8


def fold_supported(D: int, heads: int) -> bool:
    """Can a DiT of width D fold its LayerNorms into the neighbouring GEMMs?  (producer: N = D in 144-column tiles, at most 8;
    consumers: K = D; heads layout of the 128 x 144 / 256 x 288 kernels)"""
    dh = D // max(heads, 1)
    return D % FOLD_TILE == 0 and D // FOLD_TILE <= FOLD_MAX_PARTS and D % 64 == 0 and dh >= 48 and dh % 4 == 0


def fold_shapes_ok(T: int, rows_per_batch: int, D: int, heads: int) -> bool:
    """Do ALL GEMMs of a DiT block at T = batch entries x rows_per_batch token rows have a fold kernel?  The qkv projection writes a
    V^T segment, which only the 256 x 288 tile's heads epilogue does under the fold: the launch must qualify for it (the rule of
    csrc/gemm.hip launch_fold, including the switches that move it)."""
    if not fold_supported(D, heads) or rows_per_batch < 128 or not _lib.fold_available():
        return False
    if os.environ.get("PRIMX_GEMM_NOBIG") == "1" or os.environ.get("PRIMX_GEMM_LOADER") == "0":
        return False
    big_min = int(os.environ.get("PRIMX_GEMM_BIGHEADS_MIN", "160"))
    dh = D // heads
    return (big_min > 0 and D % 288 == 0 and 288 % dh == 0 and dh % 8 == 0 and dh >= 32 and rows_per_batch % 256 == 0
            and (T // 256) * (3 * D // 288) >= big_min)


def fold_workspace(rows: int, D: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """(center [2, rows, 2] fp32, part [rows, D / 144, 2] fp32) of one forward.  center[k] are the (centre, scale) = (mean, rstd)
    pairs of the rows at one LayerNorm site; the sites alternate between the two arrays (a consumer reads one, writes the other)."""
    return (torch.empty(2, rows, 2, dtype=torch.float32, device=device),
            torch.empty(rows, D // FOLD_TILE, 2, dtype=torch.float32, device=device))


def row_stats(x: torch.Tensor, eps: float, out: torch.Tensor) -> torch.Tensor:
    """out[r] = (mean(x[r, :]), 1 / sqrt(var(x[r, :]) + eps)) (fp32, [rows, 2]): the (centre, scale) pair of the first folded
    LayerNorm site of a forward."""
    rows, D = x.shape
    if tuple(out.shape) != (rows, 2):
        raise RuntimeError("row_stats: out must be [rows, 2] fp32")
    check(_lib.load().primx_row_stats(_dev(x, "x", torch.float32), rows, D, eps, _dev(out, "stats", torch.float32), _stream()),
          "primx_row_stats")
    return out


def linear_f32out(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, bias_from_row: int) -> torch.Tensor:
    """out[M, N] (fp32) = A W^T (+ bias for the rows >= bias_from_row): the fold's u rows (no bias) and v rows in one launch."""
    M, K = A.shape
    N = W.shape[0]
    if tuple(out.shape) != (M, N) or W.shape[1] != K:
        raise RuntimeError("linear_f32out: shape mismatch")
    _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_f32out(
        _dev(A, "A"), _dev(W, "W", A.dtype), _dev(bias, "bias", A.dtype) if bias is not None else None,
        _dev(out, "out", torch.float32), M, N, K, bias_from_row, dtype_code(A.dtype), _stream()), "primx_linear_f32out"))
    return out


def linear_f32out_group(problems: Sequence[Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], torch.Tensor]], bias_from_row: int) -> bool:
    """primx_linear_f32out_group (ABI 26): every (A [M, K], W [N, K], bias [N] or None, out [M, N] fp32) of `problems` - same M, K and dtype -
    from ONE launch (the fold's u / v rows of a whole planned loop).  Returns False, having launched nothing, where the grouped kernel does
    not apply (an older PRIMX_LIB build, PRIMX_UV_GROUP=0, N % 32 or K % 32 != 0): the caller then makes one linear_f32out call per problem."""
    if not problems or not _lib.f32out_group_available() or os.environ.get("PRIMX_UV_GROUP") == "0":
        return False
    M, K = problems[0][0].shape
    dt = problems[0][0].dtype
    rows, first = [], 0
    flops = 0.0
    for A, W, b, out in problems:
        N = W.shape[0]
        if (tuple(A.shape) != (M, K) or A.dtype != dt or W.shape[1] != K or tuple(out.shape) != (M, N) or N % 32 or K % 32
                or not (A.is_contiguous() and W.is_contiguous() and out.is_contiguous()) or out.data_ptr() % 16):
            return False
        rows.append([_dev(A, "A"), _dev(W, "W", dt), _dev(b, "bias", dt) if b is not None else 0, _dev(out, "out", torch.float32),
                     N | (first << 32)])                          # (N, first_wg: two little-endian ints in the struct's last 8 bytes)
        first += (N + 127) // 128
        flops += 2.0 * M * N * K
    table = torch.tensor(rows, dtype=torch.int64).to(problems[0][0].device, non_blocking=False)
    _timed(f"None {len(problems)} problems x {M} rows, K = {K}", flops, lambda: check(_lib.load().primx_linear_f32out_group(
        table.data_ptr(), len(problems), first, M, K, bias_from_row, dtype_code(dt), _stream()), "primx_linear_f32out_group"))
    return True


def _fold_args(part: torch.Tensor, u: torch.Tensor, v: torch.Tensor, center: torch.Tensor, center_out: torch.Tensor, M: int, N: int,
               K: int):
    if K % FOLD_TILE or tuple(part.shape) != (M, K // FOLD_TILE, 2) or tuple(center.shape) != (M, 2) or tuple(center_out.shape) != (M, 2):
        raise RuntimeError("fold: `part` must be [M, K / 144, 2], `center` and `center_out` [M, 2] (fp32)")
    if center.data_ptr() == center_out.data_ptr():
        raise RuntimeError("fold: a consumer writes the next (centre, scale) pairs to ANOTHER array than the one it reads")
    for t, name in ((u, "u"), (v, "v")):
        if t.dtype != torch.float32 or t.numel() != N or not t.is_contiguous() or t.data_ptr() % 16:
            raise RuntimeError(f"fold: {name} must be a contiguous, 16-byte aligned fp32 vector of N elements")
    return (_dev(part, "part", torch.float32), u.data_ptr(), v.data_ptr(), _dev(center, "center", torch.float32),
            _dev(center_out, "center_out", torch.float32))


def linear_gate_residual_fold(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], gate: torch.Tensor, x: torch.Tensor,
                              rows_per_batch: int, next_scale: torch.Tensor, center: torch.Tensor, a16: torch.Tensor,
                              part: torch.Tensor, carry: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear_gate_residual as the PRODUCER of the LayerNorm site behind it: with (c, rho_p) = center[m] also
    a16[m] = cast16((x[m] - c) rho_p cast16(1 + next_scale[b])) and the per-tile partial sums of (x - c), (x - c)^2 into `part`."""
    M, K = A.shape
    N = W.shape[0]
    if gate.stride(-1) != 1 or gate.dtype != A.dtype or not gate.is_cuda:
        raise RuntimeError("gate must be a last-dim-contiguous 16-bit device view")
    if next_scale.stride(-1) != 1 or next_scale.dtype != A.dtype or not next_scale.is_cuda or next_scale.shape[-1] != N:
        raise RuntimeError("next_scale must be a last-dim-contiguous 16-bit device view of N columns")
    if tuple(a16.shape) != (M, N) or a16.dtype != A.dtype or tuple(part.shape) != (M, N // FOLD_TILE, 2) or \
            tuple(center.shape) != (M, 2):
        raise RuntimeError("linear_gate_residual_fold: a16 must be [M, N] 16-bit, part [M, N / 144, 2], center [M, 2]")
    cp, cn = _range(carry)
    _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_gate_residual_fold(
        _dev(A, "A"), _dev(W, "W", A.dtype), _dev(bias, "bias", A.dtype) if bias is not None else None,
        gate.data_ptr(), gate.stride(0), _dev(x, "x", torch.float32), M, N, K, rows_per_batch,
        next_scale.data_ptr(), next_scale.stride(0), _dev(center, "center", torch.float32), _dev(a16, "a16"),
        _dev(part, "part", torch.float32), dtype_code(A.dtype), cp, cn, _stream()), "primx_linear_gate_residual_fold"))
    return x


def linear_heads_fold(A: torch.Tensor, W: torch.Tensor, rows_per_batch: int, heads: int, dh: int, kinds: Sequence[int],
                      dsts: Sequence[torch.Tensor], n_pad: int, part: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                      center: torch.Tensor, center_out: torch.Tensor, eps: float, scale0: float = 1.0,
                      carry: Optional[torch.Tensor] = None) -> None:
    """linear_heads as the CONSUMER of a folded LayerNorm site: A = the producer's a16, `center` the pairs the producer used; the
    Linear's bias is part of v; `center_out` receives this site's (mean, rstd) - the next producer's pairs."""
    M, K = A.shape
    N = W.shape[0]
    n_seg = len(kinds)
    kind_arr = (C.c_int * n_seg)(*kinds)
    dst_arr = (C.c_void_p * n_seg)(*[_dev(d, "dst", A.dtype) for d in dsts])
    fa = _fold_args(part, u, v, center, center_out, M, N, K)
    cp, cn = _range(carry)
    _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_heads_fold(
        _dev(A, "A"), _dev(W, "W", A.dtype), M, N, K, rows_per_batch, heads, dh, n_seg, kind_arr, dst_arr, n_pad, scale0,
        fa[0], fa[1], fa[2], fa[3], fa[4], eps, dtype_code(A.dtype), cp, cn, _stream()), "primx_linear_heads_fold"))


def fold_pair_fits(T: int, rows_per_batch: int, D: int, heads: int, kv_rows: int, kv_rows_per_batch: int, kv_K: int) -> bool:
    """Will primx_linear_heads_fold_pair run a DiT block's qkv projection (T token rows) and the next block's to_k / to_v projection
    (kv_rows conditioning rows) as ONE launch?  The rule of csrc/gemm.hip: both on the 256 x 288 heads tile, the first problem one
    round of at least PRIMX_GEMM_BIGHEADS_MIN workgroups (a multiple of 8), both together at most 256."""
    if not fold_shapes_ok(T, rows_per_batch, D, heads) or os.environ.get("PRIMX_GEMM_PROF"):
        return False
    dh = D // heads
    if (2 * D) % 288 or (heads * dh) % 288 or kv_rows_per_batch % 256 or kv_rows % kv_rows_per_batch or kv_K % 64:
        return False
    g0, g1 = (T // 256) * (3 * D // 288), (kv_rows // 256) * (2 * D // 288)
    return g0 % 8 == 0 and g0 + g1 <= 256


def linear_heads_fold_pair(fold: Optional[dict], A2: torch.Tensor, W2: torch.Tensor, bias2: Optional[torch.Tensor], rows_per_batch2: int,
                           heads2: int, dh2: int, kinds2: Sequence[int], dsts2: Sequence[torch.Tensor], n_pad2: int,
                           scale0_2: float = 1.0, real_rows2: Optional[int] = None) -> None:
    """primx_linear_heads_fold_pair (ABI 25): `fold` = the keyword arguments of linear_heads_fold (without `carry`) for problem 0 - or None:
    problem 1 alone on the 256 x 288 tile kernel - and a plain heads GEMM (linear_heads with n_rep = 1) as problem 1, from ONE launch where
    the pairing rule of the header holds (problem 1's tiles on the CUs problem 0 leaves idle), otherwise from two.  Results = the two calls'."""
    M2, K2 = A2.shape
    N2 = W2.shape[0]
    fl2 = 2.0 * (real_rows2 if real_rows2 is not None else M2) * N2 * K2     # (bench.py's FLOP credit: rows that are not zero padding)
    k2 = (C.c_int * len(kinds2))(*kinds2)
    d2 = (C.c_void_p * len(kinds2))(*[_dev(d, "dst", A2.dtype) for d in dsts2])
    tail = (_dev(A2, "A2"), _dev(W2, "W2", A2.dtype), _dev(bias2, "bias2", A2.dtype) if bias2 is not None else None, M2, N2, K2,
            rows_per_batch2, heads2, dh2, len(kinds2), k2, d2, n_pad2, scale0_2, dtype_code(A2.dtype), _stream())
    if fold is None:
        head = (None, None, 0, 0, 0, 0, 0, 0, 0, None, None, 0, 0.0, None, None, None, None, None, 0.0)
        flops, tag = fl2, f"None {M2}x{N2}x{K2}"
    else:
        A, W = fold["A"], fold["W"]
        M, K = A.shape
        N = W.shape[0]
        kinds, dsts = fold["kinds"], fold["dsts"]
        k = (C.c_int * len(kinds))(*kinds)
        d = (C.c_void_p * len(kinds))(*[_dev(t, "dst", A.dtype) for t in dsts])
        fa = _fold_args(fold["part"], fold["u"], fold["v"], fold["center"], fold["center_out"], M, N, K)
        head = (_dev(A, "A"), _dev(W, "W", A.dtype), M, N, K, fold["rows_per_batch"], fold["heads"], fold["dh"], len(kinds), k, d,
                fold["n_pad"], fold.get("scale0", 1.0), fa[0], fa[1], fa[2], fa[3], fa[4], fold["eps"])
        flops, tag = 2.0 * M * N * K + fl2, f"None {M}x{N}x{K}+{M2}x{N2}x{K2}"
    _timed(tag, flops, lambda: check(_lib.load().primx_linear_heads_fold_pair(*head, *tail), "primx_linear_heads_fold_pair"))


def linear_fold(A: torch.Tensor, W: torch.Tensor, out: torch.Tensor, part: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                center: torch.Tensor, center_out: torch.Tensor, eps: float, act: int = ACT_NONE,
                carry: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear as the CONSUMER of a folded LayerNorm site (fc1 + GELU); center / center_out as in linear_heads_fold."""
    M, K = A.shape
    N = W.shape[0]
    if tuple(out.shape) != (M, N) or out.dtype != A.dtype:
        raise RuntimeError("linear_fold: out must be a 16-bit [M, N] tensor of A's dtype")
    fa = _fold_args(part, u, v, center, center_out, M, N, K)
    cp, cn = _range(carry)
    _timed(f"None {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_fold(
        _dev(A, "A"), _dev(W, "W", A.dtype), _dev(out, "out"), M, N, K, act, fa[0], fa[1], fa[2], fa[3], fa[4], eps,
        dtype_code(A.dtype), cp, cn, _stream()), "primx_linear_fold"))
    return out


# ----------------------------------------------------------------------------- attention
KEY_MASK_VALUE = -30000.0  # finite in fp16 and bf16; times scale*log2(e) it underflows exp2 to exactly 0


def alloc_heads(B: int, H: int, n: int, dh: int, kind: int, dtype: torch.dtype, device, pad_to: int,
                role: Optional[str] = None) -> torch.Tensor:
    """Zero-initialised attention operand buffer; pads are never written by the kernels, so they keep what is put
    here.  When the head dim has a spare padded column (dh < DP, e.g. 72 -> 80) the key-padding mask lives in
    the operands: role "q" sets column dh to 1 for every query row, role "k" sets it to KEY_MASK_VALUE for the
    pad rows >= n, so q.k of a padded key is -30000 and its probability is exactly 0; a V^T buffer gets an
    all-ones row dh (valid keys only) that makes the PV MFMA produce the softmax row sum (include/primx_hip.h)."""
    DP = padded_head_dim(dh)
    n_pad = round_up(n, pad_to)
    shape = {HEADS_ROWS: (B, H, n_pad, DP), HEADS_KROWS: (B, H, n_pad, DP + 8), HEADS_VT: (B, H, DP, n_pad)}[kind]
    # (no pad rows and no pad columns - the VAE's 64-token, 32-wide heads: every element is written by the projection
    # epilogue, so the 67 MB zero fill per operand is skipped; the KROWS layout always has its 8 pad columns per row)
    exact = DP == dh and n_pad == n and kind != HEADS_KROWS
    buf = torch.empty(shape, dtype=dtype, device=device) if exact else torch.zeros(shape, dtype=dtype, device=device)
    if kind != HEADS_VT and DP > dh:
        if role == "q":
            buf[:, :, :, dh] = 1.0
        elif role == "k":
            if n_pad > n:
                buf[:, :, n:, dh] = KEY_MASK_VALUE
            if DP - dh >= 3:
                # two more constant columns: the one-wave-per-SIMD attention kernel keeps (-m_hi, -m_lo) - the running row
                # max split into two 16-bit halves - in ITS register copy of Q's columns dh+1, dh+2, so that q.k comes out
                # of the MFMA with the max already subtracted; Q holds zeros there in memory, so other readers see no change
                buf[:, :, :, dh + 1:dh + 3] = 1.0
    if kind == HEADS_VT and DP > dh:
        # row dh of V^T = 1 for every valid key (at its quad-permuted position): the PV MFMA then accumulates
        # sum_k P[k, q] - the softmax denominator - in output row dh for free
        k = torch.arange(n, device=device)
        quad = (k >> 2) & 3
        pos = (k & ~15) | ((((quad & 1) << 1) | (quad >> 1)) << 2) | (k & 3)
        buf[:, :, dh, pos] = 1.0
    return buf


def bcast_keys(nkv: int) -> int:
    """Keys a broadcast operand entry of attention(..., bcast=) holds for a sequence of nkv identical keys: one full 64-key
    tile and, when nkv % 64 != 0, the ragged last tile behind it."""
    return 64 + (nkv % 64 if nkv > 64 else 0)


def attention(Qp: torch.Tensor, Kp: Optional[torch.Tensor], Vt: Optional[torch.Tensor], nq: int, nkv: int, dh: int, scale: float,
              out: Optional[torch.Tensor] = None, bcast: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> torch.Tensor:
    """bcast = (Kb, Vb): the LAST B - Kp.shape[0] batch entries of Qp attend to nkv copies of one key / value row, held once in
    the single operand entry Kb [1, H, n_pad_b, DP + 8] / Vb [1, H, DP, n_pad_b] (bcast_keys(nkv) keys: csrc/attention.hip
    primx_attention_bcast); Kp / Vt hold the leading entries (None when every entry is a broadcast one)."""
    B, H, nq_pad, DP = Qp.shape
    if bcast is not None:
        Kb, Vb = bcast
        b_from = 0 if Kp is None else Kp.shape[0]
        nkv_pad_b = Kb.shape[2]
        if Kb.shape != (1, H, nkv_pad_b, DP + 8) or Vb.shape != (1, H, DP, nkv_pad_b) or not 0 <= b_from < B:
            raise RuntimeError("attention: broadcast operand layout mismatch")
        if Kp is not None and (Vt is None or Vt.shape != (b_from, H, DP, Kp.shape[2]) or Kp.shape[1:] != (H, Kp.shape[2], DP + 8)):
            raise RuntimeError("attention: operand layout mismatch")
        nkv_pad = Kp.shape[2] if Kp is not None else nkv_pad_b
        if out is None:
            out = torch.empty(B, nq, H * dh, dtype=Qp.dtype, device=Qp.device)
        name = f"attn_kernel<{dtype_code(Qp.dtype)}, {padded_head_dim(dh) // 16}, {(dh + 31) // 32}, {int(padded_head_dim(dh) == dh)}, 0>"
        _timed(f"{name} {B * H}x{nq}x{nkv}x{dh}", 4.0 * B * H * nq * nkv * dh, lambda: check(_lib.load().primx_attention_bcast(
            _dev(Qp, "Qp"), _dev(Kp, "Kp", Qp.dtype) if Kp is not None else None, _dev(Vt, "Vt", Qp.dtype) if Kp is not None else None,
            _dev(out, "out", Qp.dtype), B, H, nq, nq_pad, nkv, nkv_pad, dh, scale, _dev(Kb, "Kb", Qp.dtype), _dev(Vb, "Vb", Qp.dtype),
            b_from, nkv_pad_b, dtype_code(Qp.dtype), _stream()), "primx_attention_bcast"))
        return out
    nkv_pad = Kp.shape[2]
    if Vt.shape != (B, H, DP, nkv_pad) or Kp.shape != (B, H, nkv_pad, DP + 8):
        raise RuntimeError("attention: operand layout mismatch")
    if out is None:
        out = torch.empty(B, nq, H * dh, dtype=Qp.dtype, device=Qp.device)
    # algorithmic FLOPs: QK^T + PV on the unpadded head dim, softmax excluded
    small = dh == 32 and nq <= 64 and nkv <= 64 and nq_pad == 64 and nkv_pad == 64    # csrc/attention.hip: one wave per problem
    name = f"attn64_kernel<{dtype_code(Qp.dtype)}>" if small else \
        f"attn_kernel<{dtype_code(Qp.dtype)}, {padded_head_dim(dh) // 16}, {(dh + 31) // 32}, {int(padded_head_dim(dh) == dh)}, 0>"
    _timed(f"{name} {B * H}x{nq}x{nkv}x{dh}", 4.0 * B * H * nq * nkv * dh, lambda: check(_lib.load().primx_attention(
        _dev(Qp, "Qp"), _dev(Kp, "Kp", Qp.dtype), _dev(Vt, "Vt", Qp.dtype), _dev(out, "out", Qp.dtype), B, H, nq,
        nq_pad, nkv, nkv_pad, dh, scale, dtype_code(Qp.dtype), _stream()), "primx_attention"))
    return out


def pack_heads(src: torch.Tensor, kind: int, pad_to: int, role: Optional[str] = None) -> torch.Tensor:
    """src: [B, M, H, dh] view with contiguous last dim -> attention operand layout (role: see alloc_heads)."""
    B, M, H, dh = src.shape
    if src.stride(3) != 1 or not src.is_cuda:
        raise RuntimeError("pack_heads: last dim must be contiguous on a HIP device")
    dst = alloc_heads(B, H, M, dh, kind, src.dtype, src.device, pad_to, role)
    m_pad = dst.shape[3] if kind == HEADS_VT else dst.shape[2]
    check(_lib.load().primx_pack_heads(src.data_ptr(), src.stride(0), src.stride(1), src.stride(2), dst.data_ptr(),
                                       kind, B, M, H, dh, m_pad, dtype_code(src.dtype), _stream()), "primx_pack_heads")
    return dst


def memory_efficient_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_bias=None,
                               scale: Optional[float] = None) -> torch.Tensor:
    """Drop-in for ``xformers.ops.memory_efficient_attention`` on ``[B, M, H, K]`` 16-bit operands
    (the reference's call sites: models/attention.py:54,109): default scale ``K**-0.5``, no bias, p=0."""
    if attn_bias is not None:
        raise NotImplementedError("attn_bias is not used on the 3DTopia-XL path")
    B, Mq, H, dh = q.shape
    Mk = k.shape[1]
    Qp = pack_heads(q, HEADS_ROWS, BQ, "q")
    Kp = pack_heads(k, HEADS_KROWS, BKV, "k")
    Vt = pack_heads(v, HEADS_VT, BKV)
    out = attention(Qp, Kp, Vt, Mq, Mk, dh, dh ** -0.5 if scale is None else scale)
    return out.view(B, Mq, H, dh)


# ----------------------------------------------------------------------------- CFG + sampler update
def cfg_combine(model_out: torch.Tensor, cfg_scale: float) -> torch.Tensor:
    """model_out: [2B, ...] (cond half first) -> [B, ...]   (dit_crossattn.py:210-213)."""
    B2 = model_out.shape[0]
    out = torch.empty((B2 // 2,) + tuple(model_out.shape[1:]), dtype=model_out.dtype, device=model_out.device)
    check(_lib.load().primx_cfg_combine(_dev(model_out, "model_out"), out.data_ptr(), dtype_code(model_out.dtype),
                                        out.numel(), cfg_scale, _stream()), "primx_cfg_combine")
    return out


def diffusion_step(x: torch.Tensor, model_out: torch.Tensor, coef: torch.Tensor, step: int, *, mean_type: int,
                   var_type: int, ancestral: bool, clip_denoised: bool,
                   noise: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    B, nt, Cc = x.shape
    model_out = model_out.contiguous()
    sample = torch.empty_like(x)
    x0 = torch.empty_like(x)
    check(_lib.load().primx_diffusion_step(
        _dev(x, "x", torch.float32), _dev(model_out, "model_out"), dtype_code(model_out.dtype), B * nt, Cc,
        model_out.shape[-1], _dev(coef, "coef", torch.float32), step, mean_type, var_type, int(ancestral),
        int(clip_denoised), _dev(noise, "noise", torch.float32) if noise is not None else None,
        sample.data_ptr(), x0.data_ptr(), _stream()), "primx_diffusion_step")
    return sample, x0


# ----------------------------------------------------------------------------- VAE decoder
def groupnorm_silu(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float,
                   silu: bool) -> torch.Tensor:
    """x: [P, V, C] 16-bit channels-last."""
    P, V, Cc = x.shape
    out = torch.empty_like(x)
    _timed(f"groupnorm_silu {Cc}ch @{V}vox x{P}", 0.0, lambda: check(_lib.load().primx_groupnorm_silu(
        _dev(x, "x"), _dev(gamma, "gamma", torch.float32), _dev(beta, "beta", torch.float32), out.data_ptr(), P, V, Cc,
        groups, eps, int(silu), dtype_code(x.dtype), _stream()), "primx_groupnorm_silu"))
    return out


class PackedConv3:
    """Weight image of one of the activation-resident 3x3x3 kernels: kind "s4" (csrc/conv3.hip: 4^3 grid, Cin 256, Cout %
    256 == 0), "s8" (csrc/conv3s8.hip: 8^3 grid, Cin 256, Cout 32) or "s8c32" (csrc/conv3s8c32.hip: 8^3 grid, Cin 32, Cout 32
    or <= 16; the only one that can take the preceding GroupNorm + SiLU into the kernel)."""
    __slots__ = ("kind", "S", "Cin", "Cout", "Wp", "has_sc")

    def __init__(self, kind: str, S: int, Cin: int, Cout: int, Wp: torch.Tensor):
        self.kind, self.S, self.Cin, self.Cout, self.Wp, self.has_sc = kind, S, Cin, Cout, Wp, False


def pack_conv3(Wk: torch.Tensor, Cin: int, Wsc: Optional[torch.Tensor] = None) -> Optional[PackedConv3]:
    """PackedConv3 for a [Cout, 27*Cin] conv3d_k3 weight whose shape one of the activation-resident kernels covers, else
    None (also with PRIMX_CONV_REG=0, which keeps the implicit GEMM for A/B runs).  The grid edge the image is for is part
    of the result; conv3d_k3 only uses it on that grid."""
    if not Wk.is_cuda or os.environ.get("PRIMX_CONV_REG", "1") == "0":
        return None
    Cout = Wk.shape[0]
    if Cin == 32 and Wk.shape[1] >= 864 and (Cout == 32 or Cout <= 16):
        NI = 2 if Cout == 32 else 1
        Wp = torch.empty(27 * NI * 16 * 32, dtype=Wk.dtype, device=Wk.device)
        with torch.cuda.device(Wk.device):
            check(_lib.load().primx_conv3d_s8c32_pack(_dev(Wk, "Wk"), Wp.data_ptr(), Cout, Wk.shape[1], dtype_code(Wk.dtype), _stream()),
                  "primx_conv3d_s8c32_pack")
        return PackedConv3("s8c32", 8, Cin, Cout, Wp)
    if Cin != 256 or Wk.shape[1] != 27 * 256:
        return None
    if Cout % 256 != 0 and Cout != 32:
        return None
    Wp = torch.empty_like(Wk)
    with torch.cuda.device(Wk.device):
        if Cout == 32:
            # (with the ResnetBlock's 1x1 shortcut weight [32, 256] a 28th block is appended: conv3d_s8_fused)
            if Wsc is not None:
                Wp = torch.empty(28 * 8192, dtype=Wk.dtype, device=Wk.device)
            check(_lib.load().primx_conv3d_s8_pack(_dev(Wk, "Wk"), _dev(Wsc, "Wsc", Wk.dtype) if Wsc is not None else None, Wp.data_ptr(),
                                                   dtype_code(Wk.dtype), _stream()), "primx_conv3d_s8_pack")
            pc = PackedConv3("s8", 8, Cin, Cout, Wp)
            pc.has_sc = Wsc is not None
            return pc
        check(_lib.load().primx_conv3d_s4_pack(_dev(Wk, "Wk"), Wp.data_ptr(), Cout, dtype_code(Wk.dtype), _stream()), "primx_conv3d_s4_pack")
        return PackedConv3("s4", 4, Cin, Cout, Wp)


def conv3d_s8_fused(x_raw: torch.Tensor, Wp: PackedConv3, bias: Optional[torch.Tensor], part: torch.Tensor, up_bias: torch.Tensor,
                    gamma: torch.Tensor, beta: torch.Tensor, eps: float, sc_bias: Optional[torch.Tensor]):
    """(conv1(silu(group_norm(x_raw))), shortcut(x_raw)) of the 256 -> 32 ResnetBlock on the 8^3 grid in ONE kernel: x_raw is the
    upsample output [P, 512, 256], `part` / `up_bias` its partial GroupNorm sums and their shifts (convtranspose_k2s2(...,
    want_stats=True)), Wp a pack_conv3(Wk, 256, Wsc) image with the shortcut block."""
    P, V, Cin = x_raw.shape
    if not (Wp.kind == "s8" and Wp.has_sc and V == 512 and Cin == 256):
        raise ValueError("conv3d_s8_fused: needs the 8^3 256 -> 32 image with the shortcut block")
    out = torch.empty(P, V, 32, dtype=x_raw.dtype, device=x_raw.device)
    sc = torch.empty(P, V, 32, dtype=x_raw.dtype, device=x_raw.device)
    tag = f"conv3_s8c256n32_kernel<{dtype_code(x_raw.dtype)}, 1> gn+256->32+sc @8^3 x{P}"
    _timed(tag, 2.0 * P * V * 32 * 28 * 256, lambda: check(_lib.load().primx_conv3d_s8_fused(
        _dev(x_raw, "x_raw"), _dev(Wp.Wp, "Wp", x_raw.dtype), _dev(bias, "bias", x_raw.dtype) if bias is not None else None,
        _dev(part, "part", torch.float32), _dev(up_bias, "up_bias", x_raw.dtype), _dev(gamma, "gamma", torch.float32),
        _dev(beta, "beta", torch.float32), float(eps), _dev(sc_bias, "sc_bias", x_raw.dtype) if sc_bias is not None else None,
        out.data_ptr(), sc.data_ptr(), P, dtype_code(x_raw.dtype), _stream()), "primx_conv3d_s8_fused"))
    return out, sc


def conv3_takes_groupnorm(Wp: Optional[PackedConv3], S: int, groups: int) -> bool:
    """True when conv3d_k3(..., Wp=Wp, gn=...) can apply the preceding GroupNorm(groups) + SiLU inside the kernel."""
    return Wp is not None and Wp.kind == "s8c32" and Wp.S == S and groups == Wp.Cin


def conv3d_k3(x: torch.Tensor, Wk: torch.Tensor, bias: Optional[torch.Tensor], S: int,
              res: Optional[torch.Tensor] = None, res_scale: float = 1.0, Wp: Optional[PackedConv3] = None,
              gn: Optional[Tuple[torch.Tensor, torch.Tensor, float]] = None) -> torch.Tensor:
    """x: [P, S^3, Cin]; Wk: [Cout, Kpad] 16-bit (k = tap*Cin + ci); optional fused (conv + res) * res_scale.
    Wp (pack_conv3(Wk)) selects the activation-resident kernel when the grid is the one it was packed for.
    gn = (gamma, beta, eps): the input is silu(group_norm(x)) with one channel per group, computed inside the kernel - only
    where conv3_takes_groupnorm(Wp, S, groups) holds."""
    P, V, Cin = x.shape
    Cout, Kpad = Wk.shape
    out = torch.empty(P, V, Cout, dtype=x.dtype, device=x.device)
    bias_p = _dev(bias, "bias", x.dtype) if bias is not None else None
    res_p = _dev(res, "res", x.dtype) if res is not None else None
    tag, flops = f"conv3d_k3 {Cin}->{Cout} @{S}^3 x{P}", 2.0 * P * V * Cout * 27 * Cin
    if gn is not None and not conv3_takes_groupnorm(Wp, S, Cin):
        raise ValueError("conv3d_k3: gn= needs the 8^3 / 32-channel kernel (pack_conv3) and one channel per group")
    if Wp is not None and Wp.S == S and Wp.Cin == Cin and Wp.Cout == Cout and Wp.kind == "s8c32":
        tag = f"conv3_s8c32_kernel<{dtype_code(x.dtype)}, {2 if Cout == 32 else 1}> {'gn+' if gn else ''}{Cin}->{Cout} @{S}^3 x{P}"
        g_p = _dev(gn[0], "gamma", torch.float32) if gn else None
        b_p = _dev(gn[1], "beta", torch.float32) if gn else None
        _timed(tag, flops, lambda: check(_lib.load().primx_conv3d_s8c32_packed(
            _dev(x, "x"), _dev(Wp.Wp, "Wp", x.dtype), bias_p, g_p, b_p, float(gn[2]) if gn else 0.0, res_p, res_scale, out.data_ptr(),
            P, Cout, dtype_code(x.dtype), _stream()), "primx_conv3d_s8c32_packed"))
        return out
    if Wp is not None and Wp.S == S and Wp.Cin == Cin and Wp.Cout == Cout:
        # (tags of the activation-resident kernels: the kernel name as rocprofv3 prints it + the shape, like the GEMMs')
        if Wp.kind == "s4":
            tag = f"conv3_s4c256_kernel<{dtype_code(x.dtype)}, 0> {Cin}->{Cout} @{S}^3 x{P}"
            _timed(tag, flops, lambda: check(_lib.load().primx_conv3d_s4_packed(
                _dev(x, "x"), _dev(Wp.Wp, "Wp", x.dtype), bias_p, res_p, res_scale, out.data_ptr(), P, Cout, dtype_code(x.dtype),
                _stream()), "primx_conv3d_s4_packed"))
        else:
            tag = f"conv3_s8c256n32_kernel<{dtype_code(x.dtype)}, 0> {Cin}->{Cout} @{S}^3 x{P}"
            _timed(tag, flops, lambda: check(_lib.load().primx_conv3d_s8_packed(
                _dev(x, "x"), _dev(Wp.Wp, "Wp", x.dtype), bias_p, res_p, res_scale, out.data_ptr(), P, dtype_code(x.dtype),
                _stream()), "primx_conv3d_s8_packed"))
        return out
    _timed(tag, flops, lambda: check(_lib.load().primx_conv3d_k3(
        _dev(x, "x"), _dev(Wk, "Wk", x.dtype), bias_p, res_p, res_scale, out.data_ptr(), P, S, Cin, Cout, Kpad,
        dtype_code(x.dtype), _stream()), "primx_conv3d_k3"))
    return out


def linear_residual(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor],
                    res: Optional[torch.Tensor], scale: float) -> torch.Tensor:
    """out[M, N] = ((A W^T + bias) + res) * scale, 16-bit, no intermediate rounding."""
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, dtype=A.dtype, device=A.device)
    _timed(f"linear_residual {M}x{N}x{K}", 2.0 * M * N * K, lambda: check(_lib.load().primx_linear_residual(
        _dev(A, "A"), _dev(W, "W", A.dtype), _dev(bias, "bias", A.dtype) if bias is not None else None,
        _dev(res, "res", A.dtype) if res is not None else None, scale, out.data_ptr(), M, N, K, dtype_code(A.dtype),
        _stream()), "primx_linear_residual"))
    return out


def conv_in(z: torch.Tensor, pq_scale: float, pq_bias: float, W: torch.Tensor, bias: torch.Tensor, S: int,
            dtype: torch.dtype) -> torch.Tensor:
    """z: [P, S^3] fp32 latent; W: [Cout, 27] fp32."""
    P = z.shape[0]
    Cout = W.shape[0]
    out = torch.empty(P, S * S * S, Cout, dtype=dtype, device=z.device)
    check(_lib.load().primx_conv_in(_dev(z, "z", torch.float32), pq_scale, pq_bias, _dev(W, "W", torch.float32),
                                    _dev(bias, "bias", torch.float32), out.data_ptr(), P, S, Cout,
                                    dtype_code(dtype), _stream()), "primx_conv_in")
    return out


def pack_convt_s4(Wt: torch.Tensor) -> Optional[torch.Tensor]:
    """Weight image of csrc/convt.hip for the [8*256, 256] k2s2 weight, or None for other shapes / PRIMX_CONV_REG=0."""
    if tuple(Wt.shape) != (8 * 256, 256) or not Wt.is_cuda or os.environ.get("PRIMX_CONV_REG", "1") == "0":
        return None
    Wp = torch.empty_like(Wt)
    with torch.cuda.device(Wt.device):
        check(_lib.load().primx_convtranspose_s4_pack(_dev(Wt, "Wt"), Wp.data_ptr(), dtype_code(Wt.dtype), _stream()),
              "primx_convtranspose_s4_pack")
    return Wp


def convtranspose_k2s2(x: torch.Tensor, Wt: torch.Tensor, bias: torch.Tensor, S: int, Wp: Optional[torch.Tensor] = None,
                       want_stats: bool = False):
    """x: [P, S^3, Cin]; Wt: [8*Cout, Cin] 16-bit (row = tap*Cout + co) -> [P, (2S)^3, Cout].
    With Wp (pack_convt_s4(Wt)) and S == 4 the weight-stationary kernel runs; with want_stats as well the result is
    (out, part): part [P, 16, 32, 2] fp32 = partial shifted sums per group of 8 output channels (include/primx_hip.h);
    group_stats(part, bias, eps) turns them into mean / rstd."""
    P, V, Cin = x.shape
    Cout = Wt.shape[0] // 8
    out = torch.empty(P, 8 * V, Cout, dtype=x.dtype, device=x.device)
    if Wp is not None and S == 4 and Cin == 256 and Cout == 256:
        part = torch.empty(P, 16, 32, 2, dtype=torch.float32, device=x.device) if want_stats else None
        _timed(f"convt_s4c256_kernel<{dtype_code(x.dtype)}> {Cin}->{Cout} @{S}^3 x{P}", 2.0 * P * V * 8 * Cout * Cin, lambda: check(
            _lib.load().primx_convtranspose_s4_packed(_dev(x, "x"), _dev(Wp, "Wp", x.dtype), _dev(bias, "bias", x.dtype), out.data_ptr(),
                                                      part.data_ptr() if part is not None else None, P, dtype_code(x.dtype),
                                                      _stream()), "primx_convtranspose_s4_packed"))
        return (out, part) if want_stats else out
    if want_stats:
        raise ValueError("convtranspose_k2s2: statistics come only from the packed 4^3 kernel (pack_convt_s4)")
    _timed(f"convtranspose_k2s2 {Cin}->{Cout} @{S}^3 x{P}", 2.0 * P * V * 8 * Cout * Cin, lambda: check(
        _lib.load().primx_convtranspose_k2s2(_dev(x, "x"), _dev(Wt, "Wt", x.dtype), _dev(bias, "bias", x.dtype),
                                             out.data_ptr(), P, S, Cin, Cout, dtype_code(x.dtype), _stream()),
        "primx_convtranspose_k2s2"))
    return out


def group_stats(part: torch.Tensor, bias: torch.Tensor, eps: float, count: int = 4096) -> torch.Tensor:
    """[P, 16, 32, 2] partial shifted sums of convtranspose_k2s2(..., want_stats=True) -> [P, 32, 2] (mean, rstd), in the
    arithmetic of the consuming kernel (fp32, partials added in index order).  Host-side helper for tests and callers that
    want the statistics themselves; the fused convolution does this sum in its prologue."""
    acc = torch.zeros_like(part[:, 0])
    for i in range(part.shape[1]):
        acc = acc + part[:, i]
    shift = bias.float()[0::8].view(1, 32)
    m = acc[..., 0] / count
    var = (acc[..., 1] / count - m * m).clamp_min(0.0)
    return torch.stack([shift + m, torch.rsqrt(var + eps)], dim=-1)


def vae_output(x: torch.Tensor, denorm: bool, sdf_div: float = 5.0) -> torch.Tensor:
    """x: [P, V, C] 16-bit channels-last -> [P, C, V] fp32."""
    P, V, Cc = x.shape
    out = torch.empty(P, Cc, V, dtype=torch.float32, device=x.device)
    check(_lib.load().primx_vae_output(_dev(x, "x"), out.data_ptr(), P, V, Cc, int(denorm), sdf_div,
                                       dtype_code(x.dtype), _stream()), "primx_vae_output")
    return out


def latent_denorm(x: torch.Tensor, mean: torch.Tensor, std: torch.Tensor, nf: float, n_srt: int = 4):
    """x: [..., C] fp32 samples -> (srt [..., n_srt], z [..., C - n_srt]) de-normalised (inference.py:328-332)."""
    C = x.shape[-1]
    rows = x.numel() // C
    srt = torch.empty(*x.shape[:-1], n_srt, dtype=torch.float32, device=x.device)
    z = torch.empty(*x.shape[:-1], C - n_srt, dtype=torch.float32, device=x.device)
    check(_lib.load().primx_latent_denorm(_dev(x, "x", torch.float32), _dev(mean, "mean", torch.float32),
                                          _dev(std, "std", torch.float32), float(nf), srt.data_ptr(), z.data_ptr(),
                                          rows, C, n_srt, _stream()), "primx_latent_denorm")
    return srt, z


# ----------------------------------------------------------------------------- DINOv2 conditioner pieces
def vit_tokens(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, reg: Optional[torch.Tensor]) -> torch.Tensor:
    """patches [B, np, D], cls [D], pos [1+np, D], reg [R, D] or None (all fp32) -> tokens [B, 1+R+np, D] fp32
    (dinov2/models/vision_transformer.py:218-236)."""
    B, npatch, D = patches.shape
    R = 0 if reg is None else reg.shape[0]
    out = torch.empty(B, 1 + R + npatch, D, dtype=torch.float32, device=patches.device)
    check(_lib.load().primx_vit_tokens(_dev(patches, "patches", torch.float32), _dev(cls, "cls", torch.float32),
                                       _dev(pos, "pos", torch.float32),
                                       _dev(reg, "reg", torch.float32) if reg is not None else None, out.data_ptr(), B,
                                       npatch, R, D, _stream()), "primx_vit_tokens")
    return out


# ----------------------------------------------------------------------------- the reference's fp32 (no-autocast) path
def gemm_f32(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], out: Optional[torch.Tensor] = None,
             act: int = ACT_NONE, out_scale: float = 1.0, gate: Optional[torch.Tensor] = None,
             rows_per_batch: int = 0) -> torch.Tensor:
    """fp32 nn.Linear on the fp32 matrix instruction.  gate is None: out = act(A W^T + bias) * out_scale; else the
    in-place gated residual out[m] += gate[m // rows_per_batch] * (A W^T + bias)[m]  (csrc/fp32.hip)."""
    M, K = A.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise RuntimeError("gemm_f32: operand mismatch")
    if gate is not None:
        if out is None or gate.stride(-1) != 1 or gate.dtype != torch.float32 or not gate.is_cuda:
            raise RuntimeError("gemm_f32: the gated form updates `out` in place and needs a last-dim-contiguous fp32 gate")
    elif out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    _timed("gemm_f32_kernel", 2.0 * M * N * K, lambda: check(_lib.load().primx_gemm_f32(
        _dev(A, "A", torch.float32), _dev(W, "W", torch.float32),
        _dev(bias, "bias", torch.float32) if bias is not None else None, _dev(out, "out", torch.float32), M, N, K, act,
        out_scale, gate.data_ptr() if gate is not None else None, gate.stride(0) if gate is not None else 0,
        rows_per_batch, _stream()), "primx_gemm_f32"))
    return out


def attention_f32(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None) -> torch.Tensor:
    """softmax(q k^T * scale) v for fp32 [B, M, H, dh] views with a contiguous last dim (any batch / token / head
    strides) -> contiguous [B, Mq, H, dh]: the xFormers call of the reference with autocast off."""
    B, Mq, H, dh = q.shape
    Mk = k.shape[1]
    if k.shape != (B, Mk, H, dh) or v.shape != (B, Mk, H, dh):
        raise RuntimeError("attention_f32: q / k / v must be [B, M, H, dh] with equal B, H, dh")
    for t, name in ((q, "q"), (k, "k"), (v, "v")):
        if t.dtype != torch.float32 or not t.is_cuda or t.stride(3) != 1:
            raise RuntimeError(f"attention_f32: {name} must be an fp32 HIP tensor with a contiguous last dim")
    out = torch.empty(B, Mq, H, dh, dtype=torch.float32, device=q.device)
    strides = [(C.c_int64 * 3)(t.stride(0), t.stride(1), t.stride(2)) for t in (q, k, v)]
    _dev(out, "out", torch.float32)
    _timed("attn_f32_kernel", 4.0 * B * H * Mq * Mk * dh, lambda: check(_lib.load().primx_attention_f32(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), B, H, Mq, Mk, dh, strides[0], strides[1], strides[2],
        dh ** -0.5 if scale is None else scale, _stream()), "primx_attention_f32"))
    return out


def layernorm_modulate_f32(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, rows_per_batch: int,
                           eps: float = 1e-6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    rows, D = x.shape
    if shift.stride(-1) != 1 or scale.stride(-1) != 1 or shift.stride(0) != scale.stride(0) \
            or shift.dtype != torch.float32 or scale.dtype != torch.float32:
        raise RuntimeError("layernorm_modulate_f32: shift/scale must be fp32, last-dim contiguous, equal row strides")
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().primx_layernorm_modulate_f32(_dev(x, "x", torch.float32), shift.data_ptr(), scale.data_ptr(),
                                                   shift.stride(0), _dev(out, "out", torch.float32), rows, rows_per_batch,
                                                   D, eps, _stream()), "primx_layernorm_modulate_f32")
    return out


def silu_f32(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    check(_lib.load().primx_silu_f32(_dev(x, "x", torch.float32), out.data_ptr(), x.numel(), _stream()), "primx_silu_f32")
    return out
