"""ctypes binding of ``libprimx_hip.so`` (the C ABI declared in ``include/primx_hip.h``).

There is exactly one compute backend.  If the library has not been built (``__graft_entry__.build()``
or ``python 3dtopia-xl_amd/csrc/build.py``) every op raises - nothing falls back to PyTorch or the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PRIMX_LIB") or os.path.join(_HERE, "csrc", "libprimx_hip.so")  # PRIMX_LIB: A/B builds

F32, F16, BF16 = 0, 1, 2
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF = 0, 1, 2
HEADS_ROWS, HEADS_VT, HEADS_KROWS = 0, 1, 2
ABI_VERSION = 26

_p, _i, _l, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float



class DitBlockFold(C.Structure):
    """PrimxDitBlockFold (include/primx_hip.h): weights, cross-attention operands, fold tables and prefetch ranges of one DiT block."""
    _fields_ = [(n, _p) for n in ("w_q", "b_q", "w_cproj", "b_cproj", "w_qkv", "w_proj", "b_proj", "w_fc1", "w_fc2", "b_fc2",
                                  "Kc", "Vc", "Kb", "Vb", "uv_q", "uv_qkv", "uv_fc1",
                                  "carry_q", "carry_cproj", "carry_fc1", "carry_fc2")] + \
               [(n, _l) for n in ("carry_q_bytes", "carry_cproj_bytes", "carry_fc1_bytes", "carry_fc2_bytes")]


class F32outProblem(C.Structure):
    """PrimxF32outProblem (include/primx_hip.h, ABI 26): one problem of primx_linear_f32out_group; the array lives in DEVICE memory (40 bytes each)."""
    _fields_ = [("A", _p), ("W", _p), ("bias", _p), ("out", _p), ("N", _i), ("first_wg", _i)]


class DitForwardFold(C.Structure):
    """PrimxDitForwardFold (include/primx_hip.h): one forward's shapes, workspaces and tables."""
    _fields_ = [(n, _i) for n in ("dtype", "Be", "N", "D", "H", "dh", "hidden", "depth", "L", "nq_pad", "nkv_pad_c", "nkv_pad_b",
                                  "b_from", "step", "n_steps")] + \
               [("ln_eps", _f), ("scale", _f)] + \
               [(n, _p) for n in ("h", "xn", "att", "hid", "Qc", "Qs", "Ks", "Vs", "mod", "center0", "center1", "part")] + \
               [(n, _p) for n in ("kv_A", "kv_W", "kv_bias")] + [(n, _i) for n in ("kv_rows", "kv_rows_per_batch", "kv_K")]   # ABI 25


# name -> argument ctypes (return type is always int unless listed in _RESTYPES)
SIGNATURES = {
    "primx_dit_blocks_fold": [C.POINTER(DitForwardFold), C.POINTER(DitBlockFold), _p],
    "primx_abi_version": [],
    "primx_last_error": [],
    "primx_last_gemm_kernel": [],
    "primx_padded_head_dim": [_i],
    "primx_layernorm_modulate": [_p, _p, _p, _l, _p, _i, _i, _i, _i, _f, _p, _l, _p, _l, _p],
    "primx_timestep_embedding": [_p, _p, _p, _i, _i, _p],
    "primx_compute_raydirs": [_p, _p, _p, _p, _p, _f, _p, _p, _p, _i, _i, _i, _p],
    "primx_raymarch": [_p, _p, _p, _f, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p],
    "primx_primsdf_query": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "primx_vit_tokens": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "primx_point_features": [_p, _l, _p, _p, _l, _i, _i, _p],
    "primx_prefetch": [_p, _l, _p],
    "primx_silu_cast": [_p, _p, _i, _l, _p],
    "primx_cast16": [_p, _p, _i, _l, _p],
    "primx_linear_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "primx_linear": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _p, _l, _p],
    "primx_linear_gate_residual": [_p, _p, _p, _p, _l, _p, _i, _i, _i, _i, _i, _p, _l, _p],
    "primx_linear_gate_residual_ln": [_p, _p, _p, _p, _l, _p, _i, _i, _i, _i, _p, _p, _l, _p, _f, _p, _l, _i, _p, _l, _p],
    "primx_ln_sync_timeouts": [],
    "primx_linear_f32out": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "primx_row_stats": [_p, _i, _i, _f, _p, _p],
    "primx_linear_gate_residual_fold": [_p, _p, _p, _p, _l, _p, _i, _i, _i, _i, _p, _l, _p, _p, _p, _i, _p, _l, _p],
    "primx_linear_heads_fold": [_p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_p), _i, _f, _p, _p, _p, _p, _p, _f,
                                _i, _p, _l, _p],
    "primx_linear_heads_fold_pair": [_p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_p), _i, _f, _p, _p, _p, _p, _p, _f,
                                     _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_p), _i, _f, _i, _p],
    "primx_linear_f32out_group": [_p, _i, _i, _i, _i, _i, _i, _p],
    "primx_linear_fold": [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _i, _p, _l, _p],
    "primx_linear_heads": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_p), _i, _i, _i, _f, _i, _p, _l, _p],
    "primx_attention": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "primx_attention_bcast": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p, _p, _i, _i, _i, _p],
    "primx_pack_heads": [_p, _l, _l, _l, _p, _i, _i, _i, _i, _i, _i, _i, _p],
    "primx_cfg_combine": [_p, _p, _i, _l, _f, _p],
    "primx_diffusion_step": [_p, _p, _i, _l, _i, _i, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "primx_groupnorm_silu": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i, _p],
    "primx_conv3d_k3": [_p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _i, _i, _p],
    "primx_conv3d_s4_pack": [_p, _p, _i, _i, _p],
    "primx_conv3d_s4_packed": [_p, _p, _p, _p, _f, _p, _i, _i, _i, _p],
    "primx_conv3d_s8_pack": [_p, _p, _p, _i, _p],
    "primx_conv3d_s8_fused": [_p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _i, _i, _p],
    "primx_conv3d_s8_packed": [_p, _p, _p, _p, _f, _p, _i, _i, _p],
    "primx_convtranspose_s4_pack": [_p, _p, _i, _p],
    "primx_convtranspose_s4_packed": [_p, _p, _p, _p, _p, _i, _i, _p],
    "primx_conv3d_s8c32_pack": [_p, _p, _i, _i, _i, _p],
    "primx_conv3d_s8c32_packed": [_p, _p, _p, _p, _p, _f, _p, _f, _p, _i, _i, _i, _p],
    "primx_linear_residual": [_p, _p, _p, _p, _f, _p, _i, _i, _i, _i, _p],
    "primx_conv_in": [_p, _f, _f, _p, _p, _p, _i, _i, _i, _i, _p],
    "primx_convtranspose_k2s2": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p],
    "primx_vae_output": [_p, _p, _i, _i, _i, _i, _f, _i, _p],
    "primx_latent_denorm": [_p, _p, _p, _f, _p, _p, _l, _i, _i, _p],
    "primx_gemm_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _l, _i, _p],
    "primx_attention_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, C.POINTER(_l), C.POINTER(_l), C.POINTER(_l), _f, _p],
    "primx_layernorm_modulate_f32": [_p, _p, _p, _l, _p, _i, _i, _i, _f, _p],
    "primx_silu_f32": [_p, _p, _l, _p],
}
_RESTYPES = {"primx_last_error": C.c_char_p, "primx_last_gemm_kernel": C.c_char_p}
# an alternate build named by PRIMX_LIB (same-box A/B against another build) must speak the same ABI: version 21 changed the
# argument lists of the GEMM / LayerNorm entry points (explicit prefetch ranges), so older libraries cannot be bound any more;
# versions 22 / 23 only added / re-typed the LayerNorm-fold entry points, so a version-21 or -22 build can stand in as long as
# nothing folds: the fold prototypes are not bound to such a build and `fold_available()` is False (ops.fold_shapes_ok asks);
# version 24 added primx_dit_blocks_fold (one foreign call for a forward's blocks): a version-23 build lacks only that one, and the
# host then issues the launches itself (`blocks_call_available()`); version 25 added primx_linear_heads_fold_pair and the kv_* tail of
# PrimxDitForwardFold (a version-24 build ignores the tail: the host then projects K / V itself, `kv_ride_available()`); version 26 added
# primx_linear_f32out_group (`f32out_group_available()`: without it the fold's u / v rows are one launch per site)
_FOLD_ENTRY_POINTS: set = {"primx_linear_f32out", "primx_row_stats", "primx_linear_gate_residual_fold", "primx_linear_heads_fold",
                           "primx_linear_fold"}
_AB_ABI_VERSIONS: tuple = (21, 22, 23, 24, 25)
_fold_available: dict = {}
_blocks_call: dict = {}
_kv_ride: dict = {}
_f32out_group: dict = {}

_lib: Optional[C.CDLL] = None


class PrimxError(RuntimeError):
    """A C-ABI entry point returned a non-zero status."""


def load(path: Optional[str] = None) -> C.CDLL:
    """dlopen the library and attach prototypes.  Raises if it is missing or ABI-mismatched."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP library is not built. Run `python __graft_entry__.py` "
            "(build()) first - there is no PyTorch/CPU fallback for this path."
        )
    if path == LIB_PATH and not os.environ.get("PRIMX_LIB") and not os.environ.get("PRIMX_SKIP_FRESH_CHECK"):
        # the binary must have been built from exactly the sources next to it (content hashes, csrc/build.py): a
        # stale shipped .so is an error, never silently benchmarked
        import importlib.util
        spec = importlib.util.spec_from_file_location("primx_build", os.path.join(_HERE, "csrc", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if not mod.check_fresh():
            raise RuntimeError(f"{path} does not match the sources in {os.path.dirname(path)} (build_manifest.json): "
                               "run `python __graft_entry__.py` (build()) to rebuild it")
    # make sure the HIP runtime PyTorch uses is the one already mapped (same SONAME libamdhip64.so.7)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - symbol-table checks work without torch
        pass
    lib = C.CDLL(path)
    ab = bool(os.environ.get("PRIMX_LIB"))
    lib.primx_abi_version.restype = C.c_int
    got = lib.primx_abi_version()
    if got != ABI_VERSION and not (ab and got in _AB_ABI_VERSIONS):
        raise RuntimeError(f"libprimx_hip.so ABI {got} != expected {ABI_VERSION}; rebuild it")
    for name, argtypes in SIGNATURES.items():
        if got < 23 and name in _FOLD_ENTRY_POINTS:
            continue                # an older A/B build: its fold entry points (if any) have other argument lists
        if got < 24 and name == "primx_dit_blocks_fold":
            continue
        if got < 25 and name == "primx_linear_heads_fold_pair":
            continue
        if got < 26 and name == "primx_linear_f32out_group":
            continue
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _fold_available[path] = got >= 23
    _blocks_call[path] = got >= 24
    _kv_ride[path] = got >= 25
    _f32out_group[path] = got >= 26
    if path == LIB_PATH:
        _lib = lib
    return lib


def fold_available() -> bool:
    """Does the loaded library carry the LayerNorm-fold entry points of this ABI?  (False only for an older PRIMX_LIB A/B build.)"""
    load()
    return _fold_available.get(LIB_PATH, False)


def blocks_call_available() -> bool:
    """Does the loaded library carry primx_dit_blocks_fold (ABI 24)?"""
    load()
    return _blocks_call.get(LIB_PATH, False)


def kv_ride_available() -> bool:
    """Does the loaded library's primx_dit_blocks_fold project the conditioning K / V itself (ABI 25: riders on the qkv launches)?"""
    load()
    return _kv_ride.get(LIB_PATH, False)


def f32out_group_available() -> bool:
    """Does the loaded library carry primx_linear_f32out_group (ABI 26)?"""
    load()
    return _f32out_group.get(LIB_PATH, False)


def check(status: int, name: str) -> None:
    if status != 0:
        msg = load().primx_last_error()
        raise PrimxError(f"{name} failed ({status}): {msg.decode() if msg else '?'}")
