"""The C-ABI library loads and exports every symbol include/primx_hip.h declares (no compute calls:
this runs without a GPU), and the ctypes prototypes agree with the header."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "primx_hip.h")


def header_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(?:int|const char\s*\*)\s+(primx_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",")])
        fns[m.group(1)] = n
    return fns


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    import topia_xl_amd._lib as L
    return L


def test_header_is_parsed():
    fns = header_functions()
    assert len(fns) >= 20 and "primx_attention" in fns and "primx_diffusion_step" in fns


def test_every_declared_symbol_is_exported(lib):
    handle = lib.load()
    for name in header_functions():
        assert hasattr(handle, name), f"{name} declared in primx_hip.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\sT\s+(primx_\w+)", out))
    assert exported == set(header_functions()), exported ^ set(header_functions())


def test_ctypes_prototypes_match_header(lib):
    fns = header_functions()
    assert set(lib.SIGNATURES) == set(fns)
    for name, argtypes in lib.SIGNATURES.items():
        assert len(argtypes) == fns[name], name
    assert lib.load().primx_abi_version() == lib.ABI_VERSION
    assert lib.load().primx_padded_head_dim(72) == 80


def test_argument_validation_without_gpu(lib):
    """Argument checks run on the host before any launch: bad shapes return PRIMX_EINVAL + a message."""
    h = lib.load()
    assert h.primx_attention(None, None, None, None, 1, 1, 1, 128, 1, 64, 72, 1.0, 1, None) == -1
    assert b"null" in h.primx_last_error()
    assert h.primx_layernorm_modulate(1, 1, 1, 0, 1, 1, 4, 4, 101, 1e-6, None, 0, None, 0, None) == -1   # D odd
    assert h.primx_linear(1, 1, None, 1, 4, 4, 70, 1, 0, 1.0, None, 0, None) == -1      # K % 8 != 0
    assert h.primx_linear(1, 1, None, 1, 4, 4, 64, 7, 0, 1.0, None, 0, None) == -1      # bad dtype
    assert h.primx_linear(1, 1, None, 1, 4, 4, 64, 1, 0, 1.0, None, 128, None) == -1    # a prefetch range is (pointer, bytes) or (NULL, 0)
    assert b"prefetch" in h.primx_last_error()
    # the fused gate-residual + LayerNorm entry: a LayerNorm output needs its modulation vectors, sync two words per row block
    assert h.primx_linear_gate_residual_ln(1, 1, None, 1, 0, 1, 256, 1152, 64, 256, None, None, 0, 1, 1e-6, None, 0, 1, None, 0, None) == -1
    assert h.primx_linear_gate_residual_ln(1, 1, None, 1, 0, 1, 256, 1152, 64, 256, 1, 1, 0, 1, 1e-6, 1, 3, 1, None, 0, None) == -1
    assert b"sync" in h.primx_last_error()
    with pytest.raises(lib.PrimxError):
        lib.check(-1, "primx_linear")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "3dtopia-xl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "/root/reference" not in text, f
