"""Build libprimx_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python 3dtopia-xl_amd/csrc/build.py [--force]

One translation unit per .hip file.  Freshness is decided by CONTENT, not mtimes: every object is keyed on the
sha256 of (its source, the shared headers, the flags) and the library on the hashes of its objects, recorded in
``build_manifest.json`` next to the library (git-ignored like the binaries, shipped to the GPU box with them).
``check_fresh()`` recomputes the hashes without needing hipcc, so a GPU box that only received prebuilt binaries can
still prove that they match the sources it received (``_lib.load`` calls it; a stale ``.so`` raises instead of being
benchmarked).  Linked against the HIP runtime (SONAME libamdhip64.so.7 - at run time the copy PyTorch already mapped
is the one that binds).
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["rowops.hip", "gemm.hip", "attention.hip", "vae.hip", "primsdf.hip", "raymarch.hip", "fp32.hip", "conv3.hip", "conv3s8.hip", "conv3s8c32.hip", "convt.hip", "dit_host.hip"]
HEADERS = ["common.h", "ln_row.h", "gemm288q_body.inc", os.path.join("..", "..", "include", "primx_hip.h")]
LIB = os.path.join(HERE, "libprimx_hip.so")
MANIFEST = os.path.join(HERE, "build_manifest.json")
# -amdgpu-kernarg-preload-count: the leading scalar kernel arguments arrive in SGPRs at wave launch instead of through an s_load of
# the freshly written kernarg segment (a scalar-cache miss in front of every kernel's first address): -0.25 us per launch measured
# on the LayerNorm kernel (rocprofv3, 3910 launches, 8.35 -> 8.10 us); the GEMM kernels name their prologue's fields as leading
# scalars for it (gemm.hip, PRIMX_GEMM_PARAMS).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]
# Per-file additions (none at present; the hook stays for per-kernel codegen options).
EXTRA_FLAGS: dict = {}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _sha(paths, extra: str = "") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def source_hashes() -> dict:
    """{object name: hash of (source, headers, flags)} + {"lib": hash over all of them}."""
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    out = {}
    for src in SOURCES:
        out[src.replace(".hip", ".o")] = _sha([os.path.join(HERE, src)] + hdrs, " ".join(FLAGS + EXTRA_FLAGS.get(src, [])))
    out["lib"] = hashlib.sha256("".join(out[k] for k in sorted(out)).encode()).hexdigest()
    return out


def _manifest() -> dict:
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def check_fresh() -> bool:
    """True when the library on disk was built from exactly the sources on disk (no compiler needed)."""
    if not os.path.exists(LIB):
        return False
    return _manifest().get("lib") == source_hashes()["lib"]


def build(force: bool = False, verbose: bool = True) -> str:
    want, have = source_hashes(), _manifest()
    objs, jobs, names = [], [], []
    for src in SOURCES:
        name = src.replace(".hip", ".o")
        o = os.path.join(HERE, name)
        objs.append(o)
        if force or not os.path.exists(o) or have.get(name) != want[name]:
            jobs.append((name, [_hipcc(), *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(HERE, src), "-o", o]))
            names.append(name)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)

    done = dict(have)
    if jobs:
        done.pop("lib", None)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(lambda j: run(j[1]), jobs))
        for name in names:
            done[name] = want[name]
    if force or jobs or not os.path.exists(LIB) or have.get("lib") != want["lib"]:
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
        done["lib"] = want["lib"]
        with open(MANIFEST, "w") as f:
            json.dump({k: done[k] for k in sorted(done) if k in want}, f, indent=1)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
