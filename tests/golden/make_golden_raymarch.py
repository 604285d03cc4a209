"""Pin the ray marcher to the reference's OWN executable PyTorch formulation (build container only):

    python tests/golden/make_golden_raymarch.py

The CUDA extension cannot be built here, but the two `gradcheck()` self-tests the reference ships contain a pure-PyTorch
statement of what the kernels compute, next to the CUDA call they compare it with:

  * dva/mvp/extensions/mvpraymarch/mvpraymarch.py:301-461 - seeded scene (torch.manual_seed(1112); N=2, H=W=65, K=64)
    and the PyTorch march loop (lines 391-459: per step, per primitive: SRT transform by bmm, fade, F.grid_sample of
    the template, additive alpha accumulation with clamp);
  * dva/mvp/extensions/utils/utils.py:73-148 - seeded cameras (torch.manual_seed(1113)) and the PyTorch ray directions /
    unit-cube t-range.

This script runs THOSE LINES, unmodified, at generation time: it imports the reference modules (with an empty stand-in
for the compiled `mvpraymarchlib` / `utilslib` they import at the top), takes `inspect.getsource(gradcheck)`, cuts it at
the end of the "run pytorch version" section, retargets the device strings "cuda" -> "cpu", drops the
`torch.cuda.synchronize()` calls and executes it.  Nothing of the reference is copied into the repository - only the
inputs it generated and the outputs it computed are stored (tests/golden/raymarch.npz).  One size is changed: the
template resolution M = 32 -> 8 (the resolution of the shipped PrimX payload, and a 64x smaller fixture).
"""
from __future__ import annotations

import importlib.util
import inspect
import os
import re
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/dva/mvp/extensions"


def _load(path: str, stub: str):
    sys.modules.setdefault(stub, types.ModuleType(stub))       # the compiled extension the module imports at its top
    spec = importlib.util.spec_from_file_location("ref_" + stub, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _pytorch_section(fn, stop_marker: str, patches=()):
    """Source of `fn` from its first body line to the line holding `stop_marker` (inclusive), on the CPU."""
    lines = inspect.getsource(fn).splitlines()
    body_start = next(i for i, l in enumerate(lines) if l.rstrip().endswith("):")) + 1
    stop = next(i for i, l in enumerate(lines) if stop_marker in l)
    src = textwrap.dedent("\n".join(lines[body_start:stop + 1]))
    src = src.replace('"cuda"', '"cpu"').replace("torch.cuda.synchronize()", "pass")
    for a, b in patches:
        assert re.search(a, src), a
        src = re.sub(a, b, src, count=1)
    return src


def main():
    out = {}
    # ---------------- march loop (mvpraymarch.py gradcheck, PyTorch half)
    mod = _load(os.path.join(REF, "mvpraymarch", "mvpraymarch.py"), "mvpraymarchlib")
    sig = inspect.signature(mod.gradcheck)
    env = {k: v.default for k, v in sig.parameters.items()}     # usebvh, dowarp=False, fadescale=8, fadeexp=8, accum=0, algo=0 ...
    env.update(vars(mod))
    src = _pytorch_section(mod.gradcheck, "sample0 = rayrgba", patches=[(r"M = 32", "M = 8")])
    exec(compile(src, "<reference gradcheck: pytorch version>", "exec"), env)
    g = lambda k: env[k].detach().float().contiguous().numpy()
    out.update(march_raypos=g("_raypos"), march_raydir=g("_raydir"), march_tminmax=g("_tminmax"),
               march_stepsize=np.float32(env["_stepsize"]), march_template=g("template"),      # [N, K, 4, M, M, M], post-softplus
               march_primpos=g("primpos"), march_primrot=g("primrot"), march_primscale=g("primscale"),
               march_fadescale=np.float32(env["fadescale"]), march_fadeexp=np.float32(env["fadeexp"]),
               march_rgba=g("sample0"), march_steps=np.int64(env["step"]))
    print("march loop:", env["step"], "steps; alpha range", float(env["sample0"][..., 3].min()), float(env["sample0"][..., 3].max()))
    # ---------------- ray directions (utils.py gradcheck, PyTorch half)
    umod = _load(os.path.join(REF, "utils", "utils.py"), "utilslib")
    env = dict(vars(umod))
    src = _pytorch_section(umod.gradcheck, "tminmax = torch.stack([tmin, tmax], dim=-1)")
    exec(compile(src, "<reference utils gradcheck: pytorch version>", "exec"), env)
    g = lambda k: env[k].detach().float().contiguous().numpy()
    out.update(rays_viewpos=g("_viewpos"), rays_viewrot=g("_viewrot"), rays_focal=g("_focal"), rays_princpt=g("_princpt"),
               rays_pixelcoords=g("_pixelcoords"), rays_volradius=np.float32(env["volradius"]),
               rays_raydir=g("raydir"), rays_tminmax=g("tminmax"))
    path = os.path.join(HERE, "raymarch.npz")
    np.savez_compressed(path, **out)
    print("raymarch.npz", os.path.getsize(path))


if __name__ == "__main__":
    main()
