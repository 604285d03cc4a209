"""Zero-edit drop-in for a 3DTopia-XL checkout: only the modules of the hot path are replaced, everything else of the
reference keeps importing from the reference.

The reference imports, side by side (inference.py:12-21, app.py, dva/io.py:14-29 `class_name` strings):

    dva.ray_marcher  dva.io  dva.utils  dva.visualize  models.diffusion  models.dit_crossattn  models.vae3d_dib
    models.conditioner.*  models.utils  models.primsdf  utils.*

`models` and `dva` are regular packages of the reference, and `python inference.py` puts the reference root at
sys.path[0] - in front of PYTHONPATH - so neither a path-order trick nor a second `models/` package can work (a
second package would shadow `models.utils`, `dva.io`, ...).  Instead `install()` puts ONE finder at the head of
`sys.meta_path` that answers for exactly the fully-qualified names in OVERRIDES and returns None for everything else:
the parent packages and all other sub-modules stay the reference's own.

Each override is a small re-export file in `overrides/` executed under the reference's module name, so
`models.dit_crossattn.DiT is topia_xl_amd.dit.DiT`.  `models.diffusion` stays a package whose search path is the
reference's own directory: `models.diffusion.gaussian_diffusion` etc. remain importable.

Two ways to activate it without editing the reference:
    PYTHONPATH=<repo>/compat python inference.py ...        (compat/sitecustomize.py calls install())
    python <repo>/compat/run_reference.py inference.py ...  (launcher: install(), then runpy)
"""
from __future__ import annotations

import importlib.abc
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

# reference module name -> (override source in overrides/, keep the reference directory as the package search path)
OVERRIDES = {
    "models.dit_crossattn": ("models_dit_crossattn.py", False),    # DiT, DiTAdditivePosEmb (configs/inference_dit.yml:53)
    "models.vae3d_dib": ("models_vae3d_dib.py", False),            # VAE (configs/inference_dit.yml:32)
    "models.attention": ("models_attention.py", False),            # MemEffAttention / MemEffCrossAttention (no xFormers)
    "models.diffusion": ("models_diffusion.py", True),             # create_diffusion (inference.py:21)
    "models.primsdf": ("models_primsdf.py", False),                # PrimSDF (inference.py:354-369)
    "dva.ray_marcher": ("dva_ray_marcher.py", False),              # RayMarcher (inference.py:12, dva/visualize.py:7)
}


class _HotPathFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        entry = OVERRIDES.get(fullname)
        if entry is None:
            return None
        src, keep_pkg_path = entry
        search = None
        if keep_pkg_path:
            leaf = fullname.rsplit(".", 1)[1]
            search = [os.path.join(p, leaf) for p in (path or []) if os.path.isdir(os.path.join(p, leaf))]
        return importlib.util.spec_from_file_location(fullname, os.path.join(HERE, "overrides", src),
                                                      submodule_search_locations=search)


def install() -> None:
    """Idempotent.  Makes `topia_xl_amd` importable and routes the OVERRIDES names to it."""
    if REPO not in sys.path:
        sys.path.append(REPO)          # appended: never in front of the reference's own top-level names (`utils`, ...)
    if not any(isinstance(f, _HotPathFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _HotPathFinder())
    for name in OVERRIDES:             # a reference copy imported before install() would otherwise stay in effect
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, "__primx_override__", False):
            del sys.modules[name]
