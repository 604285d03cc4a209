"""Import alias for the package directory ``3dtopia-xl_amd/``.

The directory name required by the project layout is not a valid Python
identifier, so ``import topia_xl_amd`` executes this file, which loads the
package from ``3dtopia-xl_amd/`` under the name ``topia_xl_amd`` and replaces
itself in ``sys.modules``.  Sub-modules resolve normally afterwards
(``import topia_xl_amd.dit``).
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "3dtopia-xl_amd")
_spec = _ilu.spec_from_file_location(
    "topia_xl_amd",
    _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir],
)
_mod = _ilu.module_from_spec(_spec)
_sys.modules["topia_xl_amd"] = _mod
_spec.loader.exec_module(_mod)
