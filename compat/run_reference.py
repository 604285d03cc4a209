#!/usr/bin/env python
"""Launcher for an unmodified reference script with the MI355X hot path switched in:

    cd /path/to/3DTopia-XL && python /path/to/this/repo/compat/run_reference.py inference.py configs/inference_dit.yml

Installs the import finder of primx_shim.py, then runs the script exactly as `python script args...` would
(`__main__`, sys.argv, script directory at sys.path[0])."""
import os
import runpy
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import primx_shim  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit("usage: run_reference.py <script.py> [args...]")
    sys.path.pop(0)
    script = os.path.abspath(sys.argv[1])
    sys.argv = sys.argv[1:]
    sys.path.insert(0, os.path.dirname(script))
    primx_shim.install()
    runpy.run_path(script, run_name="__main__")
