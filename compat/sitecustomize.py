"""`PYTHONPATH=<repo>/compat python inference.py ...`: Python imports this file at start-up (site module), which installs
the hot-path import finder (primx_shim.py).  Set PRIMX_SHIM=0 to disable without touching PYTHONPATH."""
import os

if os.environ.get("PRIMX_SHIM", "1") != "0":
    import primx_shim

    primx_shim.install()
