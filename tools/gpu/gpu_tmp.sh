#!/bin/bash
timeout 600 python -m pytest tests/test_raymarch.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
timeout 600 python tools/raymarch_bench.py 2>&1 | grep -v amdgpu | tail -1
