#!/bin/bash
timeout 600 python -m pytest tests/test_hip_e2e.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8
timeout 900 python examples/generate.py 2>&1 | grep -v amdgpu | tail -9
