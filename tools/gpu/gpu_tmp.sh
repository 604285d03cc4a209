#!/bin/bash
L=$PWD/3dtopia-xl_amd/csrc
PRIMX_LIB=$L/libprimx_q1.so timeout 600 python -m pytest tests/test_hip_attention.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for v in q0 q1 q0 q1; do echo "--- $v"; PRIMX_LIB=$L/libprimx_$v.so REPS=50 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4 | head -3; done
