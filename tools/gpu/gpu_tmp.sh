#!/bin/bash
timeout 900 python -m pytest tests/test_hip_attention.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
PRIMX_ATTN_PROF=1 REPS=2 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | grep "segment" | awk 'NR==1'
echo "--- new"; REPS=50 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4
echo "--- prev"; (cd prev_tree && REPS=50 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4)
echo "--- new"; REPS=50 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4
