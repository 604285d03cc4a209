#!/bin/bash
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
PRIMX_GEMM_PROF=1 ONLY=proj,fc2 REPS=3 timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | grep "gemm144_dma" | awk 'NR%6==5'
for r in 1 2; do
echo "--- new"; REPS=100 ONLY=proj,fc2,qkv,kv timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | tail -4
echo "--- prev"; (cd prev_tree && REPS=100 ONLY=proj,fc2,qkv,kv timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | tail -4)
done
