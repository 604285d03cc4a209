#!/bin/bash
timeout 600 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for v in 1 0 1; do echo "BIGHEADS=$v"; PRIMX_GEMM_BIGHEADS=$v timeout 300 python tools/heads_bench.py 2>&1 | grep -v amdgpu | grep "kv_all\|k_only\|v_only"; done
