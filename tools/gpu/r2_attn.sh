#!/bin/bash
mkdir -p gpurun_out/r2c
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0; do echo "=== PRIMX_ATTN_V2=$v"; PRIMX_ATTN_V2=$v timeout 300 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2c/attn_ab.txt
PRIMX_ATTN_PROF=1 REPS=1 timeout 300 python tools/attn_bench.py 2>&1 | grep -E "attn segment profile" | head -2
timeout 900 python -m pytest tests/test_hip_attention.py tests/test_hip_dit.py tests/test_hip_fullconfig.py tests/test_hip_vae.py tests/test_hip_dinov2.py -m gpu -q --tb=short -p no:cacheprovider -x -s 2>&1 | grep -E "rel-L2|passed|failed|Error|error|assert" | tail -12
