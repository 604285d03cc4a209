#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -2
for i in 1 2; do
for v in 1 0; do echo "== PRIMX_GEMM_XCD2D=$v"; PRIMX_GEMM_XCD2D=$v ONLY=fc1,qkv,big_fc1 REPS=30 python tools/gemm_bench.py 2>&1 | grep -v amdgpu; done
done
