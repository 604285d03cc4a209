#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_D192=2 timeout 600 python tools/d192_check.py 2>&1 | grep -v amdgpu | tail -16
for d in 2 0 2 0; do
  echo "== PRIMX_GEMM_D192=$d"
  PRIMX_GEMM_D192=$d ONLY=32768 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
done
