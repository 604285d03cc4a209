#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5
for kt in 0 1 0; do
  echo "== PRIMX_GEMM_KT32=$kt"
  PRIMX_GEMM_KT32=$kt ONLY=32768 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
done
PRIMX_GEMM_PROF=1 REPS=2 ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288q_dma<" | awk 'NR%5==0' | cut -c1-330
