#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for mu in 1 0 1 0; do
  echo "== PRIMX_GEMM_MULTI=$mu"
  PRIMX_GEMM_MULTI=$mu ONLY=32768 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
done
B="python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
for mu in 1 0 1 0; do
PRIMX_GEMM_MULTI=$mu timeout 300 $B --steps 6 --warmup 2 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('MULTI=$mu batch8 ms_per_step', d['ms_per_step'], d['repeats_ms_per_step'])
"
done
