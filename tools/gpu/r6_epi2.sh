#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5
ONLY=32768 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
B="python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
for kt in 0 1 0; do
PRIMX_GEMM_KT32=$kt timeout 300 $B --steps 6 --warmup 2 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('KT32=$kt batch8 ms_per_step', d['ms_per_step'], d['repeats_ms_per_step'])
"
done
PRIMX_GEMM_PROF=1 REPS=2 ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288q_dma<" | awk 'NR%5==0' | cut -c1-330
