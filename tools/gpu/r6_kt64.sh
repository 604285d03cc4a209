#!/bin/bash
# round 6: the 256x288 kernel with 128-byte row segments (KT = 64) against its 64-byte form (PRIMX_GEMM_KT32=1): tests, micro-benchmarks, steps
OUT=gpurun_out/r6_kt64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/tests.log; tail -5 $OUT/tests.log
for kt in 0 1 0 1; do
  echo "== PRIMX_GEMM_KT32=$kt" | tee -a $OUT/bench.txt
  PRIMX_GEMM_KT32=$kt timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP | tee -a $OUT/bench.txt
done
for kt in 0 1; do
  echo "== heads PRIMX_GEMM_KT32=$kt" | tee -a $OUT/bench.txt
  PRIMX_GEMM_KT32=$kt timeout 300 python tools/heads_bench.py 2>&1 | grep -v amdgpu | tail -12 | tee -a $OUT/bench.txt
done
for kt in 0 1; do
  echo "== step PRIMX_GEMM_KT32=$kt" | tee -a $OUT/bench.txt
  PRIMX_GEMM_KT32=$kt timeout 600 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('ms_per_step', d['ms_per_step'], d['repeats_ms_per_step'], 'batch8', d.get('batch8',{}).get('ms_per_step'), 'bf16', d.get('bf16',{}).get('ms_per_step'))
print({k['kernel'][:60]: round(k['avg_us'],1) for k in d.get('kernels',[])[:12]} if isinstance(d.get('kernels'), list) else '')
" | tee -a $OUT/bench.txt
done
