#!/bin/bash
# round 5, call 2: the persistent-pass GEMM kernel (gemm144pp_dma_kernel) - its edge tests, the micro-benchmark at the large-batch shapes
# against the kernels it replaces (PRIMX_GEMM_PP_ROUNDS=0), and the batch-8 step with / without the LayerNorm fold
OUT=gpurun_out/r5_pp1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_rowops.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/tests.log; tail -15 $OUT/tests.log
for i in 1 2; do
  echo "== PRIMX_GEMM_PP_ROUNDS=0" | tee -a $OUT/gemm_big.txt
  PRIMX_GEMM_PP_ROUNDS=0 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_big.txt
  echo "== persistent" | tee -a $OUT/gemm_big.txt
  timeout 300 python tools/gemm_bench_big.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_big.txt
done
B8="--batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 6 --warmup 2"
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d.get("repeats_ms_per_step"))'
timeout 300 python bench.py $B8 2>/dev/null | python -c "$J" "b8 fold pp(default)" | tee -a $OUT/b8.txt
PRIMX_DIT_FOLD=0 timeout 300 python bench.py $B8 2>/dev/null | python -c "$J" "b8 nofold pp" | tee -a $OUT/b8.txt
PRIMX_DIT_FOLD=0 PRIMX_GEMM_PP_ROUNDS=0 timeout 300 python bench.py $B8 2>/dev/null | python -c "$J" "b8 nofold nopp" | tee -a $OUT/b8.txt
PRIMX_GEMM_PP_ROUNDS=0 timeout 300 python bench.py $B8 2>/dev/null | python -c "$J" "b8 fold nopp" | tee -a $OUT/b8.txt
