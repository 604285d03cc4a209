#!/bin/bash
# round 4, LayerNorm fold at a large batch (256 x 288 producer / Linear consumer): operator parity, the batch-8 golden trajectory
# and configs[3] end to end with the fold, the batch-8 step with and without it on the same box
OUT=gpurun_out/r4_fold4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_hip_fold.py tests/test_hip_gemm.py -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "fold + gemm suites: $(tail -1 $OUT/tests.log)"
grep -n "FAILED\|Error" $OUT/tests.log | head -20
timeout 400 python -m pytest tests/test_hip_fullconfig.py -m gpu -q -s -p no:cacheprovider -k "batch8 or configs3 or batched" > $OUT/tests_fullconfig.log 2>&1; echo "full-config (batch 8, configs[3]): $(tail -1 $OUT/tests_fullconfig.log)"
grep -n "FAILED\|rel-L2" $OUT/tests_fullconfig.log | cut -c1-260 | head -12
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --batch 8 --steps 6 --warmup 2"
for f in 0 1 0 1; do
  PRIMX_DIT_FOLD=$f timeout 200 python bench.py $B > $OUT/bench_b8_fold$f.json 2> $OUT/bench_b8_fold$f.err
  echo "batch 8 fold=$f: $(python -c "import json;r=json.load(open('$OUT/bench_b8_fold$f.json'));print(round(r['ms_per_step'],3))" 2>&1 | tail -1)"
done
