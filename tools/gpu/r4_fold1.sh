#!/bin/bash
# round 4, LayerNorm fold, first device session: operator / model parity of the fold kernels, then the headline step with and
# without PRIMX_DIT_FOLD on the same box, then a kernel trace of the folded step
OUT=gpurun_out/r4_fold1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 330 python -m pytest tests/test_hip_fold.py -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "fold suite: $(tail -1 $OUT/tests.log)"
grep -n "folded\|FAILED\|Error\|assert" $OUT/tests.log | head -60
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5"
for f in 0 1 0 1; do
  PRIMX_DIT_FOLD=$f timeout 120 python bench.py $B > $OUT/bench_fold$f.json 2> $OUT/bench_fold$f.err
  echo "fold=$f: $(python -c "import json;r=json.load(open('$OUT/bench_fold$f.json'));print(r['ms_per_step'])" 2>&1 | tail -1)"
done
PRIMX_DIT_FOLD=1 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
head -24 $(find $OUT -name "*_summary.txt" | head -1) | cut -c1-160
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete
