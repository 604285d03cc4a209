#!/bin/bash
# round 4, LayerNorm fold, later device sessions: the fold suite, the full-configuration golden tests WITH the fold, the headline
# step with and without it on the same box, a kernel trace of the folded step
OUT=gpurun_out/r4_fold2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 330 python -m pytest tests/test_hip_fold.py -m gpu -q -s -p no:cacheprovider > $OUT/tests.log 2>&1; echo "fold suite: $(tail -1 $OUT/tests.log)"
grep -n "FAILED\|Error" $OUT/tests.log | head -20
PRIMX_DIT_FOLD=1 timeout 400 python -m pytest tests/test_hip_fullconfig.py tests/test_hip_dit.py tests/test_hip_e2e.py -m gpu -q -s -p no:cacheprovider > $OUT/tests_fullconfig_fold.log 2>&1; echo "full-config + dit + e2e suites with the fold: $(tail -1 $OUT/tests_fullconfig_fold.log)"
grep -n "FAILED\|rel\|traj" $OUT/tests_fullconfig_fold.log | head -40
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5"
for f in 0 1 0 1; do
  PRIMX_DIT_FOLD=$f timeout 120 python bench.py $B > $OUT/bench_fold$f.json 2> $OUT/bench_fold$f.err
  echo "fold=$f: $(python -c "import json;r=json.load(open('$OUT/bench_fold$f.json'));print(r['ms_per_step'])" 2>&1 | tail -1)"
done
PRIMX_DIT_FOLD=1 timeout 200 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
head -12 $(find $OUT -name "*_summary.txt" | head -1) | cut -c1-130
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete
