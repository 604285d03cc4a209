#!/bin/bash
# round 4, session 2: arrival protocol / fence variants of the fused gate-residual + LayerNorm tail (PRIMX_LN_MODE), each checked
# for bit-identity against the two-launch route and timed on the step, same box
OUT=gpurun_out/r4_s2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5"
show() { python -c "import json,sys;r=json.load(open(sys.argv[1]));print('%.3f' % r['ms_per_step'], ['%.3f' % v for v in r['repeats_ms_per_step']], r['ln_in_gemm_tail'])" $1; }
for mode in 43 45 42 40 35 139; do
  PRIMX_LN_MODE=$mode timeout 300 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider -k "layernorm" > $OUT/tests_$mode.log 2>&1; echo "mode $mode tests: $(tail -1 $OUT/tests_$mode.log)"
  PRIMX_LN_MODE=$mode timeout 300 $B > $OUT/mode_$mode.json 2>> $OUT/err.txt; echo "mode $mode step: $(show $OUT/mode_$mode.json)"
done
PRIMX_DIT_FUSE_LN=0 timeout 300 $B > $OUT/unfused.json 2>> $OUT/err.txt; echo "two launches: $(show $OUT/unfused.json)"
timeout 300 $B > $OUT/default.json 2>> $OUT/err.txt; echo "default again: $(show $OUT/default.json)"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -size +20M -delete
head -12 $(find $OUT -name "*_summary.txt" | head -1) | cut -c1-160
