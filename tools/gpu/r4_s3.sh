#!/bin/bash
# round 4, session 3: how the in-kernel wait READS the arrival counter (PRIMX_LN_MODE bits 8-9): agent-scope load | returning atomic |
# L1 invalidate + plain load; same box, against the two-launch route.  Plus the GEMM / attention / VAE suites on the pruned sources.
OUT=gpurun_out/r4_s3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5"
show() { python -c "import json,sys;r=json.load(open(sys.argv[1]));print('%.3f' % r['ms_per_step'], ['%.3f' % v for v in r['repeats_ms_per_step']], r['ln_in_gemm_tail'])" $1; }
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_attention.py tests/test_hip_vae.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "suite: $(tail -1 $OUT/tests.log)"
for mode in 45 301 557 299 555; do
  PRIMX_LN_MODE=$mode timeout 300 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider -k "layernorm" > $OUT/tests_$mode.log 2>&1; echo "mode $mode tests: $(tail -1 $OUT/tests_$mode.log)"
  PRIMX_LN_MODE=$mode timeout 300 $B > $OUT/mode_$mode.json 2>> $OUT/err.txt; echo "mode $mode step: $(show $OUT/mode_$mode.json)"
done
PRIMX_DIT_FUSE_LN=0 timeout 300 $B > $OUT/unfused.json 2>> $OUT/err.txt; echo "two launches: $(show $OUT/unfused.json)"
PRIMX_LN_MODE=557 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -size +20M -delete
head -8 $(find $OUT -name "*_summary.txt" | head -1) | cut -c1-160
