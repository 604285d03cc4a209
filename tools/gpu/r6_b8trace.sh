#!/bin/bash
# round 6: kernel trace of the batch-8 step (T = 32768) on the current tree
OUT=gpurun_out/${TAG:-r6_b8trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs"
timeout 300 $B --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch8 ms_per_step', d['ms_per_step'], d['repeats_ms_per_step'])
for k in d.get('kernels', [])[:14]: print(k)
"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_b8 -- $B --steps 6 --warmup 2 --no-kernel-events > $OUT/bench_trace_b8.json 2> $OUT/bench_trace_b8.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*_kernel_trace.csv" -delete
grep -A 16 "per (kernel, grid)" $OUT/trace_b8_results_summary.txt | cut -c1-150
