#!/bin/bash
OUT=gpurun_out/touch
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 0 1; do
  PRIMX_TOUCH_W=$t timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr_$t -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 6 --warmup 2 --repeats 1 > $OUT/bench_$t.json 2> $OUT/bench_$t.err
  db=$(find $OUT -name "tr_${t}_results.db" | head -1); python tools/rocprof_summary.py $db $OUT/tr_${t}_summary.txt > /dev/null; echo "== touch=$t"; grep -A7 "per (kernel, grid)" $OUT/tr_${t}_summary.txt | cut -c1-110
done
find $OUT -name "*.db" -delete
