#!/bin/bash
OUT=gpurun_out/wpf2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_dit.py tests/test_hip_e2e.py tests/test_hip_rowops.py tests/test_hip_dinov2.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "pytest exit $?"; tail -2 $OUT/tests.log
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', ['%.3f' % x for x in d['repeats_ms_per_step']])"; }
for rep in 1 2 3; do
  PRIMX_WPREFETCH=0 run "no prefetch   "
  run "LN-carried prefetch"
done | tee $OUT/steps.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 6 --warmup 2 --repeats 1 > $OUT/bench.json 2> $OUT/bench.err
db=$(find $OUT -name "tr_results.db" | head -1); python tools/rocprof_summary.py $db $OUT/summary.txt > /dev/null; sed -n 3,9p $OUT/summary.txt | cut -c1-110
find $OUT -name "*.db" -delete
