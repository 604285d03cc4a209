#!/bin/bash
# weight prefetch carried by the GEMM launches (PRIMX_WPREFETCH=2) against the LayerNorm-carried form (1) and none (0)
OUT=gpurun_out/gpf
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_rowops.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for rep in 1 2 3; do for v in 0 1 2; do
  PRIMX_WPREFETCH=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>$OUT/bench_$v.err
  python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step wpf=$v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
for v in 1 2; do
  PRIMX_WPREFETCH=$v timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o tr_$v -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 10 --warmup 3 --repeats 2 > /dev/null 2> $OUT/tr_$v.err
  db=$(find $OUT -name "tr_${v}_results.db" | head -1); python tools/rocprof_summary.py $db $OUT/tr_${v}_summary.txt > /dev/null; echo "== wpf=$v"; sed -n 3,9p $OUT/tr_${v}_summary.txt | cut -c1-110
done | tee $OUT/trace.txt
find $OUT -name "*.db" -delete
