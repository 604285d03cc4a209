#!/bin/bash
# unconditional half's K / V rows projected once (PRIMX_NULL_KV_DEDUP, primx_attention_bcast): suite, full-configuration parity, step A/B
OUT=gpurun_out/dedup
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_attention.py tests/test_hip_dit.py tests/test_hip_e2e.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
timeout 900 python -m pytest tests/test_hip_fullconfig.py -x -q -s 2>&1 | grep -E "rel-L2|passed|failed|Error" | cut -c1-250 | tee $OUT/full.txt
for rep in 1 2 3; do for v in 0 1; do
  PRIMX_NULL_KV_DEDUP=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>$OUT/bench_$v.err
  python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step dedup=$v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
