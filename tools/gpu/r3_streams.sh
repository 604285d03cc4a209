#!/bin/bash
OUT=gpurun_out/st
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_CFG_STREAMS=1 PRIMX_GEMM_BIG_MIN=112 PRIMX_GEMM_BIGHEADS_MIN=96 timeout 600 python -m pytest tests/test_hip_dit.py tests/test_hip_fullconfig.py tests/test_hip_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -k "not ddim25" > $OUT/tests.log 2>&1; echo "pytest (streams) exit $?"; tail -3 $OUT/tests.log
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', ['%.3f' % x for x in d['repeats_ms_per_step']])"; }
for rep in 1 2; do
  run "single stream          "
  PRIMX_CFG_STREAMS=1 run "streams, thresholds std"
  PRIMX_CFG_STREAMS=1 PRIMX_GEMM_BIG_MIN=112 run "streams, big fc1       "
  PRIMX_CFG_STREAMS=1 PRIMX_GEMM_BIG_MIN=112 PRIMX_GEMM_BIGHEADS_MIN=96 run "streams, big fc1 + qkv "
done | tee $OUT/steps.txt
