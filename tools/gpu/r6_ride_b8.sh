#!/bin/bash
# round 6: the fold suite after the rider changes + the batch-8 step (riders only where they fit one round: off at T = 32768) against PRIMX_DIT_KV_RIDE=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fold.py tests/test_hip_fullconfig.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -5
for r in 1 0 1 0; do
PRIMX_DIT_KV_RIDE=$r timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch 8 KV_RIDE=$r ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
"
done
