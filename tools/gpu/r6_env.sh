#!/bin/bash
# round 6: runtime knobs that touch the launch path (kernel arguments in device memory; interrupt vs polling completion signals) - the configs[1] step, same box, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 ms_per_step', round(d['ms_per_step'],4), [round(x,4) for x in d['repeats_ms_per_step']])
"; }
for i in 1 2; do
run default
HIP_FORCE_DEV_KERNARG=1 run DEV_KERNARG=1
HIP_FORCE_DEV_KERNARG=0 run DEV_KERNARG=0
HSA_ENABLE_INTERRUPT=0 run HSA_ENABLE_INTERRUPT=0
done
