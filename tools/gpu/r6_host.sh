#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fold.py -m gpu -q --tb=short -p no:cacheprovider -k "blocks_call" 2>&1 | tail -8
timeout 600 python tools/host_bound_check.py 2>&1 | grep -v amdgpu | tail -12
for bc in 1 0 1 0; do
PRIMX_DIT_BLOCKS_CALL=$bc timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('BLOCKS_CALL=$bc ms_per_step', d['ms_per_step'], d['repeats_ms_per_step'])
"
done
