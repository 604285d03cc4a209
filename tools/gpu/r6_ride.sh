#!/bin/bash
# round 6: the to_k / to_v projection riding on the qkv launches (primx_linear_heads_fold_pair, ABI 25) - tests, then the configs[1] step with
# the riders on / off, alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fold.py -m gpu -q --tb=short -p no:cacheprovider -k "rider or pair or kv_ride or blocks_call" 2>&1 | tail -15
for r in 1 0 1 0; do
PRIMX_DIT_KV_RIDE=$r timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('KV_RIDE=$r ms_per_step', round(d['ms_per_step'],4), [round(x,4) for x in d['repeats_ms_per_step']])
"
done
