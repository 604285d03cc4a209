#!/bin/bash
# round 6: the fold's u / v rows from ONE grouped launch (primx_linear_f32out_group, ABI 26) - tests, then the configs[1] step with / without, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_fold.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
for g in 1 0 1 0; do
PRIMX_UV_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('UV_GROUP=$g ms_per_step', round(d['ms_per_step'],4), [round(x,4) for x in d['repeats_ms_per_step']])
for k,v in sorted(d['kernels'].items()):
    if 'f32out' in k or ', 9>' in k: print('    ', k, round(v['ms_per_step'],4), 'ms/step', round(v['launches_per_step'],2), 'launches/step', round(1e3*v['ms_per_step']/v['launches_per_step'],1), 'us each')
"
done
