#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_rowops.py tests/test_hip_dit.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
for e in 1 0 1 0; do PRIMX_LN_XCD=$e timeout 600 python bench.py --no-cpu-baseline --no-parity --no-kernel-events --steps 50 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('LN_XCD=$e', d['ms_per_step'])"; done
