#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REPS=40 timeout 300 python tools/attn_bench.py 2>&1 | grep -E "self_|cross_"
timeout 900 python -m pytest tests/test_hip_attention.py tests/test_hip_dit.py tests/test_hip_fullconfig.py tests/test_hip_vae.py tests/test_hip_dinov2.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -4
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline --no-kernel-events --steps 50 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('parity'))"; done
