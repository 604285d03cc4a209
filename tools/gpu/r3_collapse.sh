#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_dit.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline', d['ms_per_step']); r=d['with_reuse_cond_kv']; print('reuse', r['ms_per_step'], 'reuse+collapse', r['plus_null_cross_attention_collapse']['ms_per_step'])"
