#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_attention.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -6
for f in 1 0 1 0; do
  echo "== PRIMX_ATTN_FUSED=$f"
  PRIMX_ATTN_FUSED=$f timeout 200 python tools/attn_bench.py 2>&1 | grep -v amdgpu
done
