#!/bin/bash
# the switch combinations of r6_switches4.sh that stopped at a test guard, after the guards were generalised (tests/test_hip_fold.py only: the other suites passed)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "PRIMX_GEMM_KT64_MIN=257" "PRIMX_WPREFETCH=1" "PRIMX_CFG_STREAMS=1" "PRIMX_DIT_FUSE_LN=0" "PRIMX_DIT_LN_TAIL=1" "PRIMX_NONE=1"; do
  echo "== $kv"
  env $kv timeout 900 python -m pytest tests/test_hip_fold.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | grep -v "^\.\|amdgpu" | tail -5 | cut -c1-220
done
