#!/bin/bash
# the switch combinations that failed a TEST GUARD in r6_switches2.sh, all tests (no -x), after the guards were fixed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "PRIMX_DIT_BLOCKS_CALL=0" "PRIMX_DIT_KV_RIDE=0 PRIMX_DIT_BLOCKS_CALL=0" "PRIMX_GEMM_KT32=1" "PRIMX_NULL_KV_DEDUP=0" "PRIMX_PLAN_TIMESTEPS=0"; do
  echo "== $kv"
  env $kv timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_fullconfig.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | grep -v "^\.\|amdgpu" | tail -14 | cut -c1-250
done
