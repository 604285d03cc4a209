#!/bin/bash
# end of round 6: the remaining kernel-selection switches of DESIGN.md section 8 on the final tree (same suites as r6_switches2.sh)
OUT=gpurun_out/r6_switches4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "PRIMX_GEMM_KT64_MIN=257" "PRIMX_GEMM_LOADER=0" "PRIMX_GEMM_NOGEMV=1" "PRIMX_WPREFETCH=0" "PRIMX_WPREFETCH=1" "PRIMX_CFG_STREAMS=1" "PRIMX_DIT_FUSE_LN=0" "PRIMX_DIT_LN_TAIL=1" "PRIMX_LN_FUSE=0" "PRIMX_GEMM_XCD2D=0" "PRIMX_PLAN_TIMESTEPS=0"; do
  echo "== $kv" | tee -a $OUT/matrix.txt
  env $kv timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_fullconfig.py -m gpu -q --tb=line -p no:cacheprovider 2>&1 | grep -v "^\.\|amdgpu" | tail -6 | cut -c1-220 | tee -a $OUT/matrix.txt
done
