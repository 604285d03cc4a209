#!/bin/bash
# end of round 6 (ABI 26 tree): the GEMM / fold / DiT / full-configuration suites under the switches the round added late - the heads epilogues'
# ring, the K / V riders, the grouped u / v launch - alone and combined with the older ones they interact with
OUT=gpurun_out/r6_switches2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in "PRIMX_GEMM_HEADS_KT32=1" "PRIMX_DIT_KV_RIDE=0" "PRIMX_UV_GROUP=0" "PRIMX_DIT_KV_RIDE=0 PRIMX_DIT_BLOCKS_CALL=0" "PRIMX_DIT_BLOCKS_CALL=0" "PRIMX_GEMM_KT32=1" "PRIMX_GEMM_BIGHEADS_MIN=0" "PRIMX_GEMM_NOBIG=1" "PRIMX_NULL_KV_DEDUP=0" "PRIMX_DIT_FOLD=0" "PRIMX_PLAN_TIMESTEPS=0"; do
  echo "== $kv" | tee -a $OUT/matrix.txt
  env $kv timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_fullconfig.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/matrix.txt
done
