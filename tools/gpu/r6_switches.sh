#!/bin/bash
# rounds 5 - 6: the GEMM / fold / DiT / VAE suites under every kernel-selection switch of DESIGN.md section 8 (ABI 24 tree)
OUT=gpurun_out/r6_switches
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in PRIMX_NONE=1 PRIMX_GEMM_KT32=1 PRIMX_GEMM_KT64_MIN=257 PRIMX_DIT_BLOCKS_CALL=0 PRIMX_GEMM_LOADER=0 PRIMX_GEMM_NOBIG=1 PRIMX_GEMM_BIGHEADS_MIN=0 PRIMX_GEMM_NOGEMV=1 PRIMX_WPREFETCH=0 PRIMX_WPREFETCH=1 PRIMX_NULL_KV_DEDUP=0 PRIMX_CFG_STREAMS=1 PRIMX_DIT_FUSE_LN=0 PRIMX_DIT_FOLD=0 PRIMX_DIT_LN_TAIL=1 PRIMX_LN_FUSE=0 PRIMX_GEMM_XCD2D=0 PRIMX_PLAN_TIMESTEPS=0 PRIMX_CONV_REG=0; do
  echo "== $kv" | tee -a $OUT/matrix.txt
  env $kv timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_vae.py tests/test_hip_fullconfig.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee -a $OUT/matrix.txt
done
