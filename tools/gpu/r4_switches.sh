#!/bin/bash
# the switch matrix of DESIGN.md section 8 on the final round-4 tree: the GEMM / DiT / VAE parity tests under each alternate path
OUT=gpurun_out/r4_switches
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in PRIMX_GEMM_LOADER=0 PRIMX_GEMM_NOBIG=1 PRIMX_GEMM_P2=0 PRIMX_GEMM_BIGHEADS_MIN=0 PRIMX_GEMM_NOGEMV=1 PRIMX_WPREFETCH=0 PRIMX_WPREFETCH=1 PRIMX_NULL_KV_DEDUP=0 PRIMX_CFG_STREAMS=1 PRIMX_DIT_FUSE_LN=0 PRIMX_DIT_LN_TAIL=1 PRIMX_LN_FUSE=0 PRIMX_GEMM_XCD2D=0 PRIMX_CONV_REG=0; do
  echo "== $kv: $(env $kv timeout 300 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_vae.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1)" | tee -a $OUT/switches.txt
done
