#!/bin/bash
# the switch matrix of DESIGN_LOG.md section 8 on the final tree (the parity tests of the GEMM / DiT / VAE / attention files under each alternate path)
OUT=gpurun_out/switches
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in PRIMX_GEMM_LOADER=0 PRIMX_GEMM_NOBIG=1 PRIMX_GEMM_P2=0 PRIMX_GEMM_BIGHEADS_MIN=0 PRIMX_GEMM_NOGEMV=1 PRIMX_WPREFETCH=0 PRIMX_WPREFETCH=1 PRIMX_NULL_KV_DEDUP=0 PRIMX_CFG_STREAMS=1; do
  echo "== $kv"; env $kv timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_vae.py tests/test_hip_attention.py tests/test_hip_e2e.py -q -x -p no:cacheprovider 2>&1 | tail -1
done | tee $OUT/switches.txt
for kv in PRIMX_NULL_KV_DEDUP=0 PRIMX_CFG_STREAMS=1 PRIMX_WPREFETCH=1; do echo "== $kv (full configuration)"; env $kv timeout 600 python -m pytest tests/test_hip_fullconfig.py -q -x -p no:cacheprovider -k "forward_with_cfg or ddim_trajectory" 2>&1 | tail -1; done | tee -a $OUT/switches.txt
