#!/bin/bash
# the kernel-selection switches (DESIGN_LOG.md section 8): parity tests under each alternate path
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in PRIMX_GEMM_XCD2D=0 PRIMX_GEMM_BIGQ=0 PRIMX_GEMM_NOBIG=1 PRIMX_CONV_REG=0 PRIMX_ATTN_V2=1; do
  echo "== $kv"; env $kv timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_vae.py tests/test_hip_attention.py -q -x -p no:cacheprovider 2>&1 | tail -1
done
