#!/bin/bash
# round 4: the LayerNorm-fold suite under the kernel-selection switches that move or remove the tiles the fold kernels live on
OUT=gpurun_out/r4_switches_fold
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $OUT/matrix.txt
for kv in PRIMX_GEMM_NOBIG=1 PRIMX_GEMM_LOADER=0 PRIMX_GEMM_BIGHEADS_MIN=0 PRIMX_GEMM_P2=0 PRIMX_CFG_STREAMS=1; do
  echo "== $kv: $(env $kv timeout 150 python -m pytest tests/test_hip_fold.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1)" | tee -a $OUT/matrix.txt
done
