#!/bin/bash
# cycle timelines (PRIMX_GEMM_PROF=1) of the heads GEMMs on the two rings of the 256x288 kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for k in 0 1; do
echo "== PRIMX_GEMM_HEADS_KT32=$k"
PRIMX_GEMM_HEADS_KT32=$k PRIMX_GEMM_PROF=1 timeout 300 python tools/heads_bench.py 2>&1 | grep "gemm288q_dma<" | awk '{k=$1" "$2" "$3" "$4; c[k]++; if (c[k]==10) print}' | cut -c1-330
done
