#!/bin/bash
OUT=gpurun_out/s5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 tools/probe/write_burst 2>&1 | tee $OUT/write_burst.txt
for x in 1 0; do echo "== PRIMX_GEMM_XCD2D=$x"; ONLY=fc1,proj REPS=3 PRIMX_GEMM_XCD2D=$x PRIMX_GEMM_PROF=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "gemm|workgroup life" | tail -8; done | tee $OUT/gemm_prof.txt
