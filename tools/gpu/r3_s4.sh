#!/bin/bash
OUT=gpurun_out/s4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 tools/probe/write_burst 2>&1 | tee $OUT/write_burst.txt
ONLY=proj,fc2,fc1,qkv REPS=2 PRIMX_GEMM_PROF=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "gemm|workgroup life" | tail -16 | tee $OUT/gemm_prof.txt
