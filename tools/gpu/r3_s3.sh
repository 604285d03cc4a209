#!/bin/bash
# round 3, session 3: write-burst probe, attention segment profile, GEMM timelines with the shader clock
OUT=gpurun_out/s3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 tools/probe/write_burst 2>&1 | tee $OUT/write_burst.txt
timeout 200 python tools/attn_bench.py 2>&1 | grep -v Warn | tee $OUT/attn.txt
PRIMX_ATTN_PROF=1 timeout 200 python tools/attn_bench.py 2>&1 | grep -E "segment|us" | tail -8 | tee $OUT/attn_prof.txt
ONLY=proj,fc2,fc1,qkv REPS=2 PRIMX_GEMM_PROF=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "gemm" | awk 'NR%5==0' | tee $OUT/gemm_prof.txt
