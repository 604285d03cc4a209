#!/bin/bash
# round 6: per-round start / end times of the workgroups of the multi-round 256 x 288 launches (PRIMX_GEMM_PROF=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_PROF=1 REPS=2 ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep -A2 "dma<" | grep -v "^--" | cut -c1-400
