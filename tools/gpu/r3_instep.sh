#!/bin/bash
OUT=gpurun_out/instep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_PROF=1 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 2 --warmup 1 --repeats 1 > $OUT/bench.json 2> $OUT/prof.txt
grep -A1 "gemm288q_dma<1,0> M=4096 N=4608" $OUT/prof.txt | tail -12 | cut -c1-400
echo ...; grep -A1 "gemm144_dma<1,1> M=4096 N=1152 K=4608" $OUT/prof.txt | tail -6 | cut -c1-400
echo ...; grep -A1 "gemm144_dma<1,1> M=4096 N=1152 K=1152" $OUT/prof.txt | tail -6 | cut -c1-400
echo ...; grep -A1 "gemm288q_dma<1,2> M=4096 N=3456" $OUT/prof.txt | tail -6 | cut -c1-400
