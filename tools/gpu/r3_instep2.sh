#!/bin/bash
OUT=gpurun_out/instep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p2 in 1 0; do
PRIMX_GEMM_P2=$p2 PRIMX_GEMM_PROF=1 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events --steps 2 --warmup 1 --repeats 1 > $OUT/bench.json 2> $OUT/prof$p2.txt
echo "== P2=$p2"; grep -A1 "gemm288[pq]_dma<1,0> M=4096 N=4608" $OUT/prof$p2.txt | tail -6 | cut -c1-400
done
