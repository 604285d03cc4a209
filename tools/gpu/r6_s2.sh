#!/bin/bash
# round 6, call 2/3: what do the epilogue's stores cost at T = 32768?  probe builds of the 8-wave 256x288 kernel: skip2 = no stores at all,
# skip1 = every other tile column stores, skip3 = every tile stores into rows 0 .. 255 (the instructions without the HBM traffic)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=3dtopia-xl_amd/csrc
for v in ${VARIANTS:-hip skip2 hip skip2}; do
  echo "== $v"
  PRIMX_LIB=$PWD/$L/libprimx_$v.so ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
done
echo "== timelines"
for v in ${TVARIANTS:-hip skip2}; do
  PRIMX_LIB=$PWD/$L/libprimx_$v.so PRIMX_GEMM_PROF=1 REPS=2 ONLY=32768 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288q_dma<" | tail -3 | cut -c1-400
done
