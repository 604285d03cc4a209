#!/bin/bash
# round 4, session 8: the 8-wave 256 x 288 gate-residual epilogue with the next row group's residual / gate loads issued behind each
# column's store (in place) - same-box A/B against the library built from the previous commit (PRIMX_LIB), large-batch shapes
OUT=gpurun_out/r4_s8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "gemm suite: $(tail -1 $OUT/tests.log)"
BASE=$PWD/3dtopia-xl_amd/csrc/libprimx_base.so
for rep in 1 2; do
  echo "== previous commit"; PRIMX_LIB=$BASE timeout 200 python tools/gemm_bench_big.py 2>/dev/null | grep "g-r\|proj\|fc2" | tee -a $OUT/bench_base.txt
  echo "== pipelined epilogue"; timeout 200 python tools/gemm_bench_big.py 2>/dev/null | grep "proj\|fc2" | tee -a $OUT/bench_new.txt
done
B="python bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events"
show() { python -c "import json,sys;r=json.load(open(sys.argv[1]));print('%.3f' % r['ms_per_step'], ['%.3f' % v for v in r['repeats_ms_per_step']])" $1; }
for rep in 1 2; do
  PRIMX_LIB=$BASE timeout 300 $B > $OUT/b8_base_$rep.json 2>> $OUT/err.txt; echo "batch 8, previous commit: $(show $OUT/b8_base_$rep.json)"
  timeout 300 $B > $OUT/b8_new_$rep.json 2>> $OUT/err.txt; echo "batch 8, pipelined epilogue: $(show $OUT/b8_new_$rep.json)"
done
PRIMX_GEMM_PROF=1 REPS=2 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288q_dma<1,1>" | cut -c1-330 | awk 'NR%5==0' | tee $OUT/prof_new.txt
PRIMX_LIB=$BASE PRIMX_GEMM_PROF=1 REPS=2 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288q_dma<1,1>" | cut -c1-330 | awk 'NR%5==0' | tee $OUT/prof_base.txt
