#!/bin/bash
# round 5, call 3: where does the persistent-pass kernel's time go at T = 32768?  timeline sums, L2 counters against the 8-wave
# 256 x 288 kernel, store policies (plain / nt / write-through builds), XCD walk shapes
OUT=gpurun_out/r5_pp2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -p no:cacheprovider -x -k "persistent or large_batch" > $OUT/tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/tests.log; tail -5 $OUT/tests.log
export ONLY=32768
echo "== timeline (PRIMX_GEMM_PROF=1)" | tee -a $OUT/diag.txt
PRIMX_GEMM_PROF=1 REPS=2 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep -v amdgpu.ids | grep "gemm144pp_dma<\|TFLOP" | tail -12 | tee -a $OUT/diag.txt
PRIMX_GEMM_PP_ROUNDS=0 PRIMX_GEMM_PROF=1 REPS=2 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep -v amdgpu.ids | grep "gemm288q_dma<\|TFLOP" | tail -12 | tee -a $OUT/diag.txt
L=3dtopia-xl_amd/csrc
for v in nt wt; do
  echo "== store policy $v: persistent, then 8-wave" | tee -a $OUT/diag.txt
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP | tee -a $OUT/diag.txt
  PRIMX_GEMM_PP_ROUNDS=0 PRIMX_LIB=$L/libprimx_$v.so timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP | tee -a $OUT/diag.txt
done
echo "== walk shapes (persistent): gm x sr" | tee -a $OUT/diag.txt
for gm in 8 4 2; do for sr in 1 2 4 8 16; do
  echo "gm=$gm sr=$sr" | tee -a $OUT/diag.txt
  PRIMX_GEMM_PP_GM=$gm PRIMX_GEMM_PP_SR=$sr REPS=8 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP | tee -a $OUT/diag.txt
done; done
G="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
for mode in pp q; do
  [ $mode = q ] && export PRIMX_GEMM_PP_ROUNDS=0
  REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT -o l2_$mode -- python tools/gemm_bench_big.py > /dev/null 2> $OUT/l2_$mode.err
  f=$(find $OUT -name "l2_${mode}_counter_collection.csv" | head -1)
  [ -n "$f" ] && { echo "-- L2 counters $mode" | tee -a $OUT/diag.txt; python tools/pmc_any.py $f gemm | cut -c1-250 | tee -a $OUT/diag.txt; rm -f $f; }
done
find $OUT -name "*.csv" -size +2M -delete; find $OUT -name "*.db" -delete
