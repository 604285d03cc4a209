#!/bin/bash
# round 4, session 5: second form of the persistent 4-wave kernel (operands through registers): suite with it forced everywhere,
# micro-benchmarks against the 8-wave kernels, timelines, the batch-8 step - same box
OUT=gpurun_out/r4_s5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_W=2 timeout 600 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_w2.log 2>&1; echo "gemm suite, W everywhere: $(tail -1 $OUT/tests_w2.log)"
grep -q "passed" $OUT/tests_w2.log && ! grep -q "failed" $OUT/tests_w2.log || { tail -30 $OUT/tests_w2.log; exit 1; }
for w in 0 1; do echo "== PRIMX_GEMM_W=$w"; PRIMX_GEMM_W=$w timeout 200 python tools/gemm_bench_big.py 2>/dev/null | tee $OUT/bench_w$w.txt; done
echo "== timelines (W=2)"
PRIMX_GEMM_PROF=1 PRIMX_GEMM_W=2 REPS=2 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "gemm288w" | grep "workgroups" | cut -c1-330 | awk 'NR%5==0' | tee $OUT/prof_w2.txt
B="python bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events"
show() { python -c "import json,sys;r=json.load(open(sys.argv[1]));print('%.3f' % r['ms_per_step'], ['%.3f' % v for v in r['repeats_ms_per_step']])" $1; }
PRIMX_GEMM_W=0 timeout 300 $B > $OUT/b8_w0.json 2>> $OUT/err.txt; echo "batch 8, W=0: $(show $OUT/b8_w0.json)"
PRIMX_GEMM_W=1 timeout 300 $B > $OUT/b8_w1.json 2>> $OUT/err.txt; echo "batch 8, W=1: $(show $OUT/b8_w1.json)"
