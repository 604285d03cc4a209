#!/bin/bash
# round 4, session 4: the persistent 4-wave 256 x 288 kernel (gemm288w): suites with it forced everywhere (PRIMX_GEMM_W=2) and at its
# default selection, micro-benchmarks against the 8-wave kernels (PRIMX_GEMM_W=0) on rotating buffers, the per-workgroup timeline, and
# the batch-8 step (BASELINE configs[2] / [3] per-GPU shape) with and without it - same box.
OUT=gpurun_out/r4_s4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_W=2 timeout 900 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider > $OUT/tests_w2.log 2>&1; echo "gemm suite, W everywhere: $(tail -1 $OUT/tests_w2.log)"
grep -q "passed" $OUT/tests_w2.log && ! grep -q "failed" $OUT/tests_w2.log || { tail -30 $OUT/tests_w2.log; exit 1; }
timeout 600 python -m pytest tests/test_hip_gemm.py -m gpu -q -x -p no:cacheprovider -k "persistent or big or headline" > $OUT/tests_w1.log 2>&1; echo "default selection: $(tail -1 $OUT/tests_w1.log)"
for w in 0 1 2; do echo "== PRIMX_GEMM_W=$w"; PRIMX_GEMM_W=$w timeout 300 python tools/gemm_bench_big.py 2>/dev/null | tee $OUT/bench_w$w.txt; done
echo "== timelines"
PRIMX_GEMM_PROF=1 PRIMX_GEMM_W=2 REPS=2 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep -A1 "gemm288w\|gemm288q\|gemm288p" | grep -v "^--" | cut -c1-420 | head -40 | tee $OUT/prof_w2.txt
PRIMX_GEMM_PROF=1 PRIMX_GEMM_W=0 REPS=2 timeout 300 python tools/gemm_bench_big.py 2>&1 | grep -A1 "gemm288w\|gemm288q\|gemm288p" | grep -v "^--" | cut -c1-420 | head -40 | tee $OUT/prof_w0.txt
B="python bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events"
show() { python -c "import json,sys;r=json.load(open(sys.argv[1]));print('%.3f' % r['ms_per_step'], ['%.3f' % v for v in r['repeats_ms_per_step']])" $1; }
for rep in 1 2; do
  PRIMX_GEMM_W=0 timeout 400 $B > $OUT/b8_w0_$rep.json 2>> $OUT/err.txt; echo "batch 8, W=0: $(show $OUT/b8_w0_$rep.json)"
  PRIMX_GEMM_W=1 timeout 400 $B > $OUT/b8_w1_$rep.json 2>> $OUT/err.txt; echo "batch 8, W=1: $(show $OUT/b8_w1_$rep.json)"
done
H="python bench.py --steps 25 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events"
PRIMX_DIT_FUSE_LN=0 PRIMX_GEMM_W=1 timeout 300 $H > $OUT/h_w1.json 2>> $OUT/err.txt; echo "headline, W=1 (fc1 on gemm288p): $(show $OUT/h_w1.json)"
PRIMX_DIT_FUSE_LN=0 PRIMX_GEMM_W=2 timeout 300 $H > $OUT/h_w2.json 2>> $OUT/err.txt; echo "headline, W=2 (fc1 on gemm288w): $(show $OUT/h_w2.json)"
