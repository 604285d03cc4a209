#!/bin/bash
# same-box A/B of the default build against csrc/libprimx_prev.so: GEMM tests, isolated GEMMs, timelines, the step
OUT=gpurun_out/ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_vae.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/tests.log
for v in ${LIBS:-prev hip}; do echo "== $v timeline"; PRIMX_LIB=$CS/libprimx_$v.so ONLY=${PROF_ONLY:-fc1} REPS=3 PRIMX_GEMM_PROF=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "gemm|workgroup life" | tail -4 | cut -c1-420; done | tee $OUT/prof.txt
for rep in 1 2; do for v in ${LIBS:-prev hip}; do echo "== $v"; PRIMX_LIB=$CS/libprimx_$v.so ONLY=${BENCH_ONLY:-fc1,big_fc1} timeout 200 python tools/gemm_bench.py 2>&1 | grep TFLOP; done; done | tee $OUT/bench.txt
for rep in 1 2 3; do for v in ${LIBS:-prev hip}; do PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step $v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
