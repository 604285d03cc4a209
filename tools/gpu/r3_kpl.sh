#!/bin/bash
# kernarg preload (leading scalar kernel arguments in SGPRs at wave launch): suite, then same-box A/B of the step
OUT=gpurun_out/kpl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_rowops.py tests/test_hip_attention.py tests/test_hip_dit.py tests/test_hip_vae.py -x -q 2>&1 | tail -4 | tee $OUT/tests.txt
for rep in 1 2 3; do for v in base hip; do
  PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>$OUT/bench_$v.err
  python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step $v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
for v in base hip; do PRIMX_LIB=$CS/libprimx_$v.so ONLY=proj,fc2,qkv,fc1 timeout 200 python tools/gemm_bench.py 2>&1 | grep TFLOP; done | tee $OUT/gemm.txt
