#!/bin/bash
# sub-block tile walk inside an XCD's block (xcd_tile2d): big-tile tests, isolated big GEMMs, the K / V projection inside the step, batch 8
OUT=gpurun_out/walk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py -x -q 2>&1 | tail -2 | tee $OUT/tests.txt
for rep in 1 2; do for v in prev hip; do echo "== $v"; PRIMX_LIB=$CS/libprimx_$v.so ONLY=fc1,big_fc1,big_proj REPS=10 timeout 200 python tools/gemm_bench.py 2>&1 | grep TFLOP; done; done | tee $OUT/gemm.txt
for rep in 1 2; do for v in prev hip; do
  PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg > $OUT/bench_$v.json 2>$OUT/bench_$v.err
  python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step $v', ['%.3f' % x for x in d['repeats_ms_per_step']], 'kv proj us', [round(1e3*v['ms_per_step'],1) for k,v in d['kernels'].items() if '64512' in k])"; done; done | tee $OUT/steps.txt
for v in prev hip; do
  PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/b8_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/b8_$v.json')); print('batch 8 $v', ['%.2f' % x for x in d['repeats_ms_per_step']])"; done | tee $OUT/b8.txt
