#!/bin/bash
# store policy of the GEMM epilogues: plain / nt / write-through - per-XCD timelines, isolated GEMMs, the step
OUT=gpurun_out/s6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
for v in hip sp1 sp2; do echo "== $v timeline"; PRIMX_LIB=$CS/libprimx_$v.so ONLY=fc1,proj REPS=3 PRIMX_GEMM_PROF=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "gemm|workgroup life" | tail -4 | cut -c1-420; done | tee $OUT/prof.txt
for rep in 1 2; do for v in hip sp1 sp2; do echo "== $v"; PRIMX_LIB=$CS/libprimx_$v.so ONLY=proj,fc2,qkv,fc1 timeout 200 python tools/gemm_bench.py 2>&1 | grep TFLOP; PRIMX_LIB=$CS/libprimx_$v.so timeout 200 python tools/gemm_ksweep.py 2>&1 | grep -E "K= *(1152|4608)"; done; done | tee $OUT/bench.txt
for rep in 1 2; do for v in hip sp1 sp2; do PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step $v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
