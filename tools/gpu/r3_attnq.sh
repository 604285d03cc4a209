#!/bin/bash
# attention prologue: Q loads behind the DMA issue - suite, isolated kernel, step A/B (prev = the commit before)
OUT=gpurun_out/attnq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
timeout 900 python -m pytest tests/test_hip_attention.py tests/test_hip_dit.py -x -q 2>&1 | tail -3 | tee $OUT/tests.txt
for rep in 1 2; do for v in prev hip; do echo "== $v"; PRIMX_LIB=$CS/libprimx_$v.so timeout 200 python tools/attn_bench.py 2>&1 | grep -E "self_b1|cross_b1|self_n4096"; done; done | tee $OUT/attn.txt
for rep in 1 2 3; do for v in prev hip; do
  PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>$OUT/bench_$v.err
  python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step $v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
