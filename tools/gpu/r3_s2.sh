#!/bin/bash
# round 3, session 2: GEMM suite on the pruned tree, residual prefetch by the compute waves (XV = 4 / 6 / 8 chunks) vs default,
# per-workgroup timelines of the 256x288 and the 8-wave 128x144 kernels
OUT=gpurun_out/s2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CS=$PWD/3dtopia-xl_amd/csrc
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/gemm_tests.log 2>&1; echo "default: pytest exit $?"; tail -2 $OUT/gemm_tests.log
for v in xv4 xv6 xv8; do
  PRIMX_LIB=$CS/libprimx_$v.so timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/${v}_tests.log 2>&1; echo "$v: pytest exit $?"; tail -1 $OUT/${v}_tests.log
done
for rep in 1 2; do for v in hip xv4 xv6 xv8; do echo "== ksweep $v"; PRIMX_LIB=$CS/libprimx_$v.so timeout 200 python tools/gemm_ksweep.py 2>&1 | grep -E "K= *(128|512|1152|4608)"; done; done | tee $OUT/ksweep.txt
for rep in 1 2; do for v in hip xv4 xv6 xv8; do PRIMX_LIB=$CS/libprimx_$v.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-decode-leg --no-kernel-events > $OUT/bench_$v.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/bench_$v.json')); print('step $v', ['%.3f' % x for x in d['repeats_ms_per_step']])"; done; done | tee $OUT/steps.txt
ONLY=proj,fc2,fc1,big_fc1 REPS=3 PRIMX_GEMM_PROF=1 timeout 200 python tools/gemm_bench.py 2>&1 | grep -E "gemm|us" | tail -30 | tee $OUT/gemm_prof.txt
