#!/bin/bash
# round 6: the GELU epilogue in packed fp32 arithmetic (gelu_tanh4, one uniform branch per row group) and the uniform-base LDS-DMA
# addresses of the 128-byte ring, against the previous commit's gemm.hip (libprimx_head.so), same box, alternating
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
L=$PWD/3dtopia-xl_amd/csrc
timeout 1200 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_fullconfig.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
for v in head hip head hip; do
  echo "== $v"
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
for k,v in d['kernels'].items():
    if 'gemm' in k and v['ms_per_step']>0.3: print('   ',k, round(v['ms_per_step'],4), round(1e3*v['ms_per_step']/v['launches_per_step'],2),'us', round(v['tflops'],1))
"
  PRIMX_LIB=$L/libprimx_$v.so timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 6 --warmup 2 --no-kernel-events 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$v batch8 ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
"
done
for v in head hip; do
PRIMX_LIB=$L/libprimx_$v.so PRIMX_GEMM_PROF=1 REPS=2 ONLY=32768,4096 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep "dma<" | awk 'NR%5==0' | cut -c1-330
done
