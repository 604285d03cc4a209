#!/bin/bash
# round 6: fc1 at T = 4096 in the step: the two-pass kernel (default) against the one-pass 256x288 kernel on the 128-byte ring - per-kernel events
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25"
for cfg in ${CFGS:-"PRIMX_GEMM_P2=1" "PRIMX_GEMM_P2=0" "PRIMX_GEMM_P2=1" "PRIMX_GEMM_P2=0"}; do
env $cfg timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$cfg] ms_per_step', round(d['ms_per_step'],3), [round(v,3) for v in d['repeats_ms_per_step']], 'kernels_sum', round(d['roofline']['kernels_sum_ms_per_step'],3))
for k,v in d['kernels'].items():
    if v['ms_per_step']>0.15: print('   ',k, round(v['ms_per_step'],4), round(1e3*v['ms_per_step']/v['launches_per_step'],2),'us', round(v['tflops'] or 0,1))
"
done
