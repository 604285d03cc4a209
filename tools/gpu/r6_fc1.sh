#!/bin/bash
# round 6: fc1 at T = 4096 (one round of 256 workgroups): the two-pass kernel (default) against the one-pass 256x288 kernel on its 64-byte and 128-byte rings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25"
for cfg in "" "PRIMX_GEMM_P2=0" "PRIMX_GEMM_P2=0 PRIMX_GEMM_KT64_MIN=1" "" "PRIMX_GEMM_P2=0" "PRIMX_GEMM_P2=0 PRIMX_GEMM_KT64_MIN=1"; do
env $cfg timeout 300 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('[$cfg] ms_per_step', round(d['ms_per_step'],3), [round(v,3) for v in d['repeats_ms_per_step']])
"
done
for cfg in "" "PRIMX_GEMM_P2=0" "PRIMX_GEMM_P2=0 PRIMX_GEMM_KT64_MIN=1"; do
echo "[$cfg]"; env $cfg ONLY=4096 timeout 200 python tools/gemm_bench_big.py 2>&1 | grep TFLOP
done
