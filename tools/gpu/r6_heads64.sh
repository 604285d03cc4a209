#!/bin/bash
# end of round 6: the heads epilogues of the 256x288 kernel on the 128-byte ring (rolling fragment window, no scratch) against the 32-wide ring
# (PRIMX_GEMM_HEADS_KT32=1): GEMM / fold / attention-operand tests under both, then configs[1] and batch 8 alternating on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
for k in 0 1 0 1; do
PRIMX_GEMM_HEADS_KT32=$k timeout 300 python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 25 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('HEADS_KT32=$k ms_per_step', round(d['ms_per_step'],4), [round(x,4) for x in d['repeats_ms_per_step']])
for n,v in sorted(d['kernels'].items()):
    if '288q' in n and (', 7,' in n or ', 2,' in n or 'pair' in n): print('    ', n, round(1e3*v['ms_per_step']/v['launches_per_step'],2), 'us', round(v['tflops'],1))
"
done
for k in 0 1 0 1; do
PRIMX_GEMM_HEADS_KT32=$k timeout 300 python bench.py --batch 8 --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('batch 8 HEADS_KT32=$k ms_per_step', round(d['ms_per_step'],3), [round(x,3) for x in d['repeats_ms_per_step']])
for n,v in sorted(d['kernels'].items()):
    if '288q' in n and (', 7,' in n or ', 2,' in n): print('    ', n, round(1e3*v['ms_per_step']/v['launches_per_step'],2), 'us', round(v['tflops'],1))
"
done
