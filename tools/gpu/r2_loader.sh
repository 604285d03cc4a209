#!/bin/bash
# loader-wave 128x144 kernel (PRIMX_GEMM_LOADER=1): parity, K sweep, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_GEMM_LOADER=1 timeout 200 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -3
for v in 0 1 0 1; do
  PRIMX_GEMM_LOADER=$v timeout 100 python tools/gemm_ksweep.py 2>&1 | grep "K=  512\|K= 1152\|K= 4608" | sed "s/^/loader=$v /"
done
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain ', d['ms_per_step'])"
  PRIMX_GEMM_LOADER=1 timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('loader', d['ms_per_step'])"
done
