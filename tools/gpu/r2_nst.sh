#!/bin/bash
# A/B: 128x144 LDS-DMA GEMM ring with 4 stages (3 tiles in flight) vs 3 stages
OUT=gpurun_out/nst
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc/libprimx_nst4.so
PRIMX_LIB=$V timeout 300 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -2
for i in 1 2; do
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nst3', d['ms_per_step'])"
  PRIMX_LIB=$V timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nst4', d['ms_per_step'])"
done
timeout 120 python tools/gemm_ksweep.py 2>&1 | grep "K=" | sed 's/^/nst3 /'
PRIMX_LIB=$V timeout 120 python tools/gemm_ksweep.py 2>&1 | grep "K=" | sed 's/^/nst4 /'
