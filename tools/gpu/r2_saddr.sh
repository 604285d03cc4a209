#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc
timeout 200 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -2
for v in hip vaddr hip vaddr; do
  PRIMX_LIB=$C/libprimx_$v.so PRIMX_SKIP_FRESH_CHECK=1 timeout 100 python tools/gemm_ksweep.py 2>&1 | grep "K= 1152\|K= 4608" | sed "s/^/$v /"
done
for i in 1 2; do
  PRIMX_LIB=$C/libprimx_vaddr.so timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('vaddr', d['ms_per_step'])"
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('saddr', d['ms_per_step'])"
done
