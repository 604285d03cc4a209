#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc
for v in hip ord1 ord2 hip ord1 ord2; do
  PRIMX_LIB=$C/libprimx_$v.so PRIMX_SKIP_FRESH_CHECK=1 timeout 120 python tools/gemm_ksweep.py 2>&1 | grep "K= 1152\|K= 4608" | sed "s/^/$v /"
done
for v in ord1 ord2; do
PRIMX_LIB=$C/libprimx_$v.so timeout 300 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -1
done
