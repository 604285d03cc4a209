#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc
PRIMX_LIB=$C/libprimx_xtail.so timeout 60 python -m pytest tests/test_hip_gemm.py -m gpu -q --tb=line -x -p no:cacheprovider 2>&1 | tail -2
for v in hip xtail; do
  PRIMX_LIB=$C/libprimx_$v.so PRIMX_SKIP_FRESH_CHECK=1 timeout 40 python tools/gemm_ksweep.py 2>&1 | grep "K=  512\|K= 1152\|K= 4608" | sed "s/^/$v /"
done
