#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
C=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc
for i in 1 2 3; do
  PRIMX_LIB=$C/libprimx_ord0.so timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('issue-first', d['ms_per_step'])"
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('mfma-first ', d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_fullconfig.py tests/test_hip_vae.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -2
