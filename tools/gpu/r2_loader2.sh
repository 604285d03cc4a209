#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
  PRIMX_GEMM_LOADER=0 timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain ', d['ms_per_step'])"
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('loader', d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items() if '144' in k})"
done
