#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
for v in 1 0; do
PRIMX_GEMM_XCD2D=$v python bench.py --no-cpu-baseline --no-parity --no-kernel-events --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('XCD2D=$v', d['ms_per_step'], d['repeats_ms_per_step'])"
done
done
