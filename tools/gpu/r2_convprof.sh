#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
REPS=10 python tools/conv_bench.py
PRIMX_LIB=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc/libprimx_old.so REPS=10 python tools/conv_bench.py
done
