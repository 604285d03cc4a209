#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REPS=10 python tools/conv_bench.py
PRIMX_CONV_PROF=1 REPS=2 python tools/conv_bench.py 2>&1 | tail -2
