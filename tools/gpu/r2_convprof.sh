#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_vae.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv3d" 2>&1 | tail -3
REPS=10 python tools/conv_bench.py
REPS=10 python tools/conv_bench.py
PRIMX_CONV_PROF=1 REPS=2 python tools/conv_bench.py 2>&1 | tail -2
