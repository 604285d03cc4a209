#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_hip_vae.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv3d_k3 or decode" 2>&1 | tail -2
for i in 1 2; do
REPS=10 python tools/conv_bench.py
PRIMX_LIB=$GRAFT_REPO_ROOT/3dtopia-xl_amd/csrc/libprimx_old.so REPS=10 python tools/conv_bench.py
done
