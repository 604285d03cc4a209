#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in v2a v3a v3b v3c v2a; do echo "=== $v"; PRIMX_LIB=$PWD/3dtopia-xl_amd/csrc/libprimx_$v.so REPS=40 timeout 300 python tools/attn_bench.py 2>&1 | grep -E "self_b1|cross_b1|rel-L2" | head -4; PRIMX_ATTN_PROF=1 PRIMX_LIB=$PWD/3dtopia-xl_amd/csrc/libprimx_$v.so REPS=1 timeout 300 python tools/attn_bench.py 2>&1 | grep -E "segment profile" | head -1; done
