#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_ATTN_PROF=3 REPS=2 timeout 300 python tools/attn_bench.py 2>&1 | grep -E "attn2 profile|self_b1" | head -8
