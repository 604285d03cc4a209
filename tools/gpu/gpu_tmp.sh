#!/bin/bash
L=$PWD/3dtopia-xl_amd/csrc
for v in p0 p1 p2 p0 p1 p2; do echo "--- $v"; PRIMX_LIB=$L/libprimx_$v.so REPS=50 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | tail -4 | head -2; done
for v in p0 p1 p2; do PRIMX_LIB=$L/libprimx_$v.so PRIMX_ATTN_PROF=1 REPS=2 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | grep "segment" | head -1; done
