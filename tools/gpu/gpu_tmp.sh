#!/bin/bash
L=$PWD/3dtopia-xl_amd/csrc
for v in w8s3p0 w8s3p1; do PRIMX_LIB=$L/libprimx_$v.so PRIMX_ATTN_ABL=8 REPS=2 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | grep "phase" | head -1; done
