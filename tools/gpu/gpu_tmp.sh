#!/bin/bash
PRIMX_GEMM_PROF=1 ONLY=fc1 REPS=3 timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | grep "gemm288\|per k-tile" | tail -2
