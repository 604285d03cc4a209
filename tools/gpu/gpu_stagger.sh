#!/bin/bash
for s in 0 2 3 4 6 8; do echo "== stagger $s"; PRIMX_ATTN_STAGGER=$s REPS=30 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu; done
