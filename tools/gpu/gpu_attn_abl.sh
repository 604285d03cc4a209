#!/bin/bash
# ablation timings of the attention kernel (PRIMX_ATTN_ABL: 1 no v_exp, 2 no softmax arithmetic, 3 no PV MFMAs,
# 4 no QK^T MFMAs, 5 no LDS fragment reads, 6 no s_barrier, 7 no DMA) - results wrong by design, timing only
for a in 0 1 2 3 4 5 6 7 0; do echo "ABL=$a"; PRIMX_ATTN_ABL=$a REPS=50 timeout 120 python tools/attn_bench.py 2>&1 | grep -v amdgpu | grep "self_b1\|self_n4096"; done
