#!/bin/bash
for a in 0 1 2 3 4 5; do echo "== ABL $a  (1 noexp, 2 nosoftmax, 3 noPV, 4 noQK, 5 no-LDS-frag-reads)"; PRIMX_ATTN_ABL=$a REPS=30 timeout 120 python tools/attn_bench.py 2>&1 | grep -E "self_b1|self_b8"; done
