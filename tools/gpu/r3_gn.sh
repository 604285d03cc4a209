#!/bin/bash
# GroupNorm + SiLU folded into the 4^3 convolution operand load (conv3_s4c256_kernel<.., 1>): parity, then decode A/B.
# Needs tools/probe/conv3_gn_at_load.patch applied (the variant measured slower and is not in the tree: profiles/r3_experiments.txt section 21).
OUT=gpurun_out/gn
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_vae.py -x -q -s 2>&1 | grep -E "rel-L2|passed|failed|Error|error" | cut -c1-300 | tee $OUT/tests.txt
for rep in 1 2 3; do for v in 0 1; do
  PRIMX_CONV_GN=$v timeout 300 python bench.py --config decode --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $OUT/dec_$v.json 2>$OUT/dec_$v.err
  python -c "
import json; d=json.load(open('$OUT/dec_$v.json')); print('decode GN=$v', d['ms_per_step'], d.get('repeats_ms_per_step'))
if $rep == 1:
    for k in d.get('kernels', [])[:9]: print('   ', k)
"; done; done 2>&1 | cut -c1-400 | tee $OUT/ab.txt
