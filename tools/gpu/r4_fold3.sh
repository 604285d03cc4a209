#!/bin/bash
# round 4, LayerNorm fold: same-box A/B of the headline step - unfolded / folded with the u, v GEMMs on side streams / on the calling stream
OUT=gpurun_out/r4_fold3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5"
for rep in 1 2 3; do
for cfg in "0 4" "1 4" "1 0"; do
  set -- $cfg
  PRIMX_DIT_FOLD=$1 PRIMX_FOLD_STREAMS=$2 timeout 120 python bench.py $B > $OUT/b.json 2> $OUT/b.err
  echo "fold=$1 streams=$2: $(python -c "import json;r=json.load(open('$OUT/b.json'));print(round(r['ms_per_step'],4), r.get('measured_job'))" 2>&1 | tail -1)"
done
done
