#!/bin/bash
# round 4, last seconds of the GPU budget: the fold's u / v rows on the loader-wave 128 x 144 kernel (PRIMX_F32OUT_TILE144=1)
OUT=gpurun_out/r4_uv144
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PRIMX_F32OUT_TILE144=1 timeout 60 python -m pytest tests/test_hip_fold.py -m gpu -q -x -p no:cacheprovider -k "f32out or dit_with" > $OUT/tests.log 2>&1; echo "f32out + DiT fold tests with the 144 tile: $(tail -1 $OUT/tests.log)"
B="--no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 3 --repeats 2"
for v in 0 1; do
  PRIMX_F32OUT_TILE144=$v timeout 40 python bench.py $B > $OUT/b$v.json 2> $OUT/b$v.err
  echo "tile144=$v: $(python -c "import json;r=json.load(open('$OUT/b$v.json'));print(r['repeats_ms_per_step'])" 2>&1 | tail -1)"
done
