#!/bin/bash
# per-GPU shapes of the other BASELINE configs (configs[2]: batch 8 fp16; batch 8 bf16; configs[4]: N = 4096, batch 4, bf16; configs[1] in bf16)
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/final/bench_other.jsonl
for a in "--batch 8" "--batch 8 --dtype bf16" "--batch 4 --n-prim 4096 --dtype bf16" "--dtype bf16"; do
  timeout 600 python bench.py $a --steps 10 --warmup 2 --no-cpu-baseline --no-parity >> gpurun_out/final/bench_other.jsonl 2> /dev/null
done
python - <<'PY'
import json
for l in open('gpurun_out/final/bench_other.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:70], d['dtype'], round(d['ms_per_step'],2), round(d['value'],1), round(d.get('achieved_tflops_whole_step',0)), d['roofline']['kernel'][:50], round(d['roofline']['frac'],3))
PY
