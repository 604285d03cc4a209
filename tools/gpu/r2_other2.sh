#!/bin/bash
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > gpurun_out/final/bench_other2.jsonl
for a in "--dtype bf16" "--batch 8"; do
  timeout 55 python bench.py $a --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-parity >> gpurun_out/final/bench_other2.jsonl 2> /dev/null
done
python - <<'PY'
import json
for l in open('gpurun_out/final/bench_other2.jsonl'):
    d=json.loads(l); print(d['config']['workload'][-60:], d['dtype'], round(d['ms_per_step'],2), round(d['value'],1), round(d.get('achieved_tflops_whole_step',0)))
PY
