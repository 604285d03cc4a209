#!/bin/bash
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench exit $?"
timeout 600 python bench.py --config decode > gpurun_out/final/bench_decode.json 2> gpurun_out/final/bench_decode.err; echo "decode exit $?"
timeout 900 python bench.py --config c4 --no-cpu-baseline > gpurun_out/final/bench_c4.json 2> gpurun_out/final/bench_c4.err; echo "c4 exit $?"
python - <<'PY'
import json
for n in ('default','decode','c4'):
    d=json.load(open(f'gpurun_out/final/bench_{n}.json')); print(n, d['ms_per_step'], d['value'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'), d.get('parity'))
PY
