#!/bin/bash
mkdir -p gpurun_out/r2d
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r2d/gpu_tests.log 2>&1; tail -4 gpurun_out/r2d/gpu_tests.log
timeout 900 python bench.py > gpurun_out/r2d/bench_default.json 2> gpurun_out/r2d/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2d/bench_default.json"))
print(d["ms_per_step"], d["value"], d.get("parity"), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k,v in d["kernels"].items(): print("   ", k, round(v["ms_per_step"],4), v["tflops"] and round(v["tflops"],1))
PY
