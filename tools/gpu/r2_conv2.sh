#!/bin/bash
mkdir -p gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_vae.py -m gpu -q --tb=short -p no:cacheprovider -x -k "conv or decode or groupnorm or fused" > gpurun_out/r2e/vae_tests.log 2>&1; tail -3 gpurun_out/r2e/vae_tests.log
timeout 600 python bench.py --config decode > gpurun_out/r2e/bench_decode.json 2> gpurun_out/r2e/bench_decode.err; echo "decode exit $?"
python - <<PY
import json
d=json.load(open("gpurun_out/r2e/bench_decode.json"))
print(d["ms_per_step"], d["value"], d.get("parity"))
for k,v in d["kernels"].items(): print("   ", k, round(v["ms_per_step"],4), v["tflops"] and round(v["tflops"],1))
PY
