#!/bin/bash
# conv3 register-resident kernel: parity tests, decode bench A/B (PRIMX_CONV_REG=0 = implicit GEMM), default bench
mkdir -p gpurun_out/r2e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_vae.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/r2e/vae_tests.log 2>&1; tail -5 gpurun_out/r2e/vae_tests.log
for v in 1 0; do
  PRIMX_CONV_REG=$v timeout 600 python bench.py --config decode > gpurun_out/r2e/bench_decode_reg$v.json 2> gpurun_out/r2e/bench_decode_reg$v.err; echo "decode reg=$v exit $?"
  python - <<PY
import json
d=json.load(open("gpurun_out/r2e/bench_decode_reg$v.json"))
print(d["ms_per_step"], d["value"], d.get("parity"))
for k,v in d["kernels"].items(): print("   ", k, round(v["ms_per_step"],4), v["tflops"] and round(v["tflops"],1))
PY
done
timeout 900 python bench.py > gpurun_out/r2e/bench_default.json 2> gpurun_out/r2e/bench_default.err; echo "bench exit $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2e/bench_default.json"))
print(d["ms_per_step"], d["value"], d.get("parity"), d["roofline"]["kernel"], d["roofline"]["frac"])
print(d.get("with_reuse_cond_kv"))
PY
