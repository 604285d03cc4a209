#!/bin/bash
# round 4: last check of HEAD - the default bench line exactly as the driver runs it (wall time included)
OUT=gpurun_out/r4_last
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T0=$(date +%s.%N)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
T1=$(date +%s.%N); echo "bench wall $(python -c "print(round($T1 - $T0, 1))") s"
cut -c1-300 $OUT/bench_default.json
python - <<'PY'
import json
r = json.load(open("gpurun_out/r4_last/bench_default.json"))
print("value", r["value"], r["unit"], "ms_per_step", r["ms_per_step"], "job", r["measured_job"]["job_ms"], "batch8", r["batch8"]["ms_per_step"], "bf16", r["bf16"]["ms_per_step"])
print("roofline", {k: r["roofline"][k] for k in ("kernel", "frac", "traffic", "traffic_source")})
print({k: round(v["ms_per_step"], 4) for k, v in r["kernels"].items() if "gemm144l" in k or "ln_" in k})
PY
