#!/bin/bash
# round 6: the whole GPU suite + smoke + the default bench line on the current tree
OUT=gpurun_out/${TAG:-r6_check}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" >> $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").readline())
print("ms_per_step", d["ms_per_step"], "value", d["value"], "batch8", d.get("batch8", {}).get("ms_per_step"), "bf16", d.get("bf16", {}).get("ms_per_step"), "job_ms", d.get("measured_job", {}).get("job_ms"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms") if k in d["roofline"]})
PY
