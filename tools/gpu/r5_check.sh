#!/bin/bash
# round 5: the new bench legs and the new GPU tests, once, before the round-end artefact run
OUT=gpurun_out/r5_check
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_rowops.py tests/test_hip_e2e.py tests/test_hip_rccl.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/tests.log; tail -8 $OUT/tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_check/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "kernels sum", r["kernels_sum_ms_per_step"], "raw", r["kernels_sum_ms_per_step_raw"], "overhead us", 1e3 * r["event_pair_overhead_ms"])
print("roofline", r["kernel"], r["frac"], r["avg_launch_ms"], r["avg_launch_ms_raw_event_reading"])
print("torch_rocm_reference", json.dumps(d.get("torch_rocm_reference")))
print("cpu_baseline", d["cpu_baseline"])
print("parity", d["parity"])
print("per_rank", d["per_rank_ms_per_step"], d["world_size_seen"], d["ranks_seen"], d["weight_broadcast_ms"])
print("batch8", d["batch8"]["ms_per_step"], "bf16", d["bf16"]["ms_per_step"], "samples/s", d["samples_per_s_measured"])
PY
