#!/bin/bash
# round 4, session 1: the fused gate-residual + LayerNorm kernel on the device - the GEMM / row-op / DiT suites, then the step
# A/B on this box: fused with agent-scope fences (default) | fused with same-XCD fences | two launches (PRIMX_DIT_FUSE_LN=0)
OUT=gpurun_out/r4_s1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_dit.py tests/test_hip_e2e.py -m gpu -q --tb=short -p no:cacheprovider -x -s > $OUT/tests.log 2>&1; echo "pytest exit $?" >> $OUT/tests.log; tail -5 $OUT/tests.log
grep -q "pytest exit 0" $OUT/tests.log || { echo "tests failed: skipping the benches"; tail -40 $OUT/tests.log; exit 1; }
B="python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5"
for rep in 1 2; do
  timeout 300 $B > $OUT/fused_agent_$rep.json 2> $OUT/err.txt; echo "fused agent: $(python -c "import json;r=json.load(open('$OUT/fused_agent_$rep.json'));print(r['ms_per_step'], r['repeats_ms_per_step'], r['ln_in_gemm_tail'])")"
  PRIMX_LN_FENCE=light timeout 300 $B > $OUT/fused_light_$rep.json 2>> $OUT/err.txt; echo "fused light: $(python -c "import json;r=json.load(open('$OUT/fused_light_$rep.json'));print(r['ms_per_step'], r['repeats_ms_per_step'], r['ln_in_gemm_tail'])")"
  PRIMX_DIT_FUSE_LN=0 timeout 300 $B > $OUT/unfused_$rep.json 2>> $OUT/err.txt; echo "two launches: $(python -c "import json;r=json.load(open('$OUT/unfused_$rep.json'));print(r['ms_per_step'], r['repeats_ms_per_step'], r['ln_in_gemm_tail'])")"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $B > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*.db" -size +20M -delete
head -30 $(find $OUT -name "*_summary.txt" | head -1) | cut -c1-200
tail -3 $OUT/err.txt
