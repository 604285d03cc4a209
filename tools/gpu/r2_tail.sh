#!/bin/bash
# A/B of the LDS-DMA ring tails (PRIMX_GEMM_TAILDMA=1 = redundant clamped fetches, as before) + parity tests + fresh artefacts
OUT=gpurun_out/tail
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2; do
  PRIMX_GEMM_TAILDMA=1 timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('clamped', d['ms_per_step'], d['roofline']['achieved'])"
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('skipped', d['ms_per_step'], d['roofline']['achieved'])"
done
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_fullconfig.py tests/test_hip_e2e.py tests/test_hip_vae.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline --no-parity --steps 25 > $OUT/bench_trace.json 2> $OUT/bench_trace.err
for db in $OUT/*.db; do python tools/rocprof_summary.py $db ${db%.db}_summary.txt; done
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; cut -c1-700 $OUT/bench_default.json
