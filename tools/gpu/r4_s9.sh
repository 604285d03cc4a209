#!/bin/bash
# round 4, session 9: the token embedding writes both CFG halves itself (one launch less per forward, 16-byte stores)
OUT=gpurun_out/r4_s9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_rowops.py tests/test_hip_dit.py tests/test_hip_dinov2.py tests/test_hip_fp32.py -m gpu -q -x -p no:cacheprovider > $OUT/tests.log 2>&1; echo "suites: $(tail -1 $OUT/tests.log)"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python bench.py --no-cpu-baseline --no-parity --no-decode-leg --no-side-legs --no-kernel-events --steps 25 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
for db in $(find $OUT -name "*.db"); do python tools/rocprof_summary.py $db ${db%.db}_summary.txt > /dev/null; done
grep "linear_f32_kernel\|copy\|Copy" $(find $OUT -name "*_summary.txt" | head -1) | head -5 | cut -c1-150
python -c "import json;r=json.load(open('$OUT/bench.json'));print(r['ms_per_step'])"
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete
