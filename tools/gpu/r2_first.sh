#!/bin/bash
# round 2, first GPU pass: the whole -m gpu suite (new: fp32 route, full-config parity, pinned ray marcher), default bench,
# decode bench, kernel trace of the decode bench
mkdir -p gpurun_out/r2a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -s > gpurun_out/r2a/gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2a/gpu_tests.log; tail -5 gpurun_out/r2a/gpu_tests.log
grep -E "rel-L2|HIP march|passed|failed|Error" gpurun_out/r2a/gpu_tests.log | tail -20
timeout 900 python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err; echo "bench exit $?"; cut -c1-1500 gpurun_out/r2a/bench_default.json; tail -3 gpurun_out/r2a/bench_default.err
timeout 600 python bench.py --config decode --steps 10 > gpurun_out/r2a/bench_decode.json 2> gpurun_out/r2a/bench_decode.err; echo "decode exit $?"; cut -c1-2500 gpurun_out/r2a/bench_decode.json; tail -3 gpurun_out/r2a/bench_decode.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r2a -o dectrace -- python bench.py --config decode --steps 5 --no-cpu-baseline --no-kernel-events > /dev/null 2> gpurun_out/r2a/dectrace.err
for db in gpurun_out/r2a/*.db; do python tools/rocprof_summary.py $db gpurun_out/r2a/decode_kernel_trace_summary.txt; done
head -30 gpurun_out/r2a/decode_kernel_trace_summary.txt | cut -c1-180
rm -f gpurun_out/r2a/*.db
