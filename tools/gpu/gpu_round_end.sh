#!/bin/bash
# round-end artefacts: GPU tests + smoke + default bench, kernel trace of the bench command, PMC traffic passes
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/final/gpu_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/final/gpu_tests.log; tail -2 gpurun_out/final/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench exit $?"; cat gpurun_out/final/bench_default.json
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final -o trace -- python bench.py --no-cpu-baseline --steps 25 > gpurun_out/final/bench_trace.json 2> gpurun_out/final/bench_trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/final -o fetch -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-kernel-events > /dev/null 2> gpurun_out/final/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/final -o write -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-kernel-events > /dev/null 2> gpurun_out/final/write.err
for db in gpurun_out/final/*.db; do python tools/rocprof_summary.py $db gpurun_out/final/kernel_trace_summary.txt; done
tail -22 gpurun_out/final/kernel_trace_summary.txt | cut -c1-200
ls -la gpurun_out/final | head -30
