#!/bin/bash
# final profiles of the bench command: kernel trace (+stats) and the two PMC traffic passes
mkdir -p gpurun_out/final
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/final -o trace -- python bench.py --no-cpu-baseline --steps 25 > gpurun_out/final/bench_trace.json 2> gpurun_out/final/bench_trace.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/final -o fetch -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-kernel-events > /dev/null 2> gpurun_out/final/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/final -o write -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --no-kernel-events > /dev/null 2> gpurun_out/final/write.err
ls -la gpurun_out/final | head -20
cat gpurun_out/final/bench_trace.json | head -c 600
