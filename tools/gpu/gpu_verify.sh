#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log; tail -3 gpurun_out/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench exit $?"; cat gpurun_out/bench_default.json
