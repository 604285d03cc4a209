#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -5 gpurun_out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
echo "bench exit $?"; tail -5 gpurun_out/bench.err; cat gpurun_out/bench.json
