#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log
timeout 300 python tools/gemm_ksweep.py 2>&1 | grep "K="
timeout 300 python tools/gemm_bench.py 2>&1 | grep "M="
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err
echo "bench exit $?"; python -c "
import json; d=json.load(open('gpurun_out/bench2.json')); print(d['ms_per_step'], d['value'], d['achieved_tflops_whole_step']); print({k:(round(v['ms_per_step'],3), round(v['tflops'],1)) for k,v in d['kernels'].items()})"
