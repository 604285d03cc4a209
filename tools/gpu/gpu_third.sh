#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -6 gpurun_out/gpu_tests.log
echo "--- DMA"; timeout 300 python tools/gemm_bench.py
echo "--- regstage"; PRIMX_GEMM_REGSTAGE=1 ONLY=proj,fc2,qkv,fc1 timeout 300 python tools/gemm_bench.py
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err
echo "bench exit $?"; tail -3 gpurun_out/bench2.err; python -c "
import json; d=json.load(open('gpurun_out/bench2.json')); print(d['ms_per_step'], d['value'], d['achieved_tflops_whole_step']); print({k:(round(v['ms_per_step'],3), round(v['tflops'],1)) for k,v in d['kernels'].items()}); print(d['roofline'])"
