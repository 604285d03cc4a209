#!/bin/bash
# first GPU pass: full GPU suite without -x so that every failing kernel is reported in one call
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0), torch.version.hip)" > gpurun_out/dev.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
echo "pytest exit $?" >> gpurun_out/gpu_tests.log
tail -60 gpurun_out/gpu_tests.log
