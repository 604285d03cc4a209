#!/bin/bash
# round 4, session 6: the whole -m gpu suite on the cleaned tree (the batch-8 trajectory fixture is still being generated on the CPU)
OUT=gpurun_out/r4_s6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --deselect tests/test_hip_fullconfig.py::test_configs2_batch8_ddim5_trajectory > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" >> $OUT/gpu_tests.log
grep -v "amdgpu.ids" $OUT/gpu_tests.log | grep -i "rel-L2\|configs\[\|passed\|failed\|error\|exit" | cut -c1-400 | tail -40
