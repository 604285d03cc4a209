#!/bin/bash
# round 5: the one switch configuration r5_final.sh stopped early in (-x): PRIMX_PLAN_TIMESTEPS=0, whole matrix suite, no -x
OUT=gpurun_out/r5_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== PRIMX_PLAN_TIMESTEPS=0 (second re-run, no -x)" | tee -a $OUT/matrix_rerun2.txt
PRIMX_PLAN_TIMESTEPS=0 timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_vae.py tests/test_hip_fullconfig.py tests/test_hip_e2e.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -15 | grep -v "^\.\+ *\[" | tee -a $OUT/matrix_rerun2.txt
