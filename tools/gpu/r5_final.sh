#!/bin/bash
# round 5, last call: the three switch configurations whose test GUARDS were wrong in r5_switches.sh's first run, the two-rank bench path on
# one GPU (gloo), the whole GPU suite, the kernel suites three more times (a race shows as a flicker), the default bench line
OUT=gpurun_out/r5_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for kv in PRIMX_GEMM_NOBIG=1 PRIMX_DIT_LN_TAIL=1 PRIMX_PLAN_TIMESTEPS=0; do
  echo "== $kv (re-run with the corrected guards)" | tee -a $OUT/matrix_rerun.txt
  env $kv timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_dit.py tests/test_hip_vae.py tests/test_hip_fullconfig.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -25 | grep -v "^\.\+ *\[" | tee -a $OUT/matrix_rerun.txt
done
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/gpu_tests.log 2>&1; echo "pytest exit $?" | tee -a $OUT/gpu_tests.log; tail -4 $OUT/gpu_tests.log
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_fold.py tests/test_hip_attention.py tests/test_hip_rowops.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1 | tee -a $OUT/repeat.txt
done
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench exit $?"; cut -c1-400 $OUT/bench_default.json
