#!/bin/bash
# timestep plan (DiT.plan_timesteps): parity tests + same-box A/B of the headline bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_fullconfig.py tests/test_hip_e2e.py -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do
  PRIMX_PLAN_TIMESTEPS=0 timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('per-step', d['ms_per_step'])"
  timeout 200 python bench.py --no-cpu-baseline --no-parity --steps 25 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('planned ', d['ms_per_step'], d['kernels'].get('gemv16_kernel<1, 8> 8x292608x1152'))"
done
