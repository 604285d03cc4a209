#!/bin/bash
timeout 900 python -m pytest tests/test_hip_gemm.py tests/test_hip_dit.py tests/test_hip_vae.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
PRIMX_GEMM_PROF=1 ONLY=proj REPS=3 timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | grep "gemm144" | awk 'NR%6==5'
for r in 1 2; do
for v in 1 0; do echo "--- REGEPI=$v"; PRIMX_GEMM_REGEPI=$v REPS=100 ONLY=proj,fc2,qkv,kv timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | tail -4; done
done
for v in 1 0 1 0; do echo "--- bench REGEPI=$v"; PRIMX_GEMM_REGEPI=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k[:24]: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"; done
