#!/bin/bash
for r in 1 2; do
echo "--- new"; REPS=100 ONLY=proj,fc2,qkv,fc1 timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | tail -4
echo "--- prev"; (cd prev_tree && REPS=100 ONLY=proj,fc2,qkv,fc1 timeout 120 python tools/gemm_bench.py 2>&1 | grep -v amdgpu | tail -4)
done
echo "--- bench new"; timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"
echo "--- bench prev"; (cd prev_tree && timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: round(v['ms_per_step'],3) for k,v in d['kernels'].items()})")
